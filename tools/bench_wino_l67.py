"""The Winograd conv on the layer shapes of levels 6 / 7 at the reference's batch (6 images for G, 18 for D), beside the same layer
at 64 / 192 images: executed fraction of the fp32 MFMA peak per shape, in-graph timing (20 launches per replay)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


print("shape (plain forward + lrelu)        n=6        n=18        n=64       n=192     [us, executed fraction of 157.3 TF/s]")
for ci, co, hw in ((16, 32, 512), (32, 32, 256), (32, 48, 256), (48, 48, 128), (48, 64, 128), (64, 64, 64), (64, 80, 64), (80, 80, 32), (80, 96, 32),
                   (96, 96, 16)):
    cells = []
    for n in (6, 18, 64, 192):
        if n * ci * hw * hw >= (1 << 29) or n * co * hw * hw >= (1 << 29):
            cells.append("      -     ")
            continue
        x = torch.randn(n, ci, hw, hw, device=dev, generator=g)
        w = torch.randn(co, ci, 3, 3, device=dev, generator=g) * 0.05
        b = torch.zeros(co, device=dev)
        ww = ops.pack_wino3x3(w, False)
        y = torch.empty(n, co, hw, hw, device=dev)
        t = timed(lambda: ops.conv3x3(x, None, b, co, lrelu=True, wino=ww, out=y))
        fl = 18.0 * n * hw * hw * ci * co / 2.25
        cells.append(f"{t:7.1f} {fl / t / 1e6 / 157.3:4.2f}")
        del x, y
    print(f"{ci:3d}->{co:3d} @ {hw:3d}                 " + "   ".join(cells), flush=True)
