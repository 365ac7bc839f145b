"""Fused input transform (mg_input_transform) vs the torch tensor expressions on the GPU, batch 64 of 2x512x512, float64 (the .pt
files' type) and float32 (the packed side-car's), in-graph timing of 20 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgan_amd import ops
from musicgan_amd.utils import Grower
dev = torch.device("cuda", 0)


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
    return e0.elapsed_time(e1) / (5 * reps)


for dt in (torch.float64, torch.float32):
    x = torch.rand(64, 2, 512, 512, dtype=dt, device=dev)
    for side in (256, 128, 64, 16, 4):
        a = timed(lambda: ops.input_transform(x, side))
        by = x.numel() * x.element_size()
        print(f"{str(dt):14s} side {side:3d}: {a * 1e3:7.1f} us  = {2 * by / a / 1e9:6.2f} TB/s over two reads of the batch", flush=True)
x = torch.rand(64, 2, 512, 512, dtype=torch.float32, device=dev)
g = Grower(7, [1] * 8, [1] * 7)
while g.scale_transform.side != 128:
    g.grow(1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.scale_transform(x); e0.record()
for _ in range(5): g.scale_transform(x)
e1.record(); e1.synchronize()
print(f"torch tensor expressions on the GPU, side 128, float32: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us")
