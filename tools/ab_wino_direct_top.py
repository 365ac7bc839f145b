"""Winograd vs direct 3x3 conv on the few-channel layers at the top of levels 6 / 7 (where the Winograd transforms are amortised over
4-6 channel chunks only), timed inside a replayed HIP graph of 20 launches.  python tools/ab_wino_direct_top.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)


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


print("case                              wino us   direct us   (plain forward, lrelu)      wino+pool us  direct+pool us")
for n, ci, co, hw in ((18, 32, 48, 256), (18, 48, 32, 256), (6, 32, 48, 256), (6, 48, 32, 256), (18, 16, 32, 512), (18, 32, 16, 512),
                      (6, 16, 32, 512), (6, 32, 16, 512), (18, 48, 64, 128), (6, 48, 64, 128)):
    x = torch.randn(n, ci, hw, hw, device=dev, generator=g)
    w = torch.randn(co, ci, 3, 3, device=dev, generator=g) * 0.05
    b = torch.zeros(co, device=dev)
    wp, ww = ops.pack_conv3x3(w, False), ops.pack_wino3x3(w, False)
    y = torch.empty(n, co, hw, hw, device=dev)
    pool = torch.empty(n, co, hw // 2, hw // 2, device=dev)
    t_w = timed(lambda: ops.conv3x3(x, None, b, co, lrelu=True, wino=ww, out=y))
    t_d = timed(lambda: ops.conv3x3(x, wp, b, co, lrelu=True, out=y))
    t_wp = timed(lambda: ops.conv3x3(x, None, b, co, lrelu=True, wino=ww, out=y, pool_out=pool))
    t_dp = timed(lambda: ops.conv3x3(x, wp, b, co, lrelu=True, out=y, pool_out=pool))
    fl = 18.0 * n * hw * hw * ci * co
    print(f"{n:3d} x {ci:3d}->{co:3d} @ {hw:3d}   {t_w:9.1f} {t_d:9.1f}   ({fl / t_w / 1e6:6.1f} / {fl / t_d / 1e6:6.1f} TF/s alg)   {t_wp:9.1f} {t_dp:9.1f}", flush=True)
