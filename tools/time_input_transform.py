"""Fused input transform (mg_input_transform) vs the torch tensor expressions on the GPU, batch 64 of 2x512x512 float64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgan_amd import ops
from musicgan_amd.utils import Grower
dev = torch.device("cuda", 0)
x = torch.rand(64, 2, 512, 512, dtype=torch.float64, device=dev)
def timeit(fn, iters=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters
for side in (128, 64, 4):
    g = Grower(7, [1] * 8, [1] * 7)
    while g.scale_transform.side != side:
        g.grow(1)
    a = timeit(lambda: ops.input_transform(x, side))
    b = timeit(lambda: g.scale_transform(x.to(torch.float32)))
    print(f"side {side}: fused {a:.3f} ms ({x.numel()*8/a/1e6:.0f} GB/s of input)   torch ops {b:.3f} ms", flush=True)
