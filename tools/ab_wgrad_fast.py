"""Winograd weight gradient per layer shape, us per call (partial + reduce): run once with MG_WGRAD_FAST=0 (per-lane predicated loads) and
once with =1 (scalar-addressed loads, the default) -- the switch is read once per process.   python tools/ab_wgrad_fast.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
R = lambda *s: torch.randn(*s, device=dev, generator=g)
# (N, Cin, Cout, H, ups): level 5 at batch 64 (192 / 64 images), levels 6-7 at batch 6 (18 / 6 images)
cases = [(192, 48, 64, 128, False), (192, 64, 64, 64, False), (192, 64, 80, 64, False), (192, 80, 80, 32, False), (192, 80, 96, 32, False),
         (64, 64, 48, 128, True), (64, 80, 64, 64, True), (18, 16, 32, 512, False), (18, 32, 48, 256, False), (18, 32, 32, 256, False),
         (18, 48, 64, 128, False), (6, 32, 16, 512, True), (6, 48, 32, 256, True), (6, 32, 32, 256, False), (24, 96, 96, 16, False)]
print("MG_WGRAD_FAST =", os.environ.get("MG_WGRAD_FAST", "(default 1)"))
for (n, ci, co, h, ups) in cases:
    x = R(n, ci, h // 2 if ups else h, h // 2 if ups else h)
    gy = R(n, co, h, h)
    gw, gb = torch.empty(co, ci, 3, 3, device=dev), torch.empty(co, device=dev)
    fn = lambda: ops.conv3x3_wgrad(x, gy, gw, gb, ups=ups)
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    ref = None
    print(f"{n:4d} {ci:3d}x{co:3d} @{h:3d} {'ups' if ups else '   '}  {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us   checksum {float(gw.double().abs().sum()):.6e} {float(gb.double().abs().sum()):.6e}", flush=True)
