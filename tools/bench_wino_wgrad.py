"""Winograd vs direct 3x3 weight-gradient kernels on the dominant layer shapes (HIP events).  python tools/bench_wino_wgrad.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)

def timeit(fn, iters=20):
    for _ in range(30): fn()  # past the clock ramp that follows an idle chip
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters

print(f"{'case':34s} {'direct ms':>10s} {'TF':>7s} {'wino ms':>9s} {'alg TF':>7s} {'mfma TF':>8s}")
def case(name, ci, co, h, w, n=N):
    x = R(n, ci, h, w); gy = R(n, co, h, w)
    gw = torch.empty(co, ci, 3, 3, device=dev); gb = torch.empty(co, device=dev)
    os.environ["MG_WINO_WGRAD"] = "0"
    md = timeit(lambda: ops.conv3x3_wgrad(x, gy, gw, gb))
    os.environ["MG_WINO_WGRAD"] = "1"; os.environ["MG_WINO_WGRAD_MIN_PIXELS"] = "1"
    mw = timeit(lambda: ops.conv3x3_wgrad(x, gy, gw, gb))
    fl = 2.0 * 9 * ci * co * h * w * n
    print(f"{name:34s} {md:10.3f} {fl/md/1e9:7.1f} {mw:9.3f} {fl/mw/1e9:7.1f} {fl/2.25/mw/1e9:8.1f}", flush=True)

case("D2.0 48->64@128", 48, 64, 128, 128)
case("D2.0 48->64@128 x3N", 48, 64, 128, 128, n=3 * N)
case("D2.3 64->64@64", 64, 64, 64, 64)
case("D2.3 64->64@64 x3N", 64, 64, 64, 64, n=3 * N)
case("D3.0 64->80@64 x3N", 64, 80, 64, 64, n=3 * N)
case("D3.3 80->80@32 x3N", 80, 80, 32, 32, n=3 * N)
case("D4.0 80->96@32 x3N", 80, 96, 32, 32, n=3 * N)
case("D4.3 96->96@16 x3N", 96, 96, 16, 16, n=3 * N)
case("D5.0 96->112@16 x3N", 96, 112, 16, 16, n=3 * N)
case("D5.3 112->112@8 x3N", 112, 112, 8, 8, n=3 * N)
case("G4.0 80->80@32", 80, 80, 32, 32)
case("G3.0 96->96@16", 96, 96, 16, 16)
