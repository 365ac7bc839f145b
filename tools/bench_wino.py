"""Winograd vs direct 3x3 conv kernels on the dominant layer shapes (HIP events).  python tools/bench_wino.py [batch]"""
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

print(f"{'case':40s} {'direct ms':>10s} {'TF':>7s} {'wino ms':>9s} {'alg TF':>7s} {'mfma TF':>8s}")
def case(name, ci, co, h, w, n=N, pn=False, mask=False, pool=False):
    x = R(n, ci, h, w); wt = R(co, ci, 3, 3) * 0.05; b = None if mask else R(co)
    wp = ops.pack_conv3x3(wt, dgrad=False); up = ops.pack_wino3x3(wt, dgrad=False)
    aux = R(n, co, h, w) if mask else None
    kw = dict(lrelu=not mask, mask_aux=aux, pixnorm=pn, pool=pool)
    md = timeit(lambda: ops.conv3x3(x, wp, b, co, **kw))
    mw = timeit(lambda: ops.conv3x3(x, None, b, co, wino=up, **kw))
    fl = 2.0 * 9 * ci * co * h * w * n
    print(f"{name:40s} {md:10.3f} {fl/md/1e9:7.1f} {mw:9.3f} {fl/mw/1e9:7.1f} {fl/2.25/mw/1e9:8.1f}", flush=True)

case("D2.0 48->64@128 lrelu+pool", 48, 64, 128, 128, pool=True)
case("D2.0 48->64@128 x3N", 48, 64, 128, 128, n=3 * N, pool=True)
case("D2.0 dgrad 64->48@128 mask", 64, 48, 128, 128, mask=True)
case("D2.3 64->64@64", 64, 64, 64, 64)
case("G5.0 64->64@64 PN", 64, 64, 64, 64, pn=True)
case("D3.0 64->80@64", 64, 80, 64, 64)
case("D3.0 dgrad 80->64@64 mask", 80, 64, 64, 64, mask=True)
case("D3.3 80->80@32", 80, 80, 32, 32)
case("D4.0 80->96@32", 80, 96, 32, 32)
case("D4.3 96->96@16", 96, 96, 16, 16)
case("D4.3 96->96@16 x3N", 96, 96, 16, 16, n=3 * N)
case("D5.0 96->112@16 x3N", 96, 112, 16, 16, n=3 * N)
case("D5.3 112->112@8 x3N", 112, 112, 8, 8, n=3 * N)
