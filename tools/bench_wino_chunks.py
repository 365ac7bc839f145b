"""Time of the Winograd conv as a function of the number of 8-channel chunks (fixed tile grid): slope = cost of a chunk, intercept =
   per-workgroup fixed cost (set-up, epilogue, workgroup turnover).  python tools/bench_wino_chunks.py [N H Cout]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
CO = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda", 0)


def timeit(fn, iters=20):
    for _ in range(30): fn()  # past the clock ramp that follows an idle chip
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


wgs = N * (H // 2) * (H // 2) // 64
print(f"N={N} {H}x{H} Cout={CO}: {wgs} workgroups of 64 tiles")
pts = []
for ci in (8, 16, 32, 48, 64, 80, 96, 128):
    x = torch.randn(N, ci, H, H, device=dev)
    w = torch.randn(CO, ci, 3, 3, device=dev) * 0.05
    up = ops.pack_wino3x3(w, dgrad=False)
    b = torch.randn(CO, device=dev)
    t = timeit(lambda: ops.conv3x3(x, None, b, CO, lrelu=True, wino=up))
    pts.append((ci // 8, t))
    print(f"  Cin {ci:4d} ({ci // 8:2d} chunks): {t:8.1f} us   {t / (wgs / 256):7.2f} us per workgroup round", flush=True)
n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); icpt = (sy - slope * sx) / n
print(f"  fit: {icpt:.1f} us fixed + {slope:.1f} us per chunk  (per workgroup round: {icpt / (wgs / 256):.2f} + {slope / (wgs / 256):.2f} us)")
