"""Upsample(x2) -> Conv3x3 (+ LeakyReLU + PixelNorm) and its data gradient: the sub-pixel kernels (upconv3x3.hip) against the
9-component Winograd kernels (wino_ups.hip), us per call, same process.   python tools/ab_winoups.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
R = lambda *s: torch.randn(*s, device=dev, generator=g)


def timeit(fn, n=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# (N, Cin, Cout, Hin): level 5 batch 64; levels 6 / 7 at batch 6 and 16
cases = [(64, 64, 48, 64), (6, 64, 48, 64), (6, 48, 32, 128), (6, 32, 16, 256), (16, 32, 16, 256), (16, 48, 32, 128), (32, 64, 48, 64),
         (64, 80, 64, 32), (64, 96, 80, 16), (6, 80, 64, 32)]  # (the last three: data gradient only, in two launches of <= 3 tiles)
print(" N  cin->cout @hin    fwd: sub-pixel  winoups   ratio | dgrad: stride-2  winoups   ratio   bound us (0.25 x direct FLOP / 157.3 TF/s)")
for (n, ci, co, h) in cases:
    x, w, b = R(n, ci, h, h), R(co, ci, 3, 3) * 0.05, R(co)
    gy = R(n, co, 2 * h, 2 * h)
    wp, up = ops.pack_upconv3x3(w), ops.pack_winoups3x3(w, False)
    wpd, upd = ops.pack_upconv3x3_dgrad(w), ops.pack_winoups3x3(w, True)
    t0 = timeit(lambda: ops.upconv3x3(x, wp, b, co, lrelu=True, pixnorm=True, want_y=False))
    fwd_ok = ops.winoups3x3_supported(n, ci, co, h, h)
    t1 = timeit(lambda: ops.winoups3x3(x, up, b, co, lrelu=True, pixnorm=True, want_y=False)) if fwd_ok else float("nan")
    d0 = timeit(lambda: ops.upconv3x3_dgrad(gy, wpd, ci)) if ops.upconv3x3_dgrad_supported(h, h, gy.numel(), n) else float("nan")
    d1 = timeit(lambda: ops.winoups3x3_dgrad(gy, upd, ci))
    bound = 18.0 * n * ci * co * 4 * h * h * 0.25 / 157.3e12 * 1e6
    print(f"{n:3d} {ci:3d}->{co:3d} @{h:3d}        {t0:9.1f} {t1:9.1f} {t1 / t0:7.3f} |      {d0:9.1f} {d1:9.1f} {d1 / d0:7.3f}   {bound:7.1f}", flush=True)
