"""Fused epilogues of the Winograd conv against the separate kernels they replace (HIP events).
   python tools/bench_wino_epilogues.py [batch]      (MG_WINO_WT=2/4 and MG_WINO_CFG override the tiling)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import _lib, ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)


def timeit(fn, iters=20):
    for _ in range(30): fn()  # past the clock ramp that follows an idle chip
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def tile_mask(act):
    n, c, h, w = act.shape
    b = (act > 0).reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4).to(torch.uint8)
    return (b[..., 0] + 2 * b[..., 1] + 4 * b[..., 2] + 8 * b[..., 3]).contiguous()


def case(name, ci, co, h, w, n=N):
    x = R(n, ci, h, w); wt = R(co, ci, 3, 3) * 0.05; b = R(co)
    up = ops.pack_wino3x3(wt, dgrad=False)
    coef = torch.tensor([0.5, 0.5], device=dev)
    other = R(n, co, h, w)
    act2 = R(n, co, 2 * h, 2 * w)
    m2 = tile_mask(act2)
    m1 = tile_mask(other)
    plain = timeit(lambda: ops.conv3x3(x, None, None, co, wino=up))
    t_un = timeit(lambda: ops.conv3x3(x, None, None, co, wino=up, unpool_mask=m2))
    y = ops.conv3x3(x, None, None, co, wino=up)
    t_ap = timeit(lambda: ops.avgpool2_bwd(y, act2))
    t_apb = timeit(lambda: ops.avgpool2_bwd(y, m2))
    t_ff = timeit(lambda: ops.conv3x3_fade(x, up, b, co, _lib.MG_FADE_FWD, other, coef))
    t_f0 = timeit(lambda: ops.conv3x3(x, None, b, co, lrelu=True, wino=up))
    t_ax = timeit(lambda: ops.axpby(0.5, y, 0.5, other, coef=coef))
    t_fb = timeit(lambda: ops.conv3x3_fade(x, up, None, co, _lib.MG_FADE_BWD, other, coef, mask_in=m1))
    t_bl = timeit(lambda: ops.blend_lrelu_bwd(y, other, other, 0.5, 0.5, coef=coef))
    t_pm = timeit(lambda: ops.conv3x3(x, None, b, co, lrelu=True, pool=True, wino=up, mask_out=True))
    t_p0 = timeit(lambda: ops.conv3x3(x, None, b, co, lrelu=True, pool=True, wino=up))
    print(f"{name:24s} plain {plain:6.0f} | unpool {t_un:6.0f} vs {plain:5.0f}+{t_ap:4.0f} (bytes {t_apb:4.0f}) | fade fwd {t_ff:6.0f} vs {t_f0:5.0f}+{t_ax:4.0f}"
          f" | fade bwd {t_fb:6.0f} vs {plain:5.0f}+{t_bl:4.0f} | pool+mask {t_pm:6.0f} vs pool+y {t_p0:6.0f}   us", flush=True)


case("64->64@64 x192", 64, 64, 64, 64)
case("80->64@64 x192", 80, 64, 64, 64)
case("80->80@32 x192", 80, 80, 32, 32)
case("96->96@16 x192", 96, 96, 16, 16)
case("64->64@64 x64", 64, 64, 64, 64, n=64)
case("48->64@128 x64", 48, 64, 128, 128, n=64)
