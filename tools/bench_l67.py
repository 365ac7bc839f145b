"""The layer shapes levels 6-7 spend their time in at the reference's batch 6 (3N = 18 in the fused critic step): Winograd conv
forward / data gradient with and without the 128-tile single-channel-tile form, and the Winograd weight gradient.
    python tools/bench_l67.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)

def timeit(fn, iters=20):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters

def conv_case(name, n, ci, co, h, mask=False, pool=False):
    x = R(n, ci, h, h); wt = R(co, ci, 3, 3) * 0.05; b = None if mask else R(co)
    up = ops.pack_wino3x3(wt, dgrad=False); wp = ops.pack_conv3x3(wt, dgrad=False)
    aux = R(n, co, h, h) if mask else None
    kw = dict(lrelu=not mask, mask_aux=aux, pool=pool)
    fl = 2.0 * 9 * ci * co * h * h * n
    hbm = 4.0 * n * h * h * (ci + co * (2 if mask else 1))
    res = []
    for env in ("0", None):
        if env is None: os.environ.pop("MG_WINO_NARROW", None)
        else: os.environ["MG_WINO_NARROW"] = env
        res.append(timeit(lambda: ops.conv3x3(x, None, b, co, wino=up, **kw)))
    os.environ.pop("MG_WINO_NARROW", None)
    os.environ["MG_WINO_WT"] = "4"
    w2 = timeit(lambda: ops.conv3x3(x, None, b, co, wino=up, **kw))
    os.environ.pop("MG_WINO_WT")
    name = f"{name} [64-tile 1/CU {w2*1e3:.1f}]"
    if co <= 16:
        for v in ("2", "4"):
            os.environ["MG_WINO_NARROW_WT"] = v
            wv = timeit(lambda: ops.conv3x3(x, None, b, co, wino=up, **kw))
            name += f" [narrow wt{v} {wv*1e3:.1f}]"
        os.environ.pop("MG_WINO_NARROW_WT")
    md = timeit(lambda: ops.conv3x3(x, wp, b, co, **kw))
    print(f"{name:44s} two-tile {res[0]*1e3:8.1f} us  product {res[1]*1e3:8.1f} us  direct {md*1e3:8.1f} us | alg {fl/res[1]/1e9:6.1f} TF/s, "
          f"HBM floor {hbm/6e12*1e6:6.1f} us", flush=True)

def wgrad_case(name, n, ci, co, h, ups=False):
    x = R(n, ci, h // 2, h // 2) if ups else R(n, ci, h, h)
    gy = R(n, co, h, h)
    gw = torch.empty(co, ci, 3, 3, device=dev); gb = torch.empty(co, device=dev)
    mw = timeit(lambda: ops.conv3x3_wgrad(x, gy, gw, gb, ups=ups))
    os.environ["MG_WINO_WGRAD"] = "0"
    md = timeit(lambda: ops.conv3x3_wgrad(x, gy, gw, gb, ups=ups))
    os.environ.pop("MG_WINO_WGRAD")
    fl = 2.0 * 9 * ci * co * h * h * n
    hbm = 4.0 * n * (x.numel() / n + co * h * h)
    print(f"{name:44s} wino {mw*1e3:8.1f} us  direct {md*1e3:8.1f} us | alg {fl/mw/1e9:6.1f} TF/s, HBM floor {hbm/6e12*1e6:6.1f} us", flush=True)

conv_case("D0.0 dgrad 32->16@512 mask x18", 18, 32, 16, 512, mask=True)
conv_case("G7.4 dgrad 32->16@512 mask x6", 6, 32, 16, 512, mask=True)
conv_case("D0.0 fwd 16->32@512 lrelu+pool x18", 18, 16, 32, 512, pool=True)
conv_case("D0.3 fwd 32->32@256 lrelu x18", 18, 32, 32, 256)
conv_case("D1.0 fwd 32->48@256 lrelu+pool x18", 18, 32, 48, 256, pool=True)
conv_case("D1.0 dgrad 48->32@256 mask x18", 18, 48, 32, 256, mask=True)
conv_case("D1.3 fwd 48->48@128 lrelu x18", 18, 48, 48, 128)
wgrad_case("D0.0 wgrad 16x32@512 x18", 18, 16, 32, 512)
wgrad_case("D0.3 wgrad 32x32@256 x18", 18, 32, 32, 256)
wgrad_case("D1.0 wgrad 32x48@256 x18", 18, 32, 48, 256)
wgrad_case("D1.3 wgrad 48x48@128 x18", 18, 48, 48, 128)
wgrad_case("D2.0 wgrad 48x64@128 x18", 18, 48, 64, 128)
wgrad_case("G7.4 wgrad ups 32x16@512 x6", 6, 32, 16, 512, ups=True)
wgrad_case("G7.0 wgrad 32x32@256 x6", 6, 32, 32, 256)
wgrad_case("G6.4 wgrad ups 48x32@256 x6", 6, 48, 32, 256, ups=True)
wgrad_case("D2.0 wgrad 48x64@128 x192 (L5)", 192, 48, 64, 128)
wgrad_case("D3.0 wgrad 64x80@64 x192 (L5)", 192, 64, 80, 64)
