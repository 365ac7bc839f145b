"""mg_conv3x3_small against the direct / Winograd kernel PackCache.conv picks today, per layer shape of the <= 8x8 ends of the networks.
    python tools/ab_smallconv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops
from musicgan_amd.networks.engine import PackCache

dev = torch.device("cuda", 0)


def t_us(fn, reps=20, inner=50):
    """Per-launch time inside a replayed HIP graph of `inner` dependent launches (eager launches are host-bound at ~10 us)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * inner) * 1e3


g = torch.Generator(device=dev).manual_seed(0)
print("shape (n, cin, cout, h): old kernel us | small us | max rel diff   [plain lrelu conv]   and with pool")
shapes = ((128, 128, 4), (128, 144, 4), (144, 144, 2), (144, 160, 2), (112, 112, 8), (112, 128, 8), (32, 128, 4), (96, 112, 8))
if len(sys.argv) > 1 and sys.argv[1] == "16":
    shapes = ((96, 96, 16), (96, 112, 16), (112, 96, 16))
for (ci, co, h) in shapes:
    for n in ((6, 8, 18, 24, 32) if h == 16 else (8, 24, 64, 96, 192)):
        x = torch.randn(n, ci, h, h, device=dev, generator=g)
        w = torch.randn(co, ci, 3, 3, device=dev, generator=g) * 0.05
        b = torch.randn(co, device=dev, generator=g) * 0.1
        cache = PackCache()
        os.environ["MG_SMALLCONV"] = "0"
        wp_sn = cache.get_sn(w, False)
        old = lambda: cache.conv(x, w, False, b, co, lrelu=True)
        new = lambda: ops.conv3x3_small(x, wp_sn, b, co, lrelu=True)
        y0, y1 = old(), new()
        d = float((y0 - y1).abs().max() / y0.abs().max())
        t0, t1 = t_us(old), t_us(new)
        oldp = lambda: cache.conv(x, w, False, b, co, lrelu=True, pool=True)
        newp = lambda: ops.conv3x3_small(x, wp_sn, b, co, lrelu=True, pool=True)
        (ya, pa), (yb, pb) = oldp(), newp()
        dp = float((pa - pb).abs().max() / pa.abs().max())
        t2, t3 = t_us(oldp), t_us(newp)
        print(f"({n:3d},{ci:3d},{co:3d},{h}): {t0:6.1f} | {t1:6.1f} | {d:.1e}     pool: {t2:6.1f} | {t3:6.1f} | {dp:.1e}", flush=True)
