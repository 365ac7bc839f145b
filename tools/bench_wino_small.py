import os, sys
sys.path.insert(0, os.getcwd())
os.environ["MG_WINO_MIN_PIXELS"] = "1"
import torch
from musicgan_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)
def timeit(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (n, ci, co, h) in [(64, 112, 112, 8), (64, 112, 128, 8), (192, 128, 128, 4), (64, 128, 128, 4), (64, 96, 96, 16), (32, 96, 96, 16), (32, 80, 80, 32), (32, 112, 112, 8), (96, 112, 112, 8)]:
    x = R(n, ci, h, h); wt = R(co, ci, 3, 3) * 0.05; b = R(co)
    wp = ops.pack_conv3x3(wt, dgrad=False); up = ops.pack_wino3x3(wt, dgrad=False)
    md = timeit(lambda: ops.conv3x3(x, wp, b, co, lrelu=True))
    ok = ops.wino3x3_supported(n, co, h, h, cin=ci)
    mw = timeit(lambda: ops.conv3x3(x, None, b, co, lrelu=True, wino=up)) if ok else float("nan")
    print(f"N={n:3d} {ci}->{co}@{h}: direct {md:7.1f} us  wino {mw:7.1f} us  (pixels {n*h*h})", flush=True)
