import os, sys
sys.path.insert(0, "/root/repo")
import torch
from musicgan_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
def T(fn, it=30):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for (n, ci, co, h, mode) in ((192, 48, 64, 128, "fwd"), (192, 64, 64, 64, "plain"), (192, 64, 48, 128, "dgrad")):
    x = torch.randn(n, ci, h, h, device=dev, generator=g); w = torch.randn(co, ci, 3, 3, device=dev, generator=g) * 0.05; b = torch.randn(co, device=dev, generator=g)
    up = ops.pack_wino3x3(w, False); aux = torch.randn(n, co, h, h, device=dev, generator=g) if mode == "dgrad" else None
    if mode == "fwd": fn = lambda: ops.conv3x3(x, None, b, co, lrelu=True, pool=True, wino=up, mask_out=True)
    elif mode == "dgrad": fn = lambda: ops.conv3x3(x, None, None, co, mask_aux=aux, wino=up)
    else: fn = lambda: ops.conv3x3(x, None, b, co, lrelu=True, wino=up)
    os.environ["MG_WINO_STRIP"] = "2"
    res = []
    for abl in ("0", "2", "4", "6"):
        os.environ["MG_WINO_STRIP_ABLATE"] = abl
        res.append(T(fn))
    os.environ.pop("MG_WINO_STRIP_ABLATE")
    print(f"{n}x{ci}->{co}@{h} {mode}: full {res[0]:.1f}  no input fetch {res[1]:.1f}  no epilogue memory {res[2]:.1f}  neither {res[3]:.1f} us", flush=True)
