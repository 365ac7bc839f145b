import os, sys
sys.path.insert(0, "/root/repo")
import torch
from musicgan_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(11)
for (n, ci, co, h) in ((18, 32, 32, 256), (6, 32, 32, 256), (2, 32, 32, 64)):
    x = torch.randn(n, ci, h, h, device=dev, generator=g)
    w = torch.randn(co, ci, 3, 3, device=dev, generator=g) * 0.05
    b = torch.randn(co, device=dev, generator=g)
    wpk = ops.pack_wino3x3(w, False)
    os.environ["MG_WINO_STRIP"] = "0"
    y0 = ops.conv3x3(x, None, b, co, lrelu=True, wino=wpk)
    os.environ["MG_WINO_STRIP"] = "2"
    for t in range(2):
        y1 = ops.conv3x3(x, None, b, co, lrelu=True, wino=wpk)
        d = (y0 - y1).abs()
        nz = (d > 0).nonzero()
        print(n, ci, co, h, "trial", t, "max", float(d.max()), "count", int((d > 0).sum()), "of", d.numel())
        if nz.shape[0]:
            print("  imgs", sorted(set(nz[:, 0].tolist())), "chans", sorted(set(nz[:, 1].tolist())), "rows", sorted(set(nz[:, 2].tolist()))[:24], "cols", sorted(set(nz[:, 3].tolist()))[:40])
