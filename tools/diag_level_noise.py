"""Diagnostic: per-tensor gradient disagreement between the Winograd/sub-pixel path and the direct kernels at a given level,
next to the un-cancelled scale of the same tensor (gradient of mean(D(x_real)) alone)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from musicgan_amd.optim import FusedAdam  # noqa: E402
from musicgan_amd.train_step import ProGANStepper  # noqa: E402

level, batch = int(sys.argv[1]), int(sys.argv[2])
DEV = torch.device("cuda:0")
side = bench.LEVEL_SIDE[level]
res = {}
for mode in ("product", "direct"):
    if mode == "direct":
        os.environ.update(MG_WINO="0", MG_WINO_WGRAD="0", MG_UPCONV_DGRAD="0")
    gen, disc = bench.build_nets(level, 32, DEV)
    og = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
    od = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))
    od.step = lambda *a, **k: None
    og.step = lambda *a, **k: None
    st = ProGANStepper(gen, disc, og, od, 32)
    rng = torch.Generator(device=DEV).manual_seed(1234)
    x_real = torch.rand(batch, 2, side, side, device=DEV, generator=rng) * 2 - 1
    z = torch.randn(batch, 32, 2, 2, device=DEV, generator=rng)
    eps = torch.rand(batch, 1, 1, 1, device=DEV, generator=rng)
    st.d_step(x_real, 0.5, z=z, eps=eps)
    gd = {k: p.grad.detach().clone() for k, p in disc.named_parameters() if p.grad is not None}
    disc.zero_grad()
    disc(x_real, 0.5).mean().backward()
    gs = {k: p.grad.detach().clone() for k, p in disc.named_parameters() if p.grad is not None}
    res[mode] = (gd, gs)
ga, sa = res["product"]
gb, sb = res["direct"]
gmax = max(float(v.abs().max()) for v in gb.values())
print(f"level {level} batch {batch}: network grad max {gmax:.3e}")
for k in gb:
    e = float((ga[k] - gb[k]).abs().max())
    own = float(gb[k].abs().max())
    term = float(sb[k].abs().max())
    es = float((sa[k] - sb[k]).abs().max())
    print(f"{k:50s} own {own:.2e} err {e:.2e} err/own {e/own:.1e} err/gmax {e/gmax:.1e} | term {term:.2e} err/term {e/term:.1e} | single-term err/term {es/term:.1e}")
