"""Diagnostic (GPU box): the product's free-running trajectory next to the fp64 oracle's on the fixture's inputs -- per iteration
the loss deviations, and every gradient entry whose SIGN differs from the oracle's (Adam(beta1=0) moves such a weight by lr in the
opposite direction at its first step)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load, trajectory_inputs  # noqa: E402
from musicgan_amd.networks import Discriminator, Generator  # noqa: E402
from musicgan_amd.optim import FusedAdam  # noqa: E402
from musicgan_amd.train_step import ProGANStepper  # noqa: E402
from musicgan_amd.utils import Grower  # noqa: E402
from oracle import progan as O  # noqa: E402

DEV = 'cuda:0'
LR, BETAS = 1e-3, (0.0, 0.9)
g = load("progan_trajectory.npz")
seed, rc, batch = int(g["seed"]), int(g["rand_channels"]), int(g["batch"])
torch.manual_seed(seed)
tr = O.Trajectory(rc, O.GrowerState(7, g["fadein"].tolist(), g["train_lengths"].tolist()), dtype=torch.float64)
torch.manual_seed(seed)
gen, disc = Generator(rc).to(DEV), Discriminator(7).to(DEV)
og = FusedAdam(gen.parameters(), lr=LR, betas=BETAS)
od = FusedAdam(disc.parameters(), lr=LR, betas=BETAS)
st = ProGANStepper(gen, disc, og, od, rc)
gr = Grower(7, g["fadein"].tolist(), g["train_lengths"].tolist())
for it in range(16):
    x_real, z, z2, eps = trajectory_inputs(g, it)
    a = gr.alpha
    # snapshot the oracle's Adam state BEFORE the step to know which step count each tensor is at
    rec = tr.iteration(x_real, z, eps, z2, growth_seed=seed + 3000 + tr.gs.curr_layer)
    m = st.d_step(x_real.to(DEV), a, z=z.to(DEV), eps=eps.to(DEV))
    print(f"it {it} L{rec['level']} d_loss dev {abs(float(m['disc_loss']) - rec['disc_loss']):.1e} gp dev {abs(float(m['grad_pen']) - rec['grad_pen']):.1e}")
    for k, p in disc.named_parameters():
        if p.grad is None or k not in rec["d_grads"]:
            continue
        r = rec["d_grads"][k]
        o = p.grad.double().cpu()
        flip = (torch.sign(o) != torch.sign(r)) & (r != 0)
        if int(flip.sum()):
            idx = flip.nonzero()[:3]
            vals = [(float(r[tuple(i)]), float(o[tuple(i)])) for i in idx]
            print(f"      D {k}: {int(flip.sum())} sign flips of {r.numel()}, tensor max {float(r.abs().max()):.2e}, e.g. (fp64, ours) {vals}, max err {float((o - r).abs().max()):.2e}")
    if it % 5 == 0:
        st.g_step(batch, a, DEV, z=z2.to(DEV))
        for k, p in gen.named_parameters():
            if p.grad is None or k not in rec.get("g_grads", {}):
                continue
            r = rec["g_grads"][k]
            o = p.grad.double().cpu()
            flip = (torch.sign(o) != torch.sign(r)) & (r != 0)
            if int(flip.sum()):
                idx = flip.nonzero()[:3]
                vals = [(float(r[tuple(i)]), float(o[tuple(i)])) for i in idx]
                print(f"      G {k}: {int(flip.sum())} sign flips of {r.numel()}, tensor max {float(r.abs().max()):.2e}, e.g. {vals}")
    if gr.grow(batch) and gen.growing:
        torch.manual_seed(seed + 3000 + gen.curr_layer)
        gen.next_layer()
        disc.next_layer()
        og.add_param_group({"params": gen.end_block_params(), "lr": LR, "betas": BETAS})
        od.add_param_group({"params": disc.start_block_parameters(), "lr": LR, "betas": BETAS})
