"""Coarse wall-clock breakdown of one D step and one G step (debug aid; syncs after every phase)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from musicgan_amd import networks
from musicgan_amd.optim import FusedAdam

level = int(sys.argv[1]) if len(sys.argv) > 1 else 5
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
gen, disc = bench.build_nets(level, 32, dev)
og = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9)); od = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))
side = bench.LEVEL_SIDE[level]
x_real = torch.rand(batch, 2, side, side, device=dev) * 2 - 1
def T(msg, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); print(f"{msg}: {1e3*(t-t0):.2f} ms", flush=True); return time.perf_counter()
for it in range(3):
    print("iter", it, flush=True)
    t = time.perf_counter()
    z = torch.randn(batch, 32, 2, 2, device=dev); eps = torch.rand(batch, 1, 1, 1, device=dev)
    with torch.no_grad(): x_fake = gen(z, 0.5)
    t = T("G fwd (no grad)", t)
    out_real = disc(x_real, 0.5); t = T("D fwd real", t)
    out_fake = disc(x_fake, 0.5); t = T("D fwd fake", t)
    gp = disc.gradient_penalty_with_eps(x_real, x_fake, 0.5, eps); t = T("GP fwd (D fwd + dgrad chain)", t)
    loss = networks.wasserstein_discriminator_loss(out_real, out_fake) + gp
    disc.zero_grad(); loss.backward(); t = T("D-step backward (2x D bwd + GP 2nd order)", t)
    od.step(); t = T("Adam D", t)
    x_fake = gen(z, 0.5); t = T("G fwd", t)
    for p in disc.parameters(): p.requires_grad_(False)
    out = disc(x_fake, 0.5); t = T("D fwd", t)
    gl = networks.wasserstein_generator_loss(out); gen.zero_grad(); gl.backward(); t = T("G-step backward (D dgrad + G bwd)", t)
    for p in disc.parameters(): p.requires_grad_(True)
    og.step(); t = T("Adam G", t)
print("max mem GB", torch.cuda.max_memory_allocated()/2**30)
