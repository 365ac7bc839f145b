"""Per-tensor gradient error of the fused critic step vs the fp64 oracle (and the fp32 oracle's own error), level 4, batch 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd.networks import Discriminator, Generator
from musicgan_amd.optim import FusedAdam
from musicgan_amd.train_step import ProGANStepper
from oracle import progan as O
DEV = "cuda:0"
level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0); gs, ds = O.GenState(32), O.DiscState(7)
for _ in range(level): gs.next_layer(); ds.next_layer()
torch.manual_seed(0); gen, disc = Generator(32), Discriminator(7)
for _ in range(level): gen.next_layer(); disc.next_layer()
gen, disc = gen.to(DEV), disc.to(DEV)
rng = torch.Generator().manual_seed(1234); n = 4; side = 4 * 2 ** level
x_real = torch.rand(n, 2, side, side, generator=rng) * 2 - 1; z = torch.randn(n, 32, 2, 2, generator=rng); eps = torch.rand(n, 1, 1, 1, generator=rng)
r64 = O.d_step(gs, ds, x_real, z, eps, 0.5, dtype=torch.float64, detach_fake=True)
r32 = O.d_step(gs, ds, x_real, z, eps, 0.5, dtype=torch.float32, detach_fake=True)
od = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9)); od.step = lambda *a, **k: None
st = ProGANStepper(gen, disc, FusedAdam(gen.parameters()), od, 32, fused_d_step=True)
st.d_step(x_real.to(DEV), 0.5, z=z.to(DEV), eps=eps.to(DEV))
rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))
rows = [(rel(p.grad, r64["d_grads"][k]), rel(r32["d_grads"][k], r64["d_grads"][k]), k) for k, p in disc.named_parameters() if p.grad is not None]
for ours, ref, k in sorted(rows, reverse=True)[:6]:
    print(f"{k[-34:]:34s} ours {ours:.2e}  torch-fp32 {ref:.2e}")
