"""Diagnostic (GPU box): per-tensor deviation of the product's critic / generator gradients from the fp64 oracle, next to the
plain-PyTorch fp32 oracle's own deviation -- the table behind the gradient tolerance of tests/test_networks_gpu.py.
    python tools/diag_grad_noise.py [case ...]      (golden case names, or  L<level>b<batch>  for a synthetic one)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import PROGAN_CASES, build_oracle_states, grad_atol, load  # noqa: E402
from oracle import progan as O  # noqa: E402

DEV = "cuda:0"


def modules_for(g):
    from test_networks_gpu import build_modules
    return build_modules(g)


def synthetic(level, batch):
    import bench
    from musicgan_amd.networks import Discriminator, Generator  # noqa: F401
    torch.manual_seed(0)
    gs, ds = O.GenState(32), O.DiscState(7)
    for _ in range(level):
        gs.next_layer()
        ds.next_layer()
    gen, disc = bench.build_nets(level, 32, DEV)
    side = bench.LEVEL_SIDE[level]
    rng = torch.Generator().manual_seed(1234)
    x_real = torch.rand(batch, 2, side, side, generator=rng) * 2 - 1
    z = torch.randn(batch, 32, 2, 2, generator=rng)
    eps = torch.rand(batch, 1, 1, 1, generator=rng)
    return gs, ds, gen, disc, x_real, z, eps, 0.5, 32


def run(case):
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper
    if case.startswith("L"):
        level, batch = case[1:].split("b")
        gs, ds, gen, disc, x_real, z, eps, alpha, rc = synthetic(int(level), int(batch))
    else:
        g = load(f"progan_{case}.npz")
        gs, ds = build_oracle_states(g)
        gen, disc = modules_for(g)
        x_real, z, eps = (torch.from_numpy(g[k]) for k in ("x_real", "z", "eps"))
        alpha, rc = float(g["alpha"]), int(g["rand_channels"])
    o64 = O.d_step(gs, ds, x_real, z, eps, alpha, dtype=torch.float64, detach_fake=True)
    o32 = O.d_step(gs, ds, x_real, z, eps, alpha, dtype=torch.float32, detach_fake=True)
    terms = O.real_term_grads(ds, x_real, alpha)
    og = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
    od = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))
    od.step = lambda *a, **k: None
    og.step = lambda *a, **k: None
    st = ProGANStepper(gen, disc, og, od, rc)
    m = st.d_step(x_real.to(DEV), alpha, z=z.to(DEV), eps=eps.to(DEV))
    print(f"== {case}: grad_pen {float(m['grad_pen']):.6f} (fp64 {float(o64['grad_pen']):.6f})  modes: "
          f"MG_WINO_MIN_PIXELS={os.environ.get('MG_WINO_MIN_PIXELS', 'default')}")
    gmax = max(float(v.abs().max()) for v in o64["d_grads"].values())
    print(f"{'tensor':48s} {'max|g|':>9s} {'ours-64':>9s} {'fp32-64':>9s} {'ours/own':>9s} {'ours/fp32':>9s} {'term':>9s} {'ours/term':>9s} {'fp32/term':>9s}")
    for k, p in disc.named_parameters():
        if p.grad is None:
            continue
        r = o64["d_grads"][k]
        own = float(r.abs().max())
        e = float((p.grad.double().cpu() - r).abs().max())
        e32 = float((o32["d_grads"][k].double() - r).abs().max())
        t = float(terms[k].abs().max())
        flag = "" if e <= max(1e-3 * own, 2 * e32) else ("   (residue: within 1e-6 of the term)" if e <= grad_atol(k, o64["d_grads"], o32["d_grads"], terms) else "   <-- OUTSIDE the gate")
        print(f"{k:48s} {own:9.2e} {e:9.2e} {e32:9.2e} {e / max(own, 1e-30):9.1e} {e / max(e32, 1e-30):9.1f} {t:9.2e} {e / t:9.1e} {e32 / t:9.1e}{flag}")
    o64g = O.g_step(gs, ds, z, alpha, dtype=torch.float64)
    o32g = O.g_step(gs, ds, z, alpha, dtype=torch.float32)
    st.g_step(z.shape[0], alpha, DEV, z=z.to(DEV))
    for k, p in gen.named_parameters():
        if p.grad is None:
            continue
        r = o64g["g_grads"][k]
        own = float(r.abs().max())
        e = float((p.grad.double().cpu() - r).abs().max())
        e32 = float((o32g["g_grads"][k].double() - r).abs().max())
        flag = "" if e <= max(1e-3 * own, 2 * e32) else "   <-- OUTSIDE the gate"
        print(f"G {k:46s} {own:9.2e} {e:9.2e} {e32:9.2e} {e / max(own, 1e-30):9.1e} {e / max(e32, 1e-30):9.1f}{flag}")


if __name__ == "__main__":
    torch.set_num_threads(16)
    for c in (sys.argv[1:] or PROGAN_CASES + ["L4b4"]):
        run(c)
