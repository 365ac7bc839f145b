"""Diagnostic (GPU box): per-layer round-off of the critic's forward activations -- product (HIP kernels) and plain-PyTorch
fp32 on the CPU, each against an fp64 evaluation -- and LeakyReLU sign flips against fp64.
    python tools/diag_act_noise.py <golden case | L<level>b<batch>>"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from diag_grad_noise import synthetic  # noqa: E402
from golden_util import build_oracle_states, load  # noqa: E402
from musicgan_amd.networks import engine  # noqa: E402


def cpu_acts(ds, x, alpha, dtype):
    p = {k: v.to(dtype) for k, v in ds.params.items()}
    x = x.to(dtype)
    out = {}
    h = F.leaky_relu(F.conv2d(x, p["_Discriminator__start_block.0.weight"], p["_Discriminator__start_block.0.bias"]), 0.2)
    out["h0"] = h
    for j, i in enumerate(range(ds.curr_layer, 9)):
        pre = f"_Discriminator__conv_blocks.{i}."
        a1 = F.leaky_relu(F.conv2d(h, p[pre + "0.weight"], p[pre + "0.bias"], padding=1), 0.2)
        a2 = F.leaky_relu(F.conv2d(F.avg_pool2d(a1, 2, 2), p[pre + "3.weight"], p[pre + "3.bias"], padding=1), 0.2)
        out[f"a1_{j}"], out[f"a2_{j}"] = a1, a2
        h = a2
        if j == 0 and ds.has_last:
            o = F.leaky_relu(F.conv2d(F.avg_pool2d(x, 2, 2), p["_Discriminator__last_start_block.1.0.weight"],
                                      p["_Discriminator__last_start_block.1.0.bias"]), 0.2)
            h = alpha * a2 + (1 - alpha) * o
    out["out"] = F.linear(h.flatten(1), p["_Discriminator__clf.0.weight"], p["_Discriminator__clf.0.bias"])
    return out


case = sys.argv[1]
if case.startswith("L"):
    level, batch = case[1:].split("b")
    gs, ds, gen, disc, x_real, z, eps, alpha, rc = synthetic(int(level), int(batch))
else:
    from test_networks_gpu import build_modules
    g = load(f"progan_{case}.npz")
    gs, ds = build_oracle_states(g)
    gen, disc = build_modules(g)
    x_real, alpha = torch.from_numpy(g["x_real"]), float(g["alpha"])
with torch.no_grad():
    out, ctx = engine.disc_forward(disc._weights(), x_real.cuda().contiguous(), alpha, disc._pack_cache, save=True)
_, h0, saved, _, _, _, _ = ctx
ours = {"h0": h0, "out": out}
for j, s in enumerate(saved):
    ours[f"a1_{j}"], ours[f"a2_{j}"] = s[1], s[3]
r64, r32 = cpu_acts(ds, x_real, alpha, torch.float64), cpu_acts(ds, x_real, alpha, torch.float32)
print(f"{'layer':8s} {'shape':22s} {'rms':>9s} {'ours-64/rms':>12s} {'fp32-64/rms':>12s} {'ratio':>6s} {'flips ours':>10s} {'flips fp32':>10s}")
for k, ref in r64.items():
    a, b = ours[k].double().cpu(), r32[k].double()
    rms = float(ref.pow(2).mean().sqrt())
    ea, eb = float((a - ref).pow(2).mean().sqrt()) / rms, float((b - ref).pow(2).mean().sqrt()) / rms
    print(f"{k:8s} {str(tuple(ref.shape)):22s} {rms:9.2e} {ea:12.2e} {eb:12.2e} {ea / max(eb, 1e-30):6.1f} "
          f"{int(((a > 0) != (ref > 0)).sum()):10d} {int(((b > 0) != (ref > 0)).sum()):10d}")
