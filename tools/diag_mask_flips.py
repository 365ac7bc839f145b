import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from golden_util import load, build_oracle_states
from test_networks_gpu import build_modules
from musicgan_amd.networks import engine
from oracle import progan as O
import torch.nn.functional as F
case = sys.argv[1] if len(sys.argv) > 1 else "l2_rc16_gpnorm3"
g = load(f"progan_{case}.npz"); alpha = float(g["alpha"])
x_real, z, eps = (torch.from_numpy(g[k]) for k in ("x_real", "z", "eps"))
acts = {}
for mode in ("auto", "wino"):
    if mode == "wino":
        os.environ["MG_WINO_MIN_PIXELS"] = "1"; os.environ["MG_WINO_WGRAD_MIN_PIXELS"] = "1"
    gen, disc = build_modules(g)
    with torch.no_grad():
        xf = gen(z.cuda(), alpha)
        xi = eps.cuda() * x_real.cuda() + (1 - eps.cuda()) * xf
        xcat = torch.cat([x_real.cuda(), xf, xi]).contiguous()
        out, ctx = engine.disc_forward(disc._weights(), xcat, alpha, disc._pack_cache, save=True)
    x, h0, saved, xp, o, flat, _ = ctx
    acts[mode] = dict(h0=h0.cpu(), o=o.cpu(), **{f"a1_{i}": s[1].cpu() for i, s in enumerate(saved)}, **{f"a2_{i}": s[3].cpu() for i, s in enumerate(saved)})
    xc = xcat.cpu()
# fp64 activations of the first layers
gs, ds = build_oracle_states(g)
p = {k: v.double() for k, v in ds.params.items()}
xd = xc.double()
h0_64 = F.conv2d(xd, p["_Discriminator__start_block.0.weight"], p["_Discriminator__start_block.0.bias"])
pre = f"_Discriminator__conv_blocks.{ds.curr_layer}."
a1_64 = F.conv2d(F.leaky_relu(h0_64, 0.2), p[pre + "0.weight"], p[pre + "0.bias"], padding=1)
for k in acts["auto"]:
    a, b = acts["auto"][k], acts["wino"][k]
    flips = ((a > 0) != (b > 0))
    print(k, tuple(a.shape), "sign flips auto vs wino:", int(flips.sum()), "values:", a[flips].tolist()[:4], b[flips].tolist()[:4], "rms", float(a.pow(2).mean().sqrt()))
for name, ref in (("h0", h0_64), ("a1_0", a1_64)):
    for mode in acts:
        a = acts[mode][name]
        fl = ((a > 0) != (ref > 0))
        print(name, mode, "flips vs fp64 pre-activation:", int(fl.sum()), "fp64 values there:", ref[fl].tolist()[:5], "min|pre|", float(ref.abs().min()))
