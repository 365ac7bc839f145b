"""Shared helpers for reading tests/golden/*.npz (data produced by tools/gen_golden.py from the reference)."""
import hashlib
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FULL_MAX = 4096
NSAMP = 509

# the last two: weights scaled until the per-sample ||grad_x D(x~)|| straddles 1 / sits near 3 (penalty 0.02 / 40.8 instead of the
# ~10 of a fresh critic): (||g|| - 1) of both signs, well-conditioned penalty gradients
PROGAN_CASES = ["l0_rc8", "l1_rc8_fade", "l3_rc32_fade", "l2_direct", "l2_rc16_fade_scaled", "l2_rc16_gpnorm1",
                "l2_rc16_gpnorm3"]


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def sample_idx(numel: int) -> np.ndarray:
    if numel <= FULL_MAX:
        return np.arange(numel)
    return np.arange(NSAMP, dtype=np.int64) * (numel // NSAMP)


def check_tensor(g, name, t, rtol_max, what=""):
    """Compare tensor t with the golden record `name` (subsample + sum + l2); tolerance is max-norm relative."""
    t = t.detach().cpu().to(torch.float32).contiguous().reshape(-1)
    samp = g[name + "|samp"]
    maxabs = float(g[name + "|maxabs"])
    idx = sample_idx(t.numel())
    got = t.numpy()[idx]
    assert got.shape == samp.shape, f"{what}{name}: shape {got.shape} vs {samp.shape}"
    scale = max(maxabs, 1e-30)
    err = float(np.max(np.abs(got - samp))) / scale
    assert err <= rtol_max, f"{what}{name}: max-norm rel err {err:.3e} > {rtol_max:.1e}"
    l2 = float(g[name + "|l2"])
    got_l2 = float(t.double().norm())
    assert abs(got_l2 - l2) <= 10 * rtol_max * max(l2, 1e-30), f"{what}{name}: l2 {got_l2} vs {l2}"
    return err


def build_oracle_states(g):
    """Re-create the reference-initialised nets of a golden case with the oracle's own init (same seed, same order)."""
    from oracle import progan as O

    torch.manual_seed(int(g["seed"]))
    gs = O.GenState(int(g["rand_channels"]), end_layer=int(g["g_end_layer"]))
    ds = O.DiscState(start_layer=int(g["d_start_layer"]))
    for _ in range(int(g["n_grow"])):
        gs.next_layer()
        ds.next_layer()
    ws = float(g["wscale"]) if "wscale" in g.files else 1.0
    if ws != 1.0:
        for st in (gs, ds):
            seen = set()
            for k, v in st.params.items():
                if k.endswith("weight") and id(v) not in seen:
                    seen.add(id(v))
                    v.mul_(ws)
    return gs, ds


def trajectory_inputs(g, it):
    return tuple(torch.from_numpy(g[f"{name}|{it}"]) for name in ("x_real", "z", "z2", "eps"))


GRAD_TOL = 1e-3


def maxabs_err(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def grad_atol(k, ref64, ref32, terms64=None):
    """Absolute error budget of gradient tensor k against the fp64 oracle -- SURVEY 8(c): the larger of
      * 1e-3 of the tensor's own max-norm, and
      * twice the plain-PyTorch fp32 evaluation's own deviation from fp64 on that tensor;
    and, for critic tensors (`terms64` = oracle.real_term_grads), 1e-6 of the max-norm of ONE un-cancelled summand of that
    gradient, d mean(D(x_real))/d theta: the critic gradient is -term_real + term_fake + penalty term, and where those cancel
    (classifier weight and deep-block biases at a fresh init: |g| ~ 1e-6 .. exactly 0 from summands ~ 1e-2) the fp32 round-off
    of the summands (~2e-7 relative, tools/diag_act_noise.py) is all that is left of the tensor -- no fp32 implementation can be
    closer than that to fp64, the CPU one is at 1e-7 .. 4e-7 of the term."""
    tol = max(GRAD_TOL * float(ref64[k].abs().max()), 2.0 * maxabs_err(ref32[k], ref64[k]))
    if terms64 is not None:
        tol = max(tol, 1e-6 * float(terms64[k].abs().max()))
    return tol
