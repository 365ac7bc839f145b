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


# ---------------------------------------------------------------- BASELINE config 5 (10-minute track, 103 360 frames)
# Inputs are regenerated from numpy's PCG64 stream (the same bits on every machine) and checked by SHA-256 against the fixture;
# tools/gen_golden.py imports these very functions to feed the reference.
C5_NSAMP = 65521
C5_FRAMES = 103360


def c5_sample_idx(numel: int) -> np.ndarray:
    """Fixed strided subsample for the config-5-size tensors: C5_NSAMP indices with an ODD stride (visits every column)."""
    stride = numel // C5_NSAMP
    stride -= 1 - (stride & 1)
    return np.arange(C5_NSAMP, dtype=np.int64) * max(stride, 1)


def c5_waveform() -> np.ndarray:
    """mono U(-0.5, 0.5) track of 44 100 * 600 samples, seed 7."""
    return np.random.default_rng(7).random(44100 * 600, dtype=np.float32) - np.float32(0.5)


def c5_spectrum(frames: int = C5_FRAMES) -> np.ndarray:
    """complex64 (512, frames) input for stft_to_phase_magn whose bits do not depend on any FFT or math library, shaped like the
    STFT of a stationary signal: bin k holds a fixed phasor A_k plus noise of about its size, advanced by k quarter turns per
    frame (the phase advance of bin k's centre frequency at hop = n_fft / 4).  Quarter turns are exact (swap / negate) and every
    other step is one IEEE float32 add of PCG64 uniforms (seed 705), so every machine builds the same bits -- and the unwrapped
    phase of the odd bins runs to +-1.6e5 rad like a real 10-minute track's (ulp 2^-7), which is what this fixture is for."""
    rng = np.random.default_rng(705)
    a = (rng.random((512, 1, 2), dtype=np.float32) - np.float32(0.5))
    n = (rng.random((512, frames, 2), dtype=np.float32) - np.float32(0.5))
    v = a + n
    re, im = v[..., 0], v[..., 1]
    q = (np.arange(512, dtype=np.int64)[:, None] * np.arange(frames, dtype=np.int64)[None, :]) & 3
    out_re = np.select([q == 0, q == 1, q == 2], [re, -im, -re], im)
    out_im = np.select([q == 0, q == 1, q == 2], [im, re, -im], -re)
    out = np.empty((512, frames), dtype=np.complex64)
    out.real, out.imag = out_re, out_im
    return out


def c5_inverse_input(chunks: int = 40) -> np.ndarray:
    """(chunks, 2, 512, 512) float32 in [-1, 1), seed 706: 20 480 frames for magn_phase_to_wav."""
    return (np.random.default_rng(706).random((chunks, 2, 512, 512), dtype=np.float32) * np.float32(2.0)
            - np.float32(1.0))


def c5_phase_stats(g, case, phase):
    """Deviation of a (201, 512, 512) phase output from the reference's on the fixture's subsample + full rows.

    The codec's phase image is a discontinuous function of its input in two places, so a max-norm bound is meaningless at this
    length and the comparison is distributional:
      * `unwrapped = phi + cumsum` is rounded to float32 at magnitudes up to `unwrapped_maxabs` (1.1e5 rad: ulp 2^-7), so a
        1-ulp difference in atan2 (different libm) moves an output by 0 or by one such ulp;
      * a frame-to-frame difference within an ulp of +-pi wraps to the other sign: the output flips between -1 and +1.
    Returns (fraction within `tight`, max deviation outside flips in units of the ulp bound, number of flips, n)."""
    flat = np.asarray(phase, dtype=np.float32).reshape(-1)
    got = np.concatenate([flat[c5_sample_idx(flat.size)],
                          np.asarray(phase)[[0, 0, 0, -1, -1, -1], [0, 255, 511, 0, 255, 511], :].reshape(-1)])
    ref = np.concatenate([g[f"{case}|phase|samp"], g[f"{case}|phase|rows"].reshape(-1)])
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    rng_ = float(g[f"{case}|delta_max"]) - float(g[f"{case}|delta_min"])
    ulp = float(np.spacing(np.float32(g[f"{case}|unwrapped_maxabs"])))
    bound = 2.0 * ulp / rng_ * 2.0  # two ulps of the unwrapped phase, mapped to the [-1, 1] output range
    flips = d > 1.9
    return float(np.mean(d <= 1e-6)), float(d[~flips].max() / bound), int(flips.sum()), d.size
