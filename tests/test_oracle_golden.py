"""CPU: the oracle (oracle/progan.py, oracle/audio.py) is pinned against golden vectors captured from the reference."""
import numpy as np
import pytest
import torch

from golden_util import PROGAN_CASES, build_oracle_states, check_tensor, load, sample_idx, sha, trajectory_inputs
from oracle import audio as OA
from oracle import progan as O


@pytest.mark.parametrize("case", PROGAN_CASES)
def test_oracle_init_is_bit_exact(case):
    g = load(f"progan_{case}.npz")
    gs, ds = build_oracle_states(g)
    assert list(gs.params.keys()) == list(g["g_keys"])
    assert list(ds.params.keys()) == list(g["d_keys"])
    assert [str(tuple(v.shape)) for v in gs.params.values()] == list(g["g_shapes"])
    assert [str(tuple(v.shape)) for v in ds.params.values()] == list(g["d_shapes"])
    assert [sha(v) for v in gs.params.values()] == list(g["g_sha"])
    assert [sha(v) for v in ds.params.values()] == list(g["d_sha"])
    assert gs.curr_layer == int(g["g_curr_layer"]) and ds.curr_layer == int(g["d_curr_layer"])


@pytest.mark.parametrize("case", PROGAN_CASES)
def test_oracle_steps_match_reference(case):
    g = load(f"progan_{case}.npz")
    gs, ds = build_oracle_states(g)
    alpha = float(g["alpha"])
    z, z2 = torch.from_numpy(g["z"]), torch.from_numpy(g["z2"])
    x_real, eps = torch.from_numpy(g["x_real"]), torch.from_numpy(g["eps"])
    r = O.d_step(gs, ds, x_real, z, eps, alpha)
    assert torch.allclose(r["x_fake"], torch.from_numpy(g["x_fake"]), rtol=0, atol=2e-6)
    assert torch.allclose(r["out_real"], torch.from_numpy(g["out_real"]), rtol=1e-5, atol=1e-7)
    assert torch.allclose(r["out_fake"], torch.from_numpy(g["out_fake"]), rtol=1e-5, atol=1e-7)
    assert abs(float(r["disc_loss"]) - float(g["disc_loss"])) < 1e-6
    assert abs(float(r["grad_pen"]) - float(g["grad_pen"])) < 1e-5
    assert list(r["d_grads"].keys()) == [k for k in ds.live_keys()]
    assert sorted(r["d_grads"].keys()) == sorted(g["dstep_d_live"])
    assert sorted(r["g_grads"].keys()) == sorted(g["dstep_g_live"])
    for k, v in r["d_grads"].items():
        check_tensor(g, f"dstep_dgrad|{k}", v, 2e-4)
    for k, v in r["g_grads"].items():
        check_tensor(g, f"dstep_ggrad|{k}", v, 2e-4)
    # Adam on D (step 1 for every live tensor), then the G step against the updated critic
    for k in ds.live_keys():
        p = ds.params[k]
        new, _, _ = O.adam_update(p, r["d_grads"][k], torch.zeros_like(p), torch.zeros_like(p), 1)
        p.copy_(new)
    for k, p in ds.params.items():
        check_tensor(g, f"dstep_dparam|{k}", p, 2e-6)
    r2 = O.g_step(gs, ds, z2, alpha)
    assert torch.allclose(r2["x_fake"], torch.from_numpy(g["x_fake2"]), rtol=0, atol=2e-6)
    assert abs(float(r2["gen_loss"]) - float(g["gen_loss"])) < 1e-6
    for k, v in r2["g_grads"].items():
        check_tensor(g, f"gstep_ggrad|{k}", v, 2e-4)
    for k in gs.live_keys():
        p = gs.params[k]
        new, _, _ = O.adam_update(p, r2["g_grads"][k], torch.zeros_like(p), torch.zeros_like(p), 1)
        p.copy_(new)
    for k, p in gs.params.items():
        check_tensor(g, f"gstep_gparam|{k}", p, 2e-6)


def test_oracle_detached_d_step_gives_same_d_grads():
    """Detaching x_fake in the D step (what the product does) leaves every D gradient unchanged."""
    g = load("progan_l1_rc8_fade.npz")
    gs, ds = build_oracle_states(g)
    args = (torch.from_numpy(g["x_real"]), torch.from_numpy(g["z"]), torch.from_numpy(g["eps"]), float(g["alpha"]))
    a = O.d_step(gs, ds, *args, dtype=torch.float64)
    b = O.d_step(gs, ds, *args, dtype=torch.float64, detach_fake=True)
    for k in a["d_grads"]:
        assert torch.allclose(a["d_grads"][k], b["d_grads"][k], rtol=1e-12, atol=1e-15)
    assert len(b["g_grads"]) == 0


def test_oracle_shapes_walk_and_nonsquare():
    g = load("progan_shapes.npz")
    torch.manual_seed(5)
    gs, ds = O.GenState(8), O.DiscState(7)
    for i in range(10):
        z = torch.randn(1, 8, 2, 2)
        out = O.gen_forward(gs.params, gs.curr_layer, gs.has_last, z, 0.5)
        dout = O.disc_forward(ds.params, ds.curr_layer, ds.has_last, out, 0.5)
        assert list(out.shape) == list(g["g_out_shapes"][i])
        assert list(dout.shape) == list(g["d_out_shapes"][i])
        assert [int(gs.growing), int(ds.growing)] == list(g["growing"][i])
        gs.next_layer()
        ds.next_layer()
    assert list(gs.params.keys()) == list(g["g_keys_final"])
    assert list(ds.params.keys()) == list(g["d_keys_final"])
    torch.manual_seed(6)
    g2 = O.GenState(8, end_layer=2)
    assert list(g2.params.keys()) == list(g["ns_keys"])
    assert [sha(v) for v in g2.params.values()] == list(g["ns_sha"])
    z = torch.from_numpy(g["ns_z"])
    y = O.gen_forward(g2.params, 2, True, z, 1.0)
    assert torch.allclose(y, torch.from_numpy(g["ns_out"]), atol=2e-6, rtol=0)
    y = O.gen_forward(g2.params, 2, True, z, 0.37)
    assert torch.allclose(y, torch.from_numpy(g["ns_out_a037"]), atol=2e-6, rtol=0)


def test_gp_regime_fixtures_cover_both_signs():
    """The two scaled-weight fixtures leave the penalty-at-10 regime: per-sample ||grad_x D(x~)|| on both sides of 1 (gpnorm1)
    and well above it (gpnorm3) -- evaluated with the oracle in fp64 on the fixture's inputs."""
    norms = {}
    for case in ("l2_rc16_gpnorm1", "l2_rc16_gpnorm3"):
        g = load(f"progan_{case}.npz")
        gs, ds = build_oracle_states(g)
        x_real, z, eps = (torch.from_numpy(g[k]).double() for k in ("x_real", "z", "eps"))
        gp_, dp_ = O._leafs(gs, torch.float64), O._leafs(ds, torch.float64)
        x_fake = O.gen_forward(gp_, gs.curr_layer, gs.has_last, z, float(g["alpha"])).detach()
        xi = (eps * x_real + (1 - eps) * x_fake).requires_grad_(True)
        (gx,) = torch.autograd.grad(O.disc_forward(dp_, ds.curr_layer, ds.has_last, xi, float(g["alpha"])).sum(), xi)
        norms[case] = gx.reshape(gx.shape[0], -1).norm(dim=1)
        assert abs(float(10 * ((norms[case] - 1) ** 2).mean()) - float(g["grad_pen"])) < 1e-4
    n1, n3 = norms["l2_rc16_gpnorm1"], norms["l2_rc16_gpnorm3"]
    assert float(n1.min()) < 1.0 < float(n1.max()) and 0.5 < float(n1.min()) and float(n1.max()) < 2.0
    assert float(n3.min()) > 1.5


def _run_oracle_trajectory(g, dtype):
    torch.manual_seed(int(g["seed"]))
    tr = O.Trajectory(int(g["rand_channels"]), O.GrowerState(7, g["fadein"].tolist(), g["train_lengths"].tolist()),
                      dtype=dtype, wscale=float(g["wscale"]))
    recs = []
    for it in range(int(g["iters"])):
        x_real, z, z2, eps = trajectory_inputs(g, it)
        recs.append(tr.iteration(x_real, z, eps, z2, growth_seed=int(g["seed"]) + 3000 + tr.gs.curr_layer))
    return tr, recs


def test_oracle_trajectory_matches_reference_loop():
    """16 iterations of the reference's loop body (train.py:131-272: D step every iteration, G step every 5th, Adam with its
    second-moment memory, two growths with add_param_group and the aliased old head/stem) run by the REFERENCE in float64
    (tools/gen_golden.py::trajectory_case) against the oracle's restatement in float64: every loss of every iteration and the
    final weights / Adam moments / per-parameter step counts."""
    g = load("progan_trajectory.npz")
    tr, recs = _run_oracle_trajectory(g, torch.float64)
    assert [r["level"] for r in recs] == g["ref64|level"].astype(int).tolist() == [0] * 6 + [1] * 5 + [2] * 5
    assert [int(r["grew"]) for r in recs] == g["ref64|grew"].astype(int).tolist()
    for it, r in enumerate(recs):
        assert r["alpha"] == pytest.approx(float(g["ref64|alpha"][it]), abs=1e-15)
        for key in ("disc_loss", "grad_pen", "out_real", "out_fake"):
            ref = float(g[f"ref64|{key}"][it])
            assert abs(r[key] - ref) <= 1e-9 * max(1.0, abs(ref)), f"iteration {it} {key}: {r[key]} vs {ref}"
        ref = float(g["ref64|gen_loss"][it])
        assert (it % 5 == 0) == ("gen_loss" in r) == (not np.isnan(ref))
        if it % 5 == 0:
            assert abs(r["gen_loss"] - ref) <= 1e-9 * max(1.0, abs(ref))
    for pre, st, opt in (("g", tr.gs, tr.opt_g), ("d", tr.ds, tr.opt_d)):
        assert list(st.params.keys()) == list(g[f"{pre}_keys"])
        steps = dict(zip(g[f"adam_steps|{pre}|keys"].tolist(), g[f"adam_steps|{pre}"].tolist()))
        for k, p in st.params.items():
            if f"final64|{pre}|{k}|samp" not in g.files:  # aliased key: the reference's named_parameters() lists the object once
                continue
            idx = sample_idx(p.numel())
            assert np.max(np.abs(p.reshape(-1).numpy()[idx] - g[f"final64|{pre}|{k}|samp"])) <= 1e-9, k
            ast = opt.of(p)
            assert (ast["step"] if ast else 0) == steps[k], f"Adam step count of {k}"
            if ast:
                ref = g[f"final64|{pre}|{k}|exp_avg_sq|samp"]
                assert np.max(np.abs(ast["exp_avg_sq"].reshape(-1).numpy()[idx] - ref)) <= 1e-9 * max(1.0, float(ref.max())), k
    # the step counts the loop must produce (three growths: after iterations 5, 10 and -- with no step behind it -- 15): blocks
    # count from the iteration that brought them into the graph, heads / stems from their own growth (added param groups)
    sg = dict(zip(g["adam_steps|g|keys"].tolist(), g["adam_steps|g"].tolist()))
    assert sg["_Generator__gen_blocks.0.0.weight"] == 4 and sg["_Generator__gen_blocks.1.0.weight"] == 2
    assert sg["_Generator__gen_blocks.2.0.weight"] == 1 and sg["_Generator__gen_blocks.3.0.weight"] == 0
    assert sg["_Generator__end_block.0.weight"] == 0 and sg["_Generator__last_end_block.0.0.weight"] == 1
    sd = dict(zip(g["adam_steps|d|keys"].tolist(), g["adam_steps|d"].tolist()))
    assert sd["_Discriminator__conv_blocks.7.0.weight"] == 16 and sd["_Discriminator__conv_blocks.6.0.weight"] == 10
    assert sd["_Discriminator__conv_blocks.5.0.weight"] == 5 and sd["_Discriminator__conv_blocks.4.0.weight"] == 0
    assert sd["_Discriminator__start_block.0.weight"] == 0 and sd["_Discriminator__last_start_block.1.0.weight"] == 5
    assert g["adam_groups|g"].tolist() == [34, 2, 2, 2] and g["adam_groups|d"].tolist() == [40, 2, 2, 2]


def test_oracle_trajectory_fp32_tracks_reference_fp32():
    """Same loop in float32 against the reference's float32 run (as the reference really trains).  Both leave the float64
    trajectory by 1e-9 .. 2e-4 (stored); the oracle must stay within a few times that of the reference's float32 run."""
    g = load("progan_trajectory.npz")
    _, recs = _run_oracle_trajectory(g, torch.float32)
    for key in ("disc_loss", "grad_pen"):
        ref32, ref64 = g[f"ref32|{key}"], g[f"ref64|{key}"]
        own = np.abs(ref32 - ref64)
        for it, r in enumerate(recs):
            tol = 10 * own[it] + 2e-6 * max(1.0, abs(ref64[it]))
            assert abs(r[key] - ref32[it]) <= tol, f"iteration {it} {key}: {r[key]} vs {ref32[it]} (tol {tol:.1e})"


def test_flop_model_matches_survey():
    gf, df = O.flops_per_image(32, 5)
    assert abs(gf / 1e9 - 1.978) < 2e-3 and abs(df / 1e9 - 1.986) < 2e-3
    gf, df = O.flops_per_image(32, 4)
    assert abs(gf / 1e9 - 0.767) < 2e-3 and abs(df / 1e9 - 0.775) < 2e-3


# ------------------------------------------------------------------ audio
def test_audio_oracle_stft_matches_reference():
    g = load("audio_codec.npz")
    c = OA.stft(g["wav"])
    ref = g["stft_real"] + 1j * g["stft_imag"]
    assert c.shape == ref.shape == (512, 553)
    assert np.max(np.abs(c - ref)) <= 1e-5 * np.max(np.abs(ref))
    # explicit DFT identity at a few (bin, frame) pairs incl. the reflect-padded edges
    mono = g["wav"].mean(axis=0)
    for k, t in [(0, 0), (1, 0), (37, 1), (511, 2), (100, 300), (255, 551), (3, 552)]:
        assert abs(OA.dft_bin(mono, k, t) - ref[k, t]) <= 2e-5 * np.max(np.abs(ref))


def test_audio_oracle_codec_matches_reference():
    g = load("audio_codec.npz")
    ref = (g["stft_real"] + 1j * g["stft_imag"]).astype(np.complex64)
    magn, phase = OA.stft_to_phase_magn(ref)
    assert magn.shape == g["magn"].shape == (1, 512, 512)
    assert np.max(np.abs(magn - g["magn"])) <= 2e-6
    # phase goes through atan2 + a long fp32 cumulative sum: allow 1e-4 of the [-1,1] range
    assert np.max(np.abs(phase - g["phase"])) <= 1e-4
    s = OA.bark_scale_vector(512)
    assert np.allclose(s, g["bark_scale"], rtol=1e-6, atol=0)
    assert abs(s[0] - 4.27399e-4) < 1e-8 and abs(s[511] - 5.51122e-2) < 1e-6  # SURVEY 8(c) known answers
    assert np.allclose(OA.unwrap(g["unwrap_in"]), g["unwrap_out"], atol=2e-5)
    assert np.allclose(OA.unwrap(g["unwrap_in"]), np.unwrap(g["unwrap_in"].astype(np.float64), axis=1), atol=3e-5)


def test_audio_oracle_inverse_matches_reference():
    g = load("audio_codec.npz")
    wav = OA.magn_phase_to_wav(g["inv_in"])
    ref = g["inv_wav"].reshape(-1)
    assert wav.shape == ref.shape == (256 * 63,)
    assert np.max(np.abs(wav - ref)) <= 2e-3 * np.max(np.abs(ref))


def _c5(case):
    import hashlib
    from golden_util import c5_inverse_input, c5_spectrum, c5_waveform
    g = load("audio_config5.npz")
    x = {"wav": c5_waveform, "spec": c5_spectrum, "inv": c5_inverse_input}[case]()
    key = "wav|sha256" if case == "wav" else f"{case}|sha256"
    assert hashlib.sha256(np.ascontiguousarray(x).view(np.float32).tobytes()).hexdigest() == str(g[key]), \
        "the regenerated config-5 input is not the one the reference was given"
    return g, x


def _c5_check_magn(g, case, magn, tol):
    from golden_util import c5_sample_idx
    got = magn.reshape(-1)[c5_sample_idx(magn.size)]
    assert float(np.abs(got - g[f"{case}|magn|samp"]).max()) <= tol
    assert float(np.abs(magn[[0, 0, 0, -1, -1, -1], [0, 255, 511, 0, 255, 511], :] - g[f"{case}|magn|rows"]).max()) <= tol
    assert abs(float(magn.astype(np.float64).sum()) - float(g[f"{case}|magn|sum"])) <= tol * magn.size * 0.05


def test_audio_oracle_config5_codec_is_the_references_bit_for_bit():
    """BASELINE config 5's size (10-minute track, 103 360 frames -> 201 images; create_dataset.py:34-64, functions.py:65-94) on
    the library-independent STFT-like input: with torch's own abs / angle the oracle IS the reference, every sampled element and
    both full-row sets, deviation exactly 0 -- this pins `unwrap`'s float64 running sum (functions.py:23, torch.cumsum).  With a
    float32 running sum (the round-1/2 oracle) 0.05 % of the elements are within 1e-6 (tools/diag_unwrap_lengths.py)."""
    from golden_util import c5_phase_stats
    g, x = _c5("spec")
    magn, phase = OA.stft_to_phase_magn(x, lib="torch")
    assert magn.shape == phase.shape == (201, 512, 512)
    frac, worst, flips, n = c5_phase_stats(g, "spec", phase)
    assert (frac, worst, flips) == (1.0, 0.0, 0), (frac, worst, flips, n)
    assert abs(float(phase.astype(np.float64).sum()) - float(g["spec|phase|sum"])) <= 1e-9 * phase.size
    _c5_check_magn(g, "spec", magn, 0.0)   # (lib="torch" covers the bark vector's linspace / arcsinh / norm too)
    # the same with numpy's atan2f / hypotf: 1-ulp library differences, amplified by the exact sum, stay inside the bound
    magn, phase = OA.stft_to_phase_magn(x)
    frac, worst, flips, n = c5_phase_stats(g, "spec", phase)
    assert frac >= 0.90 and worst <= 1.0 and flips <= 3, (frac, worst, flips, n)
    _c5_check_magn(g, "spec", magn, 2e-6)


def test_audio_oracle_config5_waveform_to_codec():
    """The seed-7 U(-0.5, 0.5) 44 100 x 600-sample track of SURVEY 8(d) through the oracle's STFT (float64 FFT) and codec against
    the reference's wav_to_stft + stft_to_phase_magn: the bins agree to 1e-6 of max|X|, which moves the phase of weak bins by more
    than an ulp, so the bound on the phase image is distributional (golden_util.c5_phase_stats)."""
    from golden_util import c5_phase_stats, c5_sample_idx
    g, wav = _c5("wav")
    c = OA.stft(wav)
    assert c.shape == (512, 103360)
    cs = np.ascontiguousarray(c).view(np.float32).reshape(-1)[c5_sample_idx(2 * c.size)]
    assert float(np.abs(cs - g["wav|stft_samp"]).max()) <= 1e-6 * float(g["wav|stft_maxabs"])
    magn, phase = OA.stft_to_phase_magn(c, lib="torch")
    frac, worst, flips, n = c5_phase_stats(g, "wav", phase)
    assert frac >= 0.85 and worst <= 1.0 and flips <= 3, (frac, worst, flips, n)
    _c5_check_magn(g, "wav", magn, 2e-6)


def test_audio_oracle_config5_inverse():
    """magn_phase_to_wav over 20 480 frames (functions.py:97-139; its cumulative phase is a sequential float32 loop, :117-118)."""
    from golden_util import c5_sample_idx
    g, mp = _c5("inv")
    wav = OA.magn_phase_to_wav(mp)
    assert wav.shape == (256 * (20480 - 1),)
    scale = float(g["inv|wav|maxabs"])
    assert float(np.abs(wav[c5_sample_idx(wav.size)] - g["inv|wav|samp"]).max()) <= 2e-3 * scale
    assert float(np.abs(wav[:4096] - g["inv|wav|head"]).max()) <= 2e-3 * scale
    assert float(np.abs(wav[-4096:] - g["inv|wav|tail"]).max()) <= 2e-3 * scale


def test_stft_known_shape_30s():
    # notebook cell 7: 30 s mono at 44.1 kHz -> [513, 5168] before the Nyquist drop
    assert OA.stft(np.zeros(44100 * 30, dtype=np.float32)).shape == (512, 5168)


def test_bark_known_answer_from_the_notebook():
    # notebook cell 35: bark(22050 Hz) = 6 asinh(22050 / 600) = 25.7848 -- the un-normalised last entry of the scale vector
    assert abs(6.0 * np.arcsinh(22050.0 / 600.0) - 25.7848) < 1e-4
    s = OA.bark_scale_vector(512)
    raw = 6.0 * np.arcsinh(np.linspace(20.0, 22050.0, 512) / 600.0)
    assert np.allclose(s, raw / np.linalg.norm(raw), rtol=1e-6) and abs(raw[-1] - 25.7848) < 1e-4
