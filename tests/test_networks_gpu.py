"""GPU parity of the drop-in modules (Generator / Discriminator / gradient_penalty / FusedAdam) driven exactly like
/root/reference/music_gan/train.py:152-214, against (a) the golden vectors captured from the reference and (b) the CPU
oracle in fp64.  Tolerances are the ones stated in SURVEY 8(c) / BASELINE.md:
  forward <= 1e-5 max-norm relative; losses/GP 1e-6 abs (GP 1e-5: it is ~10 with a 1e-6 relative error);
  gradients per tensor max|d| <= 1e-3 * max|g| against the fp64 oracle; post-Adam weights 2e-6.
"""
import numpy as np
import pytest
import torch

from golden_util import GRAD_TOL, PROGAN_CASES, build_oracle_states, check_tensor, grad_atol, load, maxabs_err, sample_idx

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

FWD_TOL = 1e-5


def build_modules(g):
    from musicgan_amd.networks import Discriminator, Generator
    torch.manual_seed(int(g["seed"]))
    gen = Generator(int(g["rand_channels"]), end_layer=int(g["g_end_layer"]))
    disc = Discriminator(start_layer=int(g["d_start_layer"]))
    for _ in range(int(g["n_grow"])):
        gen.next_layer()
        disc.next_layer()
    ws = float(g["wscale"])
    if ws != 1.0:
        with torch.no_grad():
            for net in (gen, disc):
                for k, p in net.named_parameters():
                    if k.endswith("weight"):
                        p.mul_(ws)
    return gen.to(DEV), disc.to(DEV)


def maxrel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(params=["auto", "wino_everywhere"])
def conv_mode(request, monkeypatch):
    """The engine picks Winograd F(2x2,3x3) only where there are enough 2x2 tiles to fill the chip; "wino_everywhere" forces it
    onto every 3x3 convolution and weight gradient it supports (any even size) so the small golden cases exercise it end to end."""
    if request.param != "auto":
        monkeypatch.setenv("MG_WINO_MIN_PIXELS", "1")
        # "everywhere" includes the <= 8x8 maps that mg_conv3x3_small otherwise takes first ("auto" covers that kernel)
        monkeypatch.setenv("MG_SMALLCONV", "0")
        monkeypatch.setenv("MG_WINO_WGRAD_MIN_PIXELS", "1")
    return request.param


@pytest.mark.parametrize("case", PROGAN_CASES)
def test_train_step_matches_reference_golden(case, conv_mode):
    from musicgan_amd import networks
    from musicgan_amd.optim import FusedAdam
    from oracle import progan as O

    g = load(f"progan_{case}.npz")
    gen, disc = build_modules(g)
    assert list(gen.state_dict().keys()) == list(g["g_keys"])
    assert list(disc.state_dict().keys()) == list(g["d_keys"])
    alpha = float(g["alpha"])
    z, z2 = torch.from_numpy(g["z"]).to(DEV), torch.from_numpy(g["z2"]).to(DEV)
    x_real, eps = torch.from_numpy(g["x_real"]).to(DEV), torch.from_numpy(g["eps"]).to(DEV)
    optim_gen = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
    optim_disc = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))

    # fp64 oracle of the same step (the gradient yard-stick)
    gs, ds = build_oracle_states(g)
    oargs = (torch.from_numpy(g["x_real"]), torch.from_numpy(g["z"]), torch.from_numpy(g["eps"]), alpha)
    o64 = O.d_step(gs, ds, *oargs, dtype=torch.float64)
    o32 = O.d_step(gs, ds, *oargs, dtype=torch.float32)
    terms = O.real_term_grads(ds, oargs[0], alpha)

    def grad_tol(k, which="d_grads"):
        # 1e-3 * max|g| vs fp64, or twice the plain-PyTorch fp32 evaluation's own deviation from fp64
        # (the penalty's gradient cancels heavily when ||grad_x D|| << 1)
        return max(GRAD_TOL, 2.0 * maxrel(o32[which][k], o64[which][k]))

    # ---- D step exactly as train.py:152-175 (x_fake NOT detached)
    x_fake = gen(z, alpha)
    out_real = disc(x_real, alpha)
    out_fake = disc(x_fake, alpha)
    d_loss = networks.wasserstein_discriminator_loss(out_real, out_fake)
    grad_pen = disc.gradient_penalty_with_eps(x_real, x_fake, alpha, eps)
    gen.zero_grad()
    disc.zero_grad()
    (d_loss + grad_pen).backward()

    assert maxrel(x_fake, torch.from_numpy(g["x_fake"])) <= FWD_TOL
    assert maxrel(out_real, torch.from_numpy(g["out_real"])) <= 2 * FWD_TOL
    assert maxrel(out_fake, torch.from_numpy(g["out_fake"])) <= 2 * FWD_TOL
    # the loss is a difference of two output means, each carrying the forward tolerance
    out_scale = max(float(np.abs(g["out_real"]).max()), float(np.abs(g["out_fake"]).max()))
    assert abs(float(d_loss) - float(g["disc_loss"])) <= 1e-6 + 2 * FWD_TOL * out_scale
    gp_tol = 1e-5 * max(1.0, float(g["grad_pen"]) / 10.0)  # 1e-6 relative to the penalty of a fresh critic (10)
    assert abs(float(grad_pen) - float(g["grad_pen"])) <= gp_tol
    assert abs(float(grad_pen) - float(o64["grad_pen"])) <= gp_tol

    live = {k: p for k, p in disc.named_parameters() if p.grad is not None}
    assert sorted(live.keys()) == sorted(g["dstep_d_live"])
    worst = 0.0
    for k, p in live.items():
        e = maxrel(p.grad, o64["d_grads"][k])
        worst = max(worst, e)
        atol = grad_atol(k, o64["d_grads"], o32["d_grads"], terms)
        assert maxabs_err(p.grad, o64["d_grads"][k]) <= atol, f"D grad {k}: rel {e:.3e} vs fp64 oracle"
        own = max(float(o64["d_grads"][k].abs().max()), 1e-30)
        check_tensor(g, f"dstep_dgrad|{k}", p.grad, 2 * max(grad_tol(k), atol / own), what="golden ")
    # G receives (later discarded) gradients in the reference's D step; ours match those too
    live_g = {k: p for k, p in gen.named_parameters() if p.grad is not None}
    assert sorted(live_g.keys()) == sorted(g["dstep_g_live"])
    for k, p in live_g.items():
        assert maxrel(p.grad, o64["g_grads"][k]) <= grad_tol(k, "g_grads"), f"G grad (D step) {k}"

    before = {k: (p.detach().clone(), None if p.grad is None else p.grad.detach().clone())
              for k, p in disc.named_parameters()}
    optim_disc.step()
    adam_tight = adam_total = 0
    for k, p in disc.named_parameters():
        w0, gr = before[k]
        if gr is None:
            assert torch.equal(p.detach(), w0), f"{k}: parameter without gradient must not move"
            continue
        # (a) the fused kernel is torch.optim.Adam's update rule applied to OUR gradient, to fp32 round-off
        exp, _, _ = O.adam_update(w0.double().cpu(), gr.double().cpu(), torch.zeros_like(w0).double().cpu(),
                                  torch.zeros_like(w0).double().cpu(), 1)
        assert maxrel(p, exp) <= 2e-6, f"Adam kernel {k}"
        # (b) vs the reference's post-Adam weights, element by element: the first step is w - lr*g/(|g|+1e-8), so a gradient
        # error dg moves the result by lr*dg/(|g|+1e-8) -- the gradient budget of this tensor carried through the update rule
        # (tight wherever |g| is above the budget, i.e. for nearly every entry; a wrong bias correction would scale every step)
        samp, gsamp = g[f"dstep_dparam|{k}|samp"], g[f"dstep_dgrad|{k}|samp"]
        got = p.detach().cpu().reshape(-1).numpy()[sample_idx(p.numel())]
        budget = 2 * grad_atol(k, o64["d_grads"], o32["d_grads"], terms)  # ours and the reference's fp32 gradient, both vs fp64
        tol = 1e-3 * np.minimum(2.0, budget / (np.abs(gsamp) + 1e-8)) + 2e-7
        assert np.all(np.abs(got - samp) <= tol), f"post-Adam {k}: {np.abs(got - samp).max():.2e}"
        adam_tight += int(np.sum(tol < 0.1 * 1e-3))
        adam_total += tol.size

    # the element-wise bound must bite: tighter than 10% of an Adam step on a good share of the compared weights (it is slack only
    # where |g| is below the gradient budget -- there Adam's normalisation lets ANY fp32 evaluation move by up to a full step)
    assert adam_tight > 0.15 * adam_total, f"post-Adam comparison is vacuous: {adam_tight} of {adam_total}"

    # ---- G step, train.py:191-214, against the critic AS THE REFERENCE UPDATED IT: the first Adam step is lr*sign(g) for every
    # entry, so where |g| is below round-off our critic may legitimately sit 2*lr away from the reference's (bounded above); the
    # G step is compared from the reference's state -- the plain-PyTorch fp32 oracle's post-Adam critic, which the CPU suite pins
    # on the golden post-Adam weights to 2e-6 (test_oracle_steps_match_reference)
    with torch.no_grad():
        for k, p in disc.named_parameters():
            if k in o32["d_grads"]:
                w0 = ds.params[k]
                new_w, _, _ = O.adam_update(w0, o32["d_grads"][k], torch.zeros_like(w0), torch.zeros_like(w0), 1)
                p.copy_(new_w.to(DEV))
                torch.autograd.graph.increment_version(p)
    x_fake2 = gen(z2, alpha)
    out_fake2 = disc(x_fake2, alpha)
    g_loss = networks.wasserstein_generator_loss(out_fake2)
    gen.zero_grad()
    disc.zero_grad()
    g_loss.backward()
    assert maxrel(x_fake2, torch.from_numpy(g["x_fake2"])) <= FWD_TOL
    # a mean of critic scores: the forward tolerance times their scale (scores reach +-80 in the scaled-weight fixtures)
    assert abs(float(g_loss) - float(g["gen_loss"])) <= 1e-6 + FWD_TOL * float(np.abs(g["out_fake2"]).max())
    for k, p in gen.named_parameters():
        if p.grad is not None:
            check_tensor(g, f"gstep_ggrad|{k}", p.grad, GRAD_TOL, what="golden ")
    optim_gen.step()
    for k, p in gen.named_parameters():
        samp = g[f"gstep_gparam|{k}|samp"]
        got = p.detach().cpu().reshape(-1).numpy()[sample_idx(p.numel())]
        if f"gstep_ggrad|{k}|samp" not in g.files:
            assert np.array_equal(got, samp), f"{k}: parameter without gradient must not move"
            continue
        gsamp = g[f"gstep_ggrad|{k}|samp"]
        budget = 2 * GRAD_TOL * float(g[f"gstep_ggrad|{k}|maxabs"])
        tol = 1e-3 * np.minimum(2.0, budget / (np.abs(gsamp) + 1e-8)) + 2e-7
        assert np.all(np.abs(got - samp) <= tol), f"post-Adam {k}: {np.abs(got - samp).max():.2e}"
    print(f"{case}: worst D-grad rel err vs fp64 {worst:.2e}")


def test_shapes_walk_and_growth_flags():
    """The walk of the reference's networks/test_networks.py:4-38 (shapes at every level; growing flags)."""
    from musicgan_amd.networks import Discriminator, Generator
    g = load("progan_shapes.npz")
    torch.manual_seed(5)
    gen, disc = Generator(8).to(DEV), Discriminator(7).to(DEV)
    for i in range(gen.down_sample + 3):
        z = torch.randn(1, 8, 2, 2, device=DEV)
        with torch.no_grad():
            out = gen(z, 0.5)
            dout = disc(out, 0.5)
        assert list(out.shape) == list(g["g_out_shapes"][i])
        assert list(dout.shape) == list(g["d_out_shapes"][i])
        assert [int(gen.growing), int(disc.growing)] == list(g["growing"][i])
        gen.next_layer()
        disc.next_layer()
    assert list(gen.state_dict().keys()) == list(g["g_keys_final"])
    assert list(disc.state_dict().keys()) == list(g["d_keys_final"])
    assert next(gen.end_block_params()).is_cuda and next(disc.start_block_parameters()).is_cuda


def test_nonsquare_generator_forward(conv_mode):
    """generate.py:47-54 feeds a non-square latent through a directly constructed Generator(end_layer=k)."""
    from musicgan_amd.networks import Generator
    g = load("progan_shapes.npz")
    torch.manual_seed(6)
    gen = Generator(8, end_layer=2).to(DEV)
    z = torch.from_numpy(g["ns_z"]).to(DEV)
    with torch.no_grad():
        assert maxrel(gen(z, 1.0), torch.from_numpy(g["ns_out"])) <= FWD_TOL
        assert maxrel(gen(z, 0.37), torch.from_numpy(g["ns_out_a037"])) <= FWD_TOL


def test_level4_step_against_oracle(conv_mode):
    """BASELINE.json configs[1] shape family (2x64x64) at a small batch: product vs the fp64 CPU oracle."""
    from musicgan_amd import networks
    from musicgan_amd.networks import Discriminator, Generator
    from oracle import progan as O
    torch.manual_seed(0)
    gs, ds = O.GenState(32), O.DiscState(7)
    for _ in range(4):
        gs.next_layer()
        ds.next_layer()
    torch.manual_seed(0)
    gen, disc = Generator(32), Discriminator(7)
    for _ in range(4):
        gen.next_layer()
        disc.next_layer()
    gen, disc = gen.to(DEV), disc.to(DEV)
    rng = torch.Generator().manual_seed(1234)
    n = 4
    x_real = torch.rand(n, 2, 64, 64, generator=rng) * 2 - 1
    z = torch.randn(n, 32, 2, 2, generator=rng)
    eps = torch.rand(n, 1, 1, 1, generator=rng)
    ref = O.d_step(gs, ds, x_real, z, eps, 0.5, dtype=torch.float64, detach_fake=True)
    ref32 = O.d_step(gs, ds, x_real, z, eps, 0.5, dtype=torch.float32, detach_fake=True)
    terms = O.real_term_grads(ds, x_real, 0.5)
    x_fake = gen(z.to(DEV), 0.5).detach()
    out_real = disc(x_real.to(DEV), 0.5)
    out_fake = disc(x_fake, 0.5)
    loss = networks.wasserstein_discriminator_loss(out_real, out_fake) + \
        disc.gradient_penalty_with_eps(x_real.to(DEV), x_fake, 0.5, eps.to(DEV))
    disc.zero_grad()
    loss.backward()
    assert maxrel(x_fake, ref["x_fake"]) <= FWD_TOL
    for k, p in disc.named_parameters():
        if p.grad is not None:
            atol = grad_atol(k, ref["d_grads"], ref32["d_grads"], terms)
            assert maxabs_err(p.grad, ref["d_grads"][k]) <= atol, f"{k}: rel {maxrel(p.grad, ref['d_grads'][k]):.2e}"
    assert all(p.grad is None for p in gen.parameters())



_ORACLE_CACHE = {}


def _oracle_full_step(level, batch):
    """fp64 / fp32 CPU oracle of one D step + one G step (critic not yet updated) at a BASELINE.json size; cached across the
    kernel-mode parametrisation (L4 batch 32 in fp64 with the double backward takes a few seconds on the box's 16 cores)."""
    key = (level, batch)
    if key not in _ORACLE_CACHE:
        import bench
        from oracle import progan as O
        torch.set_num_threads(bench.host_cpu_share())
        torch.manual_seed(0)
        gs, ds = O.GenState(32), O.DiscState(7)
        for _ in range(level):
            gs.next_layer()
            ds.next_layer()
        side = bench.LEVEL_SIDE[level]
        rng = torch.Generator().manual_seed(1234)
        x_real = torch.rand(batch, 2, side, side, generator=rng) * 2 - 1
        z = torch.randn(batch, 32, 2, 2, generator=rng)
        z2 = torch.randn(batch, 32, 2, 2, generator=rng)
        eps = torch.rand(batch, 1, 1, 1, generator=rng)
        res = {"inputs": (x_real, z, z2, eps), "terms": O.real_term_grads(ds, x_real, 0.5)}
        for name, dt in (("64", torch.float64), ("32", torch.float32)):
            res["d" + name] = O.d_step(gs, ds, x_real, z, eps, 0.5, dtype=dt, detach_fake=True)
            res["g" + name] = O.g_step(gs, ds, z2, 0.5, dtype=dt)
        _ORACLE_CACHE[key] = res
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("level,batch", [(5, 2), (4, 32), (6, 1), (7, 1), (5, 16), (3, 8), (6, 6)])
def test_full_size_step_against_fp64_oracle(level, batch, conv_mode):
    """BASELINE.json's shapes against the ORACLE (not against another HIP path): the headline level 5 (2x128x128; batch 2 -- the
    critic then runs on 6 images, every kernel on the tiling it uses at batch 64 -- and batch 16: 48 images through the critic),
    configs[1] in full (level 4, 2x64x64, batch 32), and the levels the reference trains at (6 and 7, 2x256x256 / 2x512x512,
    one image: the few-channel tilings that only exist there -- wino3x3_mfma<1,1,2> / <1,3,2>, wino3x3_strip, the narrow
    weight-gradient kernels; reference shapes generator.py:67-76, discriminator.py:60-70), through the product's training path -- ProGANStepper's fused critic step and its generator step: generated
    images, critic scores' means, both losses, the penalty, every critic gradient and every generator gradient."""
    import bench
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper
    ref = _oracle_full_step(level, batch)
    x_real, z, z2, eps = (t.to(DEV) for t in ref["inputs"])
    gen, disc = bench.build_nets(level, 32, DEV)
    og = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
    od = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))
    od.step = lambda *a, **k: None  # keep the gradients observable and the critic un-updated for the G step, as in the oracle
    og.step = lambda *a, **k: None
    st = ProGANStepper(gen, disc, og, od, 32)
    with torch.no_grad():
        assert maxrel(gen(z, 0.5), ref["d64"]["x_fake"]) <= FWD_TOL
    m = st.d_step(x_real, 0.5, z=z, eps=eps)
    d64, d32 = ref["d64"], ref["d32"]
    out_scale = float(d64["out_real"].abs().max())
    assert abs(float(m["disc_loss"]) - float(d64["disc_loss"])) <= 1e-6 + 2 * FWD_TOL * out_scale
    assert abs(float(m["out_real_mean"]) - float(d64["out_real"].mean())) <= 1e-6 + FWD_TOL * out_scale
    assert abs(float(m["out_fake_mean"]) - float(d64["out_fake"].mean())) <= 1e-6 + FWD_TOL * out_scale
    assert abs(float(m["grad_pen"]) - float(d64["grad_pen"])) <= 1e-5
    live = {k: p.grad for k, p in disc.named_parameters() if p.grad is not None}
    assert sorted(live) == sorted(d64["d_grads"])
    worst = 0.0
    for k, gr in live.items():
        atol = grad_atol(k, d64["d_grads"], d32["d_grads"], ref["terms"])
        e = maxabs_err(gr, d64["d_grads"][k])
        worst = max(worst, e / atol)
        assert e <= atol, f"D grad {k}: {e:.3e} > {atol:.3e} (own max {float(d64['d_grads'][k].abs().max()):.2e})"
    mg = st.g_step(batch, 0.5, DEV, z=z2)
    g64, g32 = ref["g64"], ref["g32"]
    assert abs(float(mg["gen_loss"]) - float(g64["gen_loss"])) <= 1e-6 + FWD_TOL * float(g64["out_fake"].abs().max())
    live = {k: p.grad for k, p in gen.named_parameters() if p.grad is not None}
    assert sorted(live) == sorted(g64["g_grads"])
    for k, gr in live.items():
        atol = grad_atol(k, g64["g_grads"], g32["g_grads"])
        e = maxabs_err(gr, g64["g_grads"][k])
        worst = max(worst, e / atol)
        assert e <= atol, f"G grad {k}: {e:.3e} > {atol:.3e}"
    assert all(p.grad is None for p in disc.parameters())  # the G step leaves no critic gradient behind (train.py:209)
    print(f"level {level} batch {batch} [{conv_mode}]: worst gradient error / budget {worst:.2f}")


@pytest.mark.parametrize("case", ["l1_rc8_fade", "l3_rc32_fade", "l2_direct", "l2_rc16_gpnorm1", "l2_rc16_gpnorm3"])
def test_fused_d_step_equals_module_path(case, conv_mode):
    """ProGANStepper's fused critic step (one batched pass over [real|fake|interpolated], in-place tangent pass, one wgrad
    launch per layer) produces the gradients of the reference-shaped module path and of the fp64 oracle."""
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper
    from oracle import progan as O
    g = load(f"progan_{case}.npz")
    alpha = float(g["alpha"])
    z = torch.from_numpy(g["z"]).to(DEV)
    x_real, eps = torch.from_numpy(g["x_real"]).to(DEV), torch.from_numpy(g["eps"]).to(DEV)
    gs, ds = build_oracle_states(g)
    oargs = (torch.from_numpy(g["x_real"]), torch.from_numpy(g["z"]), torch.from_numpy(g["eps"]), alpha)
    o64 = O.d_step(gs, ds, *oargs, dtype=torch.float64, detach_fake=True)
    o32 = O.d_step(gs, ds, *oargs, dtype=torch.float32, detach_fake=True)
    terms = O.real_term_grads(ds, oargs[0], alpha)
    grads = {}
    for fused in (False, True):
        gen, disc = build_modules(g)
        og = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
        od = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))
        od.step = lambda *a, **k: None  # keep the gradients observable: no update
        st = ProGANStepper(gen, disc, og, od, int(g["rand_channels"]), fused_d_step=fused)
        m = st.d_step(x_real, alpha, z=z, eps=eps)
        assert abs(float(m["disc_loss"]) - float(o64["disc_loss"])) <= 1e-6 + 2e-5 * float(np.abs(g["out_real"]).max())
        assert abs(float(m["grad_pen"]) - float(o64["grad_pen"])) <= 1e-5 * max(1.0, float(o64["grad_pen"]) / 10.0)
        grads[fused] = {k: p.grad.detach().clone() for k, p in disc.named_parameters() if p.grad is not None}
        assert all(p.grad is None for p in gen.parameters())
    assert sorted(grads[True].keys()) == sorted(grads[False].keys()) == sorted(o64["d_grads"].keys())
    for k, ref in o64["d_grads"].items():
        if float(ref.abs().max()) < 1e-12:
            # clf bias: d/db of -(mean D(real) - mean D(fake)) is exactly -1 + 1 = 0; fp32 leaves one rounding of 1/N sums
            assert float(grads[True][k].abs().max()) <= 1e-6 and float(grads[False][k].abs().max()) <= 1e-6
            continue
        atol = grad_atol(k, o64["d_grads"], o32["d_grads"], terms)
        assert maxabs_err(grads[True][k], ref) <= atol, f"fused {k}: rel {maxrel(grads[True][k], ref):.2e}"
        assert maxabs_err(grads[True][k], grads[False][k]) <= 2 * atol, f"fused vs module {k}"


@pytest.mark.parametrize("level,batch", [(5, 64), (6, 8), (7, 3)])
def test_full_size_step_is_algorithm_independent(monkeypatch, level, batch):
    """BASELINE.json's headline configuration (level 5, 2x128x128, batch 64) is too large for the CPU oracle inside a test, so
    the size-independent property is used: the critic and generator gradients of one full-size D step and G step must not depend
    on WHICH convolution algorithm computed them -- Winograd F(2x2,3x3)/F(3x3,2x2) + sub-pixel kernels (the product path)
    versus the direct implicit-GEMM kernels (validated against the oracle at the small sizes above).  The two paths share no
    multiply order, so agreement to fp32 round-off at full size checks indexing, tiling, halo handling and split-K at the sizes
    the benchmark runs.  Levels 6 and 7 (256 x 256 and the final 512 x 512 maps, odd batch) cover the last two growth
    steps of a full training run, which no fixture reaches."""
    import bench
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper

    side = bench.LEVEL_SIDE[level]
    grads, terms = {}, {}
    for mode in ("product", "direct"):
        if mode == "direct":
            monkeypatch.setenv("MG_WINO", "0")
            monkeypatch.setenv("MG_WINO_WGRAD", "0")
            monkeypatch.setenv("MG_UPCONV_DGRAD", "0")
        gen, disc = bench.build_nets(level, 32, DEV)
        og = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
        od = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))
        od.step = lambda *a, **k: None  # keep the gradients observable: no update
        og.step = lambda *a, **k: None
        st = ProGANStepper(gen, disc, og, od, 32)
        rng = torch.Generator(device=DEV).manual_seed(1234)
        x_real = torch.rand(batch, 2, side, side, device=DEV, generator=rng) * 2 - 1
        z = torch.randn(batch, 32, 2, 2, device=DEV, generator=rng)
        eps = torch.rand(batch, 1, 1, 1, device=DEV, generator=rng)
        md = st.d_step(x_real, 0.5, z=z, eps=eps)
        gd = {"D." + k: p.grad.detach().clone() for k, p in disc.named_parameters() if p.grad is not None}
        mg = st.g_step(batch, 0.5, DEV, z=z)
        gg = {"G." + k: p.grad.detach().clone() for k, p in gen.named_parameters() if p.grad is not None}
        # un-cancelled scale of every critic gradient: d mean(D(x_real)) / dw alone.  At init the real and fake terms of the
        # Wasserstein loss nearly cancel in the deep blocks (own max 7e-9 against terms of 3e-3 at level 6), so fp32 round-off
        # of the TERMS -- measured 1e-7 of them -- is the floor of any comparison of the residue.
        disc.zero_grad()
        disc(x_real, 0.5).mean().backward()
        terms[mode] = {"D." + k: p.grad.detach().clone() for k, p in disc.named_parameters() if p.grad is not None}
        grads[mode] = ({**gd, **gg}, float(md["disc_loss"]), float(md["grad_pen"]), float(mg["gen_loss"]))
    (ga, la, pa, qa), (gb, lb, pb, qb) = grads["product"], grads["direct"]
    for k, t in terms["direct"].items():
        assert maxabs_err(terms["product"][k], t) <= 2e-4 * float(t.abs().max()), k
    assert ga.keys() == gb.keys() and len(ga) > 40
    assert abs(la - lb) <= 1e-5 * max(1.0, abs(lb)) and abs(pa - pb) <= 1e-4 * max(1.0, abs(pb)) and abs(qa - qb) <= 1e-5 * max(1.0, abs(qb))
    flips = []  # bias tensors with ONE element off by a LeakyReLU sign flip (below)
    for net in ("D.", "G."):
        gmax = max(float(v.abs().max()) for k, v in gb.items() if k.startswith(net))
        for k in gb:
            if k.startswith(net):
                # per tensor: 1e-3 of its own max-norm (the SURVEY 8(c) gradient gate: two fp32 algorithms each within it of fp64),
                # or (cancellation residues, see grad_atol) 2e-5 of the network's scale or of the tensor's un-cancelled term
                tol = max(1e-3 * float(gb[k].abs().max()), 2e-5 * gmax)
                if k in terms["direct"]:
                    tol = max(tol, 2e-5 * float(terms["direct"][k].abs().max()))
                err = (ga[k] - gb[k]).abs()
                over = int((err > tol).sum())
                # One pre-activation within round-off of zero can take different signs in the two algorithms: its LeakyReLU
                # derivative is then 1 in one and 0.2 in the other, and ONE summand of ONE out-channel's bias gradient changes by
                # 0.8 of itself -- 1 / sqrt(#summands) of a cancelled sum (seen at level 6 batch 8 once the Winograd kernel took the
                # 16x16 x 24-image layers: one element of a deep block's bias gradient 3.3e-5 of its un-cancelled term apart,
                # every other element of every tensor within the tolerance).  ONE such element is accepted
                # -- in the critic only (the cancelled sums are its real - fake terms), and once per configuration.
                flip = net == "D." and ga[k].dim() == 1 and over == 1 and float(err.max()) <= 4 * tol and not flips
                if flip:
                    flips.append(k)
                assert over == 0 or flip, f"{k}: {float(err.max()):.3e} > {tol:.3e} ({over} elements; sign flips accepted so far: {flips})"


@pytest.mark.parametrize("case", ["l1_rc8_fade", "l2_rc16_gpnorm1", "l3_rc32_fade"])
def test_user_code_gradient_penalty_through_autograd_equals_the_module_path(case):
    """VERDICT r03 "missing" #4: a user who writes the penalty like the reference does (discriminator.py:166-184: interpolate,
    `autograd.grad(out, x, grad_outputs=ones, create_graph=True, retain_graph=True)`, norm, `backward()`) instead of calling
    `gradient_penalty`.  `Discriminator.forward` is twice differentiable through `_DiscInputGradFn` (the closed-form second-order
    pass as the backward of the input-gradient node): penalty value and every parameter gradient must equal the module's
    `gradient_penalty_with_eps` (same kernels underneath: tight) and the reference's golden penalty; x receives zero gradient."""
    g = load(f"progan_{case}.npz")
    gen, disc = build_modules(g)
    alpha = float(g["alpha"])
    x_real, eps, z = (torch.from_numpy(g[k]).to(DEV) for k in ("x_real", "eps", "z"))
    with torch.no_grad():
        x_fake = gen(z, alpha)
    # the module's closed form
    disc.zero_grad()
    pen_mod = disc.gradient_penalty_with_eps(x_real, x_fake, alpha, eps)
    pen_mod.backward()
    want = {k: p.grad.detach().clone() for k, p in disc.named_parameters() if p.grad is not None}
    # the reference's own formulation on our modules
    disc.zero_grad()
    x_i = (eps * x_real + (1 - eps) * x_fake).requires_grad_(True)
    out = disc(x_i, alpha)
    grad = torch.autograd.grad(outputs=out, inputs=x_i, grad_outputs=torch.ones_like(out), create_graph=True, retain_graph=True)[0]
    pen = 10.0 * ((grad.reshape(grad.shape[0], -1).norm(2, dim=1) - 1) ** 2).mean()
    pen.backward()
    assert abs(float(pen) - float(pen_mod)) <= 2e-6 * max(1.0, float(pen_mod))
    assert abs(float(pen) - float(g["grad_pen"])) <= 1e-5 * max(1.0, float(g["grad_pen"]) / 10.0)
    got = {k: p.grad for k, p in disc.named_parameters() if p.grad is not None}
    for k, w in want.items():
        if float(w.abs().max()) == 0.0:  # biases: the penalty does not reach them
            assert k not in got or float(got[k].abs().max()) == 0.0, k
            continue
        assert k in got, k
        assert float((got[k] - w).abs().max()) <= 2e-5 * float(w.abs().max()), (k, float((got[k] - w).abs().max()), float(w.abs().max()))
    assert x_i.grad is None or float(x_i.grad.abs().max()) == 0.0


@pytest.mark.parametrize("case", ["l3_rc32_fade", "l2_rc16_gpnorm1"])
def test_opt_in_small_map_chains_compute_the_same_update(case, monkeypatch):
    """MG_SMALLNET=1 (the multi-layer chains of csrc/smallnet.hip: critic tail forward / data gradient / tangent and generator
    head forward / backward as one launch per pass; opt-in because slower) against the fp64 oracle with the same budgets as
    the default path: a fused critic update and a generator update, level-3 (fade-in live) and level-2 cases."""
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper
    from oracle import progan as O
    monkeypatch.setenv("MG_SMALLNET", "1")
    monkeypatch.setenv("MG_GRAPHS", "0")
    g = load(f"progan_{case}.npz")
    alpha = float(g["alpha"])
    z, z2 = torch.from_numpy(g["z"]).to(DEV), torch.from_numpy(g["z2"]).to(DEV)
    x_real, eps = torch.from_numpy(g["x_real"]).to(DEV), torch.from_numpy(g["eps"]).to(DEV)
    gs, ds = build_oracle_states(g)
    oargs = (torch.from_numpy(g["x_real"]), torch.from_numpy(g["z"]), torch.from_numpy(g["eps"]), alpha)
    o64 = O.d_step(gs, ds, *oargs, dtype=torch.float64, detach_fake=True)
    o32 = O.d_step(gs, ds, *oargs, dtype=torch.float32, detach_fake=True)
    terms = O.real_term_grads(ds, oargs[0], alpha)
    gen, disc = build_modules(g)
    from musicgan_amd.networks import engine
    if len(disc._weights().blocks) >= 4:
        assert engine.disc_tail_start(disc._weights(), x_real.shape[2], x_real.shape[3]) is not None  # the chain is really taken
    og = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
    od = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))
    od.step = og.step = lambda *a, **k: None
    st = ProGANStepper(gen, disc, og, od, int(g["rand_channels"]))
    m = st.d_step(x_real, alpha, z=z, eps=eps)
    assert abs(float(m["grad_pen"]) - float(o64["grad_pen"])) <= 1e-5 * max(1.0, float(o64["grad_pen"]) / 10.0)
    for k, ref in o64["d_grads"].items():
        got = dict(disc.named_parameters())[k].grad
        if float(ref.abs().max()) < 1e-12:
            assert float(got.abs().max()) <= 1e-6
            continue
        assert maxabs_err(got, ref) <= grad_atol(k, o64["d_grads"], o32["d_grads"], terms), k
    og64 = O.g_step(gs, ds, torch.from_numpy(g["z2"]), alpha, dtype=torch.float64)
    og32 = O.g_step(gs, ds, torch.from_numpy(g["z2"]), alpha, dtype=torch.float32)
    st.g_step(z2.shape[0], alpha, DEV, z=z2)
    for k, ref in og64["g_grads"].items():
        got = dict(gen.named_parameters())[k].grad
        tol = max(GRAD_TOL, 2.0 * maxrel(og32["g_grads"][k], ref))
        assert maxrel(got, ref) <= tol, (k, maxrel(got, ref), tol)
