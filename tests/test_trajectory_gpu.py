"""GPU: the stateful parts of the reference's training loop (/root/reference/music_gan/train.py:131-272) on the product path --
ProGANStepper (fused critic step) + FusedAdam + Grower driven through the 16-iteration trajectory the REFERENCE produced
(tests/golden/progan_trajectory.npz: D step every iteration, G step every 5th, Adam(beta1=0) second-moment memory over up to 16
steps, three growths with add_param_group and the aliased previous head / stem), and the fused Adam kernel against torch.optim.Adam
over several steps with carried state, late param groups, skipped parameters and the data-parallel gradient scale."""
import numpy as np
import pytest
import torch

from golden_util import load, sample_idx, trajectory_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LR, BETAS = 1e-3, (0.0, 0.9)


@pytest.fixture(params=["auto", "wino_everywhere", "module_path"])
def mode(request, monkeypatch):
    if request.param == "wino_everywhere":
        monkeypatch.setenv("MG_WINO_MIN_PIXELS", "1")
        # "everywhere" includes the <= 8x8 maps that mg_conv3x3_small otherwise takes first ("auto" covers that kernel)
        monkeypatch.setenv("MG_SMALLCONV", "0")
        monkeypatch.setenv("MG_WINO_WGRAD_MIN_PIXELS", "1")
    return request.param


FREE_RUN_WINDOW = 14  # iterations 0..13; see the docstring below


def _grow(seed, gen, disc, og, od):
    torch.manual_seed(seed + 3000 + gen.curr_layer)  # train.py:258-272; the fixture seeds the fresh head / stem the same way
    gen.next_layer()
    disc.next_layer()
    og.add_param_group({"params": gen.end_block_params(), "lr": LR, "betas": BETAS})
    od.add_param_group({"params": disc.start_block_parameters(), "lr": LR, "betas": BETAS})


def _build(g, fused=True):
    from musicgan_amd.networks import Discriminator, Generator
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper
    from musicgan_amd.utils import Grower
    torch.manual_seed(int(g["seed"]))
    gen, disc = Generator(int(g["rand_channels"]), end_layer=0).to(DEV), Discriminator(start_layer=7).to(DEV)
    og = FusedAdam(gen.parameters(), lr=LR, betas=BETAS)
    od = FusedAdam(disc.parameters(), lr=LR, betas=BETAS)
    stepper = ProGANStepper(gen, disc, og, od, int(g["rand_channels"]), fused_d_step=fused)
    grower = Grower(n_grow=7, fadein_lengths=g["fadein"].tolist(), train_lengths=g["train_lengths"].tolist())
    return gen, disc, og, od, stepper, grower


def test_training_trajectory_matches_reference(mode):
    """FREE-RUNNING against the fixture (no re-synchronisation): every loss of iterations 0..13 -- two growths, 14 critic and 3
    generator updates, Adam memory up to 14 steps deep -- must equal the reference's float64 run to within 20x what the
    reference's own float32 run deviates from it (+1e-5).  Measured (tools/diag_trajectory.py): 3e-6 on the critic loss and 3e-5
    on the penalty, the reference's float32 run 1e-6 / 3e-5.  Iterations 14 and 15 are NOT compared in this mode: the trajectory
    amplifies perturbations there by ~100x per iteration (the reference's own fp32 deviation jumps from 1e-7 to 2e-4 in its last
    step), and Adam(beta1=0)'s first step is lr*sign(g), so every gradient entry whose sign is below round-off (1-2 entries of
    2e6 per step here, at |g| ~ 1e-11 against a tensor maximum of 2e-3) already moves one weight by 2*lr against the reference.
    Those iterations, and the final weights / moments, are checked step by step in the lock-step test below."""
    g = load("progan_trajectory.npz")
    seed, batch = int(g["seed"]), int(g["batch"])
    gen, disc, og, od, stepper, grower = _build(g, fused=(mode != "module_path"))
    worst = {}
    for it in range(int(g["iters"])):
        x_real, z, z2, eps = (t.to(DEV) for t in trajectory_inputs(g, it))
        assert gen.curr_layer == int(g["ref64|level"][it]) and x_real.shape[-1] == 4 * 2 ** gen.curr_layer
        alpha = grower.alpha
        assert alpha == pytest.approx(float(g["ref64|alpha"][it]), abs=1e-12)
        m = stepper.d_step(x_real, alpha, z=z, eps=eps)
        got = {"disc_loss": float(m["disc_loss"]), "grad_pen": float(m["grad_pen"]), "out_real": float(m["out_real_mean"]),
               "out_fake": float(m["out_fake_mean"])}
        if it % 5 == 0:  # train.py:189
            got["gen_loss"] = float(stepper.g_step(batch, alpha, DEV, z=z2)["gen_loss"])
        for key, v in got.items():
            r64, r32 = float(g[f"ref64|{key}"][it]), float(g[f"ref32|{key}"][it])
            assert np.isfinite(v)
            if it < FREE_RUN_WINDOW:
                tol = 20 * abs(r32 - r64) + 1e-5 * max(1.0, abs(r64))
                worst[key] = max(worst.get(key, 0.0), abs(v - r64) / tol)
                assert abs(v - r64) <= tol, f"iteration {it} {key}: {v} vs reference {r64} (its own fp32 run: {r32})"
        grew = grower.grow(batch) and gen.growing
        assert bool(grew) == bool(g["ref64|grew"][it])
        if grew:
            _grow(seed, gen, disc, og, od)
    stepper.finish()
    assert list(gen.state_dict().keys()) == list(g["g_keys"]) and list(disc.state_dict().keys()) == list(g["d_keys"])
    for pre, net, opt in (("g", gen, og), ("d", disc, od)):
        assert [len(gr["params"]) for gr in opt.param_groups] == g[f"adam_groups|{pre}"].tolist()
        steps = dict(zip(g[f"adam_steps|{pre}|keys"].tolist(), g[f"adam_steps|{pre}"].tolist()))
        for k, p in net.named_parameters():
            st = opt.state.get(p, {})
            assert (int(st["step"]) if st else 0) == steps[k], f"Adam step count of {k}"
            if steps[k] == 0:  # a weight the loop never stepped is bit-identical to its same-seed initial value
                got = p.detach().cpu().reshape(-1).numpy()[sample_idx(p.numel())]
                assert np.array_equal(got, g[f"final32|{pre}|{k}|samp"]), f"{k}: untouched parameter moved"
    print(f"{mode}: worst loss deviation / tolerance over iterations 0..{FREE_RUN_WINDOW - 1}: {worst}")


def _sync_from_oracle(tr, gen, disc, og, od):
    """Product state := the fp64 oracle's (weights rounded to float32; Adam second moments and step counts)."""
    for net, st, opt_p, opt_o in ((gen, tr.gs, og, tr.opt_g), (disc, tr.ds, od, tr.opt_d)):
        for k, p in net.named_parameters():
            src = st.params[k]
            p.data.copy_(src.to(torch.float32))
            torch.autograd.graph.increment_version(p)
            ast = opt_o.of(src)
            if ast is not None:  # in place: a captured graph of the update holds the addresses of these tensors
                pst = opt_p._init_state(p)
                pst["step"].fill_(float(ast["step"]))
                pst["step_dev"].fill_(int(ast["step"]))
                pst["exp_avg"].copy_(ast["exp_avg"].to(torch.float32))
                pst["exp_avg_sq"].copy_(ast["exp_avg_sq"].to(torch.float32))


def test_training_loop_in_lock_step_with_fp64_oracle():
    """All 16 iterations (incl. the two the free-running test cannot compare), TEACHER-FORCED: before each iteration the product
    takes over the fp64 oracle's state -- the oracle's whole trajectory is pinned on the reference's float64 run to 1e-9
    (tests/test_oracle_golden.py) -- then runs the loop body on the GPU.  Checked per iteration: losses; every critic /
    generator gradient with the SURVEY 8(c) budget; and every updated weight, element by element, with that gradient budget
    carried through Adam's update (|dw| <= lr * min(2, budget / (sqrt(v_hat) + eps))): tight wherever a weight's gradient history
    is above the budget, so it pins the second-moment memory and both bias corrections at steps 1..16, the skipped generator
    steps, and the step counts of heads / stems added by add_param_group."""
    from golden_util import grad_atol, maxabs_err
    from oracle import progan as O
    g = load("progan_trajectory.npz")
    seed, batch = int(g["seed"]), int(g["batch"])
    gen, disc, og, od, stepper, grower = _build(g)
    torch.manual_seed(seed)
    tr = O.Trajectory(int(g["rand_channels"]), O.GrowerState(7, g["fadein"].tolist(), g["train_lengths"].tolist()),
                      dtype=torch.float64)
    tight = total = 0

    def check_update(net, opt_o, state, grads64, grads32, terms, what):
        nonlocal tight, total
        for k, p in net.named_parameters():
            if k not in grads64:
                continue
            atol = grad_atol(k, grads64, grads32, terms)
            assert maxabs_err(p.grad, grads64[k]) <= atol, f"iteration {it} {what} gradient {k}"
            ast = opt_o.of(state.params[k])
            bc2 = 1.0 - BETAS[1] ** ast["step"]
            denom = ast["exp_avg_sq"].sqrt() / np.sqrt(bc2) + 1e-8
            w64 = state.params[k]
            tol = LR * torch.clamp(atol / denom, max=2.0) + 2e-7 * (1.0 + w64.abs())
            err = (p.detach().double().cpu() - w64).abs()
            assert bool((err <= tol).all()), f"iteration {it} {what} weight {k} after Adam step {ast['step']}: {float((err - tol).max()):.2e} over"
            tight += int((tol < 0.02 * LR).sum())
            total += tol.numel()
            v = opt_p_state(p)["exp_avg_sq"].double().cpu()
            vmax = float(ast["exp_avg_sq"].max())  # v = 0.9 v + 0.1 g^2: a gradient error dg moves it by 0.2 |g| dg
            assert float((v - ast["exp_avg_sq"]).abs().max()) <= 2 * np.sqrt(vmax) * atol + 1e-6 * vmax + 1e-30, f"{k} exp_avg_sq"

    for it in range(int(g["iters"])):
        x_real, z, z2, eps = trajectory_inputs(g, it)
        if it > 0:
            _sync_from_oracle(tr, gen, disc, og, od)
        alpha = grower.alpha
        d32 = O.d_step(tr.gs, tr.ds, x_real, z, eps, alpha, dtype=torch.float32, detach_fake=True)
        terms = O.real_term_grads(tr.ds, x_real, alpha)
        g32 = None
        rec = tr.iteration(x_real, z, eps, z2, defer_growth=True)
        # (the oracle's record was taken with G attached, as the reference runs; D's gradients are the same either way)
        opt_p_state = lambda p: od.state[p]
        m = stepper.d_step(x_real.to(DEV), alpha, z=z.to(DEV), eps=eps.to(DEV))
        assert abs(float(m["disc_loss"]) - rec["disc_loss"]) <= 2e-5 * max(1.0, abs(rec["disc_loss"]))
        assert abs(float(m["grad_pen"]) - rec["grad_pen"]) <= 2e-5 * max(1.0, abs(rec["grad_pen"]))
        if it % 5 == 0:
            # the oracle's G step ran against ITS updated critic: give the product that critic first
            check_update(disc, tr.opt_d, tr.ds, rec["d_grads"], d32["d_grads"], terms, "critic")
            for k, p in disc.named_parameters():
                p.data.copy_(tr.ds.params[k].to(torch.float32))
                torch.autograd.graph.increment_version(p)
            opt_p_state = lambda p: og.state[p]
            mg = stepper.g_step(batch, alpha, DEV, z=z2.to(DEV))
            assert abs(float(mg["gen_loss"]) - rec["gen_loss"]) <= 2e-5 * max(1.0, abs(rec["gen_loss"]))
            # fp32 oracle G step from the same pre-step generator: rebuild it from the oracle's post-step state is not possible,
            # so the generator budget is the plain 1e-3 max-norm rule (no fp32 term)
            check_update(gen, tr.opt_g, tr.gs, rec["g_grads"], rec["g_grads"], None, "generator")
        else:
            check_update(disc, tr.opt_d, tr.ds, rec["d_grads"], d32["d_grads"], terms, "critic")
        grew = grower.grow(batch) and gen.growing
        assert bool(grew) == tr.end_of_iteration(batch, growth_seed=seed + 3000 + tr.gs.curr_layer)
        if grew:
            _grow(seed, gen, disc, og, od)
    # the element-wise weight bound is lr * budget / |g history|: tight for the entries that carry the gradient, slack for the
    # near-zero ones (where Adam's normalisation makes ANY fp32 evaluation move by up to a full step)
    assert tight > 0.25 * total, f"only {tight} of {total} weight comparisons were tighter than 2% of a step"
    print(f"lock-step: {tight} of {total} weight comparisons tighter than 2% of an Adam step")


@pytest.mark.parametrize("grad_scale", [1.0, 0.125])
def test_fused_adam_multi_step_matches_torch_adam(grad_scale):
    """mg_adam_step against torch.optim.Adam (the optimizer the reference uses, train.py:64-70) over 7 steps: second-moment
    memory and both bias corrections at step >= 2, a parameter that skips steps (its counter must not advance), a param group
    added after step 3 (counts from its own first step), gradients spanning 1e-9 .. 1e+1, and `grad_scale` (the 1/world factor of
    the data-parallel path, folded into the kernel) == torch's Adam fed the scaled gradients."""
    from musicgan_amd.optim import FusedAdam
    rng = torch.Generator().manual_seed(5)
    shapes = [(160, 144, 3, 3), (48,), (2, 64, 1, 1), (1, 160)]
    cpu = [torch.randn(s, generator=rng, dtype=torch.float64).mul(0.1).requires_grad_(True) for s in shapes]
    late = torch.randn(7, 33, generator=rng, dtype=torch.float64).mul(0.1).requires_grad_(True)
    dev = [p.detach().float().to(DEV).requires_grad_(True) for p in cpu]
    late_dev = late.detach().float().to(DEV).requires_grad_(True)
    ref = torch.optim.Adam(cpu, lr=LR, betas=BETAS)
    opt = FusedAdam(dev, lr=LR, betas=BETAS)
    opt.grad_scale = grad_scale
    for step in range(7):
        if step == 3:
            ref.add_param_group({"params": [late], "lr": LR, "betas": BETAS})
            opt.add_param_group({"params": [late_dev], "lr": LR, "betas": BETAS})
            cpu.append(late)
            dev.append(late_dev)
        for i, (pc, pd) in enumerate(zip(cpu, dev)):
            if i == 1 and step in (2, 4):  # no gradient this step: Adam skips the parameter entirely
                pc.grad = pd.grad = None
                continue
            gr = torch.randn(pc.shape, generator=rng) * 10.0 ** float(torch.randint(-9, 2, (1,), generator=rng))
            pd.grad = gr.to(DEV)
            pc.grad = (gr * grad_scale).double()  # the float32 product the kernel forms, exactly
        ref.step()
        opt.step()
        for i, (pc, pd) in enumerate(zip(cpu, dev)):
            st_c, st_d = ref.state.get(pc, {}), opt.state.get(pd, {})
            assert (int(st_d["step"]) if st_d else 0) == (int(st_c["step"]) if st_c else 0)
            # one step moves a weight by at most lr; fp32 round-off of the update is <= 2e-6 of that plus one ulp of the weight
            err = float((pd.detach().double().cpu() - pc.detach()).abs().max())
            assert err <= (step + 1) * (2e-6 * LR + 1.2e-7 * float(pc.abs().max())), f"step {step} tensor {i}: {err:.2e}"
            if st_c:
                v_c, v_d = st_c["exp_avg_sq"], st_d["exp_avg_sq"].double().cpu()
                assert float((v_d - v_c).abs().max()) <= 1e-6 * float(v_c.max()), f"step {step} exp_avg_sq {i}"
    assert int(opt.state[dev[1]]["step"]) == 5 and int(opt.state[late_dev]["step"]) == 4
    sd = opt.state_dict()  # loadable by torch.optim.Adam (the reference's checkpoint consumer)
    chk = torch.optim.Adam([p.detach().clone().requires_grad_(True) for p in dev[:-1]], lr=LR, betas=BETAS)
    chk.add_param_group({"params": [dev[-1].detach().clone().requires_grad_(True)], "lr": LR, "betas": BETAS})
    chk.load_state_dict(sd)


def test_graph_replay_is_bit_identical_to_eager_updates(monkeypatch):
    """The HIP-graph path of ProGANStepper (updates captured after two eager calls, then replayed: device-side Adam step counters,
    in-graph weight re-packing, fade-in coefficients read from device memory) against the same stepper with MG_GRAPHS=0: eight
    iterations with alpha moving every iteration and the 5:1 cadence -- every loss and every weight, Adam moment and step count
    bit for bit."""
    import bench
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper

    def run(graphs: bool):
        monkeypatch.setenv("MG_GRAPHS", "1" if graphs else "0")
        gen, disc = bench.build_nets(3, 32, DEV)
        og = FusedAdam(gen.parameters(), lr=LR, betas=BETAS)
        od = FusedAdam(disc.parameters(), lr=LR, betas=BETAS)
        st = ProGANStepper(gen, disc, og, od, 32)
        assert st.use_graphs == graphs
        rng = torch.Generator(device=DEV).manual_seed(9)
        losses = []
        for it in range(8):
            alpha = min(1.0, (1 + 4 * it) / 24.0)
            x = torch.rand(4, 2, 32, 32, device=DEV, generator=rng) * 2 - 1
            z = torch.randn(4, 32, 2, 2, device=DEV, generator=rng)
            eps = torch.rand(4, 1, 1, 1, device=DEV, generator=rng)
            m = st.d_step(x, alpha, z=z, eps=eps)
            losses += [float(m["disc_loss"]), float(m["grad_pen"])]
            if it % 5 == 0 or it >= 5:
                losses.append(float(st.g_step(4, alpha, DEV, z=z)["gen_loss"]))
        st.finish()
        if graphs:
            assert sum("graph" in e for e in st._graphs.values()) == 2  # one critic graph, one generator graph for the level
        state = {}
        for name, net, opt in (("G", gen, og), ("D", disc, od)):
            for k, p in net.named_parameters():
                state[f"{name}.{k}"] = p.detach().clone()
                if p in opt.state:
                    state[f"{name}.{k}.v"] = opt.state[p]["exp_avg_sq"].clone()
                    state[f"{name}.{k}.t"] = torch.tensor([float(opt.state[p]["step"]), float(opt.state[p]["step_dev"])])
        return losses, state

    l_graph, s_graph = run(True)
    l_eager, s_eager = run(False)
    assert l_graph == l_eager
    assert s_graph.keys() == s_eager.keys()
    for k in s_graph:
        assert torch.equal(s_graph[k], s_eager[k]), k


@pytest.mark.parametrize("how", ["raise", "invalidate"])
def test_failed_graph_capture_falls_back_to_bit_identical_eager_updates(how, monkeypatch):
    """a capture that fails half way -- here: the one-launch weight-gradient reduction raises while the stream
    is capturing ("raise": what an out-of-memory in the graph's pool looks like), or does something HIP forbids during a capture
    and thereby invalidates it ("invalidate": a device synchronisation; the runtime then fails every later call on the stream until
    the capture is ended and the error state read) -- has already marked every packed weight layout as fresh (the pack kernels
    were captured, never run) and left deferred reduction jobs that point into the discarded graph pool.  The eager fallback must
    not inherit any of it: the whole trajectory equals the MG_GRAPHS=0 one bit for bit."""
    import warnings

    import bench
    from musicgan_amd import ops
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper

    def run(fail: bool):
        monkeypatch.setenv("MG_GRAPHS", "1" if fail else "0")
        gen, disc = bench.build_nets(3, 32, DEV)
        og = FusedAdam(gen.parameters(), lr=LR, betas=BETAS)
        od = FusedAdam(disc.parameters(), lr=LR, betas=BETAS)
        st = ProGANStepper(gen, disc, og, od, 32)
        rng = torch.Generator(device=DEV).manual_seed(11)
        losses = []
        for it in range(7):
            x = torch.rand(4, 2, 32, 32, device=DEV, generator=rng) * 2 - 1
            z = torch.randn(4, 32, 2, 2, device=DEV, generator=rng)
            eps = torch.rand(4, 1, 1, 1, device=DEV, generator=rng)
            m = st.d_step(x, 0.6, z=z, eps=eps)
            losses += [float(m["disc_loss"]), float(m["grad_pen"])]
            losses.append(float(st.g_step(4, 0.6, DEV, z=z)["gen_loss"]))
        st.finish()
        if fail:
            assert all(e.get("eager") for e in st._graphs.values()) and len(st._graphs) == 2
        state = {f"{n}.{k}": p.detach().clone() for n, net in (("G", gen), ("D", disc)) for k, p in net.named_parameters()}
        return losses, state

    real_flush = ops.WgradDefer.flush

    def flush(self):
        if torch.cuda.is_current_stream_capturing():
            if how == "invalidate":
                torch.cuda.synchronize()  # illegal during a capture: raises and leaves the capture invalidated
            raise RuntimeError("injected: capture fails inside the weight-gradient sweep")
        return real_flush(self)
    l_eager, s_eager = run(False)
    monkeypatch.setattr(ops.WgradDefer, "flush", flush)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        l_fail, s_fail = run(True)
    assert l_fail == l_eager
    for k in s_eager:
        assert torch.equal(s_fail[k], s_eager[k]), k


def test_captured_graphs_follow_the_optimizer(monkeypatch):
    """lr / betas / eps travel into a captured Adam launch as scalars and the moment tensors by address, so a graph
    must not outlive a change of either.  (i) lr set to 0 in every param group after the critic graph exists: the following
    updates (two eager ones, a fresh capture, replays) must leave every critic weight bit-identical -- the round-2 stepper kept
    replaying the lr = 1e-3 graph.  (ii) `load_state_dict` replaces the moment tensors: the next updates must equal an eager
    stepper's that loaded the same state, bit for bit."""
    import bench
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper
    monkeypatch.setenv("MG_GRAPHS", "1")
    gen, disc = bench.build_nets(2, 16, DEV)
    og = FusedAdam(gen.parameters(), lr=LR, betas=BETAS)
    od = FusedAdam(disc.parameters(), lr=LR, betas=BETAS)
    st = ProGANStepper(gen, disc, og, od, 16)
    rng = torch.Generator(device=DEV).manual_seed(4)

    def d_step(stepper):
        x = torch.rand(4, 2, 16, 16, device=DEV, generator=rng) * 2 - 1
        z = torch.randn(4, 16, 2, 2, device=DEV, generator=rng)
        eps = torch.rand(4, 1, 1, 1, device=DEV, generator=rng)
        return stepper.d_step(x, 0.7, z=z, eps=eps)

    for _ in range(4):
        d_step(st)
    assert sum("graph" in e for e in st._graphs.values()) == 1
    frozen = {k: p.detach().clone() for k, p in disc.named_parameters()}
    for group in od.param_groups:
        group["lr"] = 0.0
    for _ in range(5):
        d_step(st)
    torch.cuda.synchronize()
    assert all(torch.equal(p, frozen[k]) for k, p in disc.named_parameters()), "a stale graph kept training with the old lr"
    # the lr = 0 signature has its own graph; the lr = 1e-3 one (and its private memory pool) is dropped, not kept
    assert sum("graph" in e for e in st._graphs.values()) == 1 and len(st._graphs) == 1
    # (ii) load_state_dict
    for group in od.param_groups:
        group["lr"] = LR
    sd = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in od.state_dict().items()}
    import copy
    sd = copy.deepcopy(od.state_dict())
    weights = copy.deepcopy(disc.state_dict())
    od.load_state_dict(copy.deepcopy(sd))
    rng_state = rng.get_state()
    for _ in range(4):
        d_step(st)
    st.finish()
    got = {k: p.detach().clone() for k, p in disc.named_parameters()}
    # the same four updates on an eager stepper from the same weights / optimizer state / inputs
    monkeypatch.setenv("MG_GRAPHS", "0")
    disc.load_state_dict(weights)
    od2 = FusedAdam(disc.parameters(), lr=LR, betas=BETAS)
    od2.load_state_dict(copy.deepcopy(sd))
    st2 = ProGANStepper(gen, disc, og, od2, 16)
    assert not st2.use_graphs
    rng.set_state(rng_state)
    for _ in range(4):
        d_step(st2)
    torch.cuda.synchronize()
    for k, p in disc.named_parameters():
        assert torch.equal(p, got[k]), k
