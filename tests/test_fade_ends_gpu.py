"""GPU parity of the single-launch fade-in ends (csrc/fade_ends.hip) against fp64 PyTorch restatements of the reference lines they
cover: the critic's input end (/root/reference/music_gan/networks/discriminator.py:107-113) forward, tangent and backward to x, the
generator's output end (generator.py:118-126) forward and the blend's backward; the single-launch 1x1 weight gradient; and a whole
critic + generator update with the fused ends against the same update on the separate kernels."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
SLOPE = 0.2
SHAPES = [(3, 16, 24, 8, 8), (5, 48, 80, 32, 32), (2, 16, 32, 64, 128), (1, 160, 160, 4, 4)]  # (N, C0, C1, H, W)


def _ops():
    from musicgan_amd import ops
    return ops


def _rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("shape", SHAPES)
def test_stem_pair_forward_and_tangent(shape):
    ops = _ops()
    n, c0, c1, h, w = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, 2, h, w, generator=g)
    ws, bs = torch.randn(c0, 2, 1, 1, generator=g), torch.randn(c0, generator=g)
    wo, bo = torch.randn(c1, 2, 1, 1, generator=g), torch.randn(c1, generator=g)
    xd = x.double()
    ref_h0 = F.leaky_relu(F.conv2d(xd, ws.double(), bs.double()), SLOPE)
    ref_xp = F.avg_pool2d(xd, 2)
    ref_o = F.leaky_relu(F.conv2d(ref_xp, wo.double(), bo.double()), SLOPE)
    to = lambda t: t.to(DEV)
    h0, xp, o = ops.stem_pair(to(x), to(ws), to(bs), to(wo), to(bo))
    assert _rel(h0, ref_h0) < 2e-6 and _rel(xp, ref_xp) < 2e-6 and _rel(o, ref_o) < 2e-6
    # the separate kernels give the same values (one fused multiply-add chain per output either way)
    h0_s = ops.conv1x1(to(x), to(ws), to(bs), c0, lrelu=True)
    assert torch.equal(h0, h0_s)
    # the optional tile mask of h0: one byte per 2x2 tile, bit 2i+j <-> h0[2Y+i][2X+j] > 0 (mg_wino3x3's format)
    h0_m, _, _, hm = ops.stem_pair(to(x), to(ws), to(bs), to(wo), to(bo), want_mask=True)
    b = (h0_m > 0).reshape(n, c0, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c0, h // 2, w // 2, 4).to(torch.uint8)
    assert torch.equal(h0_m, h0) and torch.equal(hm, b[..., 0] + 2 * b[..., 1] + 4 * b[..., 2] + 8 * b[..., 3])
    # tangent form: (w u) * lrelu'(activation), over the activations
    u = torch.randn(n, 2, h, w, generator=g)
    ref_t = F.conv2d(u.double(), ws.double()) * torch.where(ref_h0 > 0, 1.0, SLOPE)
    ref_up = F.avg_pool2d(u.double(), 2)
    ref_to = F.conv2d(ref_up, wo.double()) * torch.where(ref_o > 0, 1.0, SLOPE)
    th, tp, t_o = h0.clone(), torch.empty_like(xp), o.clone()
    ops.stem_pair(to(u), to(ws), None, to(wo), None, h0=th, xp=tp, o=t_o, masked=True)
    assert _rel(th, ref_t) < 2e-6 and _rel(tp, ref_up) < 2e-6 and _rel(t_o, ref_to) < 2e-6


@pytest.mark.parametrize("shape", SHAPES)
def test_stem_pair_gx(shape):
    ops = _ops()
    n, c0, c1, h, w = shape
    g = torch.Generator().manual_seed(4)
    gs, go = torch.randn(n, c0, h, w, generator=g), torch.randn(n, c1, h // 2, w // 2, generator=g)
    ws, wo = torch.randn(c0, 2, 1, 1, generator=g), torch.randn(c1, 2, 1, 1, generator=g)
    ref = F.conv_transpose2d(gs.double(), ws.double()) + \
        0.25 * F.interpolate(F.conv_transpose2d(go.double(), wo.double()), scale_factor=2, mode="nearest")
    gx = ops.stem_pair_gx(gs.to(DEV), ws.to(DEV), go.to(DEV), wo.to(DEV))
    assert _rel(gx, ref) < 3e-6


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dev_coef", [False, True])
def test_head_pair_and_blend_backward(shape, dev_coef):
    ops = _ops()
    n, c, cl, h, w = shape
    g = torch.Generator().manual_seed(5)
    x, xl = torch.randn(n, c, h, w, generator=g), torch.randn(n, cl, h // 2, w // 2, generator=g)
    wh, bh = torch.randn(2, c, 1, 1, generator=g) / c ** 0.5, torch.randn(2, generator=g)
    wo, bo = torch.randn(2, cl, 1, 1, generator=g) / cl ** 0.5, torch.randn(2, generator=g)
    a, b = 0.3, 0.7
    ref_mp = torch.tanh(F.conv2d(x.double(), wh.double(), bh.double()))
    ref_old = torch.tanh(F.conv2d(xl.double(), wo.double(), bo.double()))
    ref_out = a * ref_mp + b * F.interpolate(ref_old, scale_factor=2, mode="nearest")
    to = lambda t: t.to(DEV)
    coef = torch.tensor([a, b], device=DEV) if dev_coef else None
    sa, sb = (0.0, 0.0) if dev_coef else (a, b)  # with device coefficients the scalars are ignored
    out, mp, old = ops.head_pair(to(x), to(wh), to(bh), to(xl), to(wo), to(bo), sa, sb, coef=coef)
    assert _rel(mp, ref_mp) < 3e-6 and _rel(old, ref_old) < 3e-6 and _rel(out, ref_out) < 3e-6
    out2, mp2, old2 = ops.head_pair(to(x), to(wh), to(bh), to(xl), to(wo), to(bo), sa, sb, coef=coef, save=False)
    assert mp2 is None and old2 is None and torch.equal(out, out2)
    if w % 4 == 0:
        gout = torch.randn(n, 2, h, w, generator=g)
        gx, gy = ops.blend_up_bwd(to(gout), sa, sb, coef=coef)
        assert _rel(gx, a * gout.double()) < 1e-6
        assert _rel(gy, b * 4.0 * F.avg_pool2d(gout.double(), 2)) < 2e-6


def test_conv1x1_accumulate():
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    gy, w = torch.randn(3, 2, 16, 16, generator=g), torch.randn(2, 48, 1, 1, generator=g)
    th = torch.tanh(torch.randn(3, 2, 16, 16, generator=g))
    base = torch.randn(3, 48, 16, 16, generator=g)
    ref = base.double() + F.conv_transpose2d((gy * (1 - th * th)).double(), w.double())
    out = base.to(DEV)
    ops.conv1x1(gy.to(DEV), w.to(DEV), None, 48, transposed=True, tanh_bwd_in=th.to(DEV), out=out, accumulate=True)
    assert _rel(out, ref) < 2e-6


@pytest.mark.parametrize("case", [(24, 2, 48, 32, 32, 16), (8, 160, 2, 8, 8, 0), (6, 2, 16, 64, 64, 4), (3, 80, 2, 16, 16, 0)])
def test_conv1x1_wgrad_single_launch(case, monkeypatch):
    """(N, Cin, Cout, H, W, bias_n): the opt-in single-launch form (MG_C1_WGRAD_SINGLE=1, up to 64 workgroup columns: the last workgroup
    sums the partials); the same launch twenty times gives the same bits (the ticket counter returns to zero, the order of the sums is
    fixed)."""
    monkeypatch.setenv("MG_C1_WGRAD_SINGLE", "1")
    ops = _ops()
    n, cin, cout, h, w, bias_n = case
    g = torch.Generator().manual_seed(8)
    x, gy = torch.randn(n, cin, h, w, generator=g), torch.randn(n, cout, h, w, generator=g)
    ref_w = torch.einsum("nohw,nchw->oc", gy.double(), x.double())
    ref_b = gy[:bias_n or n].double().sum(dim=(0, 2, 3))
    xd, gyd = x.to(DEV), gy.to(DEV)
    first = None
    for _ in range(20):
        gw, gb = torch.full((cout, cin, 1, 1), 7.0, device=DEV), torch.full((cout,), 7.0, device=DEV)
        ops.conv1x1_wgrad(xd, gyd, gw, gb, bias_n=bias_n)
        if first is None:
            first = (gw.clone(), gb.clone())
            assert _rel(gw.reshape(cout, cin), ref_w) < 1e-5 and _rel(gb, ref_b) < 1e-5
        else:
            assert torch.equal(gw, first[0]) and torch.equal(gb, first[1])
    gw2, gb2 = first[0].clone(), first[1].clone()
    ops.conv1x1_wgrad(xd, gyd, gw2, gb2, bias_n=bias_n, accumulate=True)
    assert _rel(gw2.reshape(cout, cin), 2 * ref_w) < 1e-5 and _rel(gb2, 2 * ref_b) < 1e-5


def test_gp_apply_equals_finish_and_scale():
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(9)
    gx = torch.randn(7, 2, 32, 32, device=DEV, generator=g) * 0.03
    ss = ops.sumsq_per_sample(gx)
    pen, coef = ops.gp_finish(ss, 10.0, 1.0)
    want = ops.scale_per_sample(gx, coef)
    pen2, got = ops.gp_apply(gx, ss, 10.0, 1.0)
    assert torch.equal(pen, pen2)
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())


@pytest.mark.parametrize("level,batch", [(3, 8), (5, 2)])
def test_update_with_fused_ends_equals_the_separate_kernels(level, batch, monkeypatch):
    """One critic and one generator update from the same weights, noise and fade-in coefficient: every parameter gradient with the
    single-launch ends against the separate kernels (different summation grouping in the many-channel sums: 1e-5 of the largest
    gradient of the tensor)."""
    import bench
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper
    dev = torch.device(DEV)
    grads = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MG_FUSE_ENDS", mode)
        monkeypatch.setenv("MG_GRAPHS", "0")
        torch.manual_seed(0)
        gen, disc = bench.build_nets(level, 32, dev)
        st = ProGANStepper(gen, disc, FusedAdam(gen.parameters(), lr=0.0, betas=(0.0, 0.9)),
                           FusedAdam(disc.parameters(), lr=0.0, betas=(0.0, 0.9)), 32)
        side = bench.LEVEL_SIDE[level]
        rng = torch.Generator(device=dev).manual_seed(1)
        x_real = torch.rand(batch, 2, side, side, device=dev, generator=rng) * 2 - 1
        z = torch.randn(batch, 32, st.h, st.w, device=dev, generator=rng)
        eps = torch.rand(batch, 1, 1, 1, device=dev, generator=rng)
        out_d = st.d_step(x_real, 0.4, z=z, eps=eps)
        gd = [p.grad.clone() for p in disc.parameters() if p.grad is not None]
        out_g = st.g_step(batch, 0.4, dev, z=z)
        gg = [p.grad.clone() for p in gen.parameters() if p.grad is not None]
        grads[mode] = (gd, gg, {k: float(v) for k, v in {**out_d, **out_g}.items()})
    for a_, b_ in zip(grads["0"][0] + grads["0"][1], grads["1"][0] + grads["1"][1]):
        assert a_.shape == b_.shape
        assert float((a_ - b_).abs().max()) <= 1e-5 * float(a_.abs().max()) + 1e-12
    for k, v in grads["0"][2].items():
        assert abs(v - grads["1"][2][k]) <= 1e-5 * max(1.0, abs(v)), k


@pytest.mark.parametrize("shape", [(3, 48, 16, 24), (2, 64, 8, 8), (5, 32, 32, 16), (2, 16, 64, 64), (1, 48, 128, 128), (8, 80, 32, 32),
                                   (3, 96, 16, 16), (2, 128, 4, 4), (5, 20, 8, 12), (32, 112, 8, 8)])
def test_gen_head_bwd_matches_the_three_launches_and_autograd(shape):
    """mg_gen_head_bwd (head weight / data gradient + PixelNorm / LeakyReLU backward of the block in front, one pass) against the
    three launches it replaces and against fp64 autograd of tanh(conv1x1(PixelNorm(LeakyReLU(y)))) -- generator.py:31-39, 118-126."""
    ops = _ops()
    n, c, h, w = shape
    g = torch.Generator().manual_seed(23)
    y = torch.randn(n, c, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = (torch.randn(2, c, 1, 1, generator=g, dtype=torch.float64) / c ** 0.5).requires_grad_(True)
    bt = torch.randn(2, generator=g, dtype=torch.float64, requires_grad=True)
    g_mp = torch.randn(n, 2, h, w, generator=g, dtype=torch.float64)
    a = torch.nn.functional.leaky_relu(y, 0.2)
    p = a / torch.sqrt((a * a).mean(dim=1, keepdim=True) + 1e-8)
    mp = torch.tanh(torch.nn.functional.conv2d(p, wt, bt))
    (mp * g_mp).sum().backward()
    pd, mpd, gd, wd = p.detach().float().to(DEV), mp.detach().float().to(DEV), g_mp.float().to(DEV), wt.detach().float().to(DEV)
    rn = (1.0 / torch.sqrt((a * a).mean(dim=1, keepdim=True) + 1e-8)).detach().float().to(DEV)
    gw, gb = torch.full((2, c, 1, 1), 3.0, device=DEV), torch.full((2,), 3.0, device=DEV)
    assert ops.gen_head_bwd_supported(c, 2, n, h * w)
    gpre = ops.gen_head_bwd(gd, mpd, wd, pd, rn, gw, gb)
    rel = lambda got, ref: float((got.double().cpu() - ref).abs().max() / ref.abs().max())
    assert rel(gpre, y.grad) <= 5e-6 and rel(gw, wt.grad) <= 5e-6 and rel(gb, bt.grad) <= 5e-6
    # the separate launches
    gw0, gb0 = torch.empty_like(gw), torch.empty_like(gb)
    ops.conv1x1_wgrad(pd, gd, gw0, gb0, tanh_y=mpd)
    g0 = ops.conv1x1(gd, wd, None, c, transposed=True, tanh_bwd_in=mpd)
    gpre0 = ops.pixelnorm_lrelu_bwd(g0, pd, rn, from_p=True)
    assert rel(gpre, gpre0.double().cpu()) <= 2e-6 and rel(gw, gw0.double().cpu()) <= 2e-6 and rel(gb, gb0.double().cpu()) <= 2e-6
    gpre2 = ops.gen_head_bwd(gd, mpd, wd, pd, rn, gw, gb, accumulate=True)
    assert torch.equal(gpre, gpre2) and rel(gw, 2 * wt.grad) <= 5e-6 and rel(gb, 2 * bt.grad) <= 5e-6
    assert not ops.gen_head_bwd_supported(40) and not ops.gen_head_bwd_supported(48, 3) and not ops.gen_head_bwd_supported(80, 2, 64, 64 * 64)
    # a second gradient arriving at p (the old head of a fading-in level)
    g_in = torch.randn(n, c, h, w, generator=g).to(DEV)
    gw3, gb3 = torch.empty_like(gw), torch.empty_like(gb)
    gpre3 = ops.gen_head_bwd(gd, mpd, wd, pd, rn, gw3, gb3, g_in=g_in)
    ref3 = ops.pixelnorm_lrelu_bwd(g0 + g_in, pd, rn, from_p=True)
    assert rel(gpre3, ref3.double().cpu()) <= 2e-6 and rel(gw3, gw0.double().cpu()) <= 2e-6 and rel(gb3, gb0.double().cpu()) <= 2e-6
