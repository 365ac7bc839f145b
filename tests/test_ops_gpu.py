"""GPU parity of every C-ABI kernel against plain PyTorch fp32/fp64 CPU references of the same op."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from musicgan_amd import ops
    return ops


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def report(name, got, ref, tol):
    e = rel_err(got, ref)
    if not (e <= tol):
        g, r = got.detach().double().cpu(), ref.detach().double().cpu()
        bad = ((g - r).abs() > tol * r.abs().max()).nonzero()
        raise AssertionError(f"{name}: rel err {e:.3e} > {tol:.1e}; {bad.shape[0]} bad of {g.numel()}, first {bad[:6].tolist()}")
    return e


CONV_SHAPES = [
    # N, Cin, Cout, H, W
    (2, 8, 8, 2, 2), (3, 32, 128, 4, 4), (2, 128, 112, 8, 8), (2, 96, 80, 32, 32), (2, 64, 48, 64, 64),
    (1, 48, 64, 128, 128), (2, 160, 160, 1, 1), (2, 144, 160, 2, 2), (3, 16, 32, 16, 16), (1, 24, 40, 12, 20),
    (2, 80, 64, 64, 64), (5, 112, 128, 8, 8),
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_fwd_bias_lrelu(shape):
    ops = _ops()
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    b = torch.randn(co, generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1), 0.2)
    wp = ops.pack_conv3x3(wt.to(DEV), dgrad=False)
    y = ops.conv3x3(x.to(DEV), wp, b.to(DEV), co, lrelu=True)
    report("conv3x3 fwd", y, ref, 2e-6)
    # no bias, no activation
    y2 = ops.conv3x3(x.to(DEV), wp, None, co)
    report("conv3x3 plain", y2, F.conv2d(x.double(), wt.double(), None, padding=1), 2e-6)


@pytest.mark.parametrize("shape", CONV_SHAPES[:8])
def test_conv3x3_dgrad_and_mask(shape):
    ops = _ops()
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(2)
    gy = torch.randn(n, co, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    act = torch.randn(n, ci, h, w, generator=g)
    ref = F.conv_transpose2d(gy.double(), wt.double(), padding=1)
    wp = ops.pack_conv3x3(wt.to(DEV), dgrad=True)
    gx = ops.conv3x3(gy.to(DEV), wp, None, ci)
    report("conv3x3 dgrad", gx, ref, 2e-6)
    gxm = ops.conv3x3(gy.to(DEV), wp, None, ci, mask_aux=act.to(DEV))
    report("conv3x3 dgrad+mask", gxm, ref * torch.where(act > 0, 1.0, 0.2).double(), 2e-6)


@pytest.mark.parametrize("shape", [(2, 8, 8, 2, 2), (2, 32, 128, 4, 4), (2, 96, 80, 16, 16), (1, 64, 48, 64, 64),
                                   (2, 24, 40, 6, 10)])
def test_conv3x3_ups_pixnorm(shape):
    ops = _ops()
    n, ci, co, h, w = shape  # h, w = INPUT size; output is 2h x 2w
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    b = torch.randn(co, generator=g)
    xd = F.interpolate(x.double(), scale_factor=2.0, mode="nearest")
    yr = F.leaky_relu(F.conv2d(xd, wt.double(), b.double(), padding=1), 0.2)
    nr = torch.sqrt(yr.pow(2).mean(dim=1, keepdim=True) + 1e-8)
    wp = ops.pack_conv3x3(wt.to(DEV), dgrad=False)
    y, p, rn = ops.conv3x3(x.to(DEV), wp, b.to(DEV), co, ups=True, lrelu=True, pixnorm=True)
    report("ups conv y", y, yr, 2e-6)
    report("ups conv p", p, yr / nr, 3e-6)
    report("ups conv rn", rn, 1.0 / nr, 3e-6)
    # stand-alone pixelnorm kernel agrees
    p2, rn2 = ops.pixelnorm_fwd(y)
    report("pixelnorm p", p2, yr / nr, 3e-6)
    report("pixelnorm rn", rn2, 1.0 / nr, 3e-6)


WGRAD_SHAPES = [(2, 8, 8, 2, 2), (3, 32, 128, 4, 4), (4, 128, 112, 8, 8), (2, 96, 80, 32, 32), (2, 64, 48, 64, 64),
                (1, 48, 64, 128, 128), (6, 160, 160, 1, 1), (5, 144, 160, 2, 2), (3, 16, 32, 16, 16),
                (2, 24, 40, 12, 20)]


@pytest.mark.parametrize("shape", WGRAD_SHAPES)
def test_conv3x3_wgrad(shape):
    ops = _ops()
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, ci, h, w, generator=g).double().requires_grad_(False)
    gy = torch.randn(n, co, h, w, generator=g).double()
    wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x, wt, bt, padding=1) * gy).sum().backward()
    gw = torch.full((co, ci, 3, 3), 7.0, device=DEV)
    gb = torch.full((co,), 7.0, device=DEV)
    ops.conv3x3_wgrad(x.float().to(DEV), gy.float().to(DEV), gw, gb)
    report("wgrad gw", gw, wt.grad, 3e-6)
    report("wgrad gb", gb, bt.grad, 3e-6)
    ops.conv3x3_wgrad(x.float().to(DEV), gy.float().to(DEV), gw, gb, accumulate=True)
    report("wgrad gw acc", gw, 2 * wt.grad, 3e-6)
    report("wgrad gb acc", gb, 2 * bt.grad, 3e-6)


@pytest.mark.parametrize("shape", [(2, 8, 16, 2, 2), (2, 96, 80, 8, 8), (1, 64, 48, 32, 32), (3, 24, 40, 6, 10), (5, 20, 17, 1, 6)])
@pytest.mark.parametrize("wino", [False, True])
def test_conv3x3_wgrad_upsampled_input(shape, wino, monkeypatch):
    ops = _ops()
    monkeypatch.setenv("MG_WINO_WGRAD", "1" if wino else "0")
    monkeypatch.setenv("MG_WINO_WGRAD_MIN_PIXELS", "1")
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, ci, h, w, generator=g).double()
    gy = torch.randn(n, co, 2 * h, 2 * w, generator=g).double()
    wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, None, padding=1) * gy).sum().backward()
    gw = torch.empty((co, ci, 3, 3), device=DEV)
    ops.conv3x3_wgrad(x.float().to(DEV), gy.float().to(DEV), gw, None, ups=True)
    report("wgrad ups", gw, wt.grad, 3e-6)


@pytest.mark.parametrize("hw", [(4, 4), (32, 32), (3, 5)])
@pytest.mark.parametrize("c", [16, 48, 160])
def test_conv1x1_all_modes(hw, c):
    ops = _ops()
    h, w = hw
    n = 3
    g = torch.Generator().manual_seed(6)
    x2 = torch.randn(n, 2, h, w, generator=g)
    xc = torch.randn(n, c, h, w, generator=g)
    ws = torch.randn(c, 2, 1, 1, generator=g)
    bs = torch.randn(c, generator=g)
    wh = torch.randn(2, c, 1, 1, generator=g) / math.sqrt(c)
    bh = torch.randn(2, generator=g)
    # stem: 2 -> C + LReLU
    y = ops.conv1x1(x2.to(DEV), ws.to(DEV), bs.to(DEV), c, lrelu=True)
    report("stem", y, F.leaky_relu(F.conv2d(x2.double(), ws.double(), bs.double()), 0.2), 2e-6)
    # stem tangent: no bias, output mask
    act = torch.randn(n, c, h, w, generator=g)
    y = ops.conv1x1(x2.to(DEV), ws.to(DEV), None, c, mask_aux=act.to(DEV))
    report("stem tangent", y, F.conv2d(x2.double(), ws.double()) * torch.where(act > 0, 1.0, 0.2).double(), 2e-6)
    # head: C -> 2 + tanh
    mp = ops.conv1x1(xc.to(DEV), wh.to(DEV), bh.to(DEV), 2, tanh=True)
    mp_ref = torch.tanh(F.conv2d(xc.double(), wh.double(), bh.double()))
    report("head", mp, mp_ref, 3e-6)
    # stem dgrad: C -> 2 with w^T
    gyc = torch.randn(n, c, h, w, generator=g)
    gx = ops.conv1x1(gyc.to(DEV), ws.to(DEV), None, 2, transposed=True)
    report("stem dgrad", gx, F.conv_transpose2d(gyc.double(), ws.double()), 3e-6)
    # head dgrad with tanh backward on the input: 2 -> C
    gy2 = torch.randn(n, 2, h, w, generator=g)
    gxh = ops.conv1x1(gy2.to(DEV), wh.to(DEV), None, c, transposed=True, tanh_bwd_in=mp)
    ref = F.conv_transpose2d(gy2.double() * (1 - mp_ref ** 2), wh.double())
    report("head dgrad", gxh, ref, 5e-6)
    # stem wgrad
    gw = torch.empty(c, 2, 1, 1, device=DEV)
    gb = torch.empty(c, device=DEV)
    ops.conv1x1_wgrad(x2.to(DEV), gyc.to(DEV), gw, gb)
    ref_w = torch.einsum("nohw,nchw->oc", gyc.double(), x2.double())
    report("stem wgrad", gw.reshape(c, 2), ref_w, 3e-6)
    report("stem bgrad", gb, gyc.double().sum(dim=(0, 2, 3)), 3e-6)
    ops.conv1x1_wgrad(x2.to(DEV), gyc.to(DEV), gw, None, accumulate=True)
    report("stem wgrad acc", gw.reshape(c, 2), 2 * ref_w, 3e-6)
    # head wgrad with tanh backward
    gwh = torch.empty(2, c, 1, 1, device=DEV)
    gbh = torch.empty(2, device=DEV)
    ops.conv1x1_wgrad(xc.to(DEV), gy2.to(DEV), gwh, gbh, tanh_y=mp)
    gpre = gy2.double() * (1 - mp_ref ** 2)
    report("head wgrad", gwh.reshape(2, c), torch.einsum("nohw,nchw->oc", gpre, xc.double()), 5e-6)
    report("head bgrad", gbh, gpre.sum(dim=(0, 2, 3)), 5e-6)


def test_elementwise_ops():
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    for (n, c, h, w) in [(2, 48, 8, 8), (3, 7, 2, 2), (1, 160, 1, 1), (2, 16, 6, 10)]:
        y = torch.randn(n, c, h, w, generator=g)
        gp = torch.randn(n, c, h, w, generator=g)
        yd = y.double().requires_grad_(True)
        act = F.leaky_relu(yd, 0.2)
        pn = act / torch.sqrt(act.pow(2).mean(dim=1, keepdim=True) + 1e-8)
        (pn * gp.double()).sum().backward()
        a = act.detach().float().to(DEV)
        p, rn = ops.pixelnorm_fwd(a)
        report("pn fwd", p, pn, 3e-6)
        gpre = ops.pixelnorm_lrelu_bwd(gp.to(DEV), a, rn)
        report("pn+lrelu bwd", gpre, yd.grad, 1e-5)
        report("pn+lrelu bwd from p", ops.pixelnorm_lrelu_bwd(gp.to(DEV), p, rn, from_p=True), yd.grad, 1e-5)
        report("up fwd", ops.upsample2x_fwd(y.to(DEV)), F.interpolate(y, scale_factor=2.0, mode="nearest"), 0)
        g2 = torch.randn(n, c, 2 * h, 2 * w, generator=g)
        report("up bwd", ops.upsample2x_bwd(g2.to(DEV)), F.avg_pool2d(g2.double(), 2, 2) * 4, 1e-6)
        report("pool fwd", ops.avgpool2_fwd(g2.to(DEV)), F.avg_pool2d(g2.double(), 2, 2), 1e-6)
        act2 = torch.randn(n, c, 2 * h, 2 * w, generator=g)
        ref = F.interpolate(y.double(), scale_factor=2.0, mode="nearest") * 0.25
        report("pool bwd", ops.avgpool2_bwd(y.to(DEV)), ref, 1e-6)
        report("pool bwd mask", ops.avgpool2_bwd(y.to(DEV), act2.to(DEV)), ref * torch.where(act2 > 0, 1.0, 0.2), 1e-6)
        report("lrelu bwd", ops.lrelu_bwd(gp.to(DEV), y.to(DEV)), gp * torch.where(y > 0, 1.0, 0.2), 1e-7)
        report("axpby", ops.axpby(0.37, y.to(DEV), 0.63, gp.to(DEV)), 0.37 * y.double() + 0.63 * gp.double(), 1e-6)
        report("ax", ops.axpby(0.37, y.to(DEV)), 0.37 * y.double(), 1e-6)
        report("blend_up", ops.blend_up(0.37, g2.to(DEV), 0.63, y.to(DEV)),
               0.37 * g2.double() + 0.63 * F.interpolate(y.double(), scale_factor=2.0, mode="nearest"), 1e-6)
        report("chan sum", ops.channel_sum(y.to(DEV)), y.double().sum(dim=(0, 2, 3)), 1e-5)


def test_linear_and_gp_helpers():
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    n, k = 7, 160
    x = torch.randn(n, k, generator=g)
    w = torch.randn(1, k, generator=g)
    b = torch.randn(1, generator=g)
    gy = torch.randn(n, 1, generator=g)
    report("linear fwd", ops.linear1_fwd(x.to(DEV), w.to(DEV), b.to(DEV)), F.linear(x.double(), w.double(), b.double()), 2e-6)
    gw = torch.empty(1, k, device=DEV)
    gb = torch.empty(1, device=DEV)
    gx = ops.linear1_bwd(x.to(DEV), w.to(DEV), gy.to(DEV), gw=gw, gb=gb)
    report("linear gx", gx, gy.double() @ w.double(), 1e-6)
    report("linear gw", gw, gy.double().t() @ x.double(), 2e-6)
    report("linear gb", gb, gy.double().sum().reshape(1), 2e-6)
    xr = torch.randn(n, 2, 8, 8, generator=g)
    xf = torch.randn(n, 2, 8, 8, generator=g)
    eps = torch.rand(n, 1, 1, 1, generator=g)
    report("interp", ops.gp_interp(xr.to(DEV), xf.to(DEV), eps.to(DEV)), eps * xr + (1 - eps) * xf, 1e-6)
    ss = ops.sumsq_per_sample(xr.to(DEV))
    report("sumsq", ss, xr.double().pow(2).sum(dim=(1, 2, 3)), 2e-6)
    pen, coef = ops.gp_finish(ss, 10.0, 1.0)
    nrm = xr.double().reshape(n, -1).norm(dim=1)
    report("penalty", pen, 10 * ((nrm - 1) ** 2).mean(), 2e-6)
    report("coef", coef, 10 * 2 * (nrm - 1) / (n * nrm), 2e-6)
    report("scale", ops.scale_per_sample(xr.to(DEV), coef), xr.double() * (10 * 2 * (nrm - 1) / (n * nrm)).reshape(n, 1, 1, 1), 3e-6)


def test_stft_kernel_matches_oracle_and_torch():
    ops = _ops()
    from oracle import audio as OA
    g = torch.Generator().manual_seed(9)
    for length in (141_312, 256 * 40 + 17, 1024):
        wav = torch.rand(length, generator=g) - 0.5
        got = ops.stft_1024(wav.to(DEV)).cpu()
        ref = torch.from_numpy(OA.stft(wav.numpy()))
        assert got.shape == ref.shape == (512, 1 + length // 256)
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err <= 1e-5, f"stft L={length}: rel err {err:.3e}"
        tref = torch.stft(wav, 1024, 256, 1024, torch.hann_window(1024), center=True, pad_mode="reflect",
                          normalized=False, onesided=True, return_complex=True)[:-1] / math.sqrt(384.0)
        assert float((got - tref).abs().max() / tref.abs().max()) <= 1e-5
    # bin indexing is exact: a pure tone at bin 37 peaks at row 37 of every interior frame
    t = torch.arange(44100, dtype=torch.float64)
    tone = torch.sin(2 * math.pi * 37 * t / 1024).float()
    spec = ops.stft_1024(tone.to(DEV)).abs().cpu()
    assert bool((spec[:, 4:-4].argmax(dim=0) == 37).all())


@pytest.mark.parametrize("shape", [(2, 48, 64, 128, 128), (3, 64, 80, 64, 64), (4, 80, 96, 32, 32), (2, 96, 112, 16, 16),
                                   (3, 112, 128, 8, 8), (2, 144, 160, 2, 2), (1, 24, 40, 12, 20)])
def test_conv3x3_fused_avgpool_output(shape):
    """MG_CONV_POOL_OUT: the pooled tensor from the conv epilogue (or the fallback launch) equals AvgPool2d(2,2) of y, for the
    forward (bias + LeakyReLU) and the tangent (mask) forms, including in-place y over the mask."""
    ops = _ops()
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    b = torch.randn(co, generator=g)
    act = torch.randn(n, co, h, w, generator=g)
    wp = ops.pack_conv3x3(wt.to(DEV), dgrad=False)
    yr = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1), 0.2)
    y, q = ops.conv3x3(x.to(DEV), wp, b.to(DEV), co, lrelu=True, pool=True)
    report("pool y", y, yr, 2e-6)
    report("pool q", q, F.avg_pool2d(yr, 2, 2), 3e-6)
    tr = F.conv2d(x.double(), wt.double(), None, padding=1) * torch.where(act > 0, 1.0, 0.2).double()
    buf = act.to(DEV).clone()
    q2 = torch.empty(n, co, h // 2, w // 2, device=DEV)
    ops.conv3x3(x.to(DEV), wp, None, co, mask_aux=buf, out=buf, pool_out=q2)
    report("tangent y", buf, tr, 2e-6)
    report("tangent q", q2, F.avg_pool2d(tr, 2, 2), 3e-6)


@pytest.mark.parametrize("shape", [(2, 8, 16, 2, 2), (3, 32, 96, 4, 4), (2, 112, 96, 8, 8), (2, 96, 80, 16, 16),
                                   (2, 80, 64, 32, 32), (1, 64, 48, 64, 64), (2, 24, 40, 6, 10), (1, 16, 8, 3, 5)])
def test_upconv3x3_subpixel_matches_upsample_conv(shape):
    """mg_upconv3x3 (four 2x2 convs on the low-res input) == Upsample(x2 nearest) -> Conv2d(3x3) -> LeakyReLU -> PixelNorm."""
    ops = _ops()
    n, ci, co, h, w = shape  # low-res input size
    g = torch.Generator().manual_seed(12)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    b = torch.randn(co, generator=g)
    xd = F.interpolate(x.double(), scale_factor=2.0, mode="nearest")
    yr = F.leaky_relu(F.conv2d(xd, wt.double(), b.double(), padding=1), 0.2)
    nr = torch.sqrt(yr.pow(2).mean(dim=1, keepdim=True) + 1e-8)
    wp = ops.pack_upconv3x3(wt.to(DEV))
    y, p, rn = ops.upconv3x3(x.to(DEV), wp, b.to(DEV), co, lrelu=True, pixnorm=True)
    report("upconv y", y, yr, 3e-6)
    report("upconv p", p, yr / nr, 4e-6)
    report("upconv rn", rn, 1.0 / nr, 4e-6)
    y2 = ops.upconv3x3(x.to(DEV), wp, None, co)
    report("upconv plain", y2, F.conv2d(xd, wt.double(), None, padding=1), 3e-6)


def test_upconv3x3_dgrad_small_images_are_refused():
    ops = _ops()
    assert not ops.upconv3x3_dgrad_supported(2, 2) and not ops.upconv3x3_dgrad_supported(4, 4)
    assert ops.upconv3x3_dgrad_supported(8, 8) and ops.upconv3x3_dgrad_supported(64, 64)
    wp = ops.pack_upconv3x3_dgrad(torch.randn(16, 8, 3, 3, device=DEV))
    with pytest.raises(Exception, match="halo tile too large"):
        ops.upconv3x3_dgrad(torch.randn(2, 16, 4, 4, device=DEV), wp, 8)


@pytest.mark.parametrize("shape", [(3, 128, 112, 8, 8), (2, 112, 96, 8, 8), (2, 96, 80, 16, 16), (5, 20, 17, 8, 16),
                                   (2, 80, 64, 32, 32), (1, 64, 48, 64, 64), (2, 24, 40, 6, 10), (1, 16, 8, 3, 5)])
def test_upconv3x3_dgrad_matches_autograd(shape):
    """mg_upconv3x3_dgrad (one 4x4 stride-2 conv over gy) == d/dx of Conv2d(3x3)(Upsample(x2)(x))."""
    ops = _ops()
    n, ci, co, h, w = shape  # low-res input size
    g = torch.Generator().manual_seed(13)
    x = torch.randn(n, ci, h, w, generator=g).double().requires_grad_(True)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    gy = torch.randn(n, co, 2 * h, 2 * w, generator=g)
    (F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt.double(), None, padding=1) * gy.double()).sum().backward()
    wp = ops.pack_upconv3x3_dgrad(wt.to(DEV))
    gx = ops.upconv3x3_dgrad(gy.to(DEV), wp, ci)
    report("upconv dgrad", gx, x.grad, 3e-6)


WINO_SHAPES = [(2, 8, 8, 2, 2), (3, 32, 128, 4, 4), (2, 112, 96, 8, 8), (2, 96, 80, 16, 16), (2, 48, 64, 32, 32),
               (1, 64, 48, 64, 64), (2, 24, 40, 6, 10), (3, 20, 17, 2, 12), (1, 144, 160, 16, 8), (5, 16, 32, 4, 2),
               (1, 72, 112, 24, 40)]


@pytest.mark.parametrize("cfg", ["", "2", "3", "4", "2w", "3w", "4w"])
@pytest.mark.parametrize("shape", WINO_SHAPES)
def test_wino3x3_fwd_dgrad_mask_pool(shape, cfg, monkeypatch):
    """mg_wino3x3 (Winograd F(2x2,3x3)) against fp64 conv2d: forward + bias + LeakyReLU (+ fused AvgPool2d), plain, data
    gradient through the transposed/flipped pack, and the masked (tangent / dgrad) epilogue -- for every out-channel tiling."""
    ops = _ops()
    if cfg.endswith("w"):  # the 64-tile, one-workgroup-per-CU variant (picked by itself only for grids that fill the chip)
        monkeypatch.setenv("MG_WINO_WT", "4")
        cfg = cfg[:-1]
    if cfg:
        monkeypatch.setenv("MG_WINO_CFG", cfg)
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    b = torch.randn(co, generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1), 0.2)
    up = ops.pack_wino3x3(wt.to(DEV), dgrad=False)
    y, q = ops.conv3x3(x.to(DEV), None, b.to(DEV), co, lrelu=True, pool=True, wino=up)
    report("wino fwd", y, ref, 2e-6)
    report("wino fwd pooled", q, F.avg_pool2d(ref, 2), 2e-6)
    y2 = ops.conv3x3(x.to(DEV), None, None, co, wino=up)
    report("wino plain", y2, F.conv2d(x.double(), wt.double(), None, padding=1), 2e-6)
    gy = torch.randn(n, co, h, w, generator=g)
    act = torch.randn(n, ci, h, w, generator=g)
    refd = F.conv_transpose2d(gy.double(), wt.double(), padding=1)
    upd = ops.pack_wino3x3(wt.to(DEV), dgrad=True)
    gx = ops.conv3x3(gy.to(DEV), None, None, ci, wino=upd)
    report("wino dgrad", gx, refd, 2e-6)
    buf = act.to(DEV).clone()
    gxm, gq = ops.conv3x3(gy.to(DEV), None, None, ci, mask_aux=buf, out=buf, pool=True, wino=upd)  # in place over the mask
    refm = refd * torch.where(act > 0, 1.0, 0.2).double()
    report("wino dgrad+mask", gxm, refm, 2e-6)
    report("wino dgrad+mask pooled", gq, F.avg_pool2d(refm, 2), 2e-6)
    # tile masks: one byte per 2x2 tile instead of the full-resolution activation.  Same arithmetic as the fp32-mask paths above
    # => bitwise the same numbers.
    m, q2 = ops.conv3x3(x.to(DEV), None, b.to(DEV), co, lrelu=True, pool=True, wino=up, mask_out=True)
    assert m.dtype == torch.uint8 and tuple(m.shape) == (n, co, h // 2, w // 2)
    assert torch.equal(q2, q)
    bits = (y > 0).reshape(n, co, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, co, h // 2, w // 2, 4).to(torch.uint8)
    want = bits[..., 0] + 2 * bits[..., 1] + 4 * bits[..., 2] + 8 * bits[..., 3]
    assert torch.equal(m, want)
    mact = (act > 0).reshape(n, ci, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, ci, h // 2, w // 2, 4).to(torch.uint8)
    mact = (mact[..., 0] + 2 * mact[..., 1] + 4 * mact[..., 2] + 8 * mact[..., 3]).to(DEV).contiguous()
    none, gq2 = ops.conv3x3(gy.to(DEV), None, None, ci, mask_aux=mact, pool=True, wino=upd)
    assert none is None and torch.equal(gq2, gq)
    # AvgPool2d backward + LeakyReLU backward fused on the data-gradient conv: mask bytes at the conv's OUTPUT resolution
    act2 = torch.randn(n, ci, 2 * h, 2 * w, generator=g)
    m2 = (act2 > 0).reshape(n, ci, h, 2, w, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, ci, h, w, 4).to(torch.uint8)
    m2 = (m2[..., 0] + 2 * m2[..., 1] + 4 * m2[..., 2] + 8 * m2[..., 3]).to(DEV).contiguous()
    gun = ops.conv3x3(gy.to(DEV), None, None, ci, wino=upd, unpool_mask=m2)
    assert torch.equal(gun, ops.avgpool2_bwd(gx, act2.to(DEV)))
    if w % 2 == 0:  # the stand-alone kernel with a tile mask handles pairs of pooled pixels
        assert torch.equal(ops.avgpool2_bwd(gx, m2), gun)


@pytest.mark.parametrize("case", [(6, 2, 48, 16, 16), (9, 2, 64, 5, 7), (6, 96, 2, 8, 8)])
def test_conv1x1_wgrad_bias_n(case):
    """`bias_n`: the weight gradient sums all samples, the bias gradient only the first bias_n (the fused critic step hands the
    stem [real | fake | tangent] in one launch; the tangent third has no bias gradient) -- against fp64 sums."""
    ops = _ops()
    n, ci, co, h, w = case
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, ci, h, w, generator=g)
    gy = torch.randn(n, co, h, w, generator=g)
    bn = 2 * n // 3
    gw = torch.full((co, ci, 1, 1), float("nan"), device=DEV)
    gb = torch.full((co,), float("nan"), device=DEV)
    ops.conv1x1_wgrad(x.to(DEV), gy.to(DEV), gw, gb, bias_n=bn)
    ref_w = torch.einsum("nohw,nchw->oc", gy.double(), x.double())
    ref_b = gy[:bn].double().sum(dim=(0, 2, 3))
    report("conv1x1 wgrad (bias_n)", gw.reshape(co, ci), ref_w, 3e-6)
    report("conv1x1 bias grad (bias_n)", gb, ref_b, 3e-6)


def _tile_mask(act: torch.Tensor) -> torch.Tensor:
    n, c, h, w = act.shape
    b = (act > 0).reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4).to(torch.uint8)
    return (b[..., 0] + 2 * b[..., 1] + 4 * b[..., 2] + 8 * b[..., 3]).contiguous()


@pytest.mark.parametrize("cfg", ["", "2", "3w", "4w"])
@pytest.mark.parametrize("shape", [(3, 64, 64, 64, 64), (2, 24, 40, 6, 10), (3, 20, 17, 2, 12), (1, 80, 64, 32, 32), (5, 16, 32, 4, 2)])
def test_wino3x3_fade_in_epilogues_equal_the_separate_kernels(shape, cfg, monkeypatch):
    """mg_wino3x3_fade: the critic's fade-in blend (forward, tangent) and its backward fused on the Winograd conv -- bitwise the
    results of the conv followed by mg_axpby / mg_blend_lrelu_bwd, and the tile mask is the sign of the new branch."""
    ops = _ops()
    from musicgan_amd import _lib
    if cfg.endswith("w"):
        monkeypatch.setenv("MG_WINO_WT", "4")
        cfg = cfg[:-1]
    if cfg:
        monkeypatch.setenv("MG_WINO_CFG", cfg)
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, ci, h, w, generator=g).to(DEV)
    wt = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)).to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    other = torch.randn(n, co, h, w, generator=g).to(DEV)
    coef = torch.tensor([0.37, 0.63], dtype=torch.float32).to(DEV)
    up = ops.pack_wino3x3(wt, dgrad=False)
    y = ops.conv3x3(x, None, b, co, lrelu=True, wino=up)
    blend, m = ops.conv3x3_fade(x, up, b, co, _lib.MG_FADE_FWD, other, coef)
    assert torch.equal(blend, ops.axpby(0.37, y, 0.63, other, coef=coef))
    assert torch.equal(m, _tile_mask(y))
    # tangent: conv * lrelu'(new branch), blended with the old branch's tangent
    t = ops.conv3x3(x, None, None, co, mask_aux=y.clone(), wino=up)
    out = torch.empty_like(t)
    got = ops.conv3x3_fade(x, up, None, co, _lib.MG_FADE_TANGENT, other, coef, mask_in=m, out=out)
    assert got is out and torch.equal(out, ops.axpby(0.37, t, 0.63, other, coef=coef))
    # backward: x plays the gradient w.r.t. the conv after the blend (co channels), result has ci channels
    gy = torch.randn(n, co, h, w, generator=g).to(DEV)
    act_new = torch.randn(n, ci, h, w, generator=g).to(DEV)
    act_old = torch.randn(n, ci, h, w, generator=g).to(DEV)
    upd = ops.pack_wino3x3(wt, dgrad=True)
    gblend = ops.conv3x3(gy, None, None, ci, wino=upd)
    ra, ro = ops.blend_lrelu_bwd(gblend, act_new, act_old, 0.37, 0.63, coef=coef)
    ga, go = ops.conv3x3_fade(gy, upd, None, ci, _lib.MG_FADE_BWD, act_old, coef, mask_in=_tile_mask(act_new))
    assert torch.equal(ga, ra) and torch.equal(go, ro)


@pytest.mark.parametrize("wt", ["", "4"])
@pytest.mark.parametrize("shape", [(2, 8, 8, 2, 2), (2, 64, 64, 32, 32), (1, 80, 48, 64, 16), (3, 24, 33, 6, 10),
                                   (2, 32, 16, 16, 16)])
def test_wino3x3_pixnorm(shape, wt, monkeypatch):
    ops = _ops()
    if wt:
        monkeypatch.setenv("MG_WINO_WT", wt)
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(22)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    b = torch.randn(co, generator=g)
    yr = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1), 0.2)
    nr = torch.sqrt(yr.pow(2).mean(dim=1, keepdim=True) + 1e-8)
    up = ops.pack_wino3x3(wt.to(DEV), dgrad=False)
    y, p, rn = ops.conv3x3(x.to(DEV), None, b.to(DEV), co, lrelu=True, pixnorm=True, wino=up)
    report("wino pn y", y, yr, 2e-6)
    report("wino pn p", p, yr / nr, 3e-6)
    report("wino pn rn", rn, 1.0 / nr, 3e-6)
    _, p2, rn2 = ops.conv3x3(x.to(DEV), None, b.to(DEV), co, lrelu=True, pixnorm=True, want_y=False, wino=up)
    assert torch.equal(p, p2) and torch.equal(rn, rn2)


def test_wino3x3_refuses_what_it_cannot_do():
    ops = _ops()
    up = ops.pack_wino3x3(torch.randn(80, 8, 3, 3, device=DEV), dgrad=False)
    x = torch.randn(1, 8, 4, 4, device=DEV)
    with pytest.raises(Exception, match="PIXNORM needs Cout <= 64"):
        ops.conv3x3(x, None, None, 80, lrelu=True, pixnorm=True, wino=up)
    with pytest.raises(Exception, match="must be even"):
        ops.conv3x3(torch.randn(1, 8, 3, 4, device=DEV), None, None, 80, wino=up)
    with pytest.raises(Exception, match="UPS_IN unsupported"):
        ops.conv3x3(x, None, None, 80, ups=True, wino=up)


WW_SHAPES = [(2, 8, 8, 2, 2), (3, 32, 128, 4, 4), (4, 128, 112, 8, 8), (2, 96, 80, 32, 32), (2, 64, 48, 64, 64),
             (1, 48, 64, 128, 128), (5, 144, 160, 2, 2), (3, 16, 32, 16, 16), (2, 24, 40, 12, 20), (7, 20, 17, 2, 12),
             (2, 80, 80, 16, 8), (9, 64, 64, 8, 8)]


@pytest.mark.parametrize("shape", WW_SHAPES)
def test_wino_wgrad_matches_autograd(shape, monkeypatch):
    """mg_wino3x3_wgrad (Winograd F(3x3,2x2), split-K + fixed-order reduce) against fp64 autograd: weight and bias gradient,
    accumulate mode, bias restricted to the first samples (the fused critic step's use), and run-to-run determinism."""
    ops = _ops()
    monkeypatch.setenv("MG_WINO_WGRAD_MIN_PIXELS", "1")
    n, ci, co, h, w = shape
    assert ops.wino_wgrad_supported(n, ci, co, h, w)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(n, ci, h, w, generator=g).double()
    gy = torch.randn(n, co, h, w, generator=g).double()
    wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x, wt, bt, padding=1) * gy).sum().backward()
    gw = torch.full((co, ci, 3, 3), 7.0, device=DEV)
    gb = torch.full((co,), 7.0, device=DEV)
    xd, gyd = x.float().to(DEV), gy.float().to(DEV)
    ops.conv3x3_wgrad(xd, gyd, gw, gb)
    report("wino wgrad gw", gw, wt.grad, 3e-6)
    report("wino wgrad gb", gb, bt.grad, 3e-6)
    gw2 = torch.empty_like(gw)
    ops.conv3x3_wgrad(xd, gyd, gw2, None)
    assert torch.equal(gw, gw2)  # deterministic, and gb is optional
    ops.conv3x3_wgrad(xd, gyd, gw, gb, accumulate=True)
    report("wino wgrad gw acc", gw, 2 * wt.grad, 3e-6)
    report("wino wgrad gb acc", gb, 2 * bt.grad, 3e-6)
    if n > 1:
        nb = n // 2
        gb3 = torch.empty(co, device=DEV)
        ops.conv3x3_wgrad(xd, gyd, gw2, gb3, bias_n=nb)
        report("wino wgrad gb first samples", gb3, gy[:nb].sum(dim=(0, 2, 3)), 3e-6)
    # agrees with the direct kernel
    monkeypatch.setenv("MG_WINO_WGRAD", "0")
    gwd = torch.empty_like(gw2)
    ops.conv3x3_wgrad(xd, gyd, gwd, None)
    report("wino vs direct wgrad", gw2, gwd.double(), 3e-6)


@pytest.mark.parametrize("case", [(3, 16, 32, 64, 64, False, 2), (2, 48, 64, 32, 48, False, 0), (5, 96, 80, 16, 16, False, 3),
                                  (2, 32, 16, 64, 32, True, 0), (3, 64, 48, 32, 32, True, 1), (1, 112, 96, 16, 32, False, 0)])
def test_wino_wgrad_scalar_addressed_loads_equal_the_general_form(case, monkeypatch):
    """(N, Cin, Cout, H, W, ups, bias_n): on maps at least 16 wide the weight-gradient kernels address their loads per chunk instead of
    per lane (wino_wgrad.hip, FAST); the loads fetch the same values, so every sum has the same bits as with MG_WGRAD_FAST=0."""
    ops = _ops()
    monkeypatch.setenv("MG_WINO_WGRAD_MIN_PIXELS", "1")
    n, ci, co, h, w, ups, bias_n = case
    g = torch.Generator(device=DEV).manual_seed(17)
    x = torch.randn(n, ci, h // 2 if ups else h, w // 2 if ups else w, device=DEV, generator=g)
    gy = torch.randn(n, co, h, w, device=DEV, generator=g)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MG_WGRAD_FAST", mode)
        gw, gb = torch.empty(co, ci, 3, 3, device=DEV), torch.empty(co, device=DEV)
        ops.conv3x3_wgrad(x, gy, gw, gb, ups=ups, bias_n=bias_n)
        res[mode] = (gw, gb)
    assert torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][1], res["1"][1])


def test_blend_lrelu_bwd_equals_the_four_kernels_it_replaces():
    ops = _ops()
    g = torch.Generator().manual_seed(41)
    gr = torch.randn(3, 7, 5, 6, generator=g).to(DEV)
    a = torch.randn(3, 7, 5, 6, generator=g).to(DEV)
    o = torch.randn(3, 7, 5, 6, generator=g).to(DEV)
    oa, oo = ops.blend_lrelu_bwd(gr, a, o, 0.37, 0.63)
    assert torch.equal(oa, ops.lrelu_bwd(ops.axpby(0.37, gr), a)) and torch.equal(oo, ops.lrelu_bwd(ops.axpby(0.63, gr), o))
    gr4, a4, o4 = gr.reshape(-1)[:628], a.reshape(-1)[:628].clone(), o.reshape(-1)[:628].clone()  # vector path (n % 4 == 0)
    oa, oo = ops.blend_lrelu_bwd(gr4.clone(), a4, o4, 0.5, 0.5)
    assert torch.equal(oa, ops.lrelu_bwd(ops.axpby(0.5, gr4.clone()), a4))


@pytest.mark.parametrize("shape", [(33, 48, 64, 64), (9, 112, 128, 128), (8, 64, 130, 126)])
def test_pixelnorm_bwd_lds_path_is_bitwise_the_two_pass_kernel(shape, monkeypatch):
    """Large maps take the single-HBM-pass form (operands parked in LDS): same arithmetic in the same order as the two-pass
    kernel, so the outputs must be bit-identical -- both for C <= 64 (256 threads) and C > 64 (128 threads), ragged last block."""
    ops = _ops()
    n, c, h, w = shape
    g = torch.Generator(device=DEV).manual_seed(51)
    gp = torch.randn(n, c, h, w, device=DEV, generator=g)
    p = torch.randn(n, c, h, w, device=DEV, generator=g)
    rn = torch.rand(n, 1, h, w, device=DEV, generator=g) + 0.5
    for from_p in (False, True):
        monkeypatch.delenv("MG_PN_BWD_NOLDS", raising=False)
        a = ops.pixelnorm_lrelu_bwd(gp, p, rn, from_p=from_p)
        monkeypatch.setenv("MG_PN_BWD_NOLDS", "1")
        b = ops.pixelnorm_lrelu_bwd(gp, p, rn, from_p=from_p)
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [(3, 37, 5, 7), (32, 128, 4, 4), (1, 160, 2, 2), (64, 112, 8, 8), (16, 96, 32, 32), (2, 33, 16, 16)])
def test_pixelnorm_bwd_small_maps(shape):
    """Maps of <= 16 k pixels take the split-channel kernel (4 waves share a pixel's channels): against the fp64 formula, both
    conventions of `y` (post-LeakyReLU activation / the normalised output itself), ragged pixel and channel counts."""
    ops = _ops()
    n, c, h, w = shape
    g = torch.Generator().manual_seed(52)
    gp = torch.randn(n, c, h, w, generator=g)
    yv = torch.randn(n, c, h, w, generator=g)
    for from_p in (False, True):
        y64 = yv.double()
        if from_p:  # y is p = act * rn: any positive rn is consistent
            rn64 = torch.rand(n, 1, h, w, generator=g).double() + 0.5
            t = y64
        else:
            rn64 = 1.0 / torch.sqrt((y64 * y64).mean(dim=1, keepdim=True) + 1e-8)
            t = y64 * rn64
        dot = (gp.double() * t).mean(dim=1, keepdim=True)
        ref = torch.where(y64 > 0, 1.0, 0.2) * rn64 * (gp.double() - t * dot)
        got = ops.pixelnorm_lrelu_bwd(gp.to(DEV), yv.to(DEV), rn64.float().to(DEV), from_p=from_p)
        report(f"pixelnorm bwd small from_p={from_p}", got, ref, 2e-6)


@pytest.mark.parametrize("case", [(3, 512, 128, torch.float64), (2, 512, 4, torch.float64), (2, 512, 512, torch.float32),
                                  (2, 96, 32, torch.float64), (1, 100, 30, torch.float32), (2, 512, 64, torch.float32)])
def test_input_transform_matches_the_tensor_expressions(case):
    """mg_input_transform == ChannelMinMaxNorm -> ChangeRange(-1,1) -> Resize(S) evaluated by torch on the CPU (Resize = the
    bilinear + antialias interpolate torchvision's tensor path calls), for every growth level's scale and a non-power-of-two one."""
    ops = _ops()
    from musicgan_amd import audio
    n, hw, side, dt = case
    g = torch.Generator().manual_seed(61)
    x = (torch.rand(n, 2, hw, hw, generator=g, dtype=torch.float64) * 7 - 2).to(dt)
    x[0, 1] *= 1e-3  # a channel with a very different range
    xf = x.to(torch.float32)
    ref = audio.ChangeRange(-1.0, 1.0)(audio.ChannelMinMaxNorm()(xf))
    if side != hw:
        ref = F.interpolate(ref, size=(side, side), mode="bilinear", antialias=True, align_corners=False)
    got = ops.input_transform(x.to(DEV), side)
    assert got.dtype == torch.float32 and tuple(got.shape) == (n, 2, side, side)
    report("input transform", got, ref.double(), 3e-6)


def test_input_transform_matches_reference_fixture():
    """mg_input_transform against tests/golden/transforms.npz: the REFERENCE's ChannelMinMaxNorm / ChangeRange outputs
    (audio/transforms.py:4-40; `side` == input size makes the resize the identity, so this part is pinned on the reference alone,
    incl. the constant channel that divides by eps) and its Grower.scale_transform at levels 0, 3, 5 on a stored float64 sample
    (Resize through the torchvision stand-in: pinned on aten's bilinear + antialias)."""
    ops = _ops()
    from golden_util import load
    g = load("transforms.npz")
    x64 = torch.from_numpy(g["x64"])
    got = ops.input_transform(x64.to(DEV), 64)
    assert float((got.cpu() - torch.from_numpy(g["ranged"])).abs().max()) <= 5e-7
    got32 = ops.input_transform(x64.float().to(DEV), 64)
    assert torch.equal(got32, got)  # train.py:139 casts to float first: feeding float64 or float32 gives the same result
    big = torch.from_numpy(g["big32"]).double().to(DEV)
    for level, side in ((0, 4), (3, 32), (5, 128)):
        y = ops.input_transform(big, side)
        assert float((y.cpu() - torch.from_numpy(g[f"scaled_l{level}"])).abs().max()) <= 3e-6, level


def test_pack_multi_equals_single_tensor_packs():
    """mg_pack_multi (all layouts of a list of weights in one launch) writes bit for bit what the four single-tensor pack entry points
    write, for forward and data-gradient variants, odd channel counts and more records than one launch carries by value (64)."""
    ops = _ops()
    from musicgan_amd import _lib
    g = torch.Generator().manual_seed(71)
    reqs, refs = [], []
    shapes = [(48, 64), (64, 48), (80, 96), (144, 160), (32, 8), (16, 16), (112, 128)]
    for rep in range(4):
        for co, ci in shapes:
            w = torch.randn(co, ci, 3, 3, generator=g).to(DEV)
            singles = [(_lib.MG_PACK_CONV3X3, False, ops.pack_conv3x3(w, False)), (_lib.MG_PACK_CONV3X3, True, ops.pack_conv3x3(w, True)),
                       (_lib.MG_PACK_WINO3X3, False, ops.pack_wino3x3(w, False)), (_lib.MG_PACK_WINO3X3, True, ops.pack_wino3x3(w, True)),
                       (_lib.MG_PACK_UPCONV3X3, False, ops.pack_upconv3x3(w)), (_lib.MG_PACK_UPCONV3X3_DGRAD, False, ops.pack_upconv3x3_dgrad(w))]
            for kind, dgrad, ref in singles[rep::2]:
                out = torch.full((ops.packed_floats(kind, co, ci, dgrad),), float("nan"), device=DEV)
                assert out.numel() == ref.numel()
                reqs.append((kind, w, dgrad, out))
                refs.append(ref)
    assert len(reqs) > 64
    ops.pack_multi(reqs)
    for (kind, w, dgrad, out), ref in zip(reqs, refs):
        assert torch.equal(out, ref), (kind, tuple(w.shape), dgrad)


# ------------------------------------------------------------------ round-2 kernels on ragged shapes
@pytest.mark.parametrize("shape", [(3, 100, 40, 4, 4), (5, 72, 24, 2, 2), (96, 128, 128, 4, 4), (7, 136, 150, 3, 5), (2, 64, 16, 8, 8)])
def test_conv3x3_split_k_small_maps(shape, monkeypatch):
    """The split-K variant of the direct conv (maps <= 8x8, <= 1536 pixels, >= 8 chunks): ragged last chunk, ragged out-channel tile,
    odd map sizes -- against F.conv2d in fp64, bit-identical layout paths with bias + LeakyReLU and with the mask epilogue."""
    ops = _ops()
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(81)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    b = torch.randn(co, generator=g)
    wp = ops.pack_conv3x3(wt.to(DEV), dgrad=False)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1), 0.2)
    report("split-K conv + bias + lrelu", ops.conv3x3(x.to(DEV), wp, b.to(DEV), co, lrelu=True), ref, 1e-5)
    aux = torch.randn(n, co, h, w, generator=g)
    ref = F.conv2d(x.double(), wt.double(), None, padding=1) * torch.where(aux > 0, 1.0, 0.2).double()
    report("split-K conv + mask", ops.conv3x3(x.to(DEV), wp, None, co, mask_aux=aux.to(DEV)), ref, 1e-5)
    monkeypatch.setenv("MG_CONV_NOKSPLIT", "1")
    a = ops.conv3x3(x.to(DEV), wp, b.to(DEV), co, lrelu=True)
    monkeypatch.delenv("MG_CONV_NOKSPLIT")
    report("split-K vs plain direct conv", ops.conv3x3(x.to(DEV), wp, b.to(DEV), co, lrelu=True), a.double(), 2e-6)


@pytest.mark.parametrize("shape", [(3, 37, 5, 7), (32, 128, 4, 4), (1, 160, 2, 2), (64, 112, 8, 8), (2, 33, 16, 16)])
def test_pixelnorm_fwd_small_maps(shape):
    """The split-channel PixelNorm kernel (<= 2^17 pixels): pixel counts that do not fill a 64-pixel workgroup, channel counts that
    are no multiple of 4."""
    ops = _ops()
    g = torch.Generator().manual_seed(82)
    y = torch.randn(*shape, generator=g) * 3
    p, rn = ops.pixelnorm_fwd(y.to(DEV))
    ref_rn = 1.0 / torch.sqrt(y.double().pow(2).mean(dim=1, keepdim=True) + 1e-8)
    report("pixelnorm p", p, y.double() * ref_rn, 2e-6)
    report("pixelnorm rn", rn, ref_rn, 2e-6)


@pytest.mark.parametrize("case", [(3, 37, 2, 5, 9), (32, 64, 2, 64, 64), (2, 160, 1, 7, 7), (4, 16, 4, 33, 3)])
def test_conv1x1_few_out_split_channels(case):
    """The few-out 1x1 conv on < 2^20-pixel maps (4 waves split the input channels): odd pixel counts, Cin not a multiple of 4,
    with tanh (the generator head) and with the input-side mask (the critic's input gradient)."""
    ops = _ops()
    n, ci, co, h, w = case
    g = torch.Generator().manual_seed(83)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 1, 1, generator=g) / math.sqrt(ci)
    b = torch.randn(co, generator=g)
    ref = torch.tanh(F.conv2d(x.double(), wt.double(), b.double()))
    report("few-out + tanh", ops.conv1x1(x.to(DEV), wt.to(DEV), b.to(DEV), co, tanh=True), ref, 2e-6)
    aux = torch.randn(n, ci, h, w, generator=g)
    wt_t = torch.randn(ci, co, 1, 1, generator=g) / math.sqrt(ci)  # module weight of a co -> ci conv, applied transposed
    ref = F.conv2d(x.double() * torch.where(aux > 0, 1.0, 0.2).double(), wt_t.double().permute(1, 0, 2, 3))
    report("few-out transposed + input mask", ops.conv1x1(x.to(DEV), wt_t.to(DEV), None, co, transposed=True, mask_aux=aux.to(DEV)),
           ref, 2e-6)


def test_group_means_and_deferred_wgrad_reduce(monkeypatch):
    """mg_group_means (score means + Wasserstein losses, one launch) against torch, and the deferred one-launch weight-gradient
    reduction (Winograd and direct jobs mixed in one sweep) bit-identical to the immediate form (layers launched one by one:
    a grouped launch chooses other split counts, test_grouped_wgrad_launch)."""
    ops = _ops()
    monkeypatch.setenv("MG_WGRAD_GROUP", "0")
    g = torch.Generator().manual_seed(84)
    for groups, n in ((3, 5), (3, 64), (1, 300), (2, 1000)):
        s = torch.randn(groups * n, 1, generator=g) * 7
        out = ops.group_means(s.to(DEV), groups).cpu().double()
        m = s.double().reshape(groups, n).mean(dim=1)
        assert float((out[:groups] - m).abs().max()) <= 1e-6 * float(m.abs().max() + 1)
        want = float(m[1] - m[0]) if groups >= 2 else float(-m[0])
        assert abs(float(out[groups]) - want) <= 2e-6 * (abs(want) + 1)
    defer = ops.WgradDefer()
    layers = [(6, 48, 64, 64, 64, False), (6, 16, 32, 3, 5, False), (6, 64, 48, 64, 64, True), (6, 96, 112, 16, 16, False),
              (6, 144, 160, 2, 2, False), (6, 16, 32, 4, 4, False)]  # odd / < 64-pixel maps take the direct form
    want, got = [], []
    for n, ci, co, h, w, ups in layers:
        x = torch.randn(n, ci, h // 2 if ups else h, w // 2 if ups else w, generator=g).to(DEV)
        gy = torch.randn(n, co, h, w, generator=g).to(DEV)
        gw0, gb0 = torch.empty(co, ci, 3, 3, device=DEV), torch.empty(co, device=DEV)
        ops.conv3x3_wgrad(x, gy, gw0, gb0, ups=ups, bias_n=4)
        gw1, gb1 = torch.full((co, ci, 3, 3), float("nan"), device=DEV), torch.full((co,), float("nan"), device=DEV)
        ops.conv3x3_wgrad(x, gy, gw1, gb1, ups=ups, bias_n=4, defer=defer)
        want.append((gw0, gb0))
        got.append((gw1, gb1))
    assert len(defer._jobs) >= 2 and len(defer._jobs_d) >= 2  # both kernel families took part
    defer.flush()
    for (a, b), (c, d) in zip(want, got):
        assert torch.equal(a, c) and torch.equal(b, d)


@pytest.mark.parametrize("n", [24, 5])
def test_grouped_wgrad_launch(n, monkeypatch):
    """mg_wino3x3_wgrad_partial_multi: the weight gradients of a whole sweep, small layers with equal block shapes sharing one
    launch (splits sized for the group), large ones launched alone -- every layer against fp64 autograd (weight, bias of the first
    samples), against the one-launch-per-layer form, and twice for determinism.  Shapes: the <= 16x16 end of both networks
    (generator.py:15-40 incl. up-sampled inputs, discriminator.py:14-34), a 128x128 layer that must stay alone, a ragged one."""
    ops = _ops()
    g = torch.Generator().manual_seed(91)
    layers = [(144, 160, 2, 2, False), (144, 144, 4, 4, False), (128, 144, 4, 4, False), (128, 128, 8, 8, False),
              (112, 128, 8, 8, False), (128, 112, 8, 8, True), (112, 112, 8, 8, False), (96, 112, 16, 16, False),
              (112, 96, 16, 16, True), (96, 96, 16, 16, False), (48, 64, 128, 128, False), (100, 120, 6, 10, False),
              (160, 144, 4, 4, True), (144, 160, 2, 2, False)]
    data = []
    for ci, co, h, w, ups in layers:
        nn = 2 if h >= 128 else n
        x = torch.randn(nn, ci, h // 2 if ups else h, w // 2 if ups else w, generator=g)
        gy = torch.randn(nn, co, h, w, generator=g)
        xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
        wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
        (F.conv2d(xin.double(), wt, None, padding=1) * gy.double()).sum().backward()
        nb = max(1, nn // 3)
        data.append((x.to(DEV), gy.to(DEV), ups, nb, wt.grad, gy[:nb].double().sum(dim=(0, 2, 3))))

    def sweep(group):
        monkeypatch.setenv("MG_WGRAD_GROUP", str(group))
        defer = ops.WgradDefer()
        outs = []
        for x, gy, ups, nb, _, _ in data:
            gw = torch.full((gy.shape[1], x.shape[1], 3, 3), float("nan"), device=DEV)
            gb = torch.full((gy.shape[1],), float("nan"), device=DEV)
            ops.conv3x3_wgrad(x, gy, gw, gb, ups=ups, bias_n=nb, defer=defer)
            outs.append((gw, gb))
        if group > 0:  # (layers of fewer than 64 pixels take the direct form at once)
            n_wino = sum(ops.wino_wgrad_supported(gy.shape[0], x.shape[1], gy.shape[1], gy.shape[2], gy.shape[3], ups=u)
                         for x, gy, u, _, _, _ in data)
            assert len(defer._lazy) == n_wino >= 10 and not defer._jobs
        defer.flush()
        return outs

    grouped, again, single = sweep(32), sweep(32), sweep(0)
    for (gw, gb), (gw2, gb2), (gw1, gb1), (_, _, _, _, want_w, want_b) in zip(grouped, again, single, data):
        assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
        report("grouped wgrad gw", gw, want_w, 3e-6)
        report("grouped wgrad gb", gb, want_b, 3e-6)
        report("grouped vs single gw", gw, gw1.double(), 2e-6)
        assert torch.equal(gb.cpu(), gb1.cpu()) or float((gb - gb1).abs().max()) <= 2e-6 * float(gb1.abs().max())


@pytest.mark.parametrize("n,ci,co", [(24, 160, 160), (192, 160, 144), (5, 17, 33), (1, 8, 8)])
def test_conv3x3_wgrad_on_1x1_maps(n, ci, co, monkeypatch):
    """mg_conv3x3_wgrad_1x1map (the last critic conv's weight gradient, discriminator.py:14-34): centre tap = gy^T x, the other eight
    taps exactly zero, bias gradient over the first samples only, accumulate mode, inside a deferred sweep (complete before the
    flush), and against the MFMA kernel it replaces."""
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    x, gy = torch.randn(n, ci, 1, 1, generator=g), torch.randn(n, co, 1, 1, generator=g)
    want = gy.double().reshape(n, co).t() @ x.double().reshape(n, ci)
    nb = max(1, n // 3)
    gw, gb = torch.full((co, ci, 3, 3), float("nan"), device=DEV), torch.full((co,), float("nan"), device=DEV)
    defer = ops.WgradDefer()
    ops.conv3x3_wgrad(x.to(DEV), gy.to(DEV), gw, gb, bias_n=nb, defer=defer)
    assert not defer._lazy and not defer._jobs and not defer._jobs_d
    report("1x1-map wgrad centre tap", gw[:, :, 1, 1], want, 1e-6)
    rest = gw.clone()
    rest[:, :, 1, 1] = 0
    assert float(rest.abs().max()) == 0.0
    report("1x1-map wgrad bias", gb, gy[:nb].double().sum(dim=(0, 2, 3)), 1e-6)
    ops.conv3x3_wgrad(x.to(DEV), gy.to(DEV), gw, None, accumulate=True)
    report("1x1-map wgrad accumulate", gw[:, :, 1, 1], 2 * want, 1e-6)
    monkeypatch.setenv("MG_WGRAD_1X1MAP", "0")
    gw0 = torch.empty_like(gw)
    ops.conv3x3_wgrad(x.to(DEV), gy.to(DEV), gw0, None)
    report("1x1-map wgrad vs the MFMA kernel", gw0, gw.double() / 2, 3e-6)
