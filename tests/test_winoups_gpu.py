"""GPU parity of the 9-component Winograd kernels for `nn.Upsample(x2) -> nn.Conv2d(3x3)` (csrc/wino_ups.hip, round 6) through the C
ABI: forward (+ bias, LeakyReLU(0.2), PixelNorm: /root/reference/music_gan/networks/generator.py:24-39, layers.py:11-17) against
fp64 `F.interpolate(nearest) -> F.conv2d` and against the sub-pixel kernel it replaces; data gradient against fp64 autograd."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from musicgan_amd import ops
    return ops


def _rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


# (N, Cin, Cout, Hin, Win): one .. four out-channel tiles, 2 .. 8 chunks, several tile blocks per row, image edges inside a workgroup
DGRAD_ONLY = [(32, 80, 64, 16, 32), (64, 96, 80, 8, 32), (43, 96, 48, 8, 48)]  # five / six tiles of layer input channels: two launches
SHAPES = [(2, 64, 48, 16, 16), (1, 48, 32, 8, 32), (3, 32, 16, 8, 16), (2, 16, 64, 8, 16), (1, 64, 48, 64, 64), (2, 24, 32, 8, 16),
          (5, 64, 64, 8, 48)]


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_matches_fp64_and_the_subpixel_kernel(shape):
    ops = _ops()
    n, ci, co, h, w = shape
    assert ops.winoups3x3_supported(n, ci, co, h, w)
    g = torch.Generator().manual_seed(71)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    b = torch.randn(co, generator=g)
    pre = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), wt.double(), b.double(), padding=1)
    act = F.leaky_relu(pre, 0.2)
    rn_ref = 1.0 / torch.sqrt((act * act).mean(dim=1, keepdim=True) + 1e-8)
    up = ops.pack_winoups3x3(wt.to(DEV), False)
    xd, bd = x.to(DEV), b.to(DEV)
    y = ops.winoups3x3(xd, up, bd, co, lrelu=True)
    assert _rel(y, act) <= 2e-6
    y2 = ops.winoups3x3(xd, up, None, co)
    assert _rel(y2, pre - b.double().view(1, -1, 1, 1)) <= 2e-6
    yy, p, rn = ops.winoups3x3(xd, up, bd, co, lrelu=True, pixnorm=True)
    assert torch.equal(yy, y)
    assert _rel(p, act * rn_ref) <= 3e-6 and _rel(rn, rn_ref) <= 3e-6
    _, p2, rn2 = ops.winoups3x3(xd, up, bd, co, lrelu=True, pixnorm=True, want_y=False)
    assert torch.equal(p2, p) and torch.equal(rn2, rn)  # deterministic; y optional
    if ops.upconv3x3_supported(co, w, x.numel()):
        _, ps, rns = ops.upconv3x3(xd, ops.pack_upconv3x3(wt.to(DEV)), bd, co, lrelu=True, pixnorm=True, want_y=False)
        assert _rel(p, ps) <= 3e-6 and _rel(rn, rns) <= 3e-6


@pytest.mark.parametrize("shape", SHAPES + DGRAD_ONLY)
def test_data_gradient_matches_autograd(shape):
    ops = _ops()
    n, ci, co, h, w = shape
    if not ops.winoups3x3_supported(n, ci, co, h, w, dgrad=True):
        pytest.skip("data gradient: 16..64 input channels of the layer in whole tiles")
    g = torch.Generator().manual_seed(73)
    x = torch.randn(n, ci, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    gy = torch.randn(n, co, 2 * h, 2 * w, generator=g)
    (F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt.double(), None, padding=1) * gy.double()).sum().backward()
    upd = ops.pack_winoups3x3(wt.to(DEV), True)
    gx = ops.winoups3x3_dgrad(gy.to(DEV), upd, ci)
    assert _rel(gx, x.grad) <= 3e-6
    assert torch.equal(gx, ops.winoups3x3_dgrad(gy.to(DEV), upd, ci))
    # the same with the PixelNorm + LeakyReLU backward of the layer below in the epilogue
    pl = torch.randn(n, ci, h, w, generator=g).to(DEV)
    rnl = (torch.rand(n, 1, h, w, generator=g) + 0.5).to(DEV)
    if ci <= 64:
        assert ops.winoups3x3_dgrad_pn_supported(n, ci, co, h, w)
        gpre = ops.winoups3x3_dgrad_pn(gy.to(DEV), upd, pl, rnl, ci)
        ref = ops.pixelnorm_lrelu_bwd(gx, pl, rnl, from_p=True)
        assert _rel(gpre, ref.double().cpu()) <= 2e-6
    else:
        assert not ops.winoups3x3_dgrad_pn_supported(n, ci, co, h, w)


def test_unsupported_shapes_are_refused():
    ops = _ops()
    assert not ops.winoups3x3_supported(2, 96, 80, 32, 32)          # 80 out-channels: five tiles
    assert not ops.winoups3x3_supported(2, 80, 64, 32, 32)          # the filter bank of 10 chunks x 4 tiles exceeds the LDS
    assert not ops.winoups3x3_supported(2, 64, 48, 16, 8)           # low-res rows of 8 pixels
    assert not ops.winoups3x3_supported(2, 64, 48, 4, 16)           # fewer than 8 low-res rows
    with pytest.raises(Exception, match="unsupported shape"):
        ops.winoups3x3(torch.randn(2, 64, 16, 8, device=DEV), torch.empty(8, device=DEV), None, 48)


@pytest.mark.parametrize("shape", [(2, 64, 48, 16, 16), (1, 48, 32, 8, 32), (3, 32, 16, 8, 16), (2, 16, 48, 8, 16)])
def test_forward_with_the_head_in_the_epilogue(shape):
    """mg_winoups3x3_head: p, rn as mg_winoups3x3's (same bits) and mp = tanh(conv1x1(p)) against the 1x1 kernel on that p and against
    fp64 (generator.py:118-126 on the block output of :24-39); then the old head + blend from the given mp against mg_head_pair."""
    ops = _ops()
    n, ci, co, h, w = shape
    assert ops.winoups3x3_head_supported(n, ci, co, h, w) and not ops.winoups3x3_head_supported(n, 16, 64, h, w)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    b = torch.randn(co, generator=g) * 0.1
    hw = torch.randn(2, co, 1, 1, generator=g) / math.sqrt(co)
    hb = torch.randn(2, generator=g) * 0.1
    xd, bd, hwd, hbd = x.to(DEV), b.to(DEV), hw.to(DEV), hb.to(DEV)
    up = ops.pack_winoups3x3(wt.to(DEV), False)
    _, p0, rn0 = ops.winoups3x3(xd, up, bd, co, lrelu=True, pixnorm=True, want_y=False)
    y1, p1, rn1, mp = ops.winoups3x3_head(xd, up, bd, co, hwd, hbd, want_y=True)
    assert torch.equal(p0, p1) and torch.equal(rn0, rn1) and y1 is not None
    ref_k = ops.conv1x1(p0, hwd, hbd, 2, tanh=True)
    assert float((mp - ref_k).abs().max()) <= 2e-6
    ref = torch.tanh(F.conv2d(p0.double().cpu(), hw.double(), hb.double()))
    assert float((mp.double().cpu() - ref).abs().max()) <= 2e-6
    mp2 = torch.empty_like(mp)
    _, _, _, mp3 = ops.winoups3x3_head(xd, up, bd, co, hwd, None, mp_out=mp2)
    assert mp3 is mp2 and float((mp2 - ops.conv1x1(p0, hwd, None, 2, tanh=True)).abs().max()) <= 2e-6
    # the fade-in pair from the given mp
    cl = 24
    xl = torch.randn(n, cl, h, w, generator=g).to(DEV)
    wo, bo = (torch.randn(2, cl, 1, 1, generator=g) / math.sqrt(cl)).to(DEV), (torch.randn(2, generator=g) * 0.1).to(DEV)
    coef = torch.tensor([0.3, 0.7], device=DEV)
    out0, mp0, old0 = ops.head_pair(p0, hwd, hbd, xl, wo, bo, 0.3, 0.7, coef=coef)
    out1, old1 = ops.head_pair_from_mp(mp0, xl, wo, bo, 0.3, 0.7, coef=coef)
    assert torch.equal(out0, out1) and torch.equal(old0, old1)
    out2, old2 = ops.head_pair_from_mp(mp0, xl, wo, bo, 0.3, 0.7, save=False)
    assert old2 is None and float((out2 - out0).abs().max()) <= 1e-6
