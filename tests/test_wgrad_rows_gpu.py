"""The row-staged Winograd weight-gradient kernel (wino_wgrad.hip: wino_wgrad_rows_mfma, round 6) through the C ABI: against fp64
autograd of `F.conv2d` -- the reference's `aten::convolution_backward` (weight, bias) for nn.Conv2d(3x3, padding=1),
/root/reference/music_gan/networks/generator.py:15-40, discriminator.py:14-34 -- and against the chunk-staged kernels it replaces
(MG_WGRAD_ROWS=0), for every block shape <CT, OT> and the edge cases of its addressing: one stage per row (W = 32: both halo pixels
outside the image), two-row maps (every stage touches the top AND the bottom edge), channel counts that are not multiples of 16,
slabs that end inside an image, the bias gradient restricted to the first samples."""
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


# (N, Cin, Cout, H, W): block shapes <CT, OT> from <1,1> to <4,3> / <3,4> (<4,4> stays on the chunk-staged kernel), incl. channel blocks (80 = 2 x 3 tiles, 144 = 3 x 3) and ragged channels
ROWS_SHAPES = [(2, 16, 16, 32, 32), (3, 16, 32, 64, 64), (2, 32, 48, 64, 32), (1, 16, 64, 32, 64), (2, 32, 16, 32, 32), (2, 32, 32, 64, 64),
               (1, 32, 64, 32, 32), (2, 48, 16, 32, 32), (2, 48, 32, 32, 64), (3, 48, 48, 32, 32), (2, 48, 64, 64, 64), (2, 64, 16, 32, 32),
               (1, 64, 32, 32, 32), (2, 64, 48, 32, 64), (3, 64, 64, 32, 32), (2, 80, 80, 32, 32), (1, 144, 96, 32, 32), (5, 24, 40, 2, 32),
               (2, 20, 17, 4, 64), (1, 48, 64, 128, 128), (7, 64, 80, 6, 96)]


@pytest.mark.parametrize("shape", ROWS_SHAPES)
def test_rows_kernel_matches_autograd_and_the_chunk_kernels(shape, monkeypatch):
    ops = _ops()
    monkeypatch.setenv("MG_WINO_WGRAD_MIN_PIXELS", "1")
    n, ci, co, h, w = shape
    assert ops.wino_wgrad_supported(n, ci, co, h, w)
    g = torch.Generator().manual_seed(61)
    x = torch.randn(n, ci, h, w, generator=g).double()
    gy = torch.randn(n, co, h, w, generator=g).double()
    wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x, wt, bt, padding=1) * gy).sum().backward()
    xd, gyd = x.float().to(DEV), gy.float().to(DEV)
    res = {}
    for mode in ("0", "2"):  # 0: chunk-staged kernels; 2: the row-staged kernel for every block shape
        monkeypatch.setenv("MG_WGRAD_ROWS", mode)
        gw = torch.full((co, ci, 3, 3), 7.0, device=DEV)
        gb = torch.full((co,), 7.0, device=DEV)
        ops.conv3x3_wgrad(xd, gyd, gw, gb)
        assert _rel(gw, wt.grad) <= 3e-6, (mode, _rel(gw, wt.grad))
        assert _rel(gb, bt.grad) <= 3e-6, (mode, _rel(gb, bt.grad))
        gw2 = torch.empty_like(gw)
        ops.conv3x3_wgrad(xd, gyd, gw2, None)
        assert torch.equal(gw, gw2)  # deterministic; gb optional
        res[mode] = gw
        if n > 1:
            nb = n // 2
            gb3 = torch.empty(co, device=DEV)
            ops.conv3x3_wgrad(xd, gyd, gw2, gb3, bias_n=nb)
            assert _rel(gb3, gy[:nb].sum(dim=(0, 2, 3))) <= 3e-6, mode
            assert torch.equal(gw, gw2)
        ops.conv3x3_wgrad(xd, gyd, gw2, gb, accumulate=True)
        assert _rel(gw2, 2 * wt.grad) <= 3e-6 and _rel(gb, 2 * bt.grad) <= 3e-6, mode
    # the two kernels sum the same products in different orders: a few ulps of the largest element apart
    assert _rel(res["2"], res["0"]) <= 2e-6


# (N, Cin, Cout, H, W) of gy; x is (N, Cin, H/2, W/2) and the convolution input its nearest x2 up-sampling (generator.py:24-31)
UPS_SHAPES = [(2, 64, 48, 32, 32), (1, 80, 64, 64, 64), (3, 32, 16, 32, 64), (2, 48, 32, 64, 32), (2, 32, 32, 4, 32), (1, 64, 48, 128, 128),
              (5, 24, 40, 2, 32), (2, 96, 80, 32, 32), (7, 64, 64, 6, 96)]


@pytest.mark.parametrize("shape", UPS_SHAPES)
def test_rows_kernel_with_upsampled_input(shape, monkeypatch):
    """The up-sampled-input form (9 of the 16 Winograd components are identically zero there and are not computed) against fp64 autograd
    through `F.interpolate(nearest)` + `F.conv2d`, and against the chunk-staged kernels (MG_WGRAD_ROWS=0)."""
    ops = _ops()
    monkeypatch.setenv("MG_WINO_WGRAD_MIN_PIXELS", "1")
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(67)
    x = torch.randn(n, ci, h // 2, w // 2, generator=g).double()
    gy = torch.randn(n, co, h, w, generator=g).double()
    wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    (F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, bt, padding=1) * gy).sum().backward()
    xd, gyd = x.float().to(DEV), gy.float().to(DEV)
    res = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("MG_WGRAD_ROWS", mode)
        gw = torch.full((co, ci, 3, 3), 7.0, device=DEV)
        gb = torch.full((co,), 7.0, device=DEV)
        ops.conv3x3_wgrad(xd, gyd, gw, gb, ups=True)
        assert _rel(gw, wt.grad) <= 3e-6, (mode, _rel(gw, wt.grad))
        assert _rel(gb, bt.grad) <= 3e-6, (mode, _rel(gb, bt.grad))
        gw2 = torch.empty_like(gw)
        ops.conv3x3_wgrad(xd, gyd, gw2, None, ups=True)
        assert torch.equal(gw, gw2)
        if n > 1:
            gb3 = torch.empty(co, device=DEV)
            ops.conv3x3_wgrad(xd, gyd, gw2, gb3, ups=True, bias_n=n // 2)
            assert _rel(gb3, gy[:n // 2].sum(dim=(0, 2, 3))) <= 3e-6, mode
        res[mode] = gw
    assert _rel(res["2"], res["0"]) <= 2e-6


def test_rows_kernel_inside_a_deferred_sweep(monkeypatch):
    """Layers of a sweep (conv3x3_wgrad(..., defer=)): large layers take the row-staged kernel, small ones share the grouped launch."""
    ops = _ops()
    monkeypatch.setenv("MG_WINO_WGRAD_MIN_PIXELS", "1")
    g = torch.Generator(device=DEV).manual_seed(5)
    shapes = [(6, 48, 64, 64, 64), (6, 96, 96, 16, 16), (6, 112, 96, 8, 8), (6, 64, 80, 32, 32)]
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MG_WGRAD_ROWS", mode)
        g.manual_seed(5)
        d = ops.WgradDefer()
        outs = []
        for (n, ci, co, h, w) in shapes:
            x = torch.randn(n, ci, h, w, device=DEV, generator=g)
            gy = torch.randn(n, co, h, w, device=DEV, generator=g)
            gw, gb = torch.empty(co, ci, 3, 3, device=DEV), torch.empty(co, device=DEV)
            ops.conv3x3_wgrad(x, gy, gw, gb, defer=d, bias_n=4)
            outs.append((gw, gb, x, gy))
        d.flush()
        out[mode] = outs
    for (gw0, gb0, x, gy), (gw1, gb1, _, _) in zip(out["0"], out["1"]):
        assert _rel(gw1, gw0) <= 2e-6 and _rel(gb1, gb0) <= 2e-6
