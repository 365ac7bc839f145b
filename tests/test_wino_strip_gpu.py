"""GPU parity of the wave-per-tile-block Winograd kernel (csrc/wino_strip.hip: the few-channel layers of levels 6-7,
/root/reference/music_gan/networks/generator.py:67-76, discriminator.py:60-70): every epilogue kind bit for bit against the LDS-staged
kernel it replaces there (csrc/wino3x3.hip, itself checked against fp64 conv2d in test_ops_gpu.py), and directly against fp64
`F.conv2d` / `F.conv_transpose2d` at the layer shapes it is routed to."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from musicgan_amd import ops
    return ops


def _tile_mask(act):
    n, c, h, w = act.shape
    b = (act > 0).reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4).to(torch.uint8)
    return (b[..., 0] + 2 * b[..., 1] + 4 * b[..., 2] + 8 * b[..., 3]).contiguous()


def _all_kinds(ops, x, wt, b, gy, act_out, act_up, other, coef):
    """Every epilogue the engine asks of mg_wino3x3 / mg_wino3x3_fade on one layer (x: N,Ci,H,W -> Co): a dict of output tuples."""
    from musicgan_amd import _lib
    co, ci = wt.shape[0], wt.shape[1]
    up, upd = ops.pack_wino3x3(wt, dgrad=False), ops.pack_wino3x3(wt, dgrad=True)
    out = {}
    out["act"] = (ops.conv3x3(x, None, b, co, lrelu=True, wino=up),)
    out["act_nolrelu_nobias"] = (ops.conv3x3(x, None, None, co, wino=up),)
    out["act_pool"] = ops.conv3x3(x, None, b, co, lrelu=True, pool=True, wino=up)
    out["act_pool_mask_out"] = ops.conv3x3(x, None, b, co, lrelu=True, pool=True, wino=up, mask_out=True)
    m_out = _tile_mask(act_out)
    out["tangent_bytes"] = (ops.conv3x3(x, None, None, co, mask_aux=m_out, pool=True, wino=up)[1],)
    buf = act_out.clone()
    out["tangent_fp32_in_place"] = ops.conv3x3(x, None, None, co, mask_aux=buf, out=buf, pool=True, wino=up)
    out["dgrad_mask"] = (ops.conv3x3(gy, None, None, ci, mask_aux=x, wino=upd),)
    out["dgrad_unpool"] = (ops.conv3x3(gy, None, None, ci, wino=upd, unpool_mask=_tile_mask(act_up)),)
    # the LeakyReLU mask as tile bytes on a full-resolution result: a strip-kernel epilogue; the staged kernel takes the fp32 mask
    n_, _, h_, w_ = gy.shape
    bytes_ok = ops.wino3x3_mask_bytes_y_supported(n_, co, ci, h_, w_)
    out["dgrad_mask_bytes"] = (ops.conv3x3(gy, None, None, ci, mask_aux=_tile_mask(x) if bytes_ok else x, wino=upd),)
    out["fade_fwd"] = ops.conv3x3_fade(x, up, b, co, _lib.MG_FADE_FWD, other, coef)
    out["fade_tangent"] = (ops.conv3x3_fade(x, up, None, co, _lib.MG_FADE_TANGENT, other, coef, mask_in=m_out),)
    out["fade_bwd"] = ops.conv3x3_fade(gy, upd, None, ci, _lib.MG_FADE_BWD, x, coef, mask_in=_tile_mask(x))
    if co <= 32:
        out["pixnorm"] = ops.conv3x3(x, None, b, co, lrelu=True, pixnorm=True, want_y=False, wino=up)[1:]
    return out


# (N, Cin, Cout, H, W): one, two and three out-channel tiles per wave; 2..14 chunks; several tile blocks per row; ragged batch
STRIP_SHAPES = [(2, 16, 32, 32, 64), (3, 32, 16, 16, 96), (1, 48, 48, 48, 32), (2, 32, 32, 64, 32), (1, 16, 16, 16, 32),
                (2, 96, 80, 32, 32), (1, 112, 96, 16, 32)]  # (96 / 112 input channels: the bank of one out-channel tile per wave; the data gradient's of two)


@pytest.mark.parametrize("shape", STRIP_SHAPES)
def test_strip_kernel_equals_the_staged_kernel_bit_for_bit(shape, monkeypatch):
    ops = _ops()
    n, ci, co, h, w = shape
    g = torch.Generator(device=DEV).manual_seed(31)
    R = lambda *s: torch.randn(*s, device=DEV, generator=g)
    x, wt, b = R(n, ci, h, w), R(co, ci, 3, 3) / math.sqrt(9 * ci), R(co)
    gy, act_out, act_up, other = R(n, co, h, w), R(n, co, h, w), R(n, ci, 2 * h, 2 * w), R(n, co, h, w)
    coef = torch.tensor([0.37, 0.63], device=DEV)
    res = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("MG_WINO_STRIP", mode)  # 0: wino3x3.hip; 2: wino_strip.hip wherever the shape allows
        res[mode] = _all_kinds(ops, x, wt, b, gy, act_out, act_up, other, coef)
    torch.cuda.synchronize()
    for kind, staged in res["0"].items():
        for i, (p, q) in enumerate(zip(staged, res["2"][kind])):
            if p is None:
                assert q is None
                continue
            assert torch.equal(p, q), f"{kind}[{i}]: {int((p != q).sum())} of {p.numel()} elements differ"
    # the same call twice gives the same bits (the kernel's instruction-level hazards show up as run-to-run differences)
    again = _all_kinds(ops, x, wt, b, gy, act_out, act_up, other, coef)
    for kind, first in res["2"].items():
        for p, q in zip(first, again[kind]):
            assert p is None or torch.equal(p, q), kind


def test_strip_kernel_declines_what_it_cannot_do(monkeypatch):
    """Shapes outside its restrictions (ragged channels / widths, too few tile rows) keep the staged kernel and its results."""
    ops = _ops()
    monkeypatch.setenv("MG_WINO_STRIP", "2")
    g = torch.Generator().manual_seed(5)
    for (n, ci, co, h, w) in ((2, 24, 40, 6, 10), (1, 8, 32, 16, 32), (2, 16, 32, 8, 32), (1, 16, 32, 16, 48)):
        x = torch.randn(n, ci, h, w, generator=g)
        wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
        b = torch.randn(co, generator=g)
        ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1), 0.2)
        y = ops.conv3x3(x.to(DEV), None, b.to(DEV), co, lrelu=True, wino=ops.pack_wino3x3(wt.to(DEV), dgrad=False))
        err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (n, ci, co, h, w, err)


# the layers the kernel is routed to at levels 6-7, one or two images (fp64 on the host cores: seconds)
@pytest.mark.parametrize("shape", [(1, 16, 32, 512, 512), (2, 32, 32, 256, 256), (1, 32, 16, 512, 512), (1, 32, 48, 256, 256),
                                   (1, 48, 32, 256, 256)])
def test_strip_kernel_against_fp64_at_level_6_7_shapes(shape, monkeypatch):
    ops = _ops()
    monkeypatch.setenv("MG_WINO_STRIP", "2")
    n, ci, co, h, w = shape
    g = torch.Generator().manual_seed(77)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)
    b = torch.randn(co, generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1), 0.2)
    up = ops.pack_wino3x3(wt.to(DEV), dgrad=False)
    m, q = ops.conv3x3(x.to(DEV), None, b.to(DEV), co, lrelu=True, pool=True, wino=up, mask_out=True)
    scale = float(ref.abs().max())
    assert float((q.double().cpu() - F.avg_pool2d(ref, 2)).abs().max()) <= 2e-6 * scale
    # mask bits: wherever the fp64 value is not within round-off of zero the bit is its sign
    want = _tile_mask(ref)
    near = _tile_mask(ref + 1e-5 * scale) != _tile_mask(ref - 1e-5 * scale)
    assert bool(((m.cpu() == want) | near).all())
    y = ops.conv3x3(x.to(DEV), None, b.to(DEV), co, lrelu=True, wino=up)
    assert float((y.double().cpu() - ref).abs().max()) <= 2e-6 * scale
    # data gradient through the transposed pack, times the LeakyReLU mask of the layer below
    gy = torch.randn(n, co, h, w, generator=g)
    refd = F.conv_transpose2d(gy.double(), wt.double(), padding=1) * torch.where(x > 0, 1.0, 0.2).double()
    gx = ops.conv3x3(gy.to(DEV), None, None, ci, mask_aux=x.to(DEV), wino=ops.pack_wino3x3(wt.to(DEV), dgrad=True))
    assert float((gx.double().cpu() - refd).abs().max()) <= 2e-6 * float(refd.abs().max())
    # the same with the mask as one byte per 2x2 tile (what the critic keeps of its stem's output): the same bits
    gxb = ops.conv3x3(gy.to(DEV), None, None, ci, mask_aux=_tile_mask(x.to(DEV)), wino=ops.pack_wino3x3(wt.to(DEV), dgrad=True))
    assert torch.equal(gx, gxb)
