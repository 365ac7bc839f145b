"""Tensor-level wrappers over the C ABI (include/musicgan_hip.h).  Inputs are fp32 contiguous tensors on a ROCm device;
every call is asynchronous on the caller's current stream.  No fallback path exists: non-GPU tensors raise."""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib
from ._lib import (MG_C1_LRELU, MG_C1_MASK_AUX, MG_C1_TANH, MG_C1_TANH_BWD_IN, MG_C1_TRANSPOSED, MG_CONV_LRELU,
                   MG_CONV_MASK_AUX, MG_CONV_PIXNORM, MG_CONV_POOL_OUT, MG_CONV_UPS_IN, check)

SLOPE = 0.2


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.MusicGanHipError("musicgan_amd kernels need tensors on a ROCm GPU (no CPU fallback)")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise _lib.MusicGanHipError(f"expected contiguous float32, got {t.dtype} contiguous={t.is_contiguous()}")


_ws_cache = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Scratch buffer per (device, stream); grown geometrically, reused across calls on that stream."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20, 0 if buf is None else 2 * buf.numel()), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ------------------------------------------------------------------ conv 3x3
def pack_conv3x3(w: torch.Tensor, dgrad: bool) -> torch.Tensor:
    _chk(w)
    co, ci = w.shape[0], w.shape[1]
    lib = _lib.load()
    cin_call, cout_call = (co, ci) if dgrad else (ci, co)
    wp = torch.empty(lib.mg_conv3x3_packed_floats(cin_call, cout_call), dtype=torch.float32, device=w.device)
    check(lib.mg_conv3x3_pack(_p(w), _p(wp), co, ci, int(dgrad), _s()), "mg_conv3x3_pack")
    return wp


def pack_wino3x3(w: torch.Tensor, dgrad: bool) -> torch.Tensor:
    """U = G g G^T of every filter in MFMA operand order (mg_wino3x3_pack)."""
    _chk(w)
    co, ci = w.shape[0], w.shape[1]
    lib = _lib.load()
    cin_call, cout_call = (co, ci) if dgrad else (ci, co)
    up = torch.empty(lib.mg_wino3x3_packed_floats(cin_call, cout_call), dtype=torch.float32, device=w.device)
    check(lib.mg_wino3x3_pack(_p(w), _p(up), co, ci, int(dgrad), _s()), "mg_wino3x3_pack")
    return up


def packed_floats(kind: int, co: int, ci: int, dgrad: bool) -> int:
    """Size of the packed layout `kind` (_lib.MG_PACK_*) of a weight [co][ci][3][3]."""
    lib = _lib.load()
    cin_call, cout_call = (co, ci) if dgrad else (ci, co)
    if kind == _lib.MG_PACK_CONV3X3:
        return lib.mg_conv3x3_packed_floats(cin_call, cout_call)
    if kind == _lib.MG_PACK_WINO3X3:
        return lib.mg_wino3x3_packed_floats(cin_call, cout_call)
    if kind == _lib.MG_PACK_UPCONV3X3:
        return lib.mg_upconv3x3_packed_floats(ci, co)
    if kind == _lib.MG_PACK_SMALLNET:
        return lib.mg_smallnet_packed_floats(cin_call, cout_call)
    if kind == _lib.MG_PACK_WINOUPS:
        return lib.mg_winoups3x3_packed_floats(ci, co, int(dgrad))
    return lib.mg_upconv3x3_dgrad_packed_floats(ci, co)


def pack_multi(requests) -> None:
    """requests: iterable of (kind, weight [co][ci][3][3], dgrad, out) -- every layout written in ONE launch (mg_pack_multi)."""
    recs = []
    for kind, w, dgrad, out in requests:
        _chk(w, out)
        assert out.numel() == packed_floats(kind, w.shape[0], w.shape[1], dgrad)
        recs.append(_lib.PackDesc(w.data_ptr(), out.data_ptr(), kind, w.shape[0], w.shape[1], int(dgrad)))
    if not recs:
        return
    arr = (_lib.PackDesc * len(recs))(*recs)
    check(_lib.load().mg_pack_multi(ctypes.cast(arr, ctypes.c_void_p), len(recs), _s()), "mg_pack_multi")


def wino3x3_supported(n: int, cout: int, h: int, w: int, *, ups=False, pixnorm=False, cin: int = 0) -> bool:
    """Whether conv3x3(..., wino=) may be used: even sizes, enough 2x2 tiles to fill the chip, for the fused PixelNorm all
    channels of a pixel inside one workgroup, and a tile block's input and output planes within the kernel's 32-bit offsets (the long
    non-square maps of `generate` can exceed that; they then take the direct kernel)."""
    if os.environ.get("MG_WINO", "1") == "0" or ups or (h % 2) or (w % 2):
        return False
    if pixnorm and cout > 64:
        return False
    tiles = (h // 2) * (w // 2)
    if max(cin, cout, 1) * h * w * max(1, 64 // max(tiles, 1)) >= (1 << 29):  # images per 64-tile block x one image's planes
        return False
    # fewer 2x2 tiles do not fill the chip (r05 sweep 8192 / 4096 / 2048: level 3 batch 8 1.381 / 1.360 / 1.372 ms, level 4 batch 32
    # 3.843 / 3.826 / 3.832, levels 6-7 unchanged: the 16x16 layers of 24 images move from the direct kernel to this one)
    return n * h * w >= int(os.environ.get("MG_WINO_MIN_PIXELS", "4096"))


def wino3x3_mask_bytes_y_supported(n: int, cin: int, cout: int, h: int, w: int) -> bool:
    """conv3x3(..., mask_aux=<uint8 tile mask>) without pool: the strip kernel's epilogue, for the shapes it takes."""
    return bool(_lib.load().mg_wino3x3_mask_bytes_y_supported(n, cin, cout, h, w))


def _chk_tilemask(m, shape):
    if not m.is_cuda or m.dtype != torch.uint8 or not m.is_contiguous() or tuple(m.shape) != tuple(shape):
        raise _lib.MusicGanHipError(f"tile mask must be a contiguous uint8 GPU tensor of shape {tuple(shape)}, got "
                                    f"{m.dtype} {tuple(m.shape)}")


def conv3x3(x, wp, bias, cout: int, *, ups=False, lrelu=False, mask_aux=None, pixnorm=False, want_y=True, out=None,
            pool=False, pool_out=None, wino=None, mask_out=False, unpool_mask=None):
    """`wino` (optional, from pack_wino3x3): run the Winograd F(2x2,3x3) kernel instead of the direct one (`wp` is then unused).
    Returns y, or (y, p, rn) with pixnorm, or (y, pooled) with pool (AvgPool2d(2,2) of y fused in the epilogue).  Output
    spatial size = input (x2 with ups).  `out` (optional) receives y; it may alias mask_aux (the mask is read and the result
    written by the same lane); `pool_out` (optional) receives the pooled tensor.
    Tile masks (Winograd kernel only; one uint8 per 2x2 tile and channel, bit 2i+j <-> y[2Y+i, 2X+j] > 0):
    `mask_out` with pool + lrelu returns (tile mask, pooled) -- the full-resolution y is never written;
    a uint8 `mask_aux` is such a mask of this conv's own output: with pool it returns (None, pooled), without pool the masked
    result y itself (a data gradient times the LeakyReLU derivative of the layer below);
    `unpool_mask` (N,cout,H,W uint8, nothing else): returns the (N,cout,2H,2W) tensor 0.25 * up2(conv) * lrelu'(mask) --
    AvgPool2d backward + LeakyReLU backward of the layer below, fused on the data-gradient conv that feeds them."""
    if unpool_mask is not None or mask_out or (mask_aux is not None and mask_aux.dtype == torch.uint8):
        return _conv3x3_tilemask(x, bias, cout, lrelu=lrelu, mask_aux=mask_aux, pool=pool or pool_out is not None,
                                 pool_out=pool_out, wino=wino, mask_out=mask_out, unpool_mask=unpool_mask, ups=ups,
                                 pixnorm=pixnorm, out=out)
    _chk(x, wp, bias, mask_aux, out, pool_out)
    pool = pool or pool_out is not None
    n, cin, hin, win = x.shape
    h, w = (2 * hin, 2 * win) if ups else (hin, win)
    flags = (MG_CONV_UPS_IN if ups else 0) | (MG_CONV_LRELU if lrelu else 0) | \
            (MG_CONV_MASK_AUX if mask_aux is not None else 0) | (MG_CONV_PIXNORM if pixnorm else 0) | \
            (MG_CONV_POOL_OUT if pool else 0)
    y = out if out is not None else (
        torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device) if (want_y or not pixnorm) else None)
    p = rn = None
    if pixnorm:
        p = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
        rn = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
    if pool:
        p = pool_out if pool_out is not None else torch.empty((n, cout, h // 2, w // 2), dtype=torch.float32,
                                                              device=x.device)
    if wino is not None:
        _chk(wino)
        check(_lib.load().mg_wino3x3(_p(x), _p(wino), _p(bias), _p(mask_aux), _p(y), _p(p), _p(rn), n, cin, cout, h, w, flags,
                                     SLOPE, _s()), "mg_wino3x3")
    else:
        check(_lib.load().mg_conv3x3(_p(x), _p(wp), _p(bias), _p(mask_aux), _p(y), _p(p), _p(rn), n, cin, cout, h, w, flags,
                                     SLOPE, _s()), "mg_conv3x3")
    if pool:
        return y, p
    return (y, p, rn) if pixnorm else y


def _conv3x3_tilemask(x, bias, cout, *, lrelu, mask_aux, pool, pool_out, wino, mask_out, unpool_mask, ups, pixnorm, out):
    if wino is None or ups or pixnorm or out is not None:
        raise _lib.MusicGanHipError("tile masks are an epilogue of the Winograd kernel only (no ups / pixnorm / out)")
    _chk(x, wino, bias, pool_out)
    n, cin, h, w = x.shape
    lib = _lib.load()
    if unpool_mask is not None:
        if lrelu or mask_aux is not None or pool or mask_out or bias is not None:
            raise _lib.MusicGanHipError("unpool_mask excludes every other epilogue")
        _chk_tilemask(unpool_mask, (n, cout, h, w))
        y = torch.empty((n, cout, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
        check(lib.mg_wino3x3(_p(x), _p(wino), None, _p(unpool_mask), _p(y), None, None, n, cin, cout, h, w, _lib.MG_CONV_UNPOOL,
                             SLOPE, _s()), "mg_wino3x3")
        return y
    if not pool:
        if mask_aux is None or mask_out or lrelu or bias is not None:
            raise _lib.MusicGanHipError("without the fused AvgPool2d a tile mask is an input only (bias-free, no lrelu)")
        _chk_tilemask(mask_aux, (n, cout, h // 2, w // 2))
        y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
        check(lib.mg_wino3x3(_p(x), _p(wino), None, _p(mask_aux), _p(y), None, None, n, cin, cout, h, w,
                             MG_CONV_MASK_AUX | _lib.MG_CONV_MASK_BYTES, SLOPE, _s()), "mg_wino3x3")
        return y
    p = pool_out if pool_out is not None else torch.empty((n, cout, h // 2, w // 2), dtype=torch.float32, device=x.device)
    if mask_out:
        if not lrelu or mask_aux is not None:
            raise _lib.MusicGanHipError("mask_out needs lrelu and no mask_aux")
        m = torch.empty((n, cout, h // 2, w // 2), dtype=torch.uint8, device=x.device)
        flags = MG_CONV_LRELU | MG_CONV_POOL_OUT | _lib.MG_CONV_MASK_OUT
        check(lib.mg_wino3x3(_p(x), _p(wino), _p(bias), None, _p(m), _p(p), None, n, cin, cout, h, w, flags, SLOPE, _s()),
              "mg_wino3x3")
        return m, p
    if lrelu:
        raise _lib.MusicGanHipError("mask_aux excludes lrelu")
    _chk_tilemask(mask_aux, (n, cout, h // 2, w // 2))
    flags = MG_CONV_MASK_AUX | MG_CONV_POOL_OUT | _lib.MG_CONV_MASK_BYTES
    check(lib.mg_wino3x3(_p(x), _p(wino), _p(bias), _p(mask_aux), None, _p(p), None, n, cin, cout, h, w, flags, SLOPE, _s()),
          "mg_wino3x3")
    return None, p


def conv3x3_fade(x, wino, bias, cout: int, mode: int, other, coef, mask_in=None, out=None):
    """The critic's fade-in blend fused on the Winograd conv next to it (mg_wino3x3_fade; `coef` = device tensor {alpha, 1-alpha}):
    MG_FADE_FWD -> (blend, tile mask of the new branch);  MG_FADE_TANGENT (mask_in) -> blend of the tangents;
    MG_FADE_BWD (wino = data-gradient pack, mask_in, other = old branch's activation) -> (grad new branch, grad old branch)."""
    _chk(x, wino, bias, other, coef, out)
    n, cin, h, w = x.shape
    if tuple(other.shape) != (n, cout, h, w):
        raise _lib.MusicGanHipError(f"conv3x3_fade: other has shape {tuple(other.shape)}, expected {(n, cout, h, w)}")
    if mode != _lib.MG_FADE_FWD:
        _chk_tilemask(mask_in, (n, cout, h // 2, w // 2))
    y = out if out is not None else torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    out2 = None
    if mode == _lib.MG_FADE_FWD:
        out2 = torch.empty((n, cout, h // 2, w // 2), dtype=torch.uint8, device=x.device)
    elif mode == _lib.MG_FADE_BWD:
        out2 = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().mg_wino3x3_fade(_p(x), _p(wino), _p(bias), _p(mask_in), _p(other), _p(coef), _p(y), _p(out2), n, cin, cout,
                                      h, w, mode, SLOPE, _s()), "mg_wino3x3_fade")
    return y if out2 is None else (y, out2)


def pack_upconv3x3(w: torch.Tensor) -> torch.Tensor:
    """Effective sub-pixel weights of Upsample(x2) -> Conv3x3 in the kernel's LDS image layout."""
    _chk(w)
    co, ci = w.shape[0], w.shape[1]
    lib = _lib.load()
    wp = torch.empty(lib.mg_upconv3x3_packed_floats(ci, co), dtype=torch.float32, device=w.device)
    check(lib.mg_upconv3x3_pack(_p(w), _p(wp), co, ci, _s()), "mg_upconv3x3_pack")
    return wp


def upconv3x3_supported(cout: int, win: int, numel: int = 0) -> bool:
    """The sub-pixel kernel keeps 4 phases x all output channels of 16 pixels in one wave: beyond 80 channels its registers
    cost more occupancy than the 2.25x MFMA saving returns (measured: 112->96 @8x8 is slower than the direct form).
    `numel` (input elements): the kernel indexes with 31 bits."""
    return cout <= 80 and win >= 2 and numel < (1 << 31)


def upconv3x3(x, wp, bias, cout: int, *, lrelu=False, pixnorm=False, want_y=True):
    """Upsample(x2 nearest) -> Conv3x3 (+ LeakyReLU + PixelNorm) in sub-pixel form.  Returns y or (y, p, rn)."""
    _chk(x, wp, bias)
    n, cin, hin, win = x.shape
    h, w = 2 * hin, 2 * win
    flags = (MG_CONV_LRELU if lrelu else 0) | (MG_CONV_PIXNORM if pixnorm else 0)
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device) if (want_y or not pixnorm) else None
    p = rn = None
    if pixnorm:
        p = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
        rn = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().mg_upconv3x3(_p(x), _p(wp), _p(bias), _p(y), _p(p), _p(rn), n, cin, cout, hin, win, flags, SLOPE, _s()),
          "mg_upconv3x3")
    return (y, p, rn) if pixnorm else y


def winoups3x3_supported(n: int, cin: int, cout: int, hin: int, win: int, *, dgrad: bool = False) -> bool:
    """Whether Upsample(x2) -> Conv3x3 (cin -> cout, input hin x win) / its data gradient takes the 9-component Winograd kernels of
    csrc/wino_ups.hip (MG_WINOUPS=0: never -- the sub-pixel kernels then)."""
    if os.environ.get("MG_WINOUPS", "1") == "0":
        return False
    if dgrad and cin > 64:
        # 80 / 96 input channels: the data gradient as two launches of <= 3 tiles each.  Against the stride-2 kernel 107 -> 79 us
        # (80<-64 @32 x64), 83 -> 50 (96<-80 @16 x64), 61 -> 44 (x32); below ~128 wave items (N x H/8 x W/16) the other route is the
        # small-map conv + block sums, which wins there (24-35 us against 37-48): profiles/r06_ab_winoups.txt.  MG_WINOUPS_TILE_GROUPS=0: never
        if os.environ.get("MG_WINOUPS_TILE_GROUPS", "1") == "0" or n * (hin // 8) * (win // 16) < 128:
            return False
    return bool(_lib.load().mg_winoups3x3_supported(n, cin, cout, hin, win, int(dgrad)))


def pack_winoups3x3(w: torch.Tensor, dgrad: bool) -> torch.Tensor:
    _chk(w)
    out = torch.empty(packed_floats(_lib.MG_PACK_WINOUPS, w.shape[0], w.shape[1], dgrad), dtype=torch.float32, device=w.device)
    pack_multi([(_lib.MG_PACK_WINOUPS, w, dgrad, out)])
    return out


def winoups3x3(x, up, bias, cout: int, *, lrelu=False, pixnorm=False, want_y=True):
    """Upsample(x2 nearest) -> Conv3x3 (+ LeakyReLU + PixelNorm) in 9-component Winograd form.  Returns y or (y, p, rn)."""
    _chk(x, up, bias)
    n, cin, hin, win = x.shape
    h, w = 2 * hin, 2 * win
    flags = (MG_CONV_LRELU if lrelu else 0) | (MG_CONV_PIXNORM if pixnorm else 0)
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device) if (want_y or not pixnorm) else None
    p = rn = None
    if pixnorm:
        p = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
        rn = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().mg_winoups3x3(_p(x), _p(up), _p(bias), _p(y), _p(p), _p(rn), n, cin, cout, hin, win, flags, SLOPE, _s()), "mg_winoups3x3")
    return (y, p, rn) if pixnorm else y


def _head_fuse(bit: int) -> bool:
    """MG_HEAD_FUSE (default 3): bit 0 = the generator head's backward with the PixelNorm backward in front of it in one launch, bit 1 =
    the head's forward in the last conv's epilogue (A/B switch)."""
    return (int(os.environ.get("MG_HEAD_FUSE", "3")) & bit) != 0


def winoups3x3_head_supported(n: int, cin: int, cout: int, hin: int, win: int) -> bool:
    return _head_fuse(2) and winoups3x3_supported(n, cin, cout, hin, win) and bool(_lib.load().mg_winoups3x3_head_supported(n, cin, cout, hin, win))


def winoups3x3_head(x, up, bias, cout: int, hw, hb, *, want_y=False, mp_out=None):
    """winoups3x3(lrelu, pixnorm) with the generator's 1x1 head on the normalised activation in the epilogue: returns (y, p, rn, mp),
    mp = tanh(hw p + hb) of shape (N, 2, 2H, 2W) (written into `mp_out` if given)."""
    _chk(x, up, bias, hw, hb, mp_out)
    n, cin, hin, win = x.shape
    h, w = 2 * hin, 2 * win
    assert hw.shape[0] == 2 and hw.shape[1] == cout
    new = lambda c: torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    y = new(cout) if want_y else None
    p, rn = new(cout), new(1)
    mp = new(2) if mp_out is None else mp_out
    check(_lib.load().mg_winoups3x3_head(_p(x), _p(up), _p(bias), _p(y), _p(p), _p(rn), _p(hw), _p(hb), _p(mp), n, cin, cout, hin, win,
                                         SLOPE, _s()), "mg_winoups3x3_head")
    return y, p, rn, mp


def winoups3x3_dgrad(gy, up, cin: int):
    """Gradient of Upsample(x2) -> Conv3x3 w.r.t. its low-resolution input in 9-component Winograd form: (N,Cout,2H,2W) -> (N,Cin,H,W)."""
    _chk(gy, up)
    n, cout, h2, w2 = gy.shape
    gx = torch.empty((n, cin, h2 // 2, w2 // 2), dtype=torch.float32, device=gy.device)
    check(_lib.load().mg_winoups3x3_dgrad(_p(gy), _p(up), _p(gx), n, cin, cout, h2 // 2, w2 // 2, _s()), "mg_winoups3x3_dgrad")
    return gx


def winoups3x3_dgrad_pn_supported(n: int, cin: int, cout: int, hin: int, win: int) -> bool:
    """All `cin` channels of a low-res pixel in one wave (at most 64): the PixelNorm backward needs their sum."""
    return cin <= 64 and _head_fuse(1) and fuse_ends() and winoups3x3_supported(n, cin, cout, hin, win, dgrad=True)


def winoups3x3_dgrad_pn(gy, up, p, rn, cin: int):
    """winoups3x3_dgrad followed by the PixelNorm + LeakyReLU backward of the layer whose normalised output is p (rn: its 1/norm), in
    the data-gradient kernel's epilogue: returns the gradient at that layer's pre-activation (N,Cin,H,W)."""
    _chk(gy, up, p, rn)
    n, cout, h2, w2 = gy.shape
    assert p.shape == (n, cin, h2 // 2, w2 // 2)
    gpre = torch.empty_like(p)
    check(_lib.load().mg_winoups3x3_dgrad_pn(_p(gy), _p(up), _p(p), _p(rn), _p(gpre), n, cin, cout, h2 // 2, w2 // 2, SLOPE, _s()),
          "mg_winoups3x3_dgrad_pn")
    return gpre


def pack_upconv3x3_dgrad(w: torch.Tensor) -> torch.Tensor:
    _chk(w)
    co, ci = w.shape[0], w.shape[1]
    lib = _lib.load()
    wp = torch.empty(lib.mg_upconv3x3_dgrad_packed_floats(ci, co), dtype=torch.float32, device=w.device)
    check(lib.mg_upconv3x3_dgrad_pack(_p(w), _p(wp), co, ci, _s()), "mg_upconv3x3_dgrad_pack")
    return wp


def upconv3x3_dgrad_supported(hin: int, win: int, gy_numel: int = 0, n: int = 0) -> bool:
    """Mirrors the tile choice of mg_upconv3x3_dgrad: 128 low-res pixels per workgroup, the high-res halo tile of one
    8-channel chunk must fit the 24-register prefetch (images below 8x8 do not; they take the plain dgrad + block-sum path).
    `gy_numel` (elements of the output gradient): the kernel indexes with 31 bits -- larger tensors take that path too."""
    if os.environ.get("MG_UPCONV_DGRAD", "1") == "0":  # A/B switch for measurements
        return False
    if gy_numel >= (1 << 31):
        return False
    if n and n * hin * win < int(os.environ.get("MG_UPCONV_DGRAD_MIN_PIXELS", "16384")):
        # 128 low-res pixels per workgroup: fewer than ~128 workgroups leave most of the chip idle behind one long K loop (16 / 64
        # workgroups: 114 / 84 us at level 4); the plain data gradient + 2x2 block sum is sliced over out-channels and faster there
        return False
    p2 = lambda v: 1 << max(0, (v - 1).bit_length())
    tw = min(32, p2(win))
    th = min(p2(hin), 128 // tw)
    tn = 128 // (tw * th)
    return tn * (2 * th + 2) * (2 * tw + 2) <= 768


def upconv3x3_dgrad(gy, wp, cin: int):
    """Gradient of Upsample(x2) -> Conv3x3 w.r.t. its low-resolution input: (N,Cout,2H,2W) -> (N,Cin,H,W)."""
    _chk(gy, wp)
    n, cout, h2, w2 = gy.shape
    gx = torch.empty((n, cin, h2 // 2, w2 // 2), dtype=torch.float32, device=gy.device)
    check(_lib.load().mg_upconv3x3_dgrad(_p(gy), _p(wp), _p(gx), n, cin, cout, h2 // 2, w2 // 2, _s()),
          "mg_upconv3x3_dgrad")
    return gx


def wino_wgrad_supported(n: int, cin: int, cout: int, h: int, w: int, *, ups=False) -> bool:
    """Whether conv3x3_wgrad takes the Winograd F(3x3,2x2) kernel: even sizes, byte offsets within 31 bits.  The size threshold is
    low on purpose: with the slab reductions of a sweep in one launch the Winograd form beats the direct one down to 2x2 maps
    (threshold sweep 8192 -> 64: level 5 15.35 -> 15.22 ms, level 4 5.15 -> 5.02, level 3 2.73 -> 2.59)."""
    if os.environ.get("MG_WINO_WGRAD", "1") == "0" or (h % 2) or (w % 2):
        return False
    if n * max(cin, cout) * h * w >= (1 << 29):
        return False
    return n * h * w >= int(os.environ.get("MG_WINO_WGRAD_MIN_PIXELS", "64"))


def wino_wgrad_form(n: int, cin: int, cout: int, h: int, w: int, *, ups=False) -> int:
    """0 / 1 / 2: chunk-staged, row-staged, row-staged 9-component (up-sampled input) Winograd weight gradient (mg_wino3x3_wgrad_form)."""
    return int(_lib.load().mg_wino3x3_wgrad_form(n, cin, cout, h, w, MG_CONV_UPS_IN if ups else 0, wgrad_group_chunks()))


def wgrad_group_chunks() -> int:
    """MG_WGRAD_GROUP: layers of at most this many 8-tile chunks per workgroup (at one workgroup per CU) share a launch with the
    other small layers of their block shape; 0 = one launch per layer."""
    return int(os.environ.get("MG_WGRAD_GROUP", "32"))


class WgradDefer:
    """Collects the slab reductions of the Winograd weight gradients of one sweep (conv3x3_wgrad(..., defer=this)) and runs them
    in ONE launch (`flush`).  Each deferred layer keeps its own workspace alive until then; the i-th layer of a sweep reuses the
    i-th buffer of the previous sweep."""

    def __init__(self):
        self._bufs = []
        self._jobs = []      # Winograd form
        self._jobs_d = []    # direct form
        self._lazy = []      # Winograd form, matrix kernel not launched yet: (descriptor, tensors it points into)

    def workspace(self, nbytes: int, device) -> torch.Tensor:
        i = len(self._jobs) + len(self._jobs_d) + len(self._lazy)
        if i == len(self._bufs):
            self._bufs.append(torch.empty(nbytes, dtype=torch.uint8, device=device))
        elif self._bufs[i].numel() < nbytes or self._bufs[i].device != device:
            self._bufs[i] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._bufs[i]

    def add(self, job, direct: bool = False) -> None:
        (self._jobs_d if direct else self._jobs).append(job)

    def add_lazy(self, desc, keep) -> None:
        self._lazy.append((desc, keep))

    def reset(self) -> None:
        """Drop collected jobs without running them (an aborted graph capture: their pointers died with the capture's pool)."""
        self._jobs, self._jobs_d, self._lazy = [], [], []

    def flush(self) -> None:
        lib = _lib.load()
        if self._lazy:
            # the matrix kernels of the whole sweep: small layers with equal block shapes share a launch (include/musicgan_hip.h)
            descs = (_lib.WgradDesc * len(self._lazy))(*[d for d, _ in self._lazy])
            jobs = (_lib.WgradJob * len(self._lazy))()
            check(lib.mg_wino3x3_wgrad_partial_multi(ctypes.cast(descs, ctypes.c_void_p), len(self._lazy), wgrad_group_chunks(),
                                                     ctypes.cast(jobs, ctypes.c_void_p), _s()), "mg_wino3x3_wgrad_partial_multi")
            self._jobs += list(jobs)
            self._lazy = []
        for jobs, fn, what in ((self._jobs, lib.mg_wino3x3_wgrad_reduce, "mg_wino3x3_wgrad_reduce"),
                               (self._jobs_d, lib.mg_conv3x3_wgrad_reduce, "mg_conv3x3_wgrad_reduce")):
            if jobs:
                arr = (_lib.WgradJob * len(jobs))(*jobs)
                check(fn(ctypes.cast(arr, ctypes.c_void_p), len(arr), _s()), what)
        self._jobs, self._jobs_d = [], []


def conv3x3_wgrad(x, gy, gw, gb, *, ups=False, accumulate=False, bias_n: int = 0, defer: Optional[WgradDefer] = None):
    """gw[Cout,Cin,3,3] (+)= wgrad(x, gy); gb[Cout] (+)= sum gy over samples n < bias_n (0: all; gb may be None).
    `defer`: leave the Winograd kernel's slab reduction to defer.flush() (gw / gb are not valid before it)."""
    _chk(x, gy, gw, gb)
    n, cout, h, w = gy.shape
    cin = x.shape[1]
    lib = _lib.load()
    if h == 1 and w == 1 and not ups and os.environ.get("MG_WGRAD_1X1MAP", "1") != "0":
        # (complete at once: nothing is left for defer.flush())
        check(lib.mg_conv3x3_wgrad_1x1map(_p(x), _p(gy), _p(gw), _p(gb), n, cin, cout, int(accumulate), int(bias_n), _s()),
              "mg_conv3x3_wgrad_1x1map")
        return
    if defer is not None and not accumulate and wino_wgrad_supported(n, cin, cout, h, w, ups=ups):
        ws = defer.workspace(lib.mg_wino3x3_wgrad_ws_bytes(n, cin, cout, h, w), x.device)
        if wgrad_group_chunks() > 0:
            defer.add_lazy(_lib.WgradDesc(_p(x), _p(gy), _p(gw), _p(gb), _p(ws), ws.numel(), n, cin, cout, h, w,
                                          MG_CONV_UPS_IN if ups else 0, 0, int(bias_n)), (x, gy, gw, gb, ws))
            return
        job = _lib.WgradJob()
        check(lib.mg_wino3x3_wgrad_partial(_p(x), _p(gy), _p(gw), _p(gb), _p(ws), ws.numel(), n, cin, cout, h, w,
                                           MG_CONV_UPS_IN if ups else 0, 0, int(bias_n), ctypes.byref(job), _s()),
              "mg_wino3x3_wgrad_partial")
        defer.add(job)
        return
    if defer is not None and not accumulate:
        ws = defer.workspace(lib.mg_conv3x3_wgrad_ws_bytes(n, cin, cout, h, w), x.device)
        job = _lib.WgradJob()
        check(lib.mg_conv3x3_wgrad_partial(_p(x), _p(gy), _p(gw), _p(gb), _p(ws), ws.numel(), n, cin, cout, h, w,
                                           MG_CONV_UPS_IN if ups else 0, 0, int(bias_n), ctypes.byref(job), _s()),
              "mg_conv3x3_wgrad_partial")
        defer.add(job, direct=True)
        return
    if wino_wgrad_supported(n, cin, cout, h, w, ups=ups):
        ws = workspace(lib.mg_wino3x3_wgrad_ws_bytes(n, cin, cout, h, w), x.device)
        check(lib.mg_wino3x3_wgrad(_p(x), _p(gy), _p(gw), _p(gb), _p(ws), ws.numel(), n, cin, cout, h, w,
                                   MG_CONV_UPS_IN if ups else 0, int(accumulate), int(bias_n), _s()), "mg_wino3x3_wgrad")
        return
    nbytes = lib.mg_conv3x3_wgrad_ws_bytes(n, cin, cout, h, w)
    ws = workspace(nbytes, x.device)
    check(lib.mg_conv3x3_wgrad(_p(x), _p(gy), _p(gw), _p(gb), _p(ws), ws.numel(), n, cin, cout, h, w,
                               MG_CONV_UPS_IN if ups else 0, int(accumulate), int(bias_n), _s()), "mg_conv3x3_wgrad")


# ------------------------------------------------------------------ multi-layer chains on small maps
class SmallNet:
    """Builds the op list of one mg_smallnet launch (include/musicgan_hip.h): a workgroup carries `imgs_per_wg` images through
    all ops with the activations in three LDS buffers.  Methods append an op and return self; `run(n)` launches.  Global tensors
    are contiguous fp32 (N, C, H, W) (views of a leading-dimension slice are fine: only data_ptr() is passed); filters come
    from PackCache.get_sn()."""

    def __init__(self, imgs_per_wg: int = 1):
        self.g = int(imgs_per_wg)
        self.ops = []
        self.keep = []       # tensors the launch reads / writes (kept alive until run() has been issued)
        self.buf_floats = 0

    def _fit(self, c, h, w):
        self.buf_floats = max(self.buf_floats, _lib.load().mg_smallnet_buffer_floats(self.g, c, h, w))

    def _add(self, op, src=0, dst=0, C=1, C2=0, H=1, W=1, flags=0, inp=None, aux=None, bias=None, out=None, out2=None):
        _chk(inp, aux, bias, out, out2)
        self.keep += [t for t in (inp, aux, bias, out, out2) if t is not None]
        self.ops.append(_lib.SnOp(op, src, dst, C, C2, H, W, flags, _p(inp), _p(aux), _p(bias), _p(out), _p(out2)))
        self._fit(C, H, W)
        return self

    def load(self, dst, x):
        _, c, h, w = x.shape
        return self._add(_lib.MG_SN_LOAD, dst=dst, C=c, H=h, W=w, inp=x)

    def store(self, src, out):
        _, c, h, w = out.shape
        return self._add(_lib.MG_SN_STORE, src=src, C=c, H=h, W=w, out=out)

    def conv(self, src, dst, wpk, cin, cout, h, w, *, bias=None, lrelu=False, mask=None, out=None):
        self._fit(cout, h, w)
        fl = (_lib.MG_SN_LRELU if lrelu else 0) | (_lib.MG_SN_MASK_AUX if mask is not None else 0)
        return self._add(_lib.MG_SN_CONV, src=src, dst=dst, C=cin, C2=cout, H=h, W=w, flags=fl, inp=wpk, aux=mask, bias=bias,
                         out=out)

    def mask(self, src, c, h, w, aux, out=None):
        return self._add(_lib.MG_SN_MASK, src=src, C=c, H=h, W=w, aux=aux, out=out)

    def pixnorm(self, src, c, h, w, p_out=None, rn_out=None):
        return self._add(_lib.MG_SN_PIXNORM, src=src, C=c, H=h, W=w, out=p_out, out2=rn_out)

    def pnbwd(self, src, c, h, w, p, rn, out=None):
        return self._add(_lib.MG_SN_PNBWD, src=src, C=c, H=h, W=w, inp=p, aux=rn, out=out)

    def pool(self, src, dst, c, h, w, out=None):
        return self._add(_lib.MG_SN_POOL, src=src, dst=dst, C=c, H=h, W=w, out=out)

    def poolbwd(self, src, dst, c, h, w, aux, out=None, lds=True):
        if lds:
            self._fit(c, 2 * h, 2 * w)
        return self._add(_lib.MG_SN_POOLBWD, src=src, dst=dst, C=c, H=h, W=w, flags=0 if lds else _lib.MG_SN_NOLDS, aux=aux,
                         out=out)

    def up(self, src, dst, c, h, w):
        self._fit(c, 2 * h, 2 * w)
        return self._add(_lib.MG_SN_UP, src=src, dst=dst, C=c, H=h, W=w)

    def upbwd(self, src, dst, c, h, w):
        return self._add(_lib.MG_SN_UPBWD, src=src, dst=dst, C=c, H=h, W=w)

    def linear(self, src, c, w, b, out):
        return self._add(_lib.MG_SN_LINEAR, src=src, C=c, inp=w, aux=b, out=out)

    def linbwd(self, dst, c, g_out, w):
        return self._add(_lib.MG_SN_LINBWD, dst=dst, C=c, inp=g_out, aux=w)

    def run(self, n: int) -> None:
        arr = (_lib.SnOp * len(self.ops))(*self.ops)
        check(_lib.load().mg_smallnet(ctypes.cast(arr, ctypes.c_void_p), len(self.ops), int(n), self.g,
                                      (int(self.buf_floats) + 3) & ~3, SLOPE, _s()), "mg_smallnet")
        self.keep = []


def conv3x3_small_supported(n: int, cin: int, cout: int, h: int, w: int) -> bool:
    """Whether the latency-optimised one-layer kernel (mg_conv3x3_small) takes this shape and should: maps of 2x2 .. 8x8 with few
    enough pixels that the layer is latency-bound (MG_SMALLCONV=0 switches it off, MG_SMALLCONV_MAX_PIXELS moves the bound)."""
    if os.environ.get("MG_SMALLCONV", "1") == "0":
        return False
    if not _lib.load().mg_conv3x3_small_supported(n, cin, cout, h, w):
        return False
    return n * h * w <= int(os.environ.get("MG_SMALLCONV_MAX_PIXELS", "8192"))


def conv3x3_small(x, wpk, bias, cout: int, *, ups=False, lrelu=False, mask_aux=None, out=None, pool=False, upsum=False,
                  unpool_aux=None, pool_out=None, want_y=True):
    """3x3 convolution on a small map through mg_conv3x3_small (wpk: PackCache.get_sn).  Returns y; (y, p) with `pool` (p =
    AvgPool2d(2,2)(y)) or `upsum` (p = 2x2 block sums of y); with `unpool_aux` (N,cout,2H,2W) the un-pooled, masked tensor of
    that shape (AvgPool2d backward + LeakyReLU backward).  `out` receives y (may alias mask_aux); want_y=False skips y."""
    _chk(x, wpk, bias, mask_aux, out, unpool_aux, pool_out)
    n, cin, hin, win = x.shape
    h, w = (2 * hin, 2 * win) if ups else (hin, win)
    pool = pool or (pool_out is not None and not upsum)
    flags = (MG_CONV_UPS_IN if ups else 0) | (MG_CONV_LRELU if lrelu else 0) | (MG_CONV_MASK_AUX if mask_aux is not None else 0) | \
        (MG_CONV_POOL_OUT if pool else 0) | (_lib.MG_CONV_UPSUM_OUT if upsum else 0) | (_lib.MG_CONV_UNPOOL if unpool_aux is not None else 0)
    aux = unpool_aux if unpool_aux is not None else mask_aux
    p = None
    if unpool_aux is not None:
        y = torch.empty((n, cout, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    else:
        y = out if out is not None else (torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device) if want_y else None)
    if pool or upsum:
        p = pool_out if pool_out is not None else torch.empty((n, cout, h // 2, w // 2), dtype=torch.float32, device=x.device)
    check(_lib.load().mg_conv3x3_small(_p(x), _p(wpk), _p(bias), _p(aux), _p(y), _p(p), n, cin, cout, h, w, flags, SLOPE, _s()),
          "mg_conv3x3_small")
    return (y, p) if (pool or upsum) else y


def conv3x3_small_pn(x_raw, wpk, bias, cout: int, *, ups=False, lrelu=True, save=True):
    """act(conv3x3(PixelNorm(x_raw)) + bias) with the PixelNorm of the layer in front folded into the convolution's input staging
    (mg_conv3x3_small_pn).  Returns (y, p, rn): y the new activation, p = PixelNorm(x_raw) and rn = 1 / norm of x_raw's pixels
    (None, None without `save`)."""
    _chk(x_raw, wpk, bias)
    n, cin, hin, win = x_raw.shape
    h, w = (2 * hin, 2 * win) if ups else (hin, win)
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x_raw.device)
    p = torch.empty_like(x_raw) if save else None
    rn = torch.empty((n, 1, hin, win), dtype=torch.float32, device=x_raw.device) if save else None
    flags = (MG_CONV_UPS_IN if ups else 0) | (MG_CONV_LRELU if lrelu else 0)
    check(_lib.load().mg_conv3x3_small_pn(_p(x_raw), _p(wpk), _p(bias), _p(y), _p(p), _p(rn), n, cin, cout, h, w, flags, SLOPE, _s()),
          "mg_conv3x3_small_pn")
    return y, p, rn


def pack_smallnet(w: torch.Tensor, dgrad: bool) -> torch.Tensor:
    """Filters of a 3x3 convolution in the operand order mg_smallnet streams (MG_PACK_SMALLNET)."""
    _chk(w)
    out = torch.empty(packed_floats(_lib.MG_PACK_SMALLNET, w.shape[0], w.shape[1], dgrad), dtype=torch.float32, device=w.device)
    pack_multi([(_lib.MG_PACK_SMALLNET, w, dgrad, out)])
    return out


# ------------------------------------------------------------------ conv 1x1
def conv1x1(x, w, bias, cout: int, *, lrelu=False, tanh=False, mask_aux=None, transposed=False, tanh_bwd_in=None,
            out=None, accumulate=False):
    """`accumulate` (few input channels only): out += result."""
    _chk(x, w, bias, mask_aux, tanh_bwd_in, out)
    n, cin, h, wd = x.shape
    flags = (MG_C1_LRELU if lrelu else 0) | (MG_C1_TANH if tanh else 0) | (MG_C1_TRANSPOSED if transposed else 0)
    if accumulate:
        assert out is not None and cin <= 4
        flags |= _lib.MG_C1_ACCUM
    aux = None
    if mask_aux is not None:
        flags |= MG_C1_MASK_AUX
        aux = mask_aux
    if tanh_bwd_in is not None:
        assert aux is None
        flags |= MG_C1_TANH_BWD_IN
        aux = tanh_bwd_in
    y = out if out is not None else torch.empty((n, cout, h, wd), dtype=torch.float32, device=x.device)
    check(_lib.load().mg_conv1x1(_p(x), _p(w), _p(bias), _p(aux), _p(y), n, cin, cout, h * wd, flags, SLOPE, _s()),
          "mg_conv1x1")
    return y


def conv1x1_wgrad(x, gy, gw, gb, *, tanh_y=None, accumulate=False, bias_n: int = 0):
    """gw (+)= sum gy x, gb (+)= sum of gy over the samples n < bias_n (0: all)."""
    _chk(x, gy, gw, gb, tanh_y)
    n, cout, h, wd = gy.shape
    cin = x.shape[1]
    lib = _lib.load()
    nbytes = lib.mg_conv1x1_wgrad_ws_bytes(n, cin, cout, h * wd)
    ws = workspace(nbytes, x.device)
    check(lib.mg_conv1x1_wgrad(_p(x), _p(gy), _p(tanh_y), _p(gw), _p(gb), _p(ws), ws.numel(), n, cin, cout, h * wd,
                               int(accumulate), int(bias_n), _s()), "mg_conv1x1_wgrad")


def fuse_ends() -> bool:
    """The fade-in ends of both networks as single launches (csrc/fade_ends.hip); MG_FUSE_ENDS=0: the separate kernels."""
    return os.environ.get("MG_FUSE_ENDS", "1") != "0"


def stem_pair_supported(h: int, w: int) -> bool:
    return fuse_ends() and h % 2 == 0 and w % 4 == 0


# The shapes the other fused ends take (the MG_CHECK_ARGs of csrc/fade_ends.hip): the engine asks here and falls back to the separate
# kernels otherwise -- the four waves of pair_few_out_k split at least four channels on either side.
def stem_pair_gx_supported(c0: int, c1: int, h: int, w: int) -> bool:
    return fuse_ends() and c0 >= 4 and c1 >= 4 and h % 2 == 0 and w % 2 == 0


def head_pair_supported(c: int, cl: int, h: int, w: int) -> bool:
    return fuse_ends() and c >= 4 and cl >= 4 and h % 2 == 0 and w % 2 == 0


def blend_up_bwd_supported(h: int, w: int) -> bool:
    return fuse_ends() and h % 2 == 0 and w % 4 == 0


def stem_pair(x, ws, bs, wo, bo, *, lrelu=True, h0=None, xp=None, o=None, masked=False, want_xp=True, want_mask=False):
    """Critic input while a block fades in (discriminator.py:107-113): h0 = act(ws x + bs), xp = AvgPool2d(x), o = act(wo xp + bo).
    `masked`: the tangent form -- h0 / o hold the forward activations and receive (w x) * lrelu'(activation) in place."""
    _chk(x, ws, bs, wo, bo, h0, xp, o)
    n, _, h, w = x.shape
    c0, c1 = ws.shape[0], wo.shape[0]
    if masked:
        assert h0 is not None and o is not None and bs is None and bo is None
    new = lambda c, hh, ww: torch.empty((n, c, hh, ww), dtype=torch.float32, device=x.device)
    h0 = new(c0, h, w) if h0 is None else h0
    o = new(c1, h // 2, w // 2) if o is None else o
    if xp is None and want_xp:
        xp = new(2, h // 2, w // 2)
    flags = MG_C1_MASK_AUX if masked else (MG_C1_LRELU if lrelu else 0)
    hm = torch.empty((n, c0, h // 2, w // 2), dtype=torch.uint8, device=x.device) if (want_mask and not masked) else None
    check(_lib.load().mg_stem_pair(_p(x), _p(ws), _p(bs), _p(wo), _p(bo), _p(h0), _p(xp), _p(o), _p(hm), n, c0, c1, h, w, flags, SLOPE,
                                   _s()), "mg_stem_pair")
    return (h0, xp, o, hm) if want_mask else (h0, xp, o)


def stem_pair_gx(gs, ws, go, wo, out=None):
    """gx = ws^T gs + 0.25 * up2(wo^T go): the data gradient of both input branches back to the critic's input."""
    _chk(gs, ws, go, wo, out)
    n, c0, h, w = gs.shape
    c1 = go.shape[1]
    assert go.shape[0] == n and go.shape[2] == h // 2 and go.shape[3] == w // 2
    gx = torch.empty((n, 2, h, w), dtype=torch.float32, device=gs.device) if out is None else out
    check(_lib.load().mg_stem_pair_gx(_p(gs), _p(ws), _p(go), _p(wo), _p(gx), n, c0, c1, h, w, _s()), "mg_stem_pair_gx")
    return gx


def head_pair(x, wh, bh, xl, wo, bo, a: float, b: float, *, coef=None, save=True, out=None):
    """Generator output while a block fades in (generator.py:118-126): (out, mp, old) with mp = tanh(wh x + bh), old = tanh(wo xl + bo),
    out = a mp + b up2(old); `save` False: mp / old are not written (None)."""
    _chk(x, wh, bh, xl, wo, bo, coef, out)
    n, c, h, w = x.shape
    cl = xl.shape[1]
    assert xl.shape[0] == n and xl.shape[2] == h // 2 and xl.shape[3] == w // 2
    new = lambda hh, ww: torch.empty((n, 2, hh, ww), dtype=torch.float32, device=x.device)
    mp, old = (new(h, w), new(h // 2, w // 2)) if save else (None, None)
    out = new(h, w) if out is None else out
    check(_lib.load().mg_head_pair(_p(x), _p(wh), _p(bh), _p(xl), _p(wo), _p(bo), _p(coef), float(a), float(b), _p(mp), _p(old), _p(out),
                                   n, c, cl, h, w, _s()), "mg_head_pair")
    return out, mp, old


def head_pair_from_mp(mp, xl, wo, bo, a: float, b: float, *, coef=None, save=True, out=None):
    """head_pair when the new head's values mp are already there (winoups3x3_head): returns (out, old)."""
    _chk(mp, xl, wo, bo, coef, out)
    n, _, h, w = mp.shape
    cl = xl.shape[1]
    assert xl.shape[0] == n and xl.shape[2] == h // 2 and xl.shape[3] == w // 2
    new = lambda hh, ww: torch.empty((n, 2, hh, ww), dtype=torch.float32, device=mp.device)
    old = new(h // 2, w // 2) if save else None
    out = new(h, w) if out is None else out
    check(_lib.load().mg_head_pair_from_mp(_p(mp), _p(xl), _p(wo), _p(bo), _p(coef), float(a), float(b), _p(old), _p(out), n, cl, h, w,
                                           _s()), "mg_head_pair_from_mp")
    return out, old


def blend_up_bwd(g, a: float, b: float, coef=None):
    """(a g, b * 2x2 block sums of g): backward of blend_up."""
    _chk(g, coef)
    n, c, h, w = g.shape
    gx = torch.empty_like(g)
    gy = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=g.device)
    check(_lib.load().mg_blend_up_bwd(_p(g), _p(coef), float(a), float(b), _p(gx), _p(gy), n * c, h, w, _s()), "mg_blend_up_bwd")
    return gx, gy


def gen_head_bwd_supported(c: int, cout: int = 2, n: int = 0, hw: int = 0) -> bool:
    """The generator head's backward + the PixelNorm / LeakyReLU backward in front of it as one launch (MG_FUSE_ENDS=0: never).
    With (n, hw) given: also the small-map form (any channel count up to 128 on at most 32 768 pixels)."""
    if not (fuse_ends() and _head_fuse(1)):
        return False
    if n > 0 and hw > 0:
        return bool(_lib.load().mg_gen_head_bwd_supported_at(int(c), int(cout), int(n), int(hw)))
    return bool(_lib.load().mg_gen_head_bwd_supported(int(c), int(cout)))


def gen_head_bwd(g_mp, mp, w, p, rn, gw, gb, *, accumulate=False, slope: float = SLOPE, g_in=None):
    """gw (2,C) (+)= sum t p, gb (2) (+)= sum t with t = g_mp (1 - mp^2); returns the gradient at the pre-activation of the conv whose
    LeakyReLU + PixelNorm output is p (rn: its stored 1/norm): what conv1x1_wgrad(tanh_y=) + conv1x1(transposed, tanh_bwd_in=) +
    pixelnorm_lrelu_bwd(from_p=True) return, in one pass over p.  `g_in`: a second gradient at p, added to the head's."""
    _chk(g_mp, mp, w, p, rn, gw, gb, g_in)
    assert g_in is None or g_in.shape == p.shape
    n, c, h, wd = p.shape
    assert g_mp.shape == (n, 2, h, wd) and mp.shape == g_mp.shape and w.shape[0] == 2 and w.shape[1] == c
    lib = _lib.load()
    nfl = lib.mg_gen_head_bwd_ws_floats(n, c, h * wd)
    ws = workspace(4 * nfl, p.device)
    out = torch.empty_like(p)
    check(lib.mg_gen_head_bwd(_p(g_mp), _p(mp), _p(w), _p(p), _p(rn), _p(g_in), _p(out), _p(gw), _p(gb), _p(ws), ws.numel() // 4, n, c, h * wd,
                              float(slope), int(accumulate), _s()), "mg_gen_head_bwd")
    return out


def gp_apply(g, sumsq, factor: float, upstream: float = 1.0, out=None):
    """(penalty, out = g * coef[n]): gp_finish + scale_per_sample in one launch."""
    _chk(g, sumsq, out)
    n = g.shape[0]
    out = torch.empty_like(g) if out is None else out
    pen = torch.empty((), dtype=torch.float32, device=g.device)
    check(_lib.load().mg_gp_apply(_p(g), _p(sumsq), _p(pen), _p(out), n, g[0].numel(), float(factor), float(upstream), _s()), "mg_gp_apply")
    return pen, out


# ------------------------------------------------------------------ element-wise
def pixelnorm_fwd(y):
    _chk(y)
    n, c, h, w = y.shape
    p = torch.empty_like(y)
    rn = torch.empty((n, 1, h, w), dtype=torch.float32, device=y.device)
    check(_lib.load().mg_pixelnorm_fwd(_p(y), _p(p), _p(rn), n, c, h * w, _s()), "mg_pixelnorm_fwd")
    return p, rn


def pixelnorm_lrelu_bwd(gp, y, rn, slope: float = SLOPE, from_p: bool = False):
    """Backward through PixelNorm and the LeakyReLU in front of it.  y = post-LeakyReLU activation, or (from_p) the
    normalised output p = y*rn itself."""
    _chk(gp, y, rn)
    n, c, h, w = y.shape
    out = torch.empty_like(y)
    check(_lib.load().mg_pixelnorm_lrelu_bwd(_p(gp), _p(y), _p(rn), _p(out), n, c, h * w, float(slope), int(from_p),
                                             _s()), "mg_pixelnorm_lrelu_bwd")
    return out


def pixelnorm_bwd(gp, y, rn):
    """Backward through PixelNorm alone (mask slope 1 == identity)."""
    return pixelnorm_lrelu_bwd(gp, y, rn, slope=1.0)


def upsample2x_fwd(x):
    _chk(x)
    n, c, h, w = x.shape
    y = torch.empty((n, c, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    check(_lib.load().mg_upsample2x_fwd(_p(x), _p(y), n * c, h, w, _s()), "mg_upsample2x_fwd")
    return y


def upsample2x_bwd(gy):
    _chk(gy)
    n, c, h2, w2 = gy.shape
    gx = torch.empty((n, c, h2 // 2, w2 // 2), dtype=torch.float32, device=gy.device)
    check(_lib.load().mg_upsample2x_bwd(_p(gy), _p(gx), n * c, h2 // 2, w2 // 2, _s()), "mg_upsample2x_bwd")
    return gx


def avgpool2_fwd(x, out=None):
    _chk(x, out)
    n, c, h, w = x.shape
    y = out if out is not None else torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    check(_lib.load().mg_avgpool2_fwd(_p(x), _p(y), n * c, h, w, _s()), "mg_avgpool2_fwd")
    return y


def avgpool2_bwd(gy, act=None):
    """gx = 0.25 * up2(gy) * lrelu'(act) (act None: no mask; a uint8 act is a tile mask, see conv3x3)."""
    n, c, h2, w2 = gy.shape
    if act is not None and act.dtype == torch.uint8:  # tile mask of conv3x3(mask_out=True)
        _chk(gy)
        _chk_tilemask(act, gy.shape)
        gx = torch.empty((n, c, 2 * h2, 2 * w2), dtype=torch.float32, device=gy.device)
        check(_lib.load().mg_avgpool2_bwd_tilemask(_p(gy), _p(act), _p(gx), n * c, 2 * h2, 2 * w2, SLOPE, _s()),
              "mg_avgpool2_bwd_tilemask")
        return gx
    _chk(gy, act)
    gx = torch.empty((n, c, 2 * h2, 2 * w2), dtype=torch.float32, device=gy.device)
    check(_lib.load().mg_avgpool2_bwd(_p(gy), _p(act), _p(gx), n * c, 2 * h2, 2 * w2, SLOPE, _s()), "mg_avgpool2_bwd")
    return gx


def blend_lrelu_bwd(g, act_a, act_o, ca: float, co: float, coef=None):
    """(ca*g*lrelu'(act_a), co*g*lrelu'(act_o)): backward of the fade-in blend and the two LeakyReLUs in one pass.
    `coef` (all three fade-in ops): a device tensor whose first two floats replace the scalar coefficients (graph replay)."""
    _chk(g, act_a, act_o, coef)
    out_a, out_o = torch.empty_like(g), torch.empty_like(g)
    if coef is not None:
        check(_lib.load().mg_blend_lrelu_bwd_dev(_p(g), _p(act_a), _p(act_o), _p(coef), _p(out_a), _p(out_o), g.numel(), SLOPE,
                                                 _s()), "mg_blend_lrelu_bwd_dev")
    else:
        check(_lib.load().mg_blend_lrelu_bwd(_p(g), _p(act_a), _p(act_o), ca, co, _p(out_a), _p(out_o), g.numel(), SLOPE,
                                             _s()), "mg_blend_lrelu_bwd")
    return out_a, out_o


def lrelu_bwd(g, act, out=None):
    _chk(g, act, out)
    out = torch.empty_like(g) if out is None else out
    check(_lib.load().mg_lrelu_bwd(_p(g), _p(act), _p(out), g.numel(), SLOPE, _s()), "mg_lrelu_bwd")
    return out


def axpby(a: float, x, b: float = 0.0, y=None, out=None, coef=None):
    _chk(x, y, out, coef)
    out = torch.empty_like(x) if out is None else out
    if coef is not None:
        check(_lib.load().mg_axpby_dev(_p(coef), _p(x), _p(y), _p(out), x.numel(), _s()), "mg_axpby_dev")
    else:
        check(_lib.load().mg_axpby(float(a), _p(x), float(b), _p(y), _p(out), x.numel(), _s()), "mg_axpby")
    return out


def blend_up(a: float, x, b: float, ylow, out=None, coef=None):
    _chk(x, ylow, out, coef)
    n, c, h, w = x.shape
    out = torch.empty_like(x) if out is None else out
    if coef is not None:
        check(_lib.load().mg_blend_up_dev(_p(coef), _p(x), _p(ylow), _p(out), n * c, h, w, _s()), "mg_blend_up_dev")
    else:
        check(_lib.load().mg_blend_up(float(a), _p(x), float(b), _p(ylow), _p(out), n * c, h, w, _s()), "mg_blend_up")
    return out


def linear1_fwd(x, w, b):
    _chk(x, w, b)
    n, k = x.shape
    y = torch.empty((n, 1), dtype=torch.float32, device=x.device)
    check(_lib.load().mg_linear1_fwd(_p(x), _p(w), _p(b), _p(y), n, k, _s()), "mg_linear1_fwd")
    return y


def linear1_bwd(x, w, gy, *, gw=None, gb=None, need_gx=True, accumulate=False, bias_n: int = 0):
    """bias_n: the bias gradient sums the first bias_n samples only (0: all)."""
    _chk(x, w, gy, gw, gb)
    n = gy.shape[0]
    k = w.numel()
    gx = torch.empty((n, k), dtype=torch.float32, device=gy.device) if need_gx else None
    check(_lib.load().mg_linear1_bwd(_p(x), _p(w), _p(gy), _p(gx), _p(gw), _p(gb), n, k, int(accumulate), int(bias_n), _s()),
          "mg_linear1_bwd")
    return gx


def gp_interp(x_real, x_fake, eps, out=None):
    _chk(x_real, x_fake, eps, out)
    n = x_real.shape[0]
    out = torch.empty_like(x_real) if out is None else out
    check(_lib.load().mg_gp_interp(_p(x_real), _p(x_fake), _p(eps), _p(out), n, x_real[0].numel(), _s()),
          "mg_gp_interp")
    return out


def sumsq_per_sample(g):
    _chk(g)
    n = g.shape[0]
    out = torch.empty((n,), dtype=torch.float32, device=g.device)
    check(_lib.load().mg_sumsq_per_sample(_p(g), _p(out), n, g[0].numel(), _s()), "mg_sumsq_per_sample")
    return out


def scale_per_sample(g, coef, out=None):
    _chk(g, coef, out)
    n = g.shape[0]
    out = torch.empty_like(g) if out is None else out
    check(_lib.load().mg_scale_per_sample(_p(g), _p(coef), _p(out), n, g[0].numel(), _s()), "mg_scale_per_sample")
    return out


def gp_finish(sumsq, factor: float, upstream: float = 1.0, want_penalty=True, want_coef=True):
    _chk(sumsq)
    n = sumsq.numel()
    pen = torch.empty((), dtype=torch.float32, device=sumsq.device) if want_penalty else None
    coef = torch.empty((n,), dtype=torch.float32, device=sumsq.device) if want_coef else None
    check(_lib.load().mg_gp_finish(_p(sumsq), _p(pen), _p(coef), n, float(factor), float(upstream), _s()),
          "mg_gp_finish")
    return pen, coef


def group_means(scores, groups: int):
    """scores (groups*n, 1) -> float32 tensor [mean_0 .. mean_{groups-1}, loss] with loss = mean_1 - mean_0 (groups >= 2: the
    critic's Wasserstein loss for groups real | fake | ...) or -mean_0 (groups == 1: the generator's)."""
    _chk(scores)
    n = scores.numel() // groups
    assert n * groups == scores.numel()
    out = torch.empty((groups + 1,), dtype=torch.float32, device=scores.device)
    check(_lib.load().mg_group_means(_p(scores), groups, n, _p(out), _s()), "mg_group_means")
    return out


def channel_sum(x, out=None, accumulate=False):
    _chk(x, out)
    n, c, h, w = x.shape
    out = torch.empty((c,), dtype=torch.float32, device=x.device) if out is None else out
    check(_lib.load().mg_channel_sum(_p(x), _p(out), n, c, h * w, int(accumulate), _s()), "mg_channel_sum")
    return out


def stft_1024(wav_mono: torch.Tensor) -> torch.Tensor:
    """mono fp32 [L] -> complex64 [512, 1 + L//256] (audio/functions.py:38-62 without the file read)."""
    _chk(wav_mono)
    length = wav_mono.numel()
    t = 1 + length // 256
    out = torch.empty((512, t, 2), dtype=torch.float32, device=wav_mono.device)
    check(_lib.load().mg_stft_1024(_p(wav_mono), _p(out), None, length, _s()), "mg_stft_1024")
    return torch.view_as_complex(out)


_PCM_KIND = {torch.float32: _lib.MG_PCM_F32, torch.int16: _lib.MG_PCM_I16, torch.int32: _lib.MG_PCM_I32, torch.uint8: _lib.MG_PCM_U8}


def stft_1024_pcm(pcm: torch.Tensor) -> torch.Tensor:
    """PCM frames as a WAV file stores them -- (L, C) or (L,) of float32 / int16 / int32 / uint8, on the device -- -> complex64
    [512, 1 + L//256]: scaling to [-1, 1], the mono mean (audio/functions.py:43-49) and the STFT in one launch."""
    if not pcm.is_cuda or not pcm.is_contiguous() or pcm.dtype not in _PCM_KIND:
        raise _lib.MusicGanHipError(f"stft_1024_pcm: contiguous GPU tensor of float32 / int16 / int32 / uint8 expected, got {pcm.dtype}")
    length, ch = pcm.shape[0], (pcm.shape[1] if pcm.dim() == 2 else 1)
    kind = _PCM_KIND[pcm.dtype]
    lib = _lib.load()
    t = 1 + length // 256
    out = torch.empty((512, t, 2), dtype=torch.float32, device=pcm.device)
    nws = lib.mg_stft_1024_pcm_ws_bytes(length, ch, kind)
    ws = workspace(nws, pcm.device) if nws else None
    check(lib.mg_stft_1024_pcm(_p(pcm), kind, ch, _p(out), None, _p(ws), nws, length, _s()), "mg_stft_1024_pcm")
    return torch.view_as_complex(out)


def codec_fwd(stft_c: torch.Tensor, bark_scale: torch.Tensor, nb_vec: int, stacked: bool = False):
    """complex64 [512, T] -> (magn, phase) each [S, 512, nb_vec] in [-1, 1]  (audio/functions.py:65-94).
    `stacked`: one [S, 2, 512, nb_vec] tensor instead (what create_dataset.py:52-58 stacks), written in place by the kernel."""
    assert stft_c.is_complex() and stft_c.shape[0] == 512
    xr = torch.view_as_real(stft_c.contiguous())
    _chk(xr, bark_scale)
    t = stft_c.shape[1]
    s = (t - 1) // nb_vec
    lib = _lib.load()
    ws = workspace(lib.mg_codec_fwd_ws_bytes(t), xr.device)
    if stacked:
        both = torch.empty((s, 2, 512, nb_vec), dtype=torch.float32, device=xr.device)
        check(lib.mg_codec_fwd_strided(_p(xr), _p(bark_scale), _p(both), both.data_ptr() + 4 * 512 * nb_vec, 2 * 512 * nb_vec,
                                       _p(ws), ws.numel(), t, nb_vec, _s()), "mg_codec_fwd_strided")
        return both
    magn = torch.empty((s, 512, nb_vec), dtype=torch.float32, device=xr.device)
    phase = torch.empty_like(magn)
    check(lib.mg_codec_fwd(_p(xr), _p(bark_scale), _p(magn), _p(phase), _p(ws), ws.numel(), t, nb_vec, _s()),
          "mg_codec_fwd")
    return magn, phase


def pcm_to_mono(pcm: torch.Tensor) -> torch.Tensor:
    """PCM frames (L, C) / (L,) as stored -> mono float32 [L] in [-1, 1] (normalisation + mean over channels)."""
    if not pcm.is_cuda or not pcm.is_contiguous() or pcm.dtype not in _PCM_KIND:
        raise _lib.MusicGanHipError(f"pcm_to_mono: contiguous GPU tensor of float32 / int16 / int32 / uint8 expected, got {pcm.dtype}")
    length, ch = pcm.shape[0], (pcm.shape[1] if pcm.dim() == 2 else 1)
    mono = torch.empty((length,), dtype=torch.float32, device=pcm.device)
    check(_lib.load().mg_pcm_to_mono(_p(pcm), _PCM_KIND[pcm.dtype], ch, _p(mono), length, _s()), "mg_pcm_to_mono")
    return mono


def stft_generic(wav_mono: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
    """mono fp32 [L] -> complex64 [n_fft/2, 1 + L//hop] for any power-of-two n_fft in [64, 8192] (untuned path)."""
    _chk(wav_mono)
    length = wav_mono.numel()
    out = torch.empty((n_fft // 2, 1 + length // hop, 2), dtype=torch.float32, device=wav_mono.device)
    check(_lib.load().mg_stft_generic(_p(wav_mono), _p(out), length, int(n_fft), int(hop), _s()), "mg_stft_generic")
    return torch.view_as_complex(out)


def crc32_of_float64(x: torch.Tensor) -> torch.Tensor:
    """x: float32 (n, ...) on the device -> int64 tensor [n] (values < 2^32) of zlib.crc32(x[i].double().numpy().tobytes())."""
    _chk(x)
    n, per = x.shape[0], x[0].numel()
    lib = _lib.load()
    nws = lib.mg_crc32_f64_ws_bytes(n, per)
    ws = workspace(nws, x.device)
    out = torch.empty((n,), dtype=torch.int32, device=x.device)
    check(lib.mg_crc32_f64(_p(x), ctypes.c_void_p(out.data_ptr()), _p(ws), ws.numel(), n, per, _s()), "mg_crc32_f64")
    return out.to(torch.int64) & 0xFFFFFFFF


def codec_inv(magn_phase: torch.Tensor, bark_scale: torch.Tensor) -> torch.Tensor:
    """[N, 2, 512, W] -> waveform [256*(N*W-1)]  (audio/functions.py:97-139 without the file write)."""
    _chk(magn_phase, bark_scale)
    n, _, _, w = magn_phase.shape
    lib = _lib.load()
    ws = workspace(lib.mg_codec_inv_ws_bytes(n, w), magn_phase.device)
    wav = torch.empty((256 * (n * w - 1),), dtype=torch.float32, device=magn_phase.device)
    check(lib.mg_codec_inv(_p(magn_phase), _p(bark_scale), _p(wav), _p(ws), ws.numel(), n, w, _s()), "mg_codec_inv")
    return wav


# ------------------------------------------------------------------ input transform (Grower)
def input_transform(x: torch.Tensor, side: int, eps: float = 1e-8) -> torch.Tensor:
    """ChannelMinMaxNorm -> ChangeRange(-1, 1) -> Resize(side) of a (N, 2, H, W) float64 / float32 batch, one fused pass on the
    device (audio/transforms.py + utils.Grower of the reference, which run per batch on the CPU)."""
    assert x.is_cuda and x.dim() == 4 and x.shape[1] == 2 and x.is_contiguous()
    assert x.dtype in (torch.float64, torch.float32)
    n, _, h, w = x.shape
    lib = _lib.load()
    out = torch.empty((n, 2, side, side), dtype=torch.float32, device=x.device)
    ws = workspace(lib.mg_input_transform_ws_bytes(n, h, w, side), x.device)
    check(lib.mg_input_transform(_p(x), int(x.dtype == torch.float64), _p(out), _p(ws), ws.numel(), n, h, w, side, eps, _s()),
          "mg_input_transform")
    return out
