"""Layer plans of the ProGAN generator / discriminator on the HIP kernels: forward, hand-derived backward, and the WGAN-GP
second-order pass.  Pure host-side sequencing (Python) over `musicgan_amd.ops`; no autograd inside.

Math restated from /root/reference/music_gan/networks (generator.py:106-126, discriminator.py:107-124,157-184):

Generator block i (C_i -> C_o):  y1 = LReLU(conv3(x));  p1 = PN(y1);  y2 = LReLU(conv3(up2(p1)));  p2 = PN(y2)
  backward: g_pre = mask(y) * rn * (g_p - p * mean_c(g_p * p))   [PixelNorm + LeakyReLU, one kernel]
            wgrad(x_in, g_pre), dgrad(g_pre) ; up2 backward = 2x2 block sums.
Discriminator block i:  a1 = LReLU(conv3(in)); q1 = avgpool2(a1); a2 = LReLU(conv3(q1))
  backward: the LeakyReLU mask of the layer below is fused on the OUTPUT of each data-gradient conv (MG_CONV_MASK_AUX).

Gradient penalty  P = 10 * mean_n (||g_0[n]|| - 1)^2,  g_0 = d D(x~) / d x~  (first-order chain with g_out = 1):
  the chain is linear in every weight: g_{l-1} = L_l(W_l)^T h_l with h_l = mask_l * g_l (masks are piecewise constant, their
  derivative is zero -- autograd's leaky_relu double-backward returns zeros there too).  Back-propagating P through that chain
  is a *forward* pass of the bias-free network on u_0 = dP/dg_0 with the saved masks:
        u_l = mask_l * L_l(W_l) u_{l-1},        dP/dW_l = wgrad(x = u_{l-1}, gy = h_l),      dP/db = 0
  so the penalty's parameter gradient costs one extra forward-like pass and one wgrad per layer (10 D-forward equivalents per
  D step in total) and never touches the autograd engine.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from .. import _lib, ops


def gen_conv_form(n: int, cin: int, cout: int, h: int, wd: int, ups: bool) -> str:
    """Which kernels a generator `(Upsample ->) Conv3x3 -> LeakyReLU -> PixelNorm` (generator.py:9-40) runs as, for an input
    (n, cin, h, wd).  The ONE place that decides it: `PackCache.conv_lrelu_pixnorm` executes the answer and bench.py's
    executed-FLOP count reads it (returns (form, winograd?) through `gen_conv_is_wino`)."""
    ho, wo = (2 * h, 2 * wd) if ups else (h, wd)
    if n * ho * wo < int(os.environ.get("MG_PN_FUSE_MIN_PIXELS", "16384")):
        return "conv+pixnorm"
    if ups and n * h * wd < int(os.environ.get("MG_UPCONV_MIN_LOWRES_PIXELS", "16384")) and \
            ops.wino3x3_supported(n, cout, ho, wo, cin=cin):
        return "upsample+wino+pixnorm"
    if not ups and cout > 64 and os.environ.get("MG_PN_WIDE_UNFUSED", "1") == "1" and \
            ops.wino3x3_supported(n, cout, h, wd, cin=cin):
        return "wino+pixnorm"
    if ups and ops.upconv3x3_supported(cout, wd, n * cin * h * wd):
        return "subpixel-fused"
    return "fused"


def gen_conv_exec_factor(n: int, cin: int, cout: int, h: int, wd: int, ups: bool) -> float:
    """Executed / direct multiplies of that convolution's forward pass: 1, 1 / 2.25 (Winograd, sub-pixel) or 0.25 (9-component
    Winograd on the up-sampled grid)."""
    if ups and gen_conv_form(n, cin, cout, h, wd, ups) == "subpixel-fused" and ops.winoups3x3_supported(n, cin, cout, h, wd):
        return 0.25
    return 1.0 / 2.25 if gen_conv_executes_reduced(n, cin, cout, h, wd, ups) else 1.0


def gen_conv_executes_reduced(n: int, cin: int, cout: int, h: int, wd: int, ups: bool) -> bool:
    """True when that convolution's forward issues 1/2.25 of the direct-convolution multiplies (Winograd or sub-pixel form)."""
    form = gen_conv_form(n, cin, cout, h, wd, ups)
    if form in ("upsample+wino+pixnorm", "wino+pixnorm", "subpixel-fused"):
        return True
    if form == "fused":
        return ops.wino3x3_supported(n, cout, h, wd, ups=ups, pixnorm=True, cin=cin)
    return ops.wino3x3_supported(n, cout, h, wd, ups=ups, cin=cin)  # "conv+pixnorm": PackCache.conv's own choice


class PackCache:
    """Packed (LDS-image) conv3x3 weights.  A layout is packed when first asked for; afterwards its buffer is REFRESHED IN PLACE
    whenever the parameter's storage or version has changed -- one by one on demand, or all stale ones of the cache in a single
    launch through `refresh()` (the steppers call it at the top of every update: one launch instead of ~30 per network, and a
    captured HIP graph of the update re-packs into the same buffers on every replay)."""

    def __init__(self):
        self._c: Dict = {}  # (id(w), kind, dgrad) -> [tag, packed, w]

    @staticmethod
    def _tag(w: torch.Tensor):
        return (w.data_ptr(), w._version, w.device)

    def _get(self, w: torch.Tensor, kind: int, dgrad: bool) -> torch.Tensor:
        key = (id(w), kind, dgrad)
        ent = self._c.get(key)
        if ent is None:
            out = torch.empty(ops.packed_floats(kind, w.shape[0], w.shape[1], dgrad), dtype=torch.float32, device=w.device)
            ent = self._c[key] = [None, out, w]
        if ent[0] != self._tag(w):
            ops.pack_multi([(kind, w.detach(), dgrad, ent[1])])
            ent[0] = self._tag(w)
        return ent[1]

    def refresh(self, force: bool = False) -> None:
        """Re-pack every known layout whose weight has changed (`force`: all of them) in one launch."""
        stale = [(key, ent) for key, ent in self._c.items() if force or ent[0] != self._tag(ent[2])]
        ops.pack_multi([(key[1], ent[2].detach(), key[2], ent[1]) for key, ent in stale])
        for _, ent in stale:
            ent[0] = self._tag(ent[2])

    def invalidate(self) -> None:
        """Mark every layout stale (its buffer is kept and re-packed on next use / refresh)."""
        for ent in self._c.values():
            ent[0] = None

    def get(self, w: torch.Tensor, dgrad: bool) -> torch.Tensor:
        return self._get(w, _lib.MG_PACK_CONV3X3, dgrad)

    def get_wino(self, w: torch.Tensor, dgrad: bool) -> torch.Tensor:
        """Winograd-domain filters U = G g G^T."""
        return self._get(w, _lib.MG_PACK_WINO3X3, dgrad)

    def conv(self, x: torch.Tensor, w: torch.Tensor, dgrad: bool, bias, cout: int, **kw):
        """ops.conv3x3 with the kernel chosen per shape: Winograd F(2x2,3x3) where it is supported and pays (large maps, no
        fused up-sampling), the direct implicit GEMM otherwise.  Only the form that is used gets packed."""
        n, cin, h, wd = x.shape
        if self.small_ok(x, cout, **kw):
            # <= 8x8 maps of a few images: the latency-optimised one-layer kernel (split-K over the waves, the whole
            # filter share of a wave in flight at once; a fused AvgPool2d is a second output of the same launch)
            pool = kw.get("pool", False) or kw.get("pool_out") is not None
            return ops.conv3x3_small(x, self.get_sn(w, dgrad), bias, cout, ups=kw.get("ups", False), lrelu=kw.get("lrelu", False),
                                     mask_aux=kw.get("mask_aux"), out=kw.get("out"), pool=pool, pool_out=kw.get("pool_out"))
        if ops.wino3x3_supported(n, cout, h, wd, ups=kw.get("ups", False), pixnorm=kw.get("pixnorm", False), cin=cin):
            return ops.conv3x3(x, None, bias, cout, wino=self.get_wino(w, dgrad), **kw)
        return ops.conv3x3(x, self.get(w, dgrad), bias, cout, **kw)

    @staticmethod
    def small_ok(x: torch.Tensor, cout: int, **kw) -> bool:
        """Whether `conv` routes this call to ops.conv3x3_small: supported epilogue (no PixelNorm / tile masks), output map within
        MG_SMALLCONV_MAX_SIDE (default 8) and few enough pixels to be latency-bound (MG_SMALLCONV_MAX_PIXELS, default 3072 up to 4x4, 2048 at 8x8)."""
        if kw.get("pixnorm") or kw.get("mask_out") or kw.get("unpool_mask") is not None or kw.get("want_y", True) is False:
            return False
        m = kw.get("mask_aux")
        if m is not None and m.dtype != torch.float32:
            return False
        n, cin, h, wd = x.shape
        if kw.get("ups"):
            h, wd = 2 * h, 2 * wd
        side = int(os.environ.get("MG_SMALLCONV_MAX_SIDE", "16"))
        # measured against the direct / Winograd kernels (tools/ab_smallconv.py): ahead up to 192 images at 2x2 / 4x4 (whole images
        # per workgroup), up to 32 images at 8x8 (row bands); beyond that each layer is throughput- not latency-bound
        cap = int(os.environ.get("MG_SMALLCONV_MAX_PIXELS", "3072" if h * wd <= 16 else os.environ.get("MG_SMALLCONV_MAX_PIXELS_BIG", "2048")))
        if h > side or wd > side or n * h * wd > cap:
            return False
        return ops.conv3x3_small_supported(n, cin, cout, h, wd)

    def conv_unpool(self, gy: torch.Tensor, w: torch.Tensor, cout: int, act: torch.Tensor) -> torch.Tensor:
        """0.25 * up2(dgrad-conv(gy, w)) * lrelu'(act): the data gradient through conv -> LeakyReLU -> AvgPool2d back to the
        pre-activation of the layer in front of the pool (act: its fp32 activation, 2H x 2W) -- one launch on small maps."""
        if act.dtype == torch.float32 and self.small_ok(gy, cout):
            return ops.conv3x3_small(gy, self.get_sn(w, True), None, cout, unpool_aux=act)
        return ops.avgpool2_bwd(self.conv(gy, w, True, None, cout), act)

    def conv_upsum(self, gy: torch.Tensor, w: torch.Tensor, cout: int) -> torch.Tensor:
        """2x2 block sums of dgrad-conv(gy, w): the data gradient through Upsample(x2) -> conv back to the low-resolution input."""
        if self.small_ok(gy, cout):
            return ops.conv3x3_small(gy, self.get_sn(w, True), None, cout, upsum=True, want_y=False)[1]
        return ops.upsample2x_bwd(self.conv(gy, w, True, None, cout))

    def conv_lrelu_pixnorm(self, x: torch.Tensor, w: torch.Tensor, bias, cout: int, ups: bool = False, head=None):
        """conv3x3 (+ nearest x2 up-sampling of its input) + LeakyReLU + PixelNorm -> (p, 1/norm).  The fused kernels keep all
        output channels of a pixel in one wave, so the grid is output pixels / 64 workgroups; on the small maps at the start of the
        generator that is 8 .. 128 workgroups for 256 CUs, each walking all of Cin x 9 taps x Cout alone (60-80 us for 0.1-1.4
        GFLOP).  There the convolution runs sliced over out-channels (many short workgroups) and PixelNorm as its own pass over
        the (tiny) result.  `head` = [hw, hb, mp_out or None, None]: where the conv runs as the 9-component up-sampling kernel the
        generator's 1x1 head is computed in its epilogue and head[3] receives tanh(hw p + hb); elsewhere head[3] stays None."""
        n, cin, h, wd = x.shape
        form = gen_conv_form(n, cin, cout, h, wd, ups)
        if head is not None and form == "subpixel-fused" and ops.winoups3x3_head_supported(n, cin, cout, h, wd) and ops.fuse_ends():
            _, p, rn, head[3] = ops.winoups3x3_head(x, self.get_wu(w, False), bias, cout, head[0], head[1], mp_out=head[2])
            return p, rn
        if form == "upsample+wino+pixnorm":
            # an under-filled sub-pixel launch (one 4-wave workgroup per 64 low-res pixels, all channels in a wave): the up-sampled
            # tensor written out once + the Winograd conv sliced over out-channels + PixelNorm as its own pass fill the chip
            return ops.pixelnorm_fwd(self.conv(ops.upsample2x_fwd(x), w, False, bias, cout, lrelu=True))
        if form == "wino+pixnorm":
            # more than 64 channels cannot take the Winograd kernel's fused PixelNorm (all channels of a pixel in one workgroup)
            # and the direct kernel's fused form puts 5-8 channel tiles on one wave: the Winograd conv + a PixelNorm pass is faster
            return ops.pixelnorm_fwd(self.conv(x, w, False, bias, cout, lrelu=True))
        if form == "subpixel-fused":
            if ops.winoups3x3_supported(n, cin, cout, h, wd):  # Winograd on the up-sampled grid, 9 of 16 components: 4x fewer MFMAs
                _, p, rn = ops.winoups3x3(x, self.get_wu(w, False), bias, cout, lrelu=True, pixnorm=True, want_y=False)
                return p, rn
            _, p, rn = ops.upconv3x3(x, self.get_up(w), bias, cout, lrelu=True, pixnorm=True, want_y=False)  # sub-pixel form: 2.25x fewer
            return p, rn
        if form == "fused":
            _, p, rn = self.conv(x, w, False, bias, cout, ups=ups, lrelu=True, pixnorm=True, want_y=False)
            return p, rn
        return ops.pixelnorm_fwd(self.conv(x, w, False, bias, cout, ups=ups, lrelu=True))  # "conv+pixnorm"

    def get_sn(self, w: torch.Tensor, dgrad: bool) -> torch.Tensor:
        """Filters in the operand order the multi-layer small-map kernel streams (ops.SmallNet)."""
        return self._get(w, _lib.MG_PACK_SMALLNET, dgrad)

    def get_up(self, w: torch.Tensor) -> torch.Tensor:
        """Effective sub-pixel weights of Upsample(x2) -> Conv3x3 (ops.upconv3x3)."""
        return self._get(w, _lib.MG_PACK_UPCONV3X3, False)

    def get_wu(self, w: torch.Tensor, dgrad: bool) -> torch.Tensor:
        """9-component Winograd filters of Upsample(x2) -> Conv3x3 / its data gradient (ops.winoups3x3, ops.winoups3x3_dgrad)."""
        return self._get(w, _lib.MG_PACK_WINOUPS, dgrad)

    def get_up_dgrad(self, w: torch.Tensor) -> torch.Tensor:
        """4x4 stride-2 effective kernel of the upsample-conv data gradient (ops.upconv3x3_dgrad)."""
        return self._get(w, _lib.MG_PACK_UPCONV3X3_DGRAD, False)

    def clear(self):
        self._c.clear()


_UPSTREAM: Dict[tuple, torch.Tensor] = {}


def upstream_scores_grad(kind: str, n: int, device) -> torch.Tensor:
    """d loss / d critic scores, a constant of the batch size: "critic": [-1/n]*n + [+1/n]*n + [1]*n for the fused critic step's
    [real | fake | interpolated] batch; "gen": [-1/n]*n.  Built once per (kind, n, device) and only ever read."""
    key = (kind, n, str(device))
    t = _UPSTREAM.get(key)
    if t is None:
        if kind == "critic":
            t = torch.cat([torch.full((n, 1), -1.0 / n), torch.full((n, 1), 1.0 / n), torch.ones(n, 1)]).to(device)
        else:
            t = torch.full((n, 1), -1.0 / n).to(device)
        _UPSTREAM[key] = t
    return t


class FadeIn:
    """The fade-in coefficients (alpha, 1 - alpha) of generator.py:124 / discriminator.py:113 as launch scalars -- or, `dev` given
    (float32 device tensor [alpha, 1 - alpha]), read by the kernels from device memory, which keeps a captured HIP graph of an
    update valid while alpha moves.  Every engine function accepts a plain float or a FadeIn for `alpha`."""

    def __init__(self, alpha, dev: Optional[torch.Tensor] = None):
        self.a = float(alpha)
        self.b = 1.0 - float(alpha)
        self.dev = dev                                   # coefficients (a, b)
        self.dev_b = dev[1:] if dev is not None else None  # coefficient b alone (single-coefficient scalings)

    @staticmethod
    def of(alpha) -> "FadeIn":
        return alpha if isinstance(alpha, FadeIn) else FadeIn(alpha)

    def coef(self, device) -> torch.Tensor:
        """The coefficients in device memory (kernels that only take them from there): `dev`, or a filled scratch pair."""
        if self.dev is not None:
            return self.dev
        key = (self.a, str(device))
        t = _COEF.get(key)
        if t is None:
            if len(_COEF) >= 16:
                _COEF.clear()
            t = torch.empty(2, dtype=torch.float32, device=device)
            t[0:1].fill_(self.a)
            t[1:2].fill_(self.b)
            _COEF[key] = t
        return t


_COEF: Dict[tuple, torch.Tensor] = {}


# =====================================================================================================================
# Multi-layer chains on the <= 4x4 maps (ops.SmallNet / mg_smallnet): the critic's last blocks + classifier and the generator's
# first blocks as ONE launch per pass instead of 6-10 (MG_SMALLNET=0: one launch per layer everywhere)
# =====================================================================================================================
def _smallnet_on() -> bool:
    """Opt-in (MG_SMALLNET=1): measured on the MI355X, one CU per image through the whole chain (80-110 us per pass) loses to
    the per-layer launches it replaces (75-80 us with the direct kernel, ~45 with ops.conv3x3_small): a CU's fp32 matrix rate
    and its ~70 GB/s filter stream bound the chain, while a per-layer launch spreads each layer over the chip."""
    return os.environ.get("MG_SMALLNET", "0") == "1"


def _imgs_per_wg(n: int) -> int:
    """One image per workgroup while every image still gets a CU of its own (shortest chain); two beyond that (half the
    filter traffic per image)."""
    return 1 if n <= 256 else 2


def disc_tail_start(W: "DiscWeights", h: int, w: int) -> Optional[int]:
    """Index t of the critic block whose first conv runs at 8x8 (so its second conv, and every later layer, sits on <= 4x4
    maps), when the fused tail applies: square input, final map 1x1, and t >= 1 so that the fade-in blend (block 0) stays
    outside it.  The tail then covers conv2 of block t, blocks t+1 and t+2, and the classifier."""
    if not _smallnet_on() or h != w:
        return None
    t = len(W.blocks) - 3
    if t < 1 or h != (8 << t):
        return None
    return t


def _rot(cur: int):
    """The two LDS buffers other than `cur`."""
    return [b for b in (0, 1, 2) if b != cur]


def gen_head_ok(W: "GenWeights", z: torch.Tensor) -> bool:
    """The generator's first three convs (block 0 and the first conv of block 1: 2x2 and 4x4 maps) as one launch: latent maps of
    2x2 only (generate's wide latents take the per-layer path) and at least three blocks, so that neither head hangs off them."""
    return _smallnet_on() and len(W.blocks) >= 3 and tuple(z.shape[2:]) == (2, 2)


class GradSink:
    """Collects parameter gradients keyed by the parameter object; a second write to the same key accumulates.
    `flat` (optional, with `layout` = {id(param): (offset, numel)}): the gradients are carved out of ONE buffer, which is what the
    data-parallel path all-reduces -- no flatten / un-flatten copies around the exchange."""

    def __init__(self, flat: Optional[torch.Tensor] = None, layout: Optional[Dict[int, tuple]] = None):
        self.g: Dict[int, torch.Tensor] = {}
        self.flat, self.layout = flat, layout

    def slot(self, param: torch.Tensor):
        k = id(param)
        if k in self.g:
            return self.g[k], True
        if self.flat is not None and k in self.layout:
            off, n = self.layout[k]
            t = self.flat[off:off + n].view(param.shape)
        else:
            t = torch.empty_like(param, memory_format=torch.contiguous_format)
        self.g[k] = t
        return t, False

    def get(self, param: torch.Tensor) -> Optional[torch.Tensor]:
        return self.g.get(id(param))


# =====================================================================================================================
# Generator
# =====================================================================================================================
class GenWeights:
    """Views of the live generator parameters at the current level."""

    def __init__(self, blocks, head, old_head):
        self.blocks = blocks      # list of (w1, b1, w2, b2)
        self.head = head          # (w, b)
        self.old_head = old_head  # (w, b) or None

    def tensors(self) -> List[torch.Tensor]:
        out = []
        for blk in self.blocks:
            out += list(blk)
        out += list(self.head)
        if self.old_head is not None:
            out += list(self.old_head)
        return out


def gen_forward(W: GenWeights, z: torch.Tensor, alpha: float, cache: PackCache, save: bool, out=None):
    x = z.contiguous()
    saved = []
    # the 1x1 head in the last conv's epilogue where that conv is the 9-component up-sampling kernel: [hw, hb, mp buffer, result]
    head_req = [W.head[0], W.head[1], out if W.old_head is None else None, None] if W.head[0].shape[0] == 2 else None
    head = gen_head_ok(W, x)
    if head:
        # block 0 and the first conv of block 1 in one launch (activations in LDS); only what the backward pass needs is stored
        n, rc = x.shape[0], x.shape[1]
        (w1a, b1a, w2a, b2a), (w1b, b1b, _, _) = W.blocks[0], W.blocks[1]
        c0, c1 = w2a.shape[0], w1b.shape[0]
        new = lambda c, s: torch.empty((n, c, s, s), dtype=torch.float32, device=x.device)
        p1a, rn1a, p2a, rn2a = (new(rc, 2), new(1, 2), new(c0, 4), new(1, 4)) if save else (None,) * 4
        p1b, rn1b = new(c1, 4), (new(1, 4) if save else None)
        sn = ops.SmallNet(_imgs_per_wg(n))
        sn.load(0, x)
        sn.conv(0, 1, cache.get_sn(w1a, False), rc, rc, 2, 2, bias=b1a, lrelu=True).pixnorm(1, rc, 2, 2, p1a, rn1a)
        sn.up(1, 2, rc, 2, 2)
        sn.conv(2, 0, cache.get_sn(w2a, False), rc, c0, 4, 4, bias=b2a, lrelu=True).pixnorm(0, c0, 4, 4, p2a, rn2a)
        sn.conv(0, 1, cache.get_sn(w1b, False), c0, c1, 4, 4, bias=b1b, lrelu=True).pixnorm(1, c1, 4, 4, p1b, rn1b)
        sn.run(n)
    if not head and os.environ.get("MG_PN_STAGED", "1") != "0":
        # PixelNorm folded into the NEXT convolution's input staging wherever both neighbours are small-map launches: a conv leaves
        # its LeakyReLU output un-normalised ("raw") and the following mg_conv3x3_small_pn normalises while staging, writing p and
        # 1 / norm as side outputs -- four PixelNorm launches fewer per generator forward pass from level 3 on.
        convs = []
        for (w1, b1, w2, b2) in W.blocks:
            convs += [(w1, b1, w1.shape[0], False), (w2, b2, w2.shape[0], True)]
        cur, raw = x, None          # the normalised input of the next conv, or (raw) an activation whose PixelNorm is pending
        pr = [None] * len(convs)    # (p, rn) per conv
        for k, (w, b, cout, ups) in enumerate(convs):
            if raw is not None:
                probe = raw
                if PackCache.small_ok(probe, cout, ups=ups, lrelu=True):
                    # (without `save` only the input of the last block is still needed as a tensor: the old head reads it)
                    need = save or (W.old_head is not None and k - 1 == len(convs) - 3)
                    y, p_prev, rn_prev = ops.conv3x3_small_pn(raw, cache.get_sn(w, False), b, cout, ups=ups, save=need)
                    pr[k - 1] = (p_prev, rn_prev)
                    raw = y
                    continue
                cur, rn_prev = ops.pixelnorm_fwd(raw)
                pr[k - 1] = (cur, rn_prev)
                raw = None
            n_, ci_, h_, w_ = cur.shape
            if gen_conv_form(n_, ci_, cout, h_, w_, ups) == "conv+pixnorm" and PackCache.small_ok(cur, cout, ups=ups, lrelu=True):
                raw = cache.conv(cur, w, False, b, cout, ups=ups, lrelu=True)  # its PixelNorm: by the next conv, or below
            else:
                cur, rn_k = cache.conv_lrelu_pixnorm(cur, w, b, cout, ups=ups, head=head_req if k == len(convs) - 1 else None)
                pr[k] = (cur, rn_k)
        if raw is not None:
            cur, rn_k = ops.pixelnorm_fwd(raw)
            pr[-1] = (cur, rn_k)
        xin = x
        for bi in range(len(W.blocks)):
            (p1, rn1), (p2, rn2) = pr[2 * bi], pr[2 * bi + 1]
            if save:
                saved.append((xin, rn1, p1, rn2, p2))
            x_in_last, xin = xin, p2
        x = xin
        blocks_iter = ()
    else:
        blocks_iter = W.blocks
    for bi, (w1, b1, w2, b2) in enumerate(blocks_iter):
        ci, co = w1.shape[0], w2.shape[0]
        if head and bi == 0:
            if save:
                saved.append((x, rn1a, p1a, rn2a, p2a))
            x_in_last, x = x, p2a
            continue
        # only the normalised outputs p and the per-pixel 1/norm are kept: the backward derives mask and x_hat from p
        if head and bi == 1:
            p1, rn1 = p1b, rn1b
        else:
            p1, rn1 = cache.conv_lrelu_pixnorm(x, w1, b1, ci)
        p2, rn2 = cache.conv_lrelu_pixnorm(p1, w2, b2, co, ups=True, head=head_req if bi == len(W.blocks) - 1 else None)
        if save:
            saved.append((x, rn1, p1, rn2, p2))
        x_in_last, x = x, p2
    old = None
    mp_epi = head_req[3] if head_req is not None else None
    if mp_epi is not None and W.old_head is None:
        out = mp = mp_epi
    elif mp_epi is not None and ops.head_pair_supported(x.shape[1], x_in_last.shape[1], x.shape[2], x.shape[3]):
        F = FadeIn.of(alpha)
        mp = mp_epi
        out, old = ops.head_pair_from_mp(mp, x_in_last, W.old_head[0], W.old_head[1], F.a, F.b, coef=F.dev, save=save, out=out)
    elif W.old_head is not None:
        F = FadeIn.of(alpha)
        if ops.head_pair_supported(x.shape[1], x_in_last.shape[1], x.shape[2], x.shape[3]):
            # both heads, the up-sampling of the old one and the blend in one launch
            out, mp, old = ops.head_pair(x, W.head[0], W.head[1], x_in_last, W.old_head[0], W.old_head[1], F.a, F.b, coef=F.dev,
                                         save=save, out=out)
        else:
            mp = ops.conv1x1(x, W.head[0], W.head[1], 2, tanh=True)
            old = ops.conv1x1(x_in_last, W.old_head[0], W.old_head[1], 2, tanh=True)
            out = ops.blend_up(F.a, mp, F.b, old, out=out, coef=F.dev)
    else:
        out = mp = ops.conv1x1(x, W.head[0], W.head[1], 2, tanh=True, out=out)
    ctx = (saved, x, mp, old, alpha) if save else None
    return out, ctx


def gen_backward(W: GenWeights, ctx, g_out: torch.Tensor, cache: PackCache, sink: GradSink, need_gz: bool = False,
                 defer: Optional["ops.WgradDefer"] = None):
    """`defer`: the Winograd weight-gradient reductions are collected there; the caller flushes it (one launch for all layers)."""
    saved, x_last, mp, old, alpha = ctx
    F = FadeIn.of(alpha)
    g_out = g_out.contiguous()
    if W.old_head is not None and ops.blend_up_bwd_supported(g_out.shape[2], g_out.shape[3]):
        g_mp, g_old = ops.blend_up_bwd(g_out, F.a, F.b, coef=F.dev)
    elif W.old_head is not None:
        g_mp = ops.axpby(F.a, g_out, coef=F.dev)
        g_old = ops.upsample2x_bwd(g_out)
        g_old = ops.axpby(F.b, g_old, out=g_old, coef=F.dev_b)
    else:
        g_mp, g_old = g_out, None
    gw, acc = sink.slot(W.head[0])
    gb, _ = sink.slot(W.head[1])
    last = len(W.blocks) - 1
    gpre_head = None
    if ops.gen_head_bwd_supported(x_last.shape[1], W.head[0].shape[0], x_last.shape[0], x_last.shape[2] * x_last.shape[3]) \
            and saved[last][4] is x_last:
        # the head's weight gradient, its data gradient and the PixelNorm / LeakyReLU backward of the last conv: one pass over p
        gpre_head = ops.gen_head_bwd(g_mp, mp, W.head[0], x_last, saved[last][3], gw, gb, accumulate=acc)
        g = None
    else:
        ops.conv1x1_wgrad(x_last, g_mp, gw, gb, tanh_y=mp, accumulate=acc)
        g = ops.conv1x1(g_mp, W.head[0], None, x_last.shape[1], transposed=True, tanh_bwd_in=mp)
    gz = None
    old_pending = None
    for i in range(last, -1, -1):
        w1, b1, w2, b2 = W.blocks[i]
        xin, rn1, p1, rn2, p2 = saved[i]
        ci = w1.shape[0]
        if i == last and gpre_head is not None:
            gpre2 = gpre_head
        elif old_pending is not None:
            # the old head's weight / data gradient joins the conv's data gradient g inside the PixelNorm backward's pass over p2
            gwo, acc = sink.slot(W.old_head[0])
            gbo, _ = sink.slot(W.old_head[1])
            gpre2 = ops.gen_head_bwd(g_old, old, W.old_head[0], p2, rn2, gwo, gbo, accumulate=acc, g_in=g)
            old_pending = None
        else:
            gpre2 = ops.pixelnorm_lrelu_bwd(g, p2, rn2, from_p=True)
        gw2, acc = sink.slot(w2)
        gb2, _ = sink.slot(b2)
        ops.conv3x3_wgrad(p1, gpre2, gw2, gb2, ups=True, accumulate=acc, defer=defer)
        small_tail = i == 1 and gen_head_ok(W, saved[0][0])
        gpre1 = None
        if ops.winoups3x3_supported(p1.shape[0], ci, w2.shape[0], p1.shape[2], p1.shape[3], dgrad=True):
            if not small_tail and ops.winoups3x3_dgrad_pn_supported(p1.shape[0], ci, w2.shape[0], p1.shape[2], p1.shape[3]):
                # 9-component Winograd form, block sums AND the PixelNorm / LeakyReLU backward of conv 1 in the epilogue
                gp1, gpre1 = None, ops.winoups3x3_dgrad_pn(gpre2, cache.get_wu(w2, True), p1, rn1, ci)
            else:
                gp1 = ops.winoups3x3_dgrad(gpre2, cache.get_wu(w2, True), ci)  # 9-component Winograd form, block sums in the epilogue
        elif ops.upconv3x3_dgrad_supported(p1.shape[2], p1.shape[3], gpre2.numel(), p1.shape[0]):
            gp1 = ops.upconv3x3_dgrad(gpre2, cache.get_up_dgrad(w2), ci)  # stride-2 4x4 form: no high-res intermediate
        else:
            gp1 = cache.conv_upsum(gpre2, w2, ci)
        if small_tail:
            # the data-gradient chain of the generator's first three convs in one launch; their weight gradients from the
            # stored masked gradients
            z, rn1a, p1a, rn2a, p2a = saved[0]
            w1a, b1a, w2a, b2a = W.blocks[0]
            n, rc, c0 = z.shape[0], z.shape[1], w2a.shape[0]
            new = lambda c, s: torch.empty((n, c, s, s), dtype=torch.float32, device=z.device)
            gpre1, gpre2a, gpre1a = new(ci, 4), new(c0, 4), new(rc, 2)
            gz = new(rc, 2) if need_gz else None
            sn = ops.SmallNet(_imgs_per_wg(n))
            sn.load(0, gp1)
            sn.pnbwd(0, ci, 4, 4, p1, rn1, out=gpre1)
            sn.conv(0, 1, cache.get_sn(w1, True), ci, c0, 4, 4)
            sn.pnbwd(1, c0, 4, 4, p2a, rn2a, out=gpre2a)
            sn.conv(1, 2, cache.get_sn(w2a, True), c0, rc, 4, 4)
            sn.upbwd(2, 0, rc, 4, 4)
            sn.pnbwd(0, rc, 2, 2, p1a, rn1a, out=gpre1a)
            if need_gz:
                sn.conv(0, 1, cache.get_sn(w1a, True), rc, rc, 2, 2, out=gz)
            sn.run(n)
            for wt, bt, xi, gy, ups in ((w1, b1, xin, gpre1, False), (w2a, b2a, p1a, gpre2a, True), (w1a, b1a, z, gpre1a, False)):
                gwt, acc = sink.slot(wt)
                gbt, _ = sink.slot(bt)
                ops.conv3x3_wgrad(xi, gy, gwt, gbt, ups=ups, accumulate=acc, defer=defer)
            break
        if gpre1 is None:
            gpre1 = ops.pixelnorm_lrelu_bwd(gp1, p1, rn1, from_p=True)
        gw1, acc = sink.slot(w1)
        gb1, _ = sink.slot(b1)
        ops.conv3x3_wgrad(xin, gpre1, gw1, gb1, accumulate=acc, defer=defer)
        if i > 0:
            g = cache.conv(gpre1, w1, True, None, ci)
            if g_old is not None and i == last and ops.gen_head_bwd_supported(xin.shape[1], W.old_head[0].shape[0], xin.shape[0],
                                                                              xin.shape[2] * xin.shape[3]) \
                    and saved[last - 1][4] is xin:
                old_pending = True  # (handled at the top of the next turn)
            elif g_old is not None and i == last:
                gwo, acc = sink.slot(W.old_head[0])
                gbo, _ = sink.slot(W.old_head[1])
                ops.conv1x1_wgrad(xin, g_old, gwo, gbo, tanh_y=old, accumulate=acc)
                if ops.fuse_ends():  # the old head's branch joins in the conv's epilogue
                    ops.conv1x1(g_old, W.old_head[0], None, xin.shape[1], transposed=True, tanh_bwd_in=old, out=g, accumulate=True)
                else:
                    extra = ops.conv1x1(g_old, W.old_head[0], None, xin.shape[1], transposed=True, tanh_bwd_in=old)
                    g = ops.axpby(1.0, g, 1.0, extra, out=g)
        elif need_gz:
            gz = cache.conv(gpre1, w1, True, None, ci)
    return gz


# =====================================================================================================================
# Discriminator
# =====================================================================================================================
class DiscWeights:
    def __init__(self, stem, blocks, old_stem, clf):
        self.stem = stem          # (w, b): 2 -> C0
        self.blocks = blocks      # list of (w1, b1, w2, b2) for conv_blocks[curr..8]
        self.old_stem = old_stem  # (w, b) or None: 2 -> C1 of the first live block
        self.clf = clf            # (w[1,160], b[1])

    def tensors(self) -> List[torch.Tensor]:
        out = list(self.stem)
        for blk in self.blocks:
            out += list(blk)
        if self.old_stem is not None:
            out += list(self.old_stem)
        out += list(self.clf)
        return out


def _tile_mask_ok(n: int, cin: int, cout: int, h: int, w: int) -> bool:
    """Tile masks are an epilogue of the Winograd kernel: usable where it runs for the whole batch AND for the third of it that
    the penalty's tangent pass sends through the same layer again (MG_TILEMASK=0 keeps fp32 activations everywhere)."""
    if os.environ.get("MG_TILEMASK", "1") == "0" or (w % 4) or (h % 2):
        return False
    if h <= 8 and w <= 8 and _smallnet_on():
        return False  # the fused tail's un-pooling step reads the fp32 activation of the 8x8 layer
    return ops.wino3x3_supported(max(1, n // 3), cout, h, w, cin=cin)


def _fade_fused_ok(n: int, c1: int, c_next: int, h: int, w: int) -> bool:
    """Whether the fade-in blend rides on the Winograd convs around it (ops.conv3x3_fade): the block's second conv forward and
    in the tangent pass (a third of the batch), and the data-gradient conv of the next block's first conv on the way back."""
    if os.environ.get("MG_TILEMASK", "1") == "0" or c_next == 0 or (w % 4) or (h % 2):
        return False
    nmin = max(1, n // 3)
    return ops.wino3x3_supported(nmin, c1, h, w, cin=c1) and ops.wino3x3_supported(nmin, c1, h, w, cin=c_next)


def _disc_tail_forward(W: DiscWeights, t: int, q1: torch.Tensor, cache: PackCache, save: bool, masks=None):
    """conv2 of block t (4x4), blocks t+1 (4x4 -> 2x2) and t+2 (2x2 -> 1x1) and the classifier in one launch.  Returns (scores,
    [(a2_t,), (inp, a1, q1, a2) of block t+1, ... of block t+2]) -- the tuples only with `save`.
    `masks` (the tangent pass of the penalty): the saved tuples of a forward pass; every conv is then bias-free, multiplied by
    the LeakyReLU derivative of the saved activation, and written over it (q1 must be the tangent of block t's pooled output, in
    place in the saved tensor); nothing is returned."""
    n = q1.shape[0]
    dev = q1.device
    tangent = masks is not None
    new = lambda c, s: torch.empty((n, c, s, s), dtype=torch.float32, device=dev)
    sn = ops.SmallNet(_imgs_per_wg(n))
    w1, b1, w2, b2 = W.blocks[t]
    c1 = w1.shape[0]
    sn.load(0, q1)
    cur = 0
    rest = []
    a2 = masks[0][3] if tangent else (new(c1, 4) if save else None)
    nxt = _rot(cur)[0]
    sn.conv(cur, nxt, cache.get_sn(w2, False), c1, c1, 4, 4, bias=None if tangent else b2, lrelu=not tangent,
            mask=a2 if tangent else None, out=a2)
    rest.append((a2,))
    cur, cin, h, inp = nxt, c1, 4, a2
    for k, j in enumerate((t + 1, t + 2)):
        w1, b1, w2, b2 = W.blocks[j]
        c1 = w1.shape[0]
        a1 = masks[k + 1][1] if tangent else (new(c1, h) if save else None)
        qn = masks[k + 1][2] if tangent else (new(c1, h // 2) if save else None)
        a2 = masks[k + 1][3] if tangent else (new(c1, h // 2) if save else None)
        b_a, b_q = _rot(cur)
        sn.conv(cur, b_a, cache.get_sn(w1, False), cin, c1, h, h, bias=None if tangent else b1, lrelu=not tangent,
                mask=a1 if tangent else None, out=a1)
        sn.pool(b_a, b_q, c1, h, h, out=qn)
        sn.conv(b_q, cur, cache.get_sn(w2, False), c1, c1, h // 2, h // 2, bias=None if tangent else b2, lrelu=not tangent,
                mask=a2 if tangent else None, out=a2)
        rest.append((inp, a1, qn, a2))
        cin, h, inp = c1, h // 2, a2
    if tangent:
        sn.run(n)
        return None
    out = torch.empty((n, 1), dtype=torch.float32, device=dev)
    sn.linear(cur, cin, W.clf[0], W.clf[1], out)
    sn.run(n)
    return out, rest


def _disc_tail_backward(W: DiscWeights, t: int, saved, g_out: torch.Tensor, cache: PackCache):
    """The data-gradient chain of the same layers in one launch: classifier backward, LeakyReLU masks, transposed convs, AvgPool2d
    backward, down to the masked gradient in front of block t's first conv (8x8, written straight to global memory).  Returns
    {block index: (gpre1, gpre2)} for blocks t .. t+2 (what the weight gradients and the penalty's second-order pass consume)."""
    n = g_out.shape[0]
    dev = g_out.device
    new = lambda c, s: torch.empty((n, c, s, s), dtype=torch.float32, device=dev)
    sn = ops.SmallNet(_imgs_per_wg(n))
    hs = {}
    last = t + 2
    c_last = W.blocks[last][0].shape[0]
    gpre2 = new(c_last, 1)
    sn.linbwd(0, c_last, g_out, W.clf[0])
    sn.mask(0, c_last, 1, 1, saved[last][3], out=gpre2)
    cur, h = 0, 1
    for j in (last, last - 1):
        w1, b1, w2, b2 = W.blocks[j]
        c1, cin = w1.shape[0], w1.shape[1]
        inp, a1, q1, a2 = saved[j]
        gpre1 = new(c1, 2 * h)
        gprev = new(cin, 2 * h)   # masked gradient behind the previous block's second conv
        b_a, b_b = _rot(cur)
        sn.conv(cur, b_a, cache.get_sn(w2, True), c1, c1, h, h)
        sn.poolbwd(b_a, b_b, c1, h, h, a1, out=gpre1)
        sn.conv(b_b, cur, cache.get_sn(w1, True), c1, cin, 2 * h, 2 * h, mask=saved[j - 1][3], out=gprev)
        hs[j] = (gpre1, gpre2)
        gpre2, h = gprev, 2 * h
    w1, b1, w2, b2 = W.blocks[t]
    c1 = w1.shape[0]
    inp, a1, q1, a2 = saved[t]
    gpre1 = new(c1, 8)
    b_a, b_b = _rot(cur)
    sn.conv(cur, b_a, cache.get_sn(w2, True), c1, c1, 4, 4)
    sn.poolbwd(b_a, b_b, c1, 4, 4, a1, out=gpre1, lds=False)
    hs[t] = (gpre1, gpre2)
    sn.run(n)
    return hs


def disc_forward(W: DiscWeights, x: torch.Tensor, alpha: float, cache: PackCache, save: bool):
    x = x.contiguous()
    n = x.shape[0]
    c0 = W.stem[0].shape[0]
    xp = o = None
    fused_stem = W.old_stem is not None and ops.stem_pair_supported(x.shape[2], x.shape[3])
    h0m = None
    if fused_stem:  # the new block's stem, the pooled input and the old block's stem on it: one pass over x
        # of h0 the data-gradient conv in front of the stem only needs the sign: where that conv is the Winograd kernel the stem also
        # leaves one byte per 2x2 tile and channel, 1/16 of the bytes the fp32 mask costs that launch (the largest of a D / G backward)
        c1_0 = W.blocks[0][0].shape[0]
        if save and os.environ.get("MG_TILEMASK", "1") != "0" and os.environ.get("MG_STEM_TILEMASK", "1") != "0" and \
                ops.wino3x3_supported(n, c0, x.shape[2], x.shape[3], cin=c1_0) and \
                ops.wino3x3_mask_bytes_y_supported(n, c1_0, c0, x.shape[2], x.shape[3]):
            h0, xp, o, h0m = ops.stem_pair(x, W.stem[0], W.stem[1], W.old_stem[0], W.old_stem[1], want_xp=save, want_mask=True)
        else:
            h0, xp, o = ops.stem_pair(x, W.stem[0], W.stem[1], W.old_stem[0], W.old_stem[1], want_xp=save)
    else:
        h0 = ops.conv1x1(x, W.stem[0], W.stem[1], c0, lrelu=True)
    saved = []
    inp = h0
    tail = disc_tail_start(W, x.shape[2], x.shape[3])
    for i, (w1, b1, w2, b2) in enumerate(W.blocks):
        c1 = w1.shape[0]
        if tail is not None and i == tail:
            # this block's first conv (8x8) on its own, then everything behind it -- its second conv, the two last blocks and the
            # classifier -- as ONE launch with the activations in LDS; the tensors the backward passes read are stored on the way
            a1, q1 = cache.conv(inp, w1, False, b1, c1, lrelu=True, pool=True)
            out, rest = _disc_tail_forward(W, tail, q1, cache, save)
            if save:
                saved.append((inp, a1, q1, rest[0][0]))
                saved.extend(rest[1:])
            flat = rest[-1][3].reshape(n, -1) if save else None
            return out, ((x, h0, saved, xp, o, flat, alpha, h0m) if save else None)
        # AvgPool2d fused in the epilogue.  Of the full-resolution activation the backward passes only need the sign, so on the
        # large maps (Winograd kernel, for every batch slice that will come back with the mask) it is kept as one byte per
        # 2x2 tile and a1 is that uint8 tile mask (N,c1,H/2,W/2) instead of the fp32 tensor.
        a1, q1 = cache.conv(inp, w1, False, b1, c1, lrelu=True, pool=True,
                            mask_out=save and _tile_mask_ok(n, w1.shape[1], c1, inp.shape[2], inp.shape[3]))
        if i == 0 and W.old_stem is not None:  # fade-in: the block's output is blended with the old stem's
            if not fused_stem:
                xp = ops.avgpool2_fwd(x)
                o = ops.conv1x1(xp, W.old_stem[0], W.old_stem[1], c1, lrelu=True)
            F = FadeIn.of(alpha)
            c_next = W.blocks[1][0].shape[0] if len(W.blocks) > 1 else 0
            if _fade_fused_ok(n, c1, c_next, q1.shape[2], q1.shape[3]):
                # blend in the conv's epilogue; of the new branch itself only the sign is needed later: a2 = uint8 tile mask
                nxt, a2 = ops.conv3x3_fade(q1, cache.get_wino(w2, False), b2, c1, _lib.MG_FADE_FWD, o, F.coef(x.device))
            else:
                a2 = cache.conv(q1, w2, False, b2, c1, lrelu=True)
                nxt = ops.axpby(F.a, a2, F.b, o, coef=F.dev)
        else:
            a2 = nxt = cache.conv(q1, w2, False, b2, c1, lrelu=True)
        if save:
            saved.append((inp, a1, q1, a2))
        inp = nxt
    assert inp.shape[2] == 1 and inp.shape[3] == 1, \
        f"discriminator input must be square with side 2**(9-curr_layer); final map is {tuple(inp.shape)}"
    flat = inp.reshape(n, -1)
    out = ops.linear1_fwd(flat, W.clf[0], W.clf[1])
    ctx = (x, h0, saved, xp, o, flat, alpha, h0m) if save else None
    return out, ctx


def disc_backward(W: DiscWeights, ctx, g_out: torch.Tensor, cache: PackCache, sink: Optional[GradSink],
                  need_gx: bool, keep_h: bool = False, gx_from: int = 0):
    """Back-propagate g_out (N,1).  sink=None skips all parameter gradients (first-order pass of the penalty).
    keep_h returns the masked per-layer gradients h_l needed by disc_gp_param_grads()."""
    x, h0, saved, xp, o, flat, alpha, h0m = ctx
    F = FadeIn.of(alpha)
    g_out = g_out.contiguous()
    n = x.shape[0]
    nb = len(W.blocks)
    hs = {"blocks": [None] * nb, "stem": None, "old": None}
    tail = disc_tail_start(W, x.shape[2], x.shape[3])
    gpre_o = gpre_s = None
    if tail is not None:
        # classifier + the <= 4x4 layers in one launch, down to the masked gradient in front of block `tail`'s first conv; their
        # weight gradients (if wanted) from the tensors it stored
        ths = _disc_tail_backward(W, tail, saved, g_out, cache)
        if sink is not None:
            gwc, acc = sink.slot(W.clf[0])
            gbc, _ = sink.slot(W.clf[1])
            ops.linear1_bwd(flat, W.clf[0], g_out, gw=gwc, gb=gbc, need_gx=False, accumulate=acc)
            for j in range(nb - 1, tail - 1, -1):
                w1, b1, w2, b2 = W.blocks[j]
                inp, a1, q1, a2 = saved[j]
                gw2, acc = sink.slot(w2)
                gb2, _ = sink.slot(b2)
                ops.conv3x3_wgrad(q1, ths[j][1], gw2, gb2, accumulate=acc)
                gw1, acc = sink.slot(w1)
                gb1, _ = sink.slot(b1)
                ops.conv3x3_wgrad(inp, ths[j][0], gw1, gb1, accumulate=acc)
        for j in range(tail, nb):
            hs["blocks"][j] = ths[j]
        first, gpre2 = tail, None
    else:
        if sink is not None:
            gwc, acc = sink.slot(W.clf[0])
            gbc, _ = sink.slot(W.clf[1])
            g = ops.linear1_bwd(flat, W.clf[0], g_out, gw=gwc, gb=gbc, accumulate=acc)
        else:
            g = ops.linear1_bwd(flat, W.clf[0], g_out, need_gx=True)
        a2_last = saved[nb - 1][3]
        g = g.reshape(a2_last.shape)
        if nb == 1 and W.old_stem is not None:  # the blend is the classifier input
            gpre2, gpre_o = ops.blend_lrelu_bwd(g, a2_last, o, F.a, F.b, coef=F.dev)
        else:
            gpre2 = ops.lrelu_bwd(g, a2_last)
        first = nb - 1
    for i in range(first, -1, -1):
        w1, b1, w2, b2 = W.blocks[i]
        inp, a1, q1, a2 = saved[i]
        c1, cin = w1.shape[0], w1.shape[1]
        if tail is not None and i == tail:
            gpre1 = hs["blocks"][i][0]
        else:
            if sink is not None:
                gw2, acc = sink.slot(w2)
                gb2, _ = sink.slot(b2)
                ops.conv3x3_wgrad(q1, gpre2, gw2, gb2, accumulate=acc)
            if a1.dtype == torch.uint8 and ops.wino3x3_supported(n, c1, gpre2.shape[2], gpre2.shape[3], cin=c1):
                gpre1 = cache.conv(gpre2, w2, True, None, c1, unpool_mask=a1)  # AvgPool2d + LeakyReLU backward in the conv epilogue
            else:
                gpre1 = cache.conv_unpool(gpre2, w2, c1, a1)
            if sink is not None:
                gw1, acc = sink.slot(w1)
                gb1, _ = sink.slot(b1)
                ops.conv3x3_wgrad(inp, gpre1, gw1, gb1, accumulate=acc)
        if keep_h and hs["blocks"][i] is None:
            hs["blocks"][i] = (gpre1, gpre2)
        if i > 0:
            a2_prev = saved[i - 1][3]
            if i == 1 and W.old_stem is not None:  # inp is the fade-in blend of a2_prev and the old stem path
                if a2_prev.dtype == torch.uint8:  # blend + LeakyReLU backward in the data-gradient conv's epilogue
                    gpre2, gpre_o = ops.conv3x3_fade(gpre1, cache.get_wino(w1, True), None, cin, _lib.MG_FADE_BWD, o,
                                                     F.coef(x.device), mask_in=a2_prev)
                else:
                    gblend = cache.conv(gpre1, w1, True, None, cin)
                    gpre2, gpre_o = ops.blend_lrelu_bwd(gblend, a2_prev, o, F.a, F.b, coef=F.dev)
            else:
                gpre2 = cache.conv(gpre1, w1, True, None, cin, mask_aux=a2_prev)
        else:
            gpre_s = cache.conv(gpre1, w1, True, None, cin, mask_aux=h0m if h0m is not None else h0)
    if sink is not None:
        gws, acc = sink.slot(W.stem[0])
        gbs, _ = sink.slot(W.stem[1])
        ops.conv1x1_wgrad(x, gpre_s, gws, gbs, accumulate=acc)
        if W.old_stem is not None:
            gwo, acc = sink.slot(W.old_stem[0])
            gbo, _ = sink.slot(W.old_stem[1])
            ops.conv1x1_wgrad(xp, gpre_o, gwo, gbo, accumulate=acc)
    if keep_h:
        hs["stem"], hs["old"] = gpre_s, gpre_o
    gx = None
    if need_gx and W.old_stem is not None and ops.stem_pair_gx_supported(gpre_s.shape[1], gpre_o.shape[1], x.shape[2], x.shape[3]):
        gx = ops.stem_pair_gx(gpre_s[gx_from:], W.stem[0], gpre_o[gx_from:], W.old_stem[0])  # both branches and their sum, one launch
    elif need_gx:  # gx_from > 0: only samples gx_from.. are wanted (the fused critic step needs the interpolated third only)
        gx = ops.conv1x1(gpre_s[gx_from:], W.stem[0], None, 2, transposed=True)
        if W.old_stem is not None:
            gxp = ops.conv1x1(gpre_o[gx_from:], W.old_stem[0], None, 2, transposed=True)
            gx = ops.axpby(1.0, gx, 1.0, ops.avgpool2_bwd(gxp), out=gx)
    return gx, (hs if keep_h else None)


def disc_gp_param_grads(W: DiscWeights, ctx, hs, u0: torch.Tensor, cache: PackCache, sink: GradSink,
                        g_out: Optional[torch.Tensor] = None, want_t: bool = False):
    """Second-order pass of the gradient penalty: tangent-forward u_l = mask_l * L_l u_{l-1} and dP/dW_l = wgrad(u_{l-1}, h_l).
    Biases receive no gradient from the penalty."""
    x, h0, saved, xp, o, flat, alpha, _ = ctx
    n = x.shape[0]
    c0 = W.stem[0].shape[0]
    gws, acc = sink.slot(W.stem[0])
    ops.conv1x1_wgrad(u0, hs["stem"], gws, None, accumulate=acc)
    t = ops.conv1x1(u0, W.stem[0], None, c0, mask_aux=h0)
    to = None
    if W.old_stem is not None:
        u0p = ops.avgpool2_fwd(u0)
        gwo, acc = sink.slot(W.old_stem[0])
        ops.conv1x1_wgrad(u0p, hs["old"], gwo, None, accumulate=acc)
        to = ops.conv1x1(u0p, W.old_stem[0], None, W.old_stem[0].shape[0], mask_aux=o)
    for i, (w1, b1, w2, b2) in enumerate(W.blocks):
        inp, a1, q1, a2 = saved[i]
        gpre1, gpre2 = hs["blocks"][i]
        c1 = w1.shape[0]
        gw1, acc = sink.slot(w1)
        ops.conv3x3_wgrad(t, gpre1, gw1, None, accumulate=acc)
        t1, tq = cache.conv(t, w1, False, None, c1, mask_aux=a1, pool=True)
        gw2, acc = sink.slot(w2)
        ops.conv3x3_wgrad(tq, gpre2, gw2, None, accumulate=acc)
        if a2.dtype == torch.uint8:
            F = FadeIn.of(alpha)
            t = ops.conv3x3_fade(tq, cache.get_wino(w2, False), None, c1, _lib.MG_FADE_TANGENT, to, F.coef(x.device), mask_in=a2)
        else:
            t = cache.conv(tq, w2, False, None, c1, mask_aux=a2)
            if i == 0 and to is not None:
                F = FadeIn.of(alpha)
                t = ops.axpby(F.a, t, F.b, to, out=t, coef=F.dev)
    # `g_out`: the upstream score gradient the first-order chain (hs) was run with -- ones for the penalty (discriminator.py:170-176)
    up = torch.ones((n, 1), dtype=torch.float32, device=x.device) if g_out is None else g_out.contiguous()
    gwc, acc = sink.slot(W.clf[0])
    ops.linear1_bwd(t.reshape(n, -1), W.clf[0], up, gw=gwc, gb=None, need_gx=False, accumulate=acc)
    if want_t:  # the tangent of the scores themselves, J(x) u0: the derivative w.r.t. the upstream score gradient
        zero = torch.zeros(1, dtype=torch.float32, device=x.device)
        return sink, ops.linear1_fwd(t.reshape(n, -1).contiguous(), W.clf[0], zero)
    # parameters that the penalty does not reach (biases) still need a defined gradient when this sink is returned alone
    return sink


# =====================================================================================================================
# Fused discriminator step (one batched pass instead of three)
# =====================================================================================================================
def disc_step_fused(W: DiscWeights, x_real: torch.Tensor, x_fake: torch.Tensor, eps: torch.Tensor, alpha: float,
                    cache: PackCache, sink: GradSink, gp_factor: float = 10.0, xcat: Optional[torch.Tensor] = None,
                    defer: Optional["ops.WgradDefer"] = None):
    """Gradient of  -(mean D(x_real) - mean D(x_fake)) + gp_factor * mean((||grad D(x~)|| - 1)^2)  w.r.t. every live critic
    parameter, written into `sink`; returns (disc_loss, grad_pen, out, stats) with out = D([x_real; x_fake; x~]) and stats =
    [mean D(real), mean D(fake), mean D(x~), disc_loss].

    Same arithmetic as three `disc_forward` + two `disc_backward` + `disc_gp_param_grads`, organised as ONE forward and ONE
    data-gradient chain over the concatenated batch [real | fake | interpolated] (per-sample upstream -1/N, +1/N, 1), the
    penalty's tangent pass written IN PLACE over the interpolated slice of every saved activation (the conv epilogue reads
    the LeakyReLU mask and overwrites it with the tangent in the same lane), and then ONE weight-gradient launch per layer
    over all 3N samples: x = [a_real | a_fake | u],  gy = [delta_real | delta_fake | h]  sums the Wasserstein and the penalty
    gradients at once (bias gradients only from the first 2N samples: the penalty has none).
    If `xcat` (3N,2,H,W) is given, x_real / x_fake must already sit in its first two thirds."""
    n = x_real.shape[0]
    if xcat is None:
        xcat = torch.empty((3 * n,) + tuple(x_real.shape[1:]), dtype=torch.float32, device=x_real.device)
        xcat[:n].copy_(x_real)
        xcat[n:2 * n].copy_(x_fake)
    ops.gp_interp(xcat[:n], xcat[n:2 * n], eps.contiguous(), out=xcat[2 * n:])
    out, ctx = disc_forward(W, xcat, alpha, cache, save=True)
    g_out = upstream_scores_grad("critic", n, xcat.device)
    gx, hs = disc_backward(W, ctx, g_out, cache, None, need_gx=True, keep_h=True, gx_from=2 * n)  # input gradient: x~ only
    # penalty value and u_0 = dP/dg_0, written over the interpolated inputs (they are not needed any more)
    ss = ops.sumsq_per_sample(gx)
    x, h0, saved, xp, o, flat, _, _ = ctx
    sl = slice(2 * n, 3 * n)
    if ops.fuse_ends():
        grad_pen, _ = ops.gp_apply(gx, ss, gp_factor, 1.0, out=x[sl])
    else:
        grad_pen, coef = ops.gp_finish(ss, gp_factor, 1.0)
        ops.scale_per_sample(gx, coef, out=x[sl])
    # ---- tangent pass, in place over the interpolated slices
    c0 = W.stem[0].shape[0]
    if W.old_stem is not None and ops.stem_pair_supported(x.shape[2], x.shape[3]):
        ops.stem_pair(x[sl], W.stem[0], None, W.old_stem[0], None, h0=h0[sl], xp=xp[sl], o=o[sl], masked=True)
    else:
        ops.conv1x1(x[sl], W.stem[0], None, c0, mask_aux=h0[sl], out=h0[sl])
        if W.old_stem is not None:
            ops.avgpool2_fwd(x[sl], out=xp[sl])
            ops.conv1x1(xp[sl], W.old_stem[0], None, W.old_stem[0].shape[0], mask_aux=o[sl], out=o[sl])
    nb = len(W.blocks)
    tail = disc_tail_start(W, x.shape[2], x.shape[3])
    for i, (w1, b1, w2, b2) in enumerate(W.blocks):
        inp, a1, q1, a2 = saved[i]
        c1 = w1.shape[0]
        if a1.dtype == torch.uint8:
            cache.conv(inp[sl], w1, False, None, c1, mask_aux=a1[sl], pool_out=q1[sl])
        else:
            cache.conv(inp[sl], w1, False, None, c1, mask_aux=a1[sl], out=a1[sl], pool_out=q1[sl])
        if tail is not None and i == tail:  # the rest of the tangent pass (<= 4x4 maps) in one launch, in place as well
            _disc_tail_forward(W, tail, q1[sl], cache, False, masks=[tuple(u[sl] for u in saved[j]) for j in range(tail, nb)])
            break
        if a2.dtype == torch.uint8:  # fade-in block with the blend fused: straight into the next block's input slice
            F = FadeIn.of(alpha)
            ops.conv3x3_fade(q1[sl], cache.get_wino(w2, False), None, c1, _lib.MG_FADE_TANGENT, o[sl], F.coef(x.device),
                             mask_in=a2[sl], out=saved[1][0][sl])
            continue
        cache.conv(q1[sl], w2, False, None, c1, mask_aux=a2[sl], out=a2[sl])
        if i == 0 and W.old_stem is not None:
            target = saved[1][0][sl] if nb > 1 else flat[sl].reshape(a2[sl].shape)
            F = FadeIn.of(alpha)
            ops.axpby(F.a, a2[sl], F.b, o[sl], out=target, coef=F.dev)
    # ---- one weight-gradient sweep over the 3N samples
    for i, (w1, b1, w2, b2) in enumerate(W.blocks):
        inp, a1, q1, a2 = saved[i]
        gpre1, gpre2 = hs["blocks"][i]
        gw1, acc = sink.slot(w1)
        gb1, _ = sink.slot(b1)
        ops.conv3x3_wgrad(inp, gpre1, gw1, gb1, accumulate=acc, bias_n=2 * n, defer=defer)
        gw2, acc = sink.slot(w2)
        gb2, _ = sink.slot(b2)
        ops.conv3x3_wgrad(q1, gpre2, gw2, gb2, accumulate=acc, bias_n=2 * n, defer=defer)
    # x = [real | fake | u_0], gy = [delta_real | delta_fake | h_0]: one launch per stem, bias gradient from the first 2N samples
    gws, acc = sink.slot(W.stem[0])
    gbs, _ = sink.slot(W.stem[1])
    ops.conv1x1_wgrad(x, hs["stem"], gws, gbs, accumulate=acc, bias_n=2 * n)
    if W.old_stem is not None:
        gwo, acc = sink.slot(W.old_stem[0])
        gbo, _ = sink.slot(W.old_stem[1])
        ops.conv1x1_wgrad(xp, hs["old"], gwo, gbo, accumulate=acc, bias_n=2 * n)
    gwc, acc = sink.slot(W.clf[0])
    gbc, _ = sink.slot(W.clf[1])
    ops.linear1_bwd(flat, W.clf[0], g_out, gw=gwc, gb=gbc, need_gx=False, accumulate=acc, bias_n=2 * n)
    if defer is not None:
        defer.flush()  # the slab reductions of every Winograd weight gradient of the sweep, one launch
    stats = ops.group_means(out, 3)  # [mean D(real), mean D(fake), mean D(x~), -(mean D(real) - mean D(fake))]
    return stats[3], grad_pen, out, stats


# =====================================================================================================================
# Generator step without autograd
# =====================================================================================================================
def gen_step_fused(Wg: GenWeights, Wd: DiscWeights, z: torch.Tensor, alpha: float, cache_g: PackCache, cache_d: PackCache,
                   sink: GradSink, before_disc=None, defer: Optional["ops.WgradDefer"] = None):
    """Gradient of  -mean D(G(z))  (criterion.py:17-18, train.py:191-213) w.r.t. every live generator parameter, written into
    `sink`; returns (gen_loss, out_fake, stats = [mean D(G(z)), gen_loss]).  The critic's weight gradients are not evaluated (the reference computes and discards
    them, train.py:209-214): its backward pass only carries the data gradient down to the generated images.
    `before_disc` (optional callable) runs between the generator's forward pass and the critic's."""
    n = z.shape[0]
    x_fake, gctx = gen_forward(Wg, z.contiguous(), alpha, cache_g, save=True)
    if before_disc is not None:  # data-parallel: the critic's weights may still be in flight on the side stream until here
        before_disc()
    out, dctx = disc_forward(Wd, x_fake, alpha, cache_d, save=True)
    g_out = upstream_scores_grad("gen", n, z.device)
    gx, _ = disc_backward(Wd, dctx, g_out, cache_d, None, need_gx=True)
    gen_backward(Wg, gctx, gx, cache_g, sink, defer=defer)
    if defer is not None:
        defer.flush()
    stats = ops.group_means(out, 1)  # [mean D(G(z)), -mean D(G(z))]
    return stats[1], out, stats
