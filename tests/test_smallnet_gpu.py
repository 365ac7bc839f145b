"""mg_smallnet (multi-layer chains on small maps, csrc/smallnet.hip) against float64 torch chains of the reference's layers
(generator.py:15-40, discriminator.py:14-34,94-101, layers.py:11-17) -- every op, ragged batches, 1 / 2 / 4 images per workgroup."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
SLOPE = 0.2


def _rel(a, b):
    b = b.to(torch.float64)
    b = b.detach()
    return float((a.detach().double().cpu() - b.cpu()).abs().max() / b.abs().max().clamp_min(1e-30))


def _w(co, ci, gen, scale=None):
    w = torch.randn(co, ci, 3, 3, generator=gen)
    return w * (scale if scale is not None else (1.0 / (ci * 9) ** 0.5))


@pytest.mark.parametrize("n,g", [(1, 1), (5, 1), (5, 2), (7, 4), (8, 4)])
def test_conv_chain_with_pool_and_linear(n, g):
    """Critic-tail shaped chain: conv+LeakyReLU at 4x4 -> conv+LeakyReLU -> AvgPool -> 2x2 convs -> AvgPool -> 1x1-map conv ->
    Linear; channel counts that are not multiples of 16 on the way; every stored tensor compared."""
    from musicgan_amd import ops
    gen = torch.Generator().manual_seed(100 + n + g)
    chans = [24, 40, 36, 52, 48, 20]
    x = torch.randn(n, chans[0], 4, 4, generator=gen)
    ws = [_w(chans[i + 1], chans[i], gen) for i in range(5)]
    bs = [torch.randn(chans[i + 1], generator=gen) * 0.1 for i in range(5)]
    wl, bl = torch.randn(1, chans[5], generator=gen) * 0.2, torch.randn(1, generator=gen)
    # float64 reference
    xd = x.double()
    a0 = F.leaky_relu(F.conv2d(xd, ws[0].double(), bs[0].double(), padding=1), SLOPE)
    a1 = F.leaky_relu(F.conv2d(a0, ws[1].double(), bs[1].double(), padding=1), SLOPE)
    q1 = F.avg_pool2d(a1, 2)
    a2 = F.leaky_relu(F.conv2d(q1, ws[2].double(), bs[2].double(), padding=1), SLOPE)
    a3 = F.leaky_relu(F.conv2d(a2, ws[3].double(), bs[3].double(), padding=1), SLOPE)
    q3 = F.avg_pool2d(a3, 2)
    a4 = F.leaky_relu(F.conv2d(q3, ws[4].double(), bs[4].double(), padding=1), SLOPE)
    y = a4.reshape(n, -1) @ wl.double().t() + bl.double()
    # device
    d = lambda t: t.to(DEV).contiguous()
    wp = [ops.pack_smallnet(d(w), False) for w in ws]
    bd = [d(b) for b in bs]
    E = lambda c, h: torch.full((n, c, h, h), float("nan"), device=DEV)
    o0, o1, oq1, o2, o3, oq3, o4 = E(chans[1], 4), E(chans[2], 4), E(chans[2], 2), E(chans[3], 2), E(chans[4], 2), E(chans[4], 1), \
        E(chans[5], 1)
    oy = torch.full((n, 1), float("nan"), device=DEV)
    sn = ops.SmallNet(g)
    sn.load(0, d(x))
    sn.conv(0, 1, wp[0], chans[0], chans[1], 4, 4, bias=bd[0], lrelu=True, out=o0)
    sn.conv(1, 2, wp[1], chans[1], chans[2], 4, 4, bias=bd[1], lrelu=True, out=o1)
    sn.pool(2, 0, chans[2], 4, 4, out=oq1)
    sn.conv(0, 1, wp[2], chans[2], chans[3], 2, 2, bias=bd[2], lrelu=True, out=o2)
    sn.conv(1, 2, wp[3], chans[3], chans[4], 2, 2, bias=bd[3], lrelu=True, out=o3)
    sn.pool(2, 0, chans[4], 2, 2, out=oq3)
    sn.conv(0, 1, wp[4], chans[4], chans[5], 1, 1, bias=bd[4], lrelu=True, out=o4)
    sn.linear(1, chans[5], d(wl), d(bl), oy)
    sn.run(n)
    torch.cuda.synchronize()
    for name, got, want in (("a0", o0, a0), ("a1", o1, a1), ("q1", oq1, q1), ("a2", o2, a2), ("a3", o3, a3), ("q3", oq3, q3),
                            ("a4", o4, a4), ("y", oy, y)):
        assert _rel(got, want) < 2e-6, (name, _rel(got, want))


@pytest.mark.parametrize("n,g,c", [(3, 1, 32), (6, 2, 8), (5, 1, 20)])
def test_generator_head_forward_and_backward(n, g, c):
    """Generator-head shaped chain (conv -> LeakyReLU -> PixelNorm -> Upsample -> conv -> LeakyReLU -> PixelNorm -> conv -> ...)
    forward, then its data-gradient chain (PixelNorm + LeakyReLU backward, transposed convs, 2x2 block sums) against autograd
    in float64."""
    from musicgan_amd import ops
    gen = torch.Generator().manual_seed(7 + n + c)
    c1, c2 = 48, 40
    z = torch.randn(n, c, 2, 2, generator=gen)
    w0, w1, w2 = _w(c, c, gen), _w(c1, c, gen), _w(c2, c1, gen)
    b0, b1, b2 = (torch.randn(k, generator=gen) * 0.1 for k in (c, c1, c2))
    gout = torch.randn(n, c2, 4, 4, generator=gen)

    def pn(y):
        return y / torch.sqrt((y * y).mean(dim=1, keepdim=True) + 1e-8)

    zd = z.double().requires_grad_(True)
    y0 = F.leaky_relu(F.conv2d(zd, w0.double(), b0.double(), padding=1), SLOPE)
    p0 = pn(y0)
    y1 = F.leaky_relu(F.conv2d(F.interpolate(p0, scale_factor=2, mode="nearest"), w1.double(), b1.double(), padding=1), SLOPE)
    p1 = pn(y1)
    y2 = F.leaky_relu(F.conv2d(p1, w2.double(), b2.double(), padding=1), SLOPE)
    p2 = pn(y2)
    pre0 = F.conv2d(zd, w0.double(), b0.double(), padding=1)
    # gradients w.r.t. the pre-activations (what the weight-gradient kernels consume) and z
    pre = []

    def fwd_with_pre():
        a = F.conv2d(zd, w0.double(), b0.double(), padding=1)
        a.retain_grad()
        pa = pn(F.leaky_relu(a, SLOPE))
        bb = F.conv2d(F.interpolate(pa, scale_factor=2, mode="nearest"), w1.double(), b1.double(), padding=1)
        bb.retain_grad()
        pb = pn(F.leaky_relu(bb, SLOPE))
        cc = F.conv2d(pb, w2.double(), b2.double(), padding=1)
        cc.retain_grad()
        pc = pn(F.leaky_relu(cc, SLOPE))
        pre.extend([a, bb, cc])
        return pc
    out = fwd_with_pre()
    out.backward(gout.double())
    d = lambda t: t.to(DEV).contiguous()
    wp = [ops.pack_smallnet(d(w), False) for w in (w0, w1, w2)]
    wpd = [ops.pack_smallnet(d(w), True) for w in (w0, w1, w2)]
    E = lambda cc, h: torch.full((n, cc, h, h), float("nan"), device=DEV)
    P0, R0, P1, R1, P2, R2 = E(c, 2), E(1, 2), E(c1, 4), E(1, 4), E(c2, 4), E(1, 4)
    sn = ops.SmallNet(g)
    sn.load(0, d(z))
    sn.conv(0, 1, wp[0], c, c, 2, 2, bias=d(b0), lrelu=True).pixnorm(1, c, 2, 2, P0, R0)
    sn.up(1, 2, c, 2, 2)
    sn.conv(2, 0, wp[1], c, c1, 4, 4, bias=d(b1), lrelu=True).pixnorm(0, c1, 4, 4, P1, R1)
    sn.conv(0, 1, wp[2], c1, c2, 4, 4, bias=d(b2), lrelu=True).pixnorm(1, c2, 4, 4, P2, R2)
    sn.run(n)
    torch.cuda.synchronize()
    for name, got, want in (("p0", P0, p0), ("p1", P1, p1), ("p2", P2, p2)):
        assert _rel(got, want) < 3e-6, (name, _rel(got, want))
    assert _rel(R2, 1.0 / torch.sqrt((y2 * y2).mean(dim=1, keepdim=True) + 1e-8)) < 3e-6
    # backward chain
    G2, G1, G0, GZ = E(c2, 4), E(c1, 4), E(c, 2), E(c, 2)
    sb = ops.SmallNet(g)
    sb.load(0, d(gout))
    sb.pnbwd(0, c2, 4, 4, P2, R2, out=G2)
    sb.conv(0, 1, wpd[2], c2, c1, 4, 4)
    sb.pnbwd(1, c1, 4, 4, P1, R1, out=G1)
    sb.conv(1, 2, wpd[1], c1, c, 4, 4)
    sb.upbwd(2, 0, c, 4, 4)
    sb.pnbwd(0, c, 2, 2, P0, R0, out=G0)
    sb.conv(0, 1, wpd[0], c, c, 2, 2, out=GZ)
    sb.run(n)
    torch.cuda.synchronize()
    for name, got, want in (("gpre2", G2, pre[2].grad), ("gpre1", G1, pre[1].grad), ("gpre0", G0, pre[0].grad), ("gz", GZ, zd.grad)):
        assert _rel(got, want) < 2e-5, (name, _rel(got, want))


@pytest.mark.parametrize("n,g", [(4, 1), (6, 4), (3, 2)])
def test_critic_tail_data_gradient_and_tangent(n, g):
    """Critic-tail chain backwards (Linear backward, LeakyReLU masks, transposed convs, AvgPool backward incl. the final 8x8
    un-pooling written straight to global memory) and the penalty's tangent pass (bias-free convs times the saved masks, in
    place over the saved activations) against float64 autograd / jvp-by-hand."""
    from musicgan_amd import ops
    gen = torch.Generator().manual_seed(300 + n)
    ca, cb, cc = 28, 44, 36
    a_in = torch.randn(n, ca, 8, 8, generator=gen)        # pre-pool activation of the layer below (only its sign is used)
    q = F.avg_pool2d(F.leaky_relu(a_in, SLOPE), 2)
    w0, w1, w2 = _w(cb, ca, gen), _w(cc, cb, gen), _w(cc, cc, gen)
    b0, b1, b2 = (torch.randn(k, generator=gen) * 0.1 for k in (cb, cc, cc))
    wl = torch.randn(1, cc, generator=gen)
    gout = torch.randn(n, 1, generator=gen)
    a_full = a_in.double().requires_grad_(True)
    qd = F.avg_pool2d(F.leaky_relu(a_full, SLOPE), 2)
    t0 = F.conv2d(qd, w0.double(), b0.double(), padding=1)
    a0 = F.leaky_relu(t0, SLOPE)
    t1 = F.conv2d(a0, w1.double(), b1.double(), padding=1)
    a1 = F.leaky_relu(t1, SLOPE)
    q1 = F.avg_pool2d(a1, 2)
    q2 = F.avg_pool2d(q1, 2)                               # 4x4 -> 2x2 -> 1x1 to reach the linear layer
    t2 = F.conv2d(q2, w2.double(), b2.double(), padding=1)
    a2 = F.leaky_relu(t2, SLOPE)
    y = a2.reshape(n, -1) @ wl.double().t()
    for t in (t0, t1, t2):
        t.retain_grad()
    y.backward(gout.double())
    d = lambda t: t.detach().to(DEV, torch.float32).contiguous()
    wpd = [ops.pack_smallnet(d(w), True) for w in (w0, w1, w2)]
    E = lambda c, h: torch.full((n, c, h, h), float("nan"), device=DEV)
    H2, H1, H0, GA = E(cc, 1), E(cc, 4), E(cb, 4), E(ca, 8)
    sn = ops.SmallNet(g)
    sn.linbwd(0, cc, d(gout), d(wl))
    sn.mask(0, cc, 1, 1, d(a2), out=H2)                       # gradient w.r.t. t2
    sn.conv(0, 1, wpd[2], cc, cc, 1, 1)                       # -> gradient w.r.t. q2
    sn.poolbwd(1, 2, cc, 1, 1, torch.ones(n, cc, 2, 2, device=DEV))   # through the second pool (no activation in front: mask 1)
    sn.poolbwd(2, 0, cc, 2, 2, d(a1), out=H1)                 # through the first pool and LeakyReLU(t1)
    sn.conv(0, 1, wpd[1], cc, cb, 4, 4, mask=d(a0), out=H0)   # gradient w.r.t. t0
    sn.conv(1, 2, wpd[0], cb, ca, 4, 4)                       # gradient w.r.t. q
    sn.poolbwd(2, 0, ca, 4, 4, d(a_in), out=GA, lds=False)    # 8x8 un-pooling: global only
    sn.run(n)
    torch.cuda.synchronize()
    for name, got, want in (("h2", H2, t2.grad), ("h1", H1, t1.grad), ("h0", H0, t0.grad), ("g_a", GA, a_full.grad)):
        assert _rel(got, want) < 1e-5, (name, _rel(got, want))
    # tangent pass: u -> mask(a0) * conv(u, w0) -> mask(a1) * conv(., w1) -> pool, written over copies of the activations
    u = torch.randn(n, ca, 4, 4, generator=gen)
    m = lambda a: torch.where(a > 0, torch.ones_like(a), torch.full_like(a, SLOPE))
    v0 = F.conv2d(u.double(), w0.double(), padding=1) * m(a0.detach())
    v1 = F.conv2d(v0, w1.double(), padding=1) * m(a1.detach())
    vq = F.avg_pool2d(v1, 2)
    wp = [ops.pack_smallnet(d(w), False) for w in (w0, w1)]
    A0, A1, VQ = d(a0), d(a1), E(cc, 2)
    st = ops.SmallNet(g)
    st.load(0, d(u))
    st.conv(0, 1, wp[0], ca, cb, 4, 4, mask=A0, out=A0)       # in place: mask read, tangent written
    st.conv(1, 2, wp[1], cb, cc, 4, 4, mask=A1, out=A1)
    st.pool(2, 0, cc, 4, 4, out=VQ)
    st.run(n)
    torch.cuda.synchronize()
    for name, got, want in (("v0", A0, v0), ("v1", A1, v1), ("vq", VQ, vq)):
        assert _rel(got, want) < 1e-5, (name, _rel(got, want))


def test_full_width_layers_and_8x8_maps():
    """The widest layers of the networks (128 -> 144 at 4x4, 144 -> 160 at 2x2, 160 -> 160 on a 1x1 map) and an 8x8 map (64
    pixels = 4 pixel tiles per workgroup), batch 9 (ragged against 2 images per workgroup for the small maps)."""
    from musicgan_amd import ops
    gen = torch.Generator().manual_seed(5)
    for (ci, co, h, n, g) in ((128, 144, 4, 9, 1), (144, 160, 2, 9, 2), (160, 160, 1, 9, 4), (112, 128, 8, 3, 1), (128, 128, 4, 9, 2), (48, 40, 4, 9, 4)):
        x = torch.randn(n, ci, h, h, generator=gen)
        w, b = _w(co, ci, gen), torch.randn(co, generator=gen) * 0.1
        want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        want_t = F.conv_transpose2d(want, w.double(), padding=1)
        out = torch.full((n, co, h, h), float("nan"), device=DEV)
        back = torch.full((n, ci, h, h), float("nan"), device=DEV)
        sn = ops.SmallNet(g)
        sn.load(0, x.to(DEV))
        sn.conv(0, 1, ops.pack_smallnet(w.to(DEV), False), ci, co, h, h, bias=b.to(DEV), out=out)
        sn.conv(1, 2, ops.pack_smallnet(w.to(DEV), True), co, ci, h, h, out=back)
        sn.run(n)
        torch.cuda.synchronize()
        assert _rel(out, want) < 2e-6, (ci, co, h, _rel(out, want))
        assert _rel(back, want_t) < 3e-6, (ci, co, h, _rel(back, want_t))


def test_smallnet_rejects_what_does_not_fit():
    from musicgan_amd import _lib, ops
    x = torch.zeros(2, 16, 16, 16, device=DEV)
    sn = ops.SmallNet(1)
    sn.load(0, x)
    sn.conv(0, 1, ops.pack_smallnet(torch.zeros(16, 16, 3, 3, device=DEV), False), 16, 16, 16, 16)
    with pytest.raises(_lib.MusicGanHipError):
        sn.run(2)   # 256 pixels per workgroup


@pytest.mark.parametrize("n", [1, 5, 24, 70, 200])
def test_conv3x3_small_every_epilogue(n):
    """mg_conv3x3_small (one layer per launch, split-K over the waves) against float64 F.conv2d on 2x2, 4x4, 8x8 and non-square
    maps: bias + LeakyReLU, the LeakyReLU-derivative mask written in place over its source, AvgPool2d / 2x2 block sums as a second
    output, AvgPool2d-backward x mask as the output, nearest-upsampled input, data-gradient filters; batch sizes that exercise 1,
    2 and 4 pixel tiles per workgroup and ragged last workgroups."""
    from musicgan_amd import ops
    gen = torch.Generator().manual_seed(40 + n)
    d = lambda t: t.to(DEV, torch.float32).contiguous()
    lr = lambda t: F.leaky_relu(t, SLOPE)
    m = lambda a: torch.where(a > 0, torch.ones_like(a), torch.full_like(a, SLOPE))
    for (ci, co, h, w) in ((24, 40, 4, 4), (144, 160, 2, 2), (36, 20, 8, 8), (128, 128, 4, 4), (20, 36, 2, 4)):
        if n * h * w > 4096:
            continue
        x = torch.randn(n, ci, h, w, generator=gen)
        wt = _w(co, ci, gen)
        b = torch.randn(co, generator=gen) * 0.1
        wp, wpd = ops.pack_smallnet(d(wt), False), ops.pack_smallnet(d(wt), True)
        ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
        y, p = ops.conv3x3_small(d(x), wp, d(b), co, lrelu=True, pool=True)
        assert _rel(y, lr(ref)) < 2e-6 and _rel(p, F.avg_pool2d(lr(ref), 2)) < 2e-6, (ci, co, h, w)
        # tangent form: no bias, times the mask of a saved activation, written over it; pooled second output
        act = torch.randn(n, co, h, w, generator=gen)
        buf = d(act)
        pool_out = torch.full((n, co, h // 2, w // 2), float("nan"), device=DEV)
        ops.conv3x3_small(d(x), wp, None, co, mask_aux=buf, out=buf, pool_out=pool_out)
        want = F.conv2d(x.double(), wt.double(), padding=1) * m(act.double())
        assert _rel(buf, want) < 3e-6 and _rel(pool_out, F.avg_pool2d(want, 2)) < 3e-6, (ci, co, h, w)
        # data gradient (transposed, flipped filters): plain, with 2x2 block sums, and un-pooled x mask
        gy = torch.randn(n, co, h, w, generator=gen)
        gx = F.conv_transpose2d(gy.double(), wt.double(), padding=1)
        assert _rel(ops.conv3x3_small(d(gy), wpd, None, ci), gx) < 3e-6
        s4 = ops.conv3x3_small(d(gy), wpd, None, ci, upsum=True, want_y=False)[1]
        assert _rel(s4, 4 * F.avg_pool2d(gx, 2)) < 3e-6
        big = torch.randn(n, ci, 2 * h, 2 * w, generator=gen)
        un = ops.conv3x3_small(d(gy), wpd, None, ci, unpool_aux=d(big))
        assert _rel(un, 0.25 * F.interpolate(gx, scale_factor=2, mode="nearest") * m(big.double())) < 3e-6
        # nearest-upsampled input
        if h <= 4 and w <= 4:
            yu = ops.conv3x3_small(d(x), wp, d(b), co, ups=True, lrelu=True)
            assert _rel(yu, lr(F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), wt.double(), b.double(), padding=1))) < 2e-6


@pytest.mark.parametrize("n,c,h,ups", [(5, 32, 2, True), (24, 128, 4, False), (8, 128, 4, True), (3, 112, 8, False), (64, 20, 4, False)])
def test_conv3x3_small_with_pixelnorm_folded_into_its_staging(n, c, h, ups):
    """mg_conv3x3_small_pn: LeakyReLU(conv(PixelNorm(x_raw)) + bias) with p = PixelNorm(x_raw) and 1 / norm as side outputs, against
    float64 (layers.py:11-17 + F.conv2d), incl. nearest-upsampled input (the norm is the low-resolution pixel's), 8x8 row bands
    (halo rows normalised by the neighbours' workgroups too) and a channel count that is not a multiple of 16."""
    from musicgan_amd import ops
    gen = torch.Generator().manual_seed(77 + n)
    co = 48
    x = torch.randn(n, c, h, h, generator=gen)
    wt, b = _w(co, c, gen), torch.randn(co, generator=gen) * 0.1
    xd = x.double()
    rn = 1.0 / torch.sqrt((xd * xd).mean(dim=1, keepdim=True) + 1e-8)
    p = xd * rn
    src = F.interpolate(p, scale_factor=2, mode="nearest") if ups else p
    want = F.leaky_relu(F.conv2d(src, wt.double(), b.double(), padding=1), SLOPE)
    y, pg, rg = ops.conv3x3_small_pn(x.to(DEV), ops.pack_smallnet(wt.to(DEV), False), b.to(DEV), co, ups=ups)
    assert _rel(pg, p) < 2e-6 and _rel(rg, rn) < 2e-6 and _rel(y, want) < 3e-6
    y2, p2, r2 = ops.conv3x3_small_pn(x.to(DEV), ops.pack_smallnet(wt.to(DEV), False), b.to(DEV), co, ups=ups, save=False)
    assert p2 is None and r2 is None and torch.equal(y2, y)
