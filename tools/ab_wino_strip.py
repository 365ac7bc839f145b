"""A/B of the two Winograd conv kernels on the few-channel layer shapes of levels 6-7 (batch 6: 6 / 18 images): the LDS-staged
tile-block kernel (wino3x3.hip, MG_WINO_STRIP=0) against the wave-per-block strip kernel (wino_strip.hip, forced with
MG_WINO_STRIP=2), same library, interleaved timing, bitwise comparison of every output, and each launch against its bound
max(algorithmic bytes / 8 TB/s, executed FLOP / 157.3 TF/s).   python tools/ab_wino_strip.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(11)


def run(mode, x, wpk, b, co, extra):
    if mode == "fwd_pool_mask":      # critic forward: LeakyReLU + AvgPool2d + tile mask (full-resolution y never written)
        m, p = ops.conv3x3(x, None, b, co, lrelu=True, pool=True, wino=wpk, mask_out=True)
        return (m, p)
    if mode == "tangent":            # penalty tangent pass: tile-mask bytes in, pooled out
        _, p = ops.conv3x3(x, None, None, co, mask_aux=extra["mask"], pool=True, wino=wpk)
        return (p,)
    if mode == "plain":              # LeakyReLU, full-resolution output
        return (ops.conv3x3(x, None, b, co, lrelu=True, wino=wpk),)
    if mode == "dgrad_mask":         # data gradient times the fp32 LeakyReLU mask of the layer below
        return (ops.conv3x3(x, None, None, co, mask_aux=extra["act"], wino=wpk),)
    if mode == "unpool":             # data gradient -> AvgPool2d backward x LeakyReLU mask bytes (output at twice the size)
        return (ops.conv3x3(x, None, None, co, wino=wpk, unpool_mask=extra["umask"]),)
    if mode == "pn":                 # generator: LeakyReLU + PixelNorm fused
        _, p, rn = ops.conv3x3(x, None, b, co, lrelu=True, pixnorm=True, want_y=False, wino=wpk)
        return (p, rn)
    if mode == "act_pool":           # LeakyReLU -> y and the pooled tensor
        return ops.conv3x3(x, None, b, co, lrelu=True, pool=True, wino=wpk)
    if mode == "maskf_pool":         # tangent pass on fp32 masks: masked result + pooled
        return ops.conv3x3(x, None, None, co, mask_aux=extra["act"], pool=True, wino=wpk)
    if mode in ("fade_fwd", "fade_tan", "fade_bwd"):
        from musicgan_amd import _lib
        m = {"fade_fwd": _lib.MG_FADE_FWD, "fade_tan": _lib.MG_FADE_TANGENT, "fade_bwd": _lib.MG_FADE_BWD}[mode]
        r = ops.conv3x3_fade(x, wpk, b if mode == "fade_fwd" else None, co, m, extra["act"], extra["coef"],
                             mask_in=None if mode == "fade_fwd" else extra["mask"])
        return r if isinstance(r, tuple) else (r,)
    raise ValueError(mode)


def bound_us(mode, n, ci, co, h):
    px = n * h * h
    fl = 18.0 * px * ci * co / 2.25
    by = 4.0 * px * ci
    by += {"fwd_pool_mask": px * co * (4 / 4 + 1 / 4), "tangent": px * co * (4 / 4 + 1 / 4), "plain": 4.0 * px * co,
           "dgrad_mask": 8.0 * px * co, "unpool": px * co * (16 + 1), "pn": px * (4.0 * co + 4), "act_pool": 5.0 * px * co,
           "maskf_pool": 9.0 * px * co, "fade_fwd": 8.25 * px * co, "fade_tan": 8.25 * px * co, "fade_bwd": 12.25 * px * co}[mode]
    return max(by / 8e12, fl / 157.3e12) * 1e6, by, fl


cases = [(18, 16, 32, 512, "fwd_pool_mask"), (6, 16, 32, 512, "tangent"), (18, 32, 16, 512, "dgrad_mask"),
         (18, 32, 32, 256, "plain"), (18, 32, 32, 256, "unpool"), (18, 32, 48, 256, "fwd_pool_mask"), (18, 48, 32, 256, "plain"),
         (18, 48, 48, 128, "plain"), (18, 48, 48, 128, "unpool"), (6, 32, 32, 256, "pn"), (6, 48, 48, 128, "plain"),
         (6, 32, 16, 512, "dgrad_mask"), (6, 32, 32, 256, "plain"), (18, 16, 32, 512, "act_pool"), (6, 16, 32, 512, "maskf_pool"),
         (18, 32, 32, 256, "fade_fwd"), (6, 32, 32, 256, "fade_tan"), (18, 48, 32, 256, "fade_bwd"), (18, 32, 48, 256, "plain"),
         (4, 16, 16, 64, "plain"), (3, 8, 48, 96, "pn"), (5, 24, 32, 64, "unpool")]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    cases = cases[:4]
if len(sys.argv) > 1 and sys.argv[1] == "odd":  # three, five and six out-channel tiles: two per wave + a padding tile, or three per wave
    cases = [(192, 64, 48, 128, "dgrad_mask"), (18, 32, 48, 256, "fwd_pool_mask"), (18, 32, 48, 256, "plain"), (18, 48, 48, 128, "unpool"),
             (192, 64, 80, 64, "fwd_pool_mask"), (64, 64, 80, 64, "tangent"), (192, 80, 64, 64, "unpool"), (192, 80, 80, 32, "plain"),
             (192, 80, 96, 32, "fwd_pool_mask"), (18, 64, 48, 128, "fade_bwd"), (18, 64, 80, 64, "fwd_pool_mask")]
if len(sys.argv) > 1 and sys.argv[1] == "pad":  # odd tile counts at the sizes where the padded row's share of the grid matters
    cases = [(192, 64, 48, 128, "dgrad_mask"), (18, 32, 48, 256, "plain"), (18, 48, 48, 128, "unpool"), (192, 64, 80, 64, "fwd_pool_mask"),
             (18, 64, 48, 128, "fade_bwd"), (64, 64, 48, 128, "dgrad_mask")]
if len(sys.argv) > 1 and sys.argv[1] == "l5":  # the 64-out-channel layers of level 5 at batch 64 (192 images through the critic)
    cases = [(192, 48, 64, 128, "fwd_pool_mask"), (64, 48, 64, 128, "tangent"), (192, 64, 48, 128, "dgrad_mask"), (192, 64, 64, 64, "fade_fwd"),
             (192, 64, 64, 64, "unpool"), (64, 48, 64, 128, "fwd_pool_mask"), (192, 64, 64, 64, "plain"), (192, 64, 80, 64, "fwd_pool_mask"),
             (192, 80, 64, 64, "unpool")]
if len(sys.argv) > 1 and sys.argv[1] == "nt1":  # 16 out-channels: one tile per wave
    cases = [(18, 32, 16, 512, "dgrad_mask"), (6, 32, 16, 512, "dgrad_mask"), (18, 16, 16, 512, "plain"), (6, 32, 16, 256, "pn"),
             (18, 48, 16, 256, "dgrad_mask")]
if len(sys.argv) > 1 and sys.argv[1] == "small":  # the 32x32 layers of level 5 (192 / 64 images) and the 64x64 ones of level 6 (18 / 6): < 2 048 tile blocks or Cin >= 96
    cases = [(192, 96, 80, 32, "dgrad_mask"), (192, 80, 80, 32, "unpool"), (192, 80, 96, 32, "fwd_pool_mask"), (192, 80, 80, 32, "plain"),
             (64, 80, 80, 32, "unpool"), (64, 96, 80, 32, "dgrad_mask"), (64, 80, 80, 32, "dgrad_mask"), (64, 80, 96, 32, "tangent"),
             (64, 80, 96, 32, "fwd_pool_mask"), (64, 80, 80, 32, "plain"), (18, 64, 80, 64, "fwd_pool_mask"), (18, 80, 64, 64, "dgrad_mask"),
             (18, 64, 64, 64, "plain"), (6, 64, 64, 64, "plain"), (6, 64, 80, 64, "tangent"), (96, 64, 80, 64, "fwd_pool_mask"), (32, 64, 64, 64, "plain")]
if len(sys.argv) > 1 and sys.argv[1] == "l4":  # level 4 at batch 32 (96 / 32 images): the 32x32 layers have 1 536 / 512 tile blocks
    cases = [(96, 80, 96, 32, "fwd_pool_mask"), (96, 96, 80, 32, "dgrad_mask"), (96, 80, 80, 32, "unpool"), (96, 80, 80, 32, "plain"),
             (32, 80, 96, 32, "tangent"), (32, 80, 80, 32, "plain"), (32, 96, 80, 32, "dgrad_mask"), (32, 80, 80, 32, "unpool"),
             (96, 96, 96, 16, "plain"), (96, 96, 112, 16, "fwd_pool_mask")]
# variants: "0" the staged kernel; "2" strip, all out-channel tiles in a wave; "2n1" strip, one tile per wave (tiles on grid.y)
VARIANTS = ["0", "2"] + [v for v in sys.argv[1:] if v.startswith("2")]


def setenv(v):
    os.environ["MG_WINO_STRIP"] = v[0]
    for k in ("MG_WINO_STRIP_NIW", "MG_WINO_STRIP_WGS"):
        os.environ.pop(k, None)
    if "n" in v:
        os.environ["MG_WINO_STRIP_NIW"] = v[v.index("n") + 1]
    if "w" in v:
        os.environ["MG_WINO_STRIP_WGS"] = v[v.index("w") + 1]


print("n  cin->cout @side mode            staged us  " + "  ".join(f"strip[{v}]" for v in VARIANTS[1:]) + "   best/staged   bound us   best/bound   bitwise")
for (n, ci, co, h, mode) in cases:
    x = torch.randn(n, ci, h, h, device=dev, generator=g)
    w = torch.randn(co, ci, 3, 3, device=dev, generator=g) * 0.05
    b = torch.randn(co, device=dev, generator=g)
    wpk = ops.pack_wino3x3(w, False)
    extra = {}
    if mode in ("tangent", "fade_tan", "fade_bwd"):
        extra["mask"] = torch.randint(0, 16, (n, co, h // 2, h // 2), device=dev, generator=g).to(torch.uint8)
    if mode.startswith("fade"):
        extra["coef"] = torch.tensor([0.37, 0.63], device=dev)
    if mode in ("dgrad_mask", "maskf_pool", "fade_fwd", "fade_tan", "fade_bwd"):
        extra["act"] = torch.randn(n, co, h, h, device=dev, generator=g)
    if mode == "unpool":
        extra["umask"] = torch.randint(0, 16, (n, co, h, h), device=dev, generator=g).to(torch.uint8)
    outs, times = {}, {k: 0.0 for k in VARIANTS}
    for env in VARIANTS:
        setenv(env)
        outs[env] = run(mode, x, wpk, b, co, extra)
    torch.cuda.synchronize()
    same = all(all(torch.equal(p, q) for p, q in zip(outs["0"], outs[k])) for k in VARIANTS)
    for rep in range(3):
        for env in VARIANTS:
            setenv(env)
            for _ in range(5):
                run(mode, x, wpk, b, co, extra)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(mode, x, wpk, b, co, extra)
            e1.record(); e1.synchronize()
            times[env] += e0.elapsed_time(e1) / 20 / 3 * 1e3
    bd, _, _ = bound_us(mode, n, ci, co, h)
    best = min(times[k] for k in VARIANTS if k != "0")
    print(f"{n:3d} {ci:3d}->{co:3d} @{h:3d} {mode:14s} {times['0']:9.1f} " + " ".join(f"{times[k]:9.1f}" for k in VARIANTS if k != "0") +
          f" {best / times['0']:7.3f} {bd:9.1f} {best / bd:9.2f}        {same}", flush=True)
    del x, outs
for k in ("MG_WINO_STRIP", "MG_WINO_STRIP_NIW", "MG_WINO_STRIP_WGS"):
    os.environ.pop(k, None)
