"""The memory-bound launches of a level-5 step on their own: us per call, algorithmic bytes, TB/s (a device-to-device copy reaches ~6.3
on this part).   python tools/bench_hbm_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from musicgan_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
R = lambda *s: torch.randn(*s, device=dev, generator=g)


def timeit(fn, n=30):
    for _ in range(8): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def row(name, us, nbytes):
    print(f"{name:64s} {us:8.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / us / 1e6:6.2f} TB/s", flush=True)


big = R(256 * 1024 * 1024)
dst = torch.empty_like(big)
row("copy 1 GiB (torch)", timeit(lambda: dst.copy_(big)), 2 * big.numel() * 4)
del big, dst
# 1x1 weight gradients: critic stem (x few, gy many), generator head (x many, gy few, tanh derivative)
for (n, c_many, hw, stem) in [(192, 48, 128, True), (192, 64, 64, True), (64, 48, 128, False), (64, 64, 64, False)]:
    if stem:
        x, gy = R(n, 2, hw, hw), R(n, c_many, hw, hw)
        gw, gb = torch.empty(c_many, 2, 1, 1, device=dev), torch.empty(c_many, device=dev)
        f = lambda: ops.conv1x1_wgrad(x, gy, gw, gb, bias_n=n // 3)
        nb = (x.numel() + gy.numel()) * 4
    else:
        x, gy, t = R(n, c_many, hw, hw), R(n, 2, hw, hw), torch.tanh(R(n, 2, hw, hw))
        gw, gb = torch.empty(2, c_many, 1, 1, device=dev), torch.empty(2, device=dev)
        f = lambda: ops.conv1x1_wgrad(x, gy, gw, gb, tanh_y=t)
        nb = (x.numel() + 2 * gy.numel()) * 4
    row(f"conv1x1_wgrad {'stem' if stem else 'head'} {n}x{c_many}@{hw}", timeit(f), nb)
# PixelNorm + LeakyReLU backward (generator)
for (n, c, hw) in [(64, 48, 128), (64, 64, 64)]:
    gp, p = R(n, c, hw, hw), R(n, c, hw, hw)
    rn = torch.rand(n, 1, hw, hw, device=dev, generator=g) + 0.5
    row(f"pixelnorm_lrelu_bwd from_p {n}x{c}@{hw}", timeit(lambda: ops.pixelnorm_lrelu_bwd(gp, p, rn, from_p=True)), (3 * gp.numel() + rn.numel()) * 4)
# generator head backward: three launches (1x1 weight gradient, transposed 1x1, PixelNorm + LeakyReLU backward) against the fused one
for (n, c, hw) in [(64, 48, 128), (6, 32, 256), (6, 16, 512), (16, 16, 512), (32, 64, 64)]:
    p, gm, mp = R(n, c, hw, hw), R(n, 2, hw, hw), torch.tanh(R(n, 2, hw, hw))
    rn = torch.rand(n, 1, hw, hw, device=dev, generator=g) + 0.5
    w = R(2, c, 1, 1) * 0.1
    gw, gb = torch.empty(2, c, 1, 1, device=dev), torch.empty(2, device=dev)

    def three():
        ops.conv1x1_wgrad(p, gm, gw, gb, tanh_y=mp)
        gg = ops.conv1x1(gm, w, None, c, transposed=True, tanh_bwd_in=mp)
        return ops.pixelnorm_lrelu_bwd(gg, p, rn, from_p=True)

    row(f"head backward, three launches {n}x{c}@{hw}", timeit(three), (5 * p.numel()) * 4)
    if ops.gen_head_bwd_supported(c):
        row(f"head backward, fused {n}x{c}@{hw}", timeit(lambda: ops.gen_head_bwd(gm, mp, w, p, rn, gw, gb)), (2 * p.numel()) * 4)
# the old head of a fading-in level: 1x1 weight gradient + transposed 1x1 accumulated into the conv's data gradient + PixelNorm backward,
# against the fused launch with that data gradient as second input
for (n, c, hw) in [(64, 64, 64), (6, 48, 128), (6, 32, 256)]:
    p, gm, mp, gin = R(n, c, hw, hw), R(n, 2, hw, hw), torch.tanh(R(n, 2, hw, hw)), R(n, c, hw, hw)
    rn = torch.rand(n, 1, hw, hw, device=dev, generator=g) + 0.5
    w = R(2, c, 1, 1) * 0.1
    gw, gb = torch.empty(2, c, 1, 1, device=dev), torch.empty(2, device=dev)
    gacc = gin.clone()

    def three_old():
        ops.conv1x1_wgrad(p, gm, gw, gb, tanh_y=mp)
        ops.conv1x1(gm, w, None, c, transposed=True, tanh_bwd_in=mp, out=gacc, accumulate=True)
        return ops.pixelnorm_lrelu_bwd(gacc, p, rn, from_p=True)

    row(f"old head backward, three launches {n}x{c}@{hw}", timeit(three_old), (5 * p.numel()) * 4)
    row(f"old head backward, fused {n}x{c}@{hw}", timeit(lambda: ops.gen_head_bwd(gm, mp, w, p, rn, gw, gb, g_in=gin)), (3 * p.numel()) * 4)
