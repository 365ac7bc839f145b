"""Parameter holders and PixelNorm for the MI355X ProGAN networks.

The arithmetic lives in libmusicgan_hip.so; these modules only own `nn.Parameter`s with the reference's names, shapes and
RNG draw order so that same-seed initialisation and checkpoints are interchangeable with
/root/reference/music_gan/networks (generator.py:83-104, discriminator.py:81-105, layers.py:5-17).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops


class ConvParams(nn.Module):
    """weight/bias of a k x k convolution (k=0: a Linear), initialised exactly like torch's nn.Conv2d / nn.Linear default
    (kaiming_uniform(a=sqrt 5) on the weight, then U(+-1/sqrt(fan_in)) on the bias; weight drawn first)."""

    def __init__(self, in_channels: int, out_channels: int, kernel: int):
        super().__init__()
        shape = (out_channels, in_channels, kernel, kernel) if kernel > 0 else (out_channels, in_channels)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = in_channels * max(kernel, 1) ** 2
        bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
        nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self) -> str:
        return f"weight={tuple(self.weight.shape)}"


class Holder(nn.Module):
    """Numbered container reproducing the child names an nn.Sequential would give ("0", "1", "3", "4", ...)."""

    def __init__(self, **children: nn.Module):
        super().__init__()
        for name, mod in children.items():
            self.add_module(name.lstrip("_"), mod)

    def child(self, name: str) -> nn.Module:
        return self._modules[name]


class _PixelNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        p, rn = ops.pixelnorm_fwd(x.contiguous())
        ctx.save_for_backward(x, rn)
        return p

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gp):
        x, rn = ctx.saved_tensors
        return ops.pixelnorm_bwd(gp.contiguous(), x.contiguous(), rn)


class PixelNorm(nn.Module):
    """x / sqrt(mean_c(x^2) + eps), eps fixed at the reference's 1e-8 (layers.py:5-17)."""

    def __init__(self, epsilon: float = 1e-8):
        super().__init__()
        assert epsilon == 1e-8, "the HIP kernel hard-codes the reference epsilon 1e-8"
        self.__epsilon = epsilon

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _PixelNormFn.apply(x)

    def __repr__(self):
        return f"PixelNorm(eps={self.__epsilon})"

    def __str__(self):
        return self.__repr__()
