"""Progressive-growing generator with the reference's constructor / forward / growth API
(/root/reference/music_gan/networks/generator.py:55-171) running on the MI355X kernels.

Drop-in facts kept: class name (so the name-mangled state_dict keys `_Generator__gen_blocks.{i}.{0,4}.*`,
`_Generator__end_block.0.*`, `_Generator__last_end_block.0.0.*` match), parameter creation order (same-seed init equals the
reference's), all 8 blocks registered up front, `next_layer()` re-using the current head object as the previous head,
`zero_grad()` setting grads to None, fully-convolutional forward (non-square latents allowed).
"""
from __future__ import annotations

from typing import Iterator

import torch as th
import torch.nn as nn

from . import engine
from .layers import ConvParams, Holder

_TAIL = (128, 112, 96, 80, 64, 48, 32, 16)


class _GenFn(th.autograd.Function):
    @staticmethod
    def forward(ctx, z, alpha, net, *params):
        W = net._weights()
        need = any(ctx.needs_input_grad)  # grad mode is off inside Function.forward; this reflects the caller's
        out, saved = engine.gen_forward(W, z.detach(), alpha, net._pack_cache, save=need)
        ctx.net, ctx.W, ctx.saved = net, W, saved
        ctx.need_gz = ctx.needs_input_grad[0]
        return out

    @staticmethod
    @th.autograd.function.once_differentiable
    def backward(ctx, g_out):
        sink = engine.GradSink()
        gz = engine.gen_backward(ctx.W, ctx.saved, g_out, ctx.net._pack_cache, sink, need_gz=ctx.need_gz)
        ctx.saved = None
        return (gz, None, None) + tuple(sink.get(p) for p in ctx.W.tensors())


class Generator(nn.Module):
    def __init__(self, rand_channels: int, end_layer: int = 0):
        super().__init__()
        self.__curr_layer = end_layer
        self.__nb_downsample = 7
        ins = (rand_channels,) + _TAIL[:-1]
        channels = list(zip(ins, _TAIL))
        self.__channels = channels
        assert 0 <= end_layer < len(channels), f"0 <= {end_layer} < {len(channels)}"

        # child names "0" and "4" = positions of the two convs in the reference's Block(nn.Sequential)
        self.__gen_blocks = nn.ModuleList([
            Holder(_0=ConvParams(ci, ci, 3), _4=ConvParams(ci, co, 3)) for ci, co in channels
        ])
        self.__end_block = Holder(_0=ConvParams(channels[end_layer][1], 2, 1))
        self.__last_end_block = None if end_layer == 0 else Holder(
            _0=Holder(_0=ConvParams(channels[end_layer - 1][1], 2, 1)))
        self._pack_cache = engine.PackCache()

    # ------------------------------------------------------------------ engine plumbing
    def _weights(self) -> engine.GenWeights:
        blocks = []
        for i in range(self.__curr_layer + 1):
            b = self.__gen_blocks[i]
            c0, c4 = b.child("0"), b.child("4")
            blocks.append((c0.weight, c0.bias, c4.weight, c4.bias))
        head = self.__end_block.child("0")
        old = None
        if self.__last_end_block is not None:
            o = self.__last_end_block.child("0").child("0")
            old = (o.weight, o.bias)
        return engine.GenWeights(blocks, (head.weight, head.bias), old)

    def forward(self, z: th.Tensor, alpha: float) -> th.Tensor:
        W = self._weights()
        return _GenFn.apply(z, float(alpha), self, *W.tensors())

    # ------------------------------------------------------------------ growth
    def next_layer(self) -> bool:
        if self.growing:
            self.__curr_layer += 1
            self.__last_end_block = Holder(_0=self.__end_block)
            device = next(self.__gen_blocks.parameters()).device
            self.__end_block = Holder(_0=ConvParams(self.__channels[self.curr_layer][1], 2, 1)).to(device)
            return True
        return False

    @property
    def down_sample(self) -> int:
        return self.__nb_downsample

    @property
    def curr_layer(self) -> int:
        return self.__curr_layer

    @property
    def growing(self) -> bool:
        return self.curr_layer < len(self.__gen_blocks) - 1

    def end_block_params(self) -> Iterator[nn.Parameter]:
        return self.__end_block.parameters()

    def zero_grad(self, set_to_none: bool = False) -> None:
        for p in self.parameters():
            p.grad = None
