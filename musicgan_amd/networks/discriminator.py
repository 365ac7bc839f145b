"""Progressive-growing critic with the reference's API (/root/reference/music_gan/networks/discriminator.py:53-191) on the
MI355X kernels.  `gradient_penalty()` evaluates the WGAN-GP term and its parameter gradient with the hand-derived
second-order pass of `engine.disc_gp_param_grads` instead of autograd's create_graph double backward; user code that
differentiates `forward()` twice itself (`autograd.grad(out, x, create_graph=True)` as discriminator.py:170-176 does) reaches the
same closed form through `_DiscInputGradFn`."""
from __future__ import annotations

from typing import Iterator

import torch as th
import torch.nn as nn

from . import engine
from .. import ops
from .layers import ConvParams, Holder

_CHANNELS = ((16, 32), (32, 48), (48, 64), (64, 80), (80, 96), (96, 112), (112, 128), (128, 144), (144, 160))
GRAD_PEN_FACTOR = 10.0


class _DiscFn(th.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha, net, *params):
        W = net._weights()
        need = any(ctx.needs_input_grad)  # grad mode is off inside Function.forward; this reflects the caller's
        out, saved = engine.disc_forward(W, x.detach(), alpha, net._pack_cache, save=need)
        ctx.net, ctx.W, ctx.saved = net, W, saved
        ctx.need_gx = ctx.needs_input_grad[0]
        ctx.need_gp = any(ctx.needs_input_grad[3:])
        return out

    @staticmethod
    def backward(ctx, g_out):
        if th.is_grad_enabled() and ctx.need_gx:
            # autograd.grad(..., create_graph=True) (discriminator.py:170-176 does this inside gradient_penalty; user code may do
            # it itself): the input gradient becomes a node whose own backward is the closed-form second-order pass
            state = _DiscState(ctx.net, ctx.W, ctx.saved, ctx.need_gp)
            gx = _DiscInputGradFn.apply(g_out, state, *ctx.W.tensors())
            return (gx, None, None) + state.pg
        sink = engine.GradSink() if ctx.need_gp else None
        gx, _ = engine.disc_backward(ctx.W, ctx.saved, g_out, ctx.net._pack_cache, sink, need_gx=ctx.need_gx)
        ctx.saved = None
        pg = tuple(sink.get(p) for p in ctx.W.tensors()) if sink is not None else (None,) * len(ctx.W.tensors())
        return (gx, None, None) + pg


class _DiscState:
    def __init__(self, net, W, saved, need_gp):
        self.net, self.W, self.saved, self.need_gp = net, W, saved, need_gp
        self.pg = (None,) * len(W.tensors())


class _DiscInputGradFn(th.autograd.Function):
    """g_x = J(x)^T g_out, the critic's input gradient, as a function of (g_out, parameters).  The critic is piecewise linear in x
    and linear in each weight along this chain, so a loss L(g_x) has  dL/d theta = the penalty pass of engine.disc_gp_param_grads
    with u_0 = dL/dg_x,  dL/dg_out = J(x) u_0 (the tangent of the scores)  and  dL/dx = 0 -- what autograd's double backward
    through the reference's modules yields (leaky_relu's second derivative is zero almost everywhere).  The first-order
    parameter gradients that `_DiscFn.backward` returns alongside are constants of this node (differentiating THEM again is not
    supported)."""

    @staticmethod
    def forward(ctx, g_out, state, *params):
        sink = engine.GradSink() if state.need_gp else None
        gx, hs = engine.disc_backward(state.W, state.saved, g_out.detach(), state.net._pack_cache, sink, need_gx=True, keep_h=True)
        if sink is not None:
            state.pg = tuple(sink.get(p) for p in state.W.tensors())
        ctx.state, ctx.hs, ctx.g_out = state, hs, g_out.detach()
        return gx

    @staticmethod
    @th.autograd.function.once_differentiable
    def backward(ctx, u):
        st = ctx.state
        sink = engine.GradSink()
        _, t_out = engine.disc_gp_param_grads(st.W, st.saved, ctx.hs, u.contiguous(), st.net._pack_cache, sink, g_out=ctx.g_out,
                                              want_t=True)
        return (t_out, None) + tuple(sink.get(p) for p in st.W.tensors())


class _GradPenFn(th.autograd.Function):
    """penalty = 10 * mean((||grad_x D(x~)||_2 - 1)^2); backward yields d penalty / d theta (zero w.r.t. x_real/x_gen and the
    biases, exactly what autograd's double backward produces for this piecewise-linear critic)."""

    @staticmethod
    def forward(ctx, x_real, x_gen, eps, alpha, net, *params):
        W = net._weights()
        cache = net._pack_cache
        x_i = ops.gp_interp(x_real.detach().contiguous(), x_gen.detach().contiguous(), eps.contiguous())
        out, saved = engine.disc_forward(W, x_i, alpha, cache, save=True)
        g0, hs = engine.disc_backward(W, saved, th.ones_like(out), cache, None, need_gx=True, keep_h=True)
        ss = ops.sumsq_per_sample(g0)
        pen, _ = ops.gp_finish(ss, GRAD_PEN_FACTOR, want_coef=False)
        ctx.net, ctx.W, ctx.saved, ctx.hs, ctx.g0, ctx.ss = net, W, saved, hs, g0, ss
        return pen

    @staticmethod
    @th.autograd.function.once_differentiable
    def backward(ctx, g_pen):
        _, coef = ops.gp_finish(ctx.ss, GRAD_PEN_FACTOR, 1.0, want_penalty=False)
        coef = (coef * g_pen.reshape(())).contiguous()
        u0 = ops.scale_per_sample(ctx.g0, coef)
        sink = engine.GradSink()
        engine.disc_gp_param_grads(ctx.W, ctx.saved, ctx.hs, u0, ctx.net._pack_cache, sink)
        ctx.saved = ctx.hs = ctx.g0 = None
        return (None, None, None, None, None) + tuple(sink.get(p) for p in ctx.W.tensors())


class Discriminator(nn.Module):
    def __init__(self, start_layer: int = 7):
        super().__init__()
        self.__channels = list(_CHANNELS)
        self.__curr_layer = start_layer
        self.__nb_layer = len(_CHANNELS)
        assert 0 <= start_layer <= len(_CHANNELS)

        # child names "0" and "3" = positions of the two convs in the reference's ConvBlock(nn.Sequential)
        self.__conv_blocks = nn.ModuleList([
            Holder(_0=ConvParams(ci, co, 3), _3=ConvParams(co, co, 3)) for ci, co in _CHANNELS
        ])
        self.__last_start_block = None
        self.__start_block = Holder(_0=ConvParams(2, _CHANNELS[self.curr_layer][0], 1))
        # the reference's  160 * 512 // 2**9 * 512 // 2**9  evaluates left to right to 160
        out_size = _CHANNELS[-1][1] * 512 // 2 ** self.__nb_layer * 512 // 2 ** self.__nb_layer
        self.__clf = Holder(_0=ConvParams(out_size, 1, 0))
        self._pack_cache = engine.PackCache()

    # ------------------------------------------------------------------ engine plumbing
    def _weights(self) -> engine.DiscWeights:
        s = self.__start_block.child("0")
        blocks = []
        for i in range(self.__curr_layer, len(self.__conv_blocks)):
            b = self.__conv_blocks[i]
            c0, c3 = b.child("0"), b.child("3")
            blocks.append((c0.weight, c0.bias, c3.weight, c3.bias))
        old = None
        if self.__last_start_block is not None:
            o = self.__last_start_block.child("1").child("0")
            old = (o.weight, o.bias)
        c = self.__clf.child("0")
        return engine.DiscWeights((s.weight, s.bias), blocks, old, (c.weight, c.bias))

    def forward(self, x: th.Tensor, alpha: float) -> th.Tensor:
        W = self._weights()
        return _DiscFn.apply(x, float(alpha), self, *W.tensors())

    # ------------------------------------------------------------------ growth
    def next_layer(self) -> bool:
        if self.growing:
            self.__curr_layer -= 1
            # child "1": the reference wraps (AvgPool2d, old start block) in a Sequential
            self.__last_start_block = Holder(_1=self.__start_block)
            device = next(self.__conv_blocks.parameters()).device
            self.__start_block = Holder(_0=ConvParams(2, self.__channels[self.curr_layer][0], 1)).to(device)
            return True
        return False

    @property
    def curr_layer(self) -> int:
        return self.__curr_layer

    @property
    def growing(self) -> bool:
        return self.__curr_layer > 0

    def gradient_penalty(self, x_real: th.Tensor, x_gen: th.Tensor, alpha: float) -> th.Tensor:
        device = next(self.parameters()).device
        batch_size = x_real.size()[0]
        eps = th.rand(batch_size, 1, 1, 1, device=device)
        return self.gradient_penalty_with_eps(x_real, x_gen, alpha, eps)

    def gradient_penalty_with_eps(self, x_real, x_gen, alpha: float, eps: th.Tensor) -> th.Tensor:
        """Same as gradient_penalty() with the interpolation coefficients injected (tests, reproducible runs)."""
        W = self._weights()
        return _GradPenFn.apply(x_real, x_gen, eps, float(alpha), self, *W.tensors())

    def start_block_parameters(self) -> Iterator[nn.Parameter]:
        return self.__start_block.parameters()

    def zero_grad(self, set_to_none: bool = False) -> None:
        for p in self.parameters():
            p.grad = None
