"""CPU oracle for the ProGAN WGAN-GP training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) restatement of the reference's
algorithm.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package ``musicgan_amd`` never does.

Parity pin: ``tools/gen_golden.py`` imports the reference's own ``music_gan.networks``
in the build container, runs it on seeded inputs and commits the results under
``tests/golden/progan_*.npz``; ``tests/test_oracle_golden.py`` checks this file against
those vectors (weights bit-exact by hash, outputs/grads to fp32 round-off).

Reference lines restated (paths relative to /root/reference/music_gan):
  networks/generator.py:9-40     Block = conv3x3 . LReLU(0.2) . PixelNorm . Up x2 . conv3x3 . LReLU . PixelNorm
  networks/generator.py:43-52    ToMagnPhaseLayer = conv1x1(C->2) . tanh
  networks/generator.py:55-126   Generator ctor (creation order = RNG order) and forward with fade-in
  networks/generator.py:128-152  next_layer(): old head becomes last_end_block[0], fresh head drawn
  networks/discriminator.py:8-50 ConvBlock = conv3x3 . LReLU . AvgPool2 . conv3x3 . LReLU ; MagPhaseLayer = conv1x1(2->C) . LReLU
  networks/discriminator.py:53-124  Discriminator ctor / forward with fade-in, Linear(160,1)
  networks/discriminator.py:157-184 gradient_penalty (eps~U[0,1), autograd.grad(create_graph), 10*mean((|g|-1)^2))
  networks/layers.py:5-17        PixelNorm: x / sqrt(mean_c(x^2) + 1e-8)
  networks/criterion.py:12-18    Wasserstein losses
  train.py:135-221               D step (G not detached) and G step, Adam(lr 1e-3, betas (0, 0.9))
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

G_TAIL = [128, 112, 96, 80, 64, 48, 32, 16]  # generator.py:67-76 (out channels of block i)
D_CHANNELS = [(16, 32), (32, 48), (48, 64), (64, 80), (80, 96), (96, 112), (112, 128), (128, 144),
              (144, 160)]  # discriminator.py:60-70
LRELU = 0.2
PN_EPS = 1e-8
GP_FACTOR = 10.0


def g_channels(rand_channels: int):
    ins = [rand_channels] + G_TAIL[:-1]
    return list(zip(ins, G_TAIL))


# --------------------------------------------------------------------------- init
def _conv_init(out_c: int, in_c: int, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Stock nn.Conv2d/nn.Linear default init (kaiming_uniform a=sqrt(5), then bias U(+-1/sqrt(fan_in)));
    the draw order weight->bias is what makes same-seed weights equal the reference's."""
    shape = (out_c, in_c, k, k) if k > 0 else (out_c, in_c)
    w = torch.empty(shape)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    fan_in = in_c * max(k, 1) * max(k, 1)
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
    b = torch.empty(out_c)
    torch.nn.init.uniform_(b, -bound, bound)
    return w, b


class GenState:
    """Generator parameters keyed exactly like the reference state_dict (generator.py:55-104)."""

    def __init__(self, rand_channels: int, end_layer: int = 0):
        self.rand_channels = rand_channels
        self.channels = g_channels(rand_channels)
        assert 0 <= end_layer < len(self.channels)
        self.curr_layer = end_layer
        p = OrderedDict()
        for i, (ci, co) in enumerate(self.channels):
            p[f"_Generator__gen_blocks.{i}.0.weight"], p[f"_Generator__gen_blocks.{i}.0.bias"] = _conv_init(ci, ci, 3)
            p[f"_Generator__gen_blocks.{i}.4.weight"], p[f"_Generator__gen_blocks.{i}.4.bias"] = _conv_init(co, ci, 3)
        p["_Generator__end_block.0.weight"], p["_Generator__end_block.0.bias"] = _conv_init(
            2, self.channels[end_layer][1], 1)
        self.has_last = end_layer > 0
        if self.has_last:
            p["_Generator__last_end_block.0.0.weight"], p["_Generator__last_end_block.0.0.bias"] = _conv_init(
                2, self.channels[end_layer - 1][1], 1)
        self.params: "OrderedDict[str, torch.Tensor]" = p

    @property
    def growing(self) -> bool:
        return self.curr_layer < len(self.channels) - 1

    def next_layer(self) -> bool:
        if not self.growing:
            return False
        self.curr_layer += 1
        p = self.params
        # the old head object is re-used as last_end_block[0] (aliased), generator.py:132-138
        p["_Generator__last_end_block.0.0.weight"] = p["_Generator__end_block.0.weight"]
        p["_Generator__last_end_block.0.0.bias"] = p["_Generator__end_block.0.bias"]
        w, b = _conv_init(2, self.channels[self.curr_layer][1], 1)
        p["_Generator__end_block.0.weight"], p["_Generator__end_block.0.bias"] = w, b
        # state_dict order of the reference: gen_blocks, end_block, last_end_block
        p.move_to_end("_Generator__last_end_block.0.0.weight")
        p.move_to_end("_Generator__last_end_block.0.0.bias")
        self.has_last = True
        return True

    def live_keys(self):
        """Parameters that receive a gradient at the current level."""
        keys = []
        for i in range(self.curr_layer + 1):
            for j in (0, 4):
                keys += [f"_Generator__gen_blocks.{i}.{j}.weight", f"_Generator__gen_blocks.{i}.{j}.bias"]
        keys += ["_Generator__end_block.0.weight", "_Generator__end_block.0.bias"]
        if self.has_last:
            keys += ["_Generator__last_end_block.0.0.weight", "_Generator__last_end_block.0.0.bias"]
        return keys


class DiscState:
    """Discriminator parameters keyed like the reference state_dict (discriminator.py:53-105)."""

    def __init__(self, start_layer: int = 7):
        assert 0 <= start_layer <= len(D_CHANNELS)
        self.curr_layer = start_layer
        p = OrderedDict()
        for i, (ci, co) in enumerate(D_CHANNELS):
            p[f"_Discriminator__conv_blocks.{i}.0.weight"], p[f"_Discriminator__conv_blocks.{i}.0.bias"] = \
                _conv_init(co, ci, 3)
            p[f"_Discriminator__conv_blocks.{i}.3.weight"], p[f"_Discriminator__conv_blocks.{i}.3.bias"] = \
                _conv_init(co, co, 3)
        p["_Discriminator__start_block.0.weight"], p["_Discriminator__start_block.0.bias"] = _conv_init(
            D_CHANNELS[start_layer][0], 2, 1)
        # discriminator.py:94-101: 160 * 512 // 2**9 * 512 // 2**9 == 160
        p["_Discriminator__clf.0.weight"], p["_Discriminator__clf.0.bias"] = _conv_init(1, 160, 0)
        self.has_last = False
        self.params: "OrderedDict[str, torch.Tensor]" = p

    @property
    def growing(self) -> bool:
        return self.curr_layer > 0

    def next_layer(self) -> bool:
        if not self.growing:
            return False
        self.curr_layer -= 1
        p = self.params
        p["_Discriminator__last_start_block.1.0.weight"] = p["_Discriminator__start_block.0.weight"]
        p["_Discriminator__last_start_block.1.0.bias"] = p["_Discriminator__start_block.0.bias"]
        w, b = _conv_init(D_CHANNELS[self.curr_layer][0], 2, 1)
        p["_Discriminator__start_block.0.weight"], p["_Discriminator__start_block.0.bias"] = w, b
        self.has_last = True
        return True

    def live_keys(self):
        keys = []
        for i in range(self.curr_layer, len(D_CHANNELS)):
            for j in (0, 3):
                keys += [f"_Discriminator__conv_blocks.{i}.{j}.weight", f"_Discriminator__conv_blocks.{i}.{j}.bias"]
        keys += ["_Discriminator__start_block.0.weight", "_Discriminator__start_block.0.bias"]
        if self.has_last:
            keys += ["_Discriminator__last_start_block.1.0.weight", "_Discriminator__last_start_block.1.0.bias"]
        keys += ["_Discriminator__clf.0.weight", "_Discriminator__clf.0.bias"]
        return keys


# --------------------------------------------------------------------------- forward
def pixel_norm(x: torch.Tensor) -> torch.Tensor:  # layers.py:11-17
    return x / torch.sqrt(x.pow(2.0).mean(dim=1, keepdim=True) + PN_EPS)


def gen_forward(p: Dict[str, torch.Tensor], curr_layer: int, has_last: bool, z: torch.Tensor,
                alpha: float) -> torch.Tensor:  # generator.py:106-126
    def block(i, x):
        pre = f"_Generator__gen_blocks.{i}."
        x = F.conv2d(x, p[pre + "0.weight"], p[pre + "0.bias"], padding=1)
        x = pixel_norm(F.leaky_relu(x, LRELU))
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        x = F.conv2d(x, p[pre + "4.weight"], p[pre + "4.bias"], padding=1)
        return pixel_norm(F.leaky_relu(x, LRELU))

    out = z
    for i in range(curr_layer):
        out = block(i, out)
    out_block = block(curr_layer, out)
    out_mp = torch.tanh(F.conv2d(out_block, p["_Generator__end_block.0.weight"], p["_Generator__end_block.0.bias"]))
    if has_last:
        old = torch.tanh(F.conv2d(out, p["_Generator__last_end_block.0.0.weight"],
                                  p["_Generator__last_end_block.0.0.bias"]))
        old = F.interpolate(old, scale_factor=2.0, mode="nearest")
        return alpha * out_mp + (1.0 - alpha) * old
    return out_mp


def disc_forward(p: Dict[str, torch.Tensor], curr_layer: int, has_last: bool, x: torch.Tensor,
                 alpha: float) -> torch.Tensor:  # discriminator.py:107-124
    def block(i, h):
        pre = f"_Discriminator__conv_blocks.{i}."
        h = F.leaky_relu(F.conv2d(h, p[pre + "0.weight"], p[pre + "0.bias"], padding=1), LRELU)
        h = F.avg_pool2d(h, 2, 2)
        return F.leaky_relu(F.conv2d(h, p[pre + "3.weight"], p[pre + "3.bias"], padding=1), LRELU)

    h = F.leaky_relu(F.conv2d(x, p["_Discriminator__start_block.0.weight"], p["_Discriminator__start_block.0.bias"]),
                     LRELU)
    h = block(curr_layer, h)
    if has_last:
        old = F.avg_pool2d(x, 2, 2)
        old = F.leaky_relu(F.conv2d(old, p["_Discriminator__last_start_block.1.0.weight"],
                                    p["_Discriminator__last_start_block.1.0.bias"]), LRELU)
        h = alpha * h + (1 - alpha) * old
    for i in range(curr_layer + 1, len(D_CHANNELS)):
        h = block(i, h)
    h = h.flatten(1, -1)
    return F.linear(h, p["_Discriminator__clf.0.weight"], p["_Discriminator__clf.0.bias"])


def gradient_penalty(dp, d_layer, d_has_last, x_real, x_gen, alpha, eps):  # discriminator.py:157-184
    n = x_real.shape[0]
    x_i = eps * x_real + (1 - eps) * x_gen
    if not x_i.requires_grad:
        x_i.requires_grad_(True)
    out = disc_forward(dp, d_layer, d_has_last, x_i, alpha)
    (g,) = torch.autograd.grad(out, x_i, grad_outputs=torch.ones_like(out), create_graph=True, retain_graph=True)
    gnorm = g.reshape(n, -1).norm(2, dim=1)
    return GP_FACTOR * ((gnorm - 1.0) ** 2.0).mean()


def w_disc_loss(y_real, y_fake):  # criterion.py:12-14
    return -(torch.mean(y_real) - torch.mean(y_fake))


def w_gen_loss(y_fake):  # criterion.py:17-18
    return -torch.mean(y_fake)


# --------------------------------------------------------------------------- steps
def _leafs(state, dtype):
    """Fresh leaf tensors (aliased keys share one leaf, as the reference's aliased Parameters do)."""
    by_id, out = {}, OrderedDict()
    for k, v in state.params.items():
        if id(v) not in by_id:
            by_id[id(v)] = v.detach().to(dtype).clone().requires_grad_(True)
        out[k] = by_id[id(v)]
    return out


def d_step(gs: GenState, ds: DiscState, x_real, z, eps, alpha, dtype=torch.float32, detach_fake=False):
    """train.py:152-174: returns forward values and the gradients .backward() leaves on both nets.

    detach_fake=False reproduces the reference exactly (G also receives gradients, later discarded)."""
    gp_, dp_ = _leafs(gs, dtype), _leafs(ds, dtype)
    x_real, z, eps = x_real.to(dtype), z.to(dtype), eps.to(dtype)
    x_fake = gen_forward(gp_, gs.curr_layer, gs.has_last, z, alpha)
    if detach_fake:
        x_fake = x_fake.detach()
    out_real = disc_forward(dp_, ds.curr_layer, ds.has_last, x_real, alpha)
    out_fake = disc_forward(dp_, ds.curr_layer, ds.has_last, x_fake, alpha)
    d_loss = w_disc_loss(out_real, out_fake)
    gp = gradient_penalty(dp_, ds.curr_layer, ds.has_last, x_real, x_fake, alpha, eps)
    (d_loss + gp).backward()
    return {
        "x_fake": x_fake.detach(), "out_real": out_real.detach(), "out_fake": out_fake.detach(),
        "disc_loss": d_loss.detach(), "grad_pen": gp.detach(),
        "d_grads": OrderedDict((k, dp_[k].grad.detach()) for k in ds.live_keys()),
        "g_grads": OrderedDict((k, gp_[k].grad.detach()) for k in gs.live_keys()
                               if gp_[k].grad is not None),
    }


def real_term_grads(ds: DiscState, x_real, alpha, dtype=torch.float64):
    """d mean(D(x_real)) / d theta: ONE of the summands of the critic gradient (criterion.py:12-14).  At a fresh init the real and
    the fake term nearly cancel in the deep blocks and in the classifier; the un-cancelled term is the scale fp32 round-off of such
    a residue has to be judged on."""
    dp_ = _leafs(ds, dtype)
    disc_forward(dp_, ds.curr_layer, ds.has_last, x_real.to(dtype), alpha).mean().backward()
    return OrderedDict((k, dp_[k].grad.detach()) for k in ds.live_keys())


def g_step(gs: GenState, ds: DiscState, z, alpha, dtype=torch.float32):
    """train.py:191-213."""
    gp_, dp_ = _leafs(gs, dtype), _leafs(ds, dtype)
    x_fake = gen_forward(gp_, gs.curr_layer, gs.has_last, z.to(dtype), alpha)
    out_fake = disc_forward(dp_, ds.curr_layer, ds.has_last, x_fake, alpha)
    loss = w_gen_loss(out_fake)
    loss.backward()
    return {
        "x_fake": x_fake.detach(), "out_fake": out_fake.detach(), "gen_loss": loss.detach(),
        "g_grads": OrderedDict((k, gp_[k].grad.detach()) for k in gs.live_keys()),
        "d_grads": OrderedDict((k, dp_[k].grad.detach()) for k in ds.live_keys()),
    }


def adam_update(param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, beta1=0.0, beta2=0.9, eps=1e-8):
    """One torch.optim.Adam step (defaults of train.py:64-70), restated from the published update rule:
    m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    exp_avg = beta1 * exp_avg + (1 - beta1) * grad
    exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = exp_avg_sq.sqrt() / math.sqrt(bc2) + eps
    return param - (lr / bc1) * exp_avg / denom, exp_avg, exp_avg_sq


def flops_per_image(rand_channels: int, level: int) -> Tuple[float, float]:
    """conv+linear MACs x2 of one G / one D forward at `level` (1.978 / 1.986 GFLOP at level 5)."""
    ch = g_channels(rand_channels)
    g = 0.0
    s = 2
    for i in range(level + 1):
        ci, co = ch[i]
        g += 2 * 9 * ci * ci * s * s
        s *= 2
        g += 2 * 9 * ci * co * s * s
    g += 2 * ch[level][1] * 2 * s * s
    if level > 0:
        g += 2 * ch[level - 1][1] * 2 * (s // 2) * (s // 2)
    dl = 7 - level
    d = 2 * 2 * D_CHANNELS[dl][0] * s * s
    if level > 0:
        d += 2 * 2 * D_CHANNELS[dl][1] * (s // 2) * (s // 2)
    t = s
    for i in range(dl, 9):
        ci, co = D_CHANNELS[i]
        d += 2 * 9 * ci * co * t * t
        t //= 2
        d += 2 * 9 * co * co * t * t
    d += 2 * 160
    return g, d


# --------------------------------------------------------------------------- the stateful loop (train.py:101-272)
class GrowerState:
    """utils.py:14-68 restated: sample counters -> (grow?, alpha).  `train_lengths` are cumulated (utils.py:39-43); growth when
    the cumulated length is STRICTLY below the samples seen (utils.py:52); alpha read before grow() (train.py:152 vs :258)."""

    def __init__(self, n_grow: int, fadein_lengths, train_lengths):
        assert len(fadein_lengths) == n_grow + 1 and len(train_lengths) == n_grow
        self.n_grow, self.fade = n_grow, list(fadein_lengths)
        self.train_l, acc = [], 0
        for t in train_lengths:
            acc += t
            self.train_l.append(acc)
        self.curr_grow = self.sample_idx = self.step_sample_idx = 0

    def grow(self, viewed: int) -> bool:
        self.sample_idx += viewed
        self.step_sample_idx += viewed
        if self.curr_grow >= self.n_grow:
            return False
        if self.train_l[self.curr_grow] < self.sample_idx:
            self.step_sample_idx = 0
            self.curr_grow += 1
            return True
        return False

    @property
    def alpha(self) -> float:
        return min(1.0, (1.0 + self.step_sample_idx) / self.fade[self.curr_grow])


class AdamState:
    """torch.optim.Adam as train.py:64-70,175,214,262-272 uses it: state per parameter OBJECT (aliased keys share it), created at
    the first step in which the parameter has a gradient, `step` counted per parameter, parameters without gradient skipped.
    Param groups only matter through that per-parameter step count (all groups share lr / betas)."""

    def __init__(self, lr=1e-3, betas=(0.0, 0.9), eps=1e-8):
        self.lr, self.betas, self.eps = lr, betas, eps
        self.state: Dict[int, dict] = {}

    def step(self, params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor]) -> None:
        done = set()
        for k, g in grads.items():
            p = params[k]
            if id(p) in done:
                continue
            done.add(id(p))
            # "param" keeps the tensor alive (as torch's optimizer does): a dropped head's id() must not be recycled
            st = self.state.setdefault(id(p), {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p),
                                               "param": p})
            st["step"] += 1
            new, st["exp_avg"], st["exp_avg_sq"] = adam_update(p, g.to(p.dtype), st["exp_avg"], st["exp_avg_sq"], st["step"],
                                                               self.lr, self.betas[0], self.betas[1], self.eps)
            p.copy_(new)

    def of(self, p: torch.Tensor) -> Optional[dict]:
        return self.state.get(id(p))


class Trajectory:
    """The body of the reference's training loop (train.py:135-272) on oracle states, one call of `iteration()` per loader batch:
    D step every iteration (G not detached), G step when iter_idx % 5 == 0, Adam on the stepped net only, then Grower.grow(batch)
    and -- when it fires and the nets still grow -- next_layer() on both (fresh head/stem drawn from the global RNG, seeded by the
    caller through `growth_seed`)."""

    def __init__(self, rand_channels: int, grower: GrowerState, dtype=torch.float32, wscale: float = 1.0):
        self.gs, self.ds = GenState(rand_channels), DiscState(7)
        self.dtype, self.grower = dtype, grower
        for st in (self.gs, self.ds):
            for k in list(st.params.keys()):
                v = st.params[k]
                st.params[k] = (v * wscale if k.endswith("weight") else v).to(dtype)
        self.opt_g, self.opt_d = AdamState(), AdamState()
        self.iter_idx = 0

    def _retype_new(self, st, keys):
        for k in keys:
            st.params[k] = st.params[k].to(self.dtype)

    def iteration(self, x_real, z, eps, z2=None, growth_seed: Optional[int] = None, defer_growth: bool = False) -> dict:
        """`defer_growth`: stop in front of train.py:258 (state inspection by the lock-step test); the caller then runs
        `end_of_iteration(batch, growth_seed)` itself."""
        gs, ds, dt = self.gs, self.ds, self.dtype
        alpha = self.grower.alpha
        rec = {"iter": self.iter_idx, "level": gs.curr_layer, "alpha": alpha}
        d = d_step(gs, ds, x_real, z, eps, alpha, dtype=dt)
        self.opt_d.step(ds.params, d["d_grads"])
        rec.update(disc_loss=float(d["disc_loss"]), grad_pen=float(d["grad_pen"]), out_real=float(d["out_real"].mean()),
                   out_fake=float(d["out_fake"].mean()), d_grads=d["d_grads"])
        if self.iter_idx % 5 == 0:
            g = g_step(gs, ds, z2, alpha, dtype=dt)
            self.opt_g.step(gs.params, g["g_grads"])
            rec.update(gen_loss=float(g["gen_loss"]), g_grads=g["g_grads"])
        self.iter_idx += 1
        if not defer_growth:
            rec["grew"] = self.end_of_iteration(x_real.shape[0], growth_seed)
        return rec

    def end_of_iteration(self, batch: int, growth_seed: Optional[int] = None) -> bool:
        """train.py:258-272."""
        gs, ds = self.gs, self.ds
        if self.grower.grow(batch) and gs.growing:
            if growth_seed is not None:
                torch.manual_seed(growth_seed)
            gs.next_layer()
            ds.next_layer()
            self._retype_new(gs, ["_Generator__end_block.0.weight", "_Generator__end_block.0.bias"])
            self._retype_new(ds, ["_Discriminator__start_block.0.weight", "_Discriminator__start_block.0.bias"])
            return True
        return False
