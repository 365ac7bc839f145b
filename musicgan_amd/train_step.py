"""One discriminator update + one generator update of the WGAN-GP ProGAN loop, restating the body of
/root/reference/music_gan/train.py:135-221 on the MI355X modules without its wasted work:

  * the D step runs G under no_grad (the reference back-propagates into G and then discards those gradients, train.py:152,
    170-174,209 -- D's gradients are identical either way, tests/test_oracle_golden.py::test_oracle_detached...);
  * the G step freezes D's parameters, so only D's data gradient is evaluated (the reference also computes and discards D's
    weight gradients, train.py:209-214);
  * no .item() host syncs: losses come back as device tensors.

With torch.distributed initialised (one process per GPU, backend "nccl" = RCCL) the per-rank gradients are summed with one
flat all-reduce per network on a side stream, the fused Adam runs behind it on that stream, and the main stream meanwhile
runs the next forward pass that does not depend on the updated weights (G forward of the G step; D(x_real) of the next D step).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import networks
from .dist import GradBucket, is_distributed


class ProGANStepper:
    def __init__(self, gen, disc, optim_gen, optim_disc, rand_channels: int, height: int = 2, width: int = 2,
                 fused_d_step: bool = True, noise: Optional[torch.Generator] = None):
        """`noise`: device generator the latents and the penalty's epsilon are drawn from when they are not injected (None: the
        default generator, as the reference does).  Data-parallel runs give every rank its own stream."""
        self.gen, self.disc = gen, disc
        self.noise = noise
        self.optim_gen, self.optim_disc = optim_gen, optim_disc
        self.rand_channels, self.h, self.w = rand_channels, height, width
        self.fused_d_step = fused_d_step
        self.dp = is_distributed()
        self.bucket_d = GradBucket()
        self.bucket_g = GradBucket()
        if hasattr(optim_gen, "grad_scale"):
            optim_gen.grad_scale = self.bucket_g.grad_scale
            optim_disc.grad_scale = self.bucket_d.grad_scale

    def _latent(self, n: int, device, generator=None) -> torch.Tensor:
        return torch.randn(n, self.rand_channels, self.h, self.w, device=device,
                           generator=generator if generator is not None else self.noise)

    def _update(self, bucket: GradBucket, net, optim) -> None:
        if self.dp:
            bucket.launch(net.parameters())
            with torch.cuda.stream(bucket.stream()):
                optim.step()
        else:
            optim.step()

    def d_step(self, x_real: torch.Tensor, alpha: float, z: Optional[torch.Tensor] = None,
               eps: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        n = x_real.shape[0]
        if z is None:
            z = self._latent(n, x_real.device)
        if self.fused_d_step:
            return self._d_step_fused(x_real, alpha, z, eps)
        if self.dp:
            self.bucket_d.wait()  # D weights final (Adam of the previous D step)
        out_real = self.disc(x_real, alpha)  # independent of G: overlaps the G exchange of the previous G step
        if self.dp:
            self.bucket_g.wait()  # G weights final
        with torch.no_grad():
            x_fake = self.gen(z, alpha)
        out_fake = self.disc(x_fake, alpha)
        disc_loss = networks.wasserstein_discriminator_loss(out_real, out_fake)
        if eps is None and self.noise is not None:
            eps = torch.rand(n, 1, 1, 1, device=x_real.device, generator=self.noise)
        if eps is None:
            grad_pen = self.disc.gradient_penalty(x_real, x_fake, alpha)
        else:
            grad_pen = self.disc.gradient_penalty_with_eps(x_real, x_fake, alpha, eps)
        self.gen.zero_grad()
        self.disc.zero_grad()
        (disc_loss + grad_pen).backward()
        self._update(self.bucket_d, self.disc, self.optim_disc)
        return {"disc_loss": disc_loss.detach(), "grad_pen": grad_pen.detach(),
                "out_real_mean": out_real.detach().mean(), "out_fake_mean": out_fake.detach().mean()}

    def _d_step_fused(self, x_real, alpha, z, eps) -> Dict[str, torch.Tensor]:
        """Same update as the module path above through `engine.disc_step_fused`: one batched critic pass over
        [real | fake | interpolated] instead of three, one weight-gradient launch per layer."""
        from .networks import engine
        n = x_real.shape[0]
        dev = x_real.device
        if eps is None:
            eps = torch.rand(n, 1, 1, 1, device=dev, generator=self.noise)
        if self.dp:
            self.bucket_d.wait()
            self.bucket_g.wait()
        xcat = torch.empty((3 * n,) + tuple(x_real.shape[1:]), dtype=torch.float32, device=dev)
        xcat[:n].copy_(x_real)
        with torch.no_grad():
            engine.gen_forward(self.gen._weights(), z.contiguous(), alpha, self.gen._pack_cache, save=False,
                               out=xcat[n:2 * n])
            W = self.disc._weights()
            sink = engine.GradSink()
            disc_loss, grad_pen, out = engine.disc_step_fused(W, xcat[:n], xcat[n:2 * n], eps, alpha,
                                                              self.disc._pack_cache, sink, xcat=xcat)
        self.gen.zero_grad()
        self.disc.zero_grad()
        for p in W.tensors():
            p.grad = sink.get(p)
        self._update(self.bucket_d, self.disc, self.optim_disc)
        return {"disc_loss": disc_loss, "grad_pen": grad_pen, "out_real_mean": out[:n].mean(),
                "out_fake_mean": out[n:2 * n].mean()}

    def g_step(self, batch_size: int, alpha: float, device, z: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        if z is None:
            z = self._latent(batch_size, device)
        if self.dp:
            self.bucket_g.wait()
        x_fake = self.gen(z, alpha)  # overlaps D's gradient exchange + Adam on the side stream
        if self.dp:
            self.bucket_d.wait()
        d_params = list(self.disc.parameters())
        for p in d_params:
            p.requires_grad_(False)
        try:
            out_fake = self.disc(x_fake, alpha)
            gen_loss = networks.wasserstein_generator_loss(out_fake)
            self.gen.zero_grad()
            self.disc.zero_grad()
            gen_loss.backward()
        finally:
            for p in d_params:
                p.requires_grad_(True)
        self._update(self.bucket_g, self.gen, self.optim_gen)
        return {"gen_loss": gen_loss.detach(), "out_fake_mean": out_fake.detach().mean()}

    def finish(self) -> None:
        """Join the side streams (call before reading weights / checkpointing)."""
        if self.dp:
            self.bucket_d.wait()
            self.bucket_g.wait()
