"""One discriminator update + one generator update of the WGAN-GP ProGAN loop, restating the body of
/root/reference/music_gan/train.py:135-221 on the MI355X modules without its wasted work:

  * the D step runs G under no_grad (the reference back-propagates into G and then discards those gradients, train.py:152,
    170-174,209 -- D's gradients are identical either way, tests/test_oracle_golden.py::test_oracle_detached...);
  * the G step freezes D's parameters, so only D's data gradient is evaluated (the reference also computes and discards D's
    weight gradients, train.py:209-214);
  * no .item() host syncs: losses come back as device tensors.

Single-process runs replay each update as a HIP graph: after two eager calls with the same (level, batch, alpha) the update --
forward, backward, weight re-packing and the fused Adam with its device-side step counter -- is captured once and replayed, so
the ~150 launches of an update cost the host one call (the small maps at the ends of both networks are otherwise launch-bound:
15 us of Python + ctypes per launch against 5-20 us of GPU time).  The fade-in coefficient alpha, which changes every iteration
of a fade-in, is read by the captured kernels from device memory, so one graph per (level, batch) serves the whole level.
`MG_GRAPHS=0` disables it.

With torch.distributed initialised (one process per GPU, backend "nccl" = RCCL) the per-rank gradients are summed with one
flat all-reduce per network on a side stream, the fused Adam runs behind it on that stream, and the main stream meanwhile
runs the next forward pass that does not depend on the updated weights: every update starts with a forward pass through G (the
fake batch of a critic update, the generator update's own forward), which needs G's weights only, so the critic's exchange + Adam
of the previous update overlap it -- eagerly by where the waits sit, under graph replay by capturing each update as two graphs
with the wait between them.
"""
from __future__ import annotations

import os
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
        self.use_graphs = fused_d_step and os.environ.get("MG_GRAPHS", "1") != "0"
        self._graphs: Dict[tuple, dict] = {}
        from . import ops
        self._defer_d, self._defer_g = ops.WgradDefer(), ops.WgradDefer()  # one-launch weight-gradient reductions per sweep
        self._fade: Optional[torch.Tensor] = None   # [alpha, 1 - alpha] read by the captured fade-in kernels
        self._fade_value: Optional[float] = None
        self.bucket_d = GradBucket()
        self.bucket_g = GradBucket()
        if hasattr(optim_gen, "grad_scale"):
            optim_gen.grad_scale = self.bucket_g.grad_scale
            optim_disc.grad_scale = self.bucket_d.grad_scale

    def _latent(self, n: int, device, generator=None) -> torch.Tensor:
        return torch.randn(n, self.rand_channels, self.h, self.w, device=device,
                           generator=generator if generator is not None else self.noise)

    def _latent_into(self, out: torch.Tensor) -> None:
        torch.randn(out.shape, device=out.device, generator=self.noise, out=out)

    def _eps_into(self, out: torch.Tensor) -> None:
        torch.rand(out.shape, device=out.device, generator=self.noise, out=out)

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
        if self.fused_d_step and self.use_graphs:
            # noise that is not injected is drawn straight into the replayed graph's input buffers (same generator stream, two
            # copy launches fewer per update: at levels 3-4 an update is ~100 launches of 5-20 us)
            return self._graphed("D", (x_real, z, eps), alpha,
                                 draw=(None, lambda out: self._latent_into(out), lambda out: self._eps_into(out)),
                                 shapes=(tuple(x_real.shape), (n, self.rand_channels, self.h, self.w), (n, 1, 1, 1)))
        if z is None:
            z = self._latent(n, x_real.device)
        if self.fused_d_step:
            if eps is None:
                eps = torch.rand(n, 1, 1, 1, device=x_real.device, generator=self.noise)
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

    def _d_step_fused(self, x_real, alpha, z, eps, update_in_line: bool = True, split=None) -> Dict[str, torch.Tensor]:
        """Same update as the module path above through `engine.disc_step_fused`: one batched critic pass over
        [real | fake | interpolated] instead of three, one weight-gradient launch per layer.
        Data-parallel: the update has two halves -- the generator's forward pass for the fake batch, which needs G's weights only,
        and the critic's passes -- so the critic's gradient exchange + Adam of the PREVIOUS update (side stream) overlap the
        first half: eagerly the waits sit exactly there; under graph capture `split()` is called between the halves (it ends the
        first graph and begins the second, `_graphed`)."""
        from .networks import engine
        n = x_real.shape[0]
        dev = x_real.device
        if eps is None:
            eps = torch.rand(n, 1, 1, 1, device=dev, generator=self.noise)
        eager_dp = self.dp and update_in_line
        force = torch.cuda.is_current_stream_capturing()
        if eager_dp:
            self.bucket_g.wait()  # G's weights final; the critic's exchange + Adam may still be running
        xcat = torch.empty((3 * n,) + tuple(x_real.shape[1:]), dtype=torch.float32, device=dev)
        xcat[:n].copy_(x_real)
        self.gen._pack_cache.refresh(force)
        with torch.no_grad():
            engine.gen_forward(self.gen._weights(), z.contiguous(), alpha, self.gen._pack_cache, save=False,
                               out=xcat[n:2 * n])
        if eager_dp:
            self.bucket_d.wait()  # the critic's weights final
        if split is not None:
            split()
        self.disc._pack_cache.refresh(torch.cuda.is_current_stream_capturing())
        with torch.no_grad():
            W = self.disc._weights()
            sink = engine.GradSink(*self.bucket_d.flat_sink(W.tensors())) if self.dp else engine.GradSink()
            disc_loss, grad_pen, out, stats = engine.disc_step_fused(W, xcat[:n], xcat[n:2 * n], eps, alpha,
                                                                     self.disc._pack_cache, sink, xcat=xcat,
                                                                     defer=self._defer_d)
        self.gen.zero_grad()
        self.disc.zero_grad()
        for p in W.tensors():
            p.grad = sink.get(p)
        if update_in_line:
            self._update(self.bucket_d, self.disc, self.optim_disc)
        return {"disc_loss": disc_loss, "grad_pen": grad_pen, "out_real_mean": stats[0], "out_fake_mean": stats[1]}

    def g_step(self, batch_size: int, alpha: float, device, z: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        if self.use_graphs:
            return self._graphed("G", (z,), alpha, draw=(lambda out: self._latent_into(out),),
                                 shapes=((batch_size, self.rand_channels, self.h, self.w),), device=device)
        if z is None:
            z = self._latent(batch_size, device)
        if self.fused_d_step:
            return self._g_step_fused(z, alpha)
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

    def _g_step_fused(self, z, alpha, update_in_line: bool = True, split=None) -> Dict[str, torch.Tensor]:
        """The generator update through `engine.gen_step_fused` (no autograd graph, no critic weight gradients).  Data-parallel:
        G's forward runs while the critic's gradient exchange + Adam are still on the side stream; the critic's weights (and their
        packed layouts) are only touched behind `bucket_d.wait()` -- eagerly through the waits below, under graph capture
        through `split()` between the two graphs of the update (`_graphed`)."""
        from .networks import engine
        before_disc = None
        if self.dp and not update_in_line:
            self.gen._pack_cache.refresh(torch.cuda.is_current_stream_capturing())

            def before_disc():
                if split is not None:
                    split()
                self.disc._pack_cache.refresh(torch.cuda.is_current_stream_capturing())
        elif self.dp:
            self.bucket_g.wait()
            self.gen._pack_cache.refresh()

            def before_disc():
                self.bucket_d.wait()
                self.disc._pack_cache.refresh()
        else:
            self._refresh_packs()
        with torch.no_grad():
            Wg, Wd = self.gen._weights(), self.disc._weights()
            sink = engine.GradSink(*self.bucket_g.flat_sink(Wg.tensors())) if self.dp else engine.GradSink()
            gen_loss, out, stats = engine.gen_step_fused(Wg, Wd, z, alpha, self.gen._pack_cache, self.disc._pack_cache, sink,
                                                         before_disc=before_disc, defer=self._defer_g)
        self.gen.zero_grad()
        self.disc.zero_grad()
        for p in Wg.tensors():
            p.grad = sink.get(p)
        if update_in_line:
            self._update(self.bucket_g, self.gen, self.optim_gen)
        return {"gen_loss": gen_loss, "out_fake_mean": stats[0]}

    def _refresh_packs(self) -> None:
        """Every packed weight layout either network has used so far, re-packed in one launch per network if its weight changed
        (all of them while a graph is being captured: a replay must not depend on what was fresh at capture time)."""
        force = torch.cuda.is_current_stream_capturing()
        self.gen._pack_cache.refresh(force)
        self.disc._pack_cache.refresh(force)

    # ------------------------------------------------------------------ HIP-graph replay of whole updates
    _WARM_CALLS = 2

    def _graphed(self, kind: str, inputs, alpha: float, draw=None, shapes=None, device=None) -> Dict[str, torch.Tensor]:
        """`inputs` may hold None where the caller did not inject noise: `draw[i](buffer)` then fills it -- into a fresh tensor while
        the update still runs eagerly, straight into the graph's static input afterwards."""
        if any(t is None for t in inputs):
            dev = device if device is not None else next(t.device for t in inputs if t is not None)
            have = self._graphs_entry_inputs(kind, shapes)
            filled = []
            for i, t in enumerate(inputs):
                if t is None:
                    t = have[i] if have is not None else torch.empty(shapes[i], dtype=torch.float32, device=dev)
                    draw[i](t)
                filled.append(t)
            inputs = tuple(filled)
        net, other = (self.disc, self.gen) if kind == "D" else (self.gen, self.disc)
        opt_sig = (self.optim_disc if kind == "D" else self.optim_gen)
        opt_sig = opt_sig.capture_signature() if hasattr(opt_sig, "capture_signature") else ()
        key = (kind, self.gen.curr_layer, tuple(tuple(t.shape) for t in inputs), tuple(id(p) for p in net.parameters()), opt_sig)

        # Data-parallel: the graph ends with the gradients in the bucket's flat buffer; the exchange (RCCL, side stream) and Adam
        # behind it stay outside, so a replay is bracketed by "join the side streams" and "launch the exchange".
        in_line = not self.dp

        def run(fade, *a, captured=False, split=None):
            upd = in_line or not captured
            if kind == "D":
                return self._d_step_fused(a[0], fade, a[1], a[2], upd, split)
            return self._g_step_fused(a[0], fade, upd, split)
        ent = self._graphs.get(key)
        if ent is None:
            # growth or a new batch shape: graphs of other levels (and their private memory pools, GBs at the large levels)
            # are never replayed again
            # -- and neither are graphs of this update captured under other optimizer hyper-parameters (a scheduler that moves
            # lr would otherwise keep one private activation pool alive per value)
            for k in [k for k in self._graphs if k[1] != key[1] or (k[:4] == key[:4] and k[4] != key[4])]:
                del self._graphs[k]
            if len(self._graphs) > 8:
                self._graphs.clear()
            ent = self._graphs[key] = {"calls": 0}
        if "graph" not in ent:
            ent["calls"] += 1
            if ent["calls"] <= self._WARM_CALLS or ent.get("eager"):
                return run(alpha, *inputs)
            opt = self.optim_disc if kind == "D" else self.optim_gen
            mirrors = [(st, st["step"].clone()) for st in opt.state.values() if "step" in st]
            caller_stream = torch.cuda.current_stream()
            try:
                self._capture(ent, kind, net, other, inputs, alpha, in_line, run)
            except (RuntimeError, torch.cuda.OutOfMemoryError) as e:
                # A capture that HIP itself invalidated (an illegal call while capturing) leaves torch.cuda.graph's exit half done:
                # the current stream is still the capture stream, which stays unusable.  Back to the caller's stream first.
                torch.cuda.set_stream(caller_stream)
                # e.g. the graph's private activation pool does not fit next to the other update's: run this update eagerly
                # from now on rather than fail the training run (the eager path is the same kernels, launched one by one)
                import warnings
                warnings.warn(f"HIP-graph capture of the {kind} update failed ({e}); it runs eagerly from here on")
                for k in ("graph", "first", "inputs", "out", "grads", "names"):
                    ent.pop(k, None)
                ent["eager"] = True
                torch.cuda.synchronize()
                # host state the aborted capture already changed: the pack caches recorded every layout as fresh although the
                # pack kernels were only captured, never run; deferred weight-gradient reductions point into the discarded pool
                self._defer_d.reset()
                self._defer_g.reset()
                self.gen._pack_cache.invalidate()
                self.disc._pack_cache.invalidate()
                for st, v in mirrors:  # a captured-but-never-run optimizer step may have advanced the host step mirrors
                    st["step"].copy_(v)
                return run(alpha, *inputs)
        for dst, src in zip(ent["inputs"], inputs):
            if dst is not src:  # (noise drawn above already sits in the static buffer)
                dst.copy_(src)
        if self._fade_value != float(alpha):
            self._fade[0:1].fill_(float(alpha))
            self._fade[1:2].fill_(1.0 - float(alpha))
            self._fade_value = float(alpha)
        if self.dp:
            # G's weights final -> [generator forward] -> the critic's weights final -> [the rest]: the critic's exchange + Adam of
            # the previous update (RCCL + one kernel on the side stream) run under the generator's forward pass
            self.bucket_g.wait()
            ent["first"].replay()
            self.bucket_d.wait()
        ent["graph"].replay()
        self.gen.zero_grad()
        self.disc.zero_grad()
        for p, g in ent["grads"]:
            p.grad = g
        if in_line:
            (self.optim_disc if kind == "D" else self.optim_gen).note_replay([p for p, _ in ent["grads"]])
        else:
            self._update(self.bucket_d if kind == "D" else self.bucket_g, net, self.optim_disc if kind == "D" else self.optim_gen)
        out = ent["out"].clone()
        return {k: out[i] for i, k in enumerate(ent["names"])}

    def _graphs_entry_inputs(self, kind: str, shapes):
        """The static input buffers of the captured graph this call will replay, if it exists (same key as `_graphed` builds)."""
        net = self.disc if kind == "D" else self.gen
        opt = self.optim_disc if kind == "D" else self.optim_gen
        sig = opt.capture_signature() if hasattr(opt, "capture_signature") else ()
        ent = self._graphs.get((kind, self.gen.curr_layer, tuple(tuple(s) for s in shapes), tuple(id(p) for p in net.parameters()), sig))
        return ent["inputs"] if ent is not None and "graph" in ent else None

    def _capture(self, ent, kind, net, other, inputs, alpha, in_line, run) -> None:
        # capture: static copies of the inputs, every weight form re-packed inside the graph (caches emptied first), the
        # gradients and the four scalars the caller reads live in the graph's pool
        ent["inputs"] = [t.detach().clone().contiguous() for t in inputs]
        if self.dp:
            self.finish()
        net.zero_grad()
        other.zero_grad()
        torch.cuda.synchronize()
        # the fade-in coefficients live in device memory inside the graph: alpha changes every iteration of a fade-in
        if self._fade is None:
            self._fade = torch.zeros(2, dtype=torch.float32, device=inputs[0].device)
        from .networks.engine import FadeIn
        graph = torch.cuda.CUDAGraph()
        if in_line:
            # thread_local: loader threads (pinned-memory staging, uploads on their own stream) keep working during the capture
            # (a capture stream of its own: torch's shared default one would stay broken after an invalidated capture)
            with torch.cuda.graph(graph, stream=torch.cuda.Stream(device=inputs[0].device), capture_error_mode="thread_local"):
                m = run(FadeIn(alpha, dev=self._fade), *ent["inputs"], captured=True)
                ent["names"] = list(m.keys())
                ent["out"] = torch.stack([m[k].reshape(()) for k in ent["names"]])
        else:
            # data-parallel: TWO graphs sharing one memory pool -- [generator forward] | [everything that reads the critic's
            # weights] -- so that a replay can wait for the critic's exchange + Adam between them instead of in front
            first, pool = torch.cuda.CUDAGraph(), torch.cuda.graph_pool_handle()
            cap = torch.cuda.Stream(device=inputs[0].device)
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap):
                first.capture_begin(pool, capture_error_mode="thread_local")
                state = {"open": first}

                def split():
                    first.capture_end()
                    graph.capture_begin(pool, capture_error_mode="thread_local")
                    state["open"] = graph
                try:
                    m = run(FadeIn(alpha, dev=self._fade), *ent["inputs"], captured=True, split=split)
                    assert state["open"] is graph, "the update never reached its split point"
                    ent["names"] = list(m.keys())
                    ent["out"] = torch.stack([m[k].reshape(()) for k in ent["names"]])
                finally:
                    state["open"].capture_end()
            torch.cuda.current_stream().wait_stream(cap)
            ent["first"] = first
        ent["graph"] = graph
        ent["grads"] = [(p, p.grad) for p in net.parameters() if p.grad is not None]
        opt = self.optim_disc if kind == "D" else self.optim_gen
        if in_line:
            for p, _ in ent["grads"]:  # the captured optimizer step advanced the host mirrors, but nothing ran yet
                opt.state[p]["step"] -= 1
        # the captured run itself did not execute: fall through to the first replay with the caller's inputs

    def finish(self) -> None:
        """Join the side streams (call before reading weights / checkpointing)."""
        if self.dp:
            self.bucket_d.wait()
            self.bucket_g.wait()
