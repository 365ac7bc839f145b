"""Data-parallel gradient exchange for the ProGAN step: one process per GPU, RCCL (torch.distributed backend "nccl") over
xGMI, one flat bucket per network, launched on a side HIP stream so the exchange overlaps the next forward pass.

The reference is single-GPU (no torch.distributed anywhere); the contract here is "N ranks with per-rank batch B
produce the gradients of one rank with batch N*B" -- every loss is a batch mean (criterion.py:12-18) and the penalty is a
mean of per-sample terms (discriminator.py:178-180), so averaging the per-rank gradients is exact.

Payloads at level 5 are 6.4 MB (D) and 3.4 MB (G): latency-bound on 7 x ~153 GB/s xGMI links, so the whole network goes in
ONE all-reduce (no per-layer buckets) and the 1/world scaling is folded into the fused Adam kernel (`grad_scale`).
Works unchanged on CPU tensors with the gloo backend (used by the world_size-2 tests).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    """True when gradients must be exchanged.  MG_FORCE_DP=1 takes the data-parallel code path (side stream, flat bucket,
    all-reduce, Adam behind it) even with a single rank -- used to rehearse that path on a one-GPU box."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("MG_FORCE_DP", "0") == "1"


class GradBucket:
    """Flattens the live gradients of one network into a single buffer, all-reduces it (sum) and re-points each
    `param.grad` at its slice of the reduced buffer (no copy back)."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group) if is_distributed() else 1
        self._flat: Optional[torch.Tensor] = None
        self._own: Optional[torch.Tensor] = None   # flat buffer handed out by flat_sink()
        self._own_key = None
        self._work = None
        self._stream: Optional[torch.cuda.Stream] = None
        self._done: Optional[torch.cuda.Event] = None
        self.copied = False

    @property
    def grad_scale(self) -> float:
        """Factor turning the summed gradient into the global-batch mean gradient (fold into FusedAdam.grad_scale)."""
        return 1.0 / self.world

    def _side_stream(self, device) -> torch.cuda.Stream:
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def flat_sink(self, tensors: List[torch.Tensor]):
        """(flat, layout) for engine.GradSink: one buffer the gradients of `tensors` (unique, in this order) are written into
        directly, reused from step to step (the caller waits for the previous exchange before it writes again)."""
        uniq, seen = [], set()
        for p in tensors:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        total = sum(p.numel() for p in uniq)
        key = tuple(id(p) for p in uniq)
        if self._own is None or self._own_key != key:
            self._own = torch.empty(total, dtype=torch.float32, device=uniq[0].device)
            self._own_key = key
        layout, off = {}, 0
        for p in uniq:
            layout[id(p)] = (off, p.numel())
            off += p.numel()
        return self._own, layout

    def launch(self, params: Iterable[torch.nn.Parameter]) -> None:
        """Start the exchange of every non-None .grad.  On GPU it runs on a side stream ordered after the current stream's
        work so far; the caller continues issuing independent work and later calls wait()."""
        plist: List[torch.nn.Parameter] = [p for p in params if p.grad is not None]
        if not plist:
            return
        dev = plist[0].grad.device
        if dev.type == "cuda":
            main = torch.cuda.current_stream(dev)
            side = self._side_stream(dev)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                for p in plist:
                    # the gradients were allocated on the main stream and are dropped (re-pointed) below while the side
                    # stream may not have read them yet: tell the caching allocator they are in use there
                    p.grad.record_stream(side)
                flat = self._as_own_flat(plist)
                if flat is None:
                    flat = torch.cat([p.grad.reshape(-1) for p in plist])
                if is_distributed():
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                self._done = torch.cuda.Event()
                self._done.record(side)
        else:
            flat = torch.cat([p.grad.reshape(-1) for p in plist])
            if self.world > 1:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        if flat is not self._own:  # (slices of the own buffer already ARE the gradients, in the layout's order)
            off = 0
            for p in plist:
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n
        self._flat = flat
        self.copied = flat is not self._own  # (tests: the data-parallel steppers must never flatten by copy)

    def _as_own_flat(self, plist) -> Optional[torch.Tensor]:
        """The flat_sink() buffer if the gradients of `plist` are exactly its slices (then nothing is copied).  In ANY order:
        `launch` walks `net.parameters()` (registration order) while `flat_sink` lays the buffer out in the engine's order
        (`GenWeights.tensors()`: blocks, head, old head) -- a consecutive-order test fails for the generator and every update then
        pays a torch.cat on the side stream."""
        if self._own is None:
            return None
        base, size = self._own.data_ptr(), self._own.numel()
        spans = []
        for p in plist:
            off = p.grad.data_ptr() - base
            if off < 0 or off % 4 or not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                return None
            spans.append((off // 4, p.numel()))
        spans.sort()
        end = 0
        for off, n in spans:  # disjoint slices that tile the buffer
            if off != end:
                return None
            end = off + n
        return self._own if end == size else None

    def stream(self) -> Optional[torch.cuda.Stream]:
        return self._stream

    def wait(self) -> None:
        """Make the current stream wait for the exchange (and anything queued behind it on the side stream)."""
        if self._stream is not None:
            ev = torch.cuda.Event()
            ev.record(self._stream)
            torch.cuda.current_stream().wait_event(ev)


def check_world(expected: int, device=None) -> int:
    """A sum of ones over the data path's backend: the ranks that really take part.  Raises SystemExit(3) when that is not
    `expected` (WORLD_SIZE) -- a rank that came up on the wrong device, or alone, must not train a model nobody asked for."""
    seen = 1
    if dist.is_available() and dist.is_initialized():
        ones = torch.ones(1, dtype=torch.float32, device=device if device is not None else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        seen = int(round(float(ones.item())))
    if seen != expected:
        raise SystemExit(f"musicgan_amd: WORLD_SIZE={expected} but {seen} rank(s) take part in the gradient exchange")
    return seen


def broadcast_parameters(modules: Iterable[torch.nn.Module], src: int = 0) -> None:
    """Identical replicas at start (same seed already gives this; the broadcast makes it independent of host RNG state)."""
    if not is_distributed():
        return
    for m in modules:
        for p in m.parameters():
            dist.broadcast(p.data, src=src)
