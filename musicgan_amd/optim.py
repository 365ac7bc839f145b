"""Fused multi-tensor Adam on the HIP kernel `mg_adam_step_dev` -- one launch (+ a one-wave counter tick) per optimizer step.

Semantics and state layout are torch.optim.Adam's (amsgrad=False, weight_decay=0, maximize=False), which is what the
reference uses (/root/reference/music_gan/train.py:64-70,175,214,262-272): per-parameter `step`, `exp_avg`, `exp_avg_sq`;
parameters whose .grad is None are skipped (their step count does not advance); param groups added later start at step 0.
`state_dict()` is loadable by torch.optim.Adam and vice versa.

The step count the kernel reads lives in device memory (`step_dev`, int32) and the bias corrections are formed on the device,
so a captured HIP graph of the step stays valid from one replay to the next; `state["step"]` is the host mirror torch's format
wants (advanced here on every eager step, and by `note_replay()` after a graph replay).
"""
from __future__ import annotations

import ctypes
from typing import Iterable, List

import torch

from . import _lib
from ._lib import AdamTensorDev, check


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self.grad_scale = 1.0  # multiplied into every gradient inside the kernel (data-parallel averaging)
        self._state_epoch = 0  # bumped whenever the state tensors are replaced (load_state_dict)

    def capture_signature(self) -> tuple:
        """Everything a captured HIP graph of `step()` has baked in besides the parameters themselves: the hyper-parameters travel
        as launch scalars and the moment / step-counter tensors by address.  ProGANStepper keys its graphs on this, so a changed
        `param_group['lr']` (a scheduler) or a `load_state_dict` leads to a fresh capture instead of being silently ignored."""
        return (self._state_epoch, float(self.grad_scale),
                tuple((float(g["lr"]), tuple(float(b) for b in g["betas"]), float(g["eps"])) for g in self.param_groups))

    def _init_state(self, p: torch.Tensor) -> dict:
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        if "step_dev" not in st:
            st["step_dev"] = torch.full((), int(st["step"]), dtype=torch.int32, device=p.device)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr, eps = group["lr"], group["eps"]
            recs: List[AdamTensorDev] = []
            touched = []
            device = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise _lib.MusicGanHipError("FusedAdam needs parameters on a ROCm GPU (no CPU fallback)")
                g = p.grad
                if not g.is_contiguous():
                    g = g.contiguous()
                    p.grad = g
                st = self._init_state(p)
                st["step"] += 1
                recs.append(AdamTensorDev(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                          p.numel(), st["step_dev"].data_ptr()))
                touched.append(p)
                device = p.device
            if not recs:
                continue
            arr = (AdamTensorDev * len(recs))(*recs)  # host records; the library hands them to the kernel by value
            with torch.cuda.device(device):
                stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                check(lib.mg_adam_step_dev(ctypes.cast(arr, ctypes.c_void_p), len(recs), lr, beta1, beta2, eps,
                                           float(self.grad_scale), stream), "mg_adam_step_dev")
            for p in touched:  # the kernel wrote behind autograd's back: bump versions so packed-weight caches refresh
                torch.autograd.graph.increment_version(p)
        return loss

    def note_replay(self, params: Iterable[torch.Tensor]) -> None:
        """A captured graph containing this optimizer's step over `params` was replayed: the device counters advanced by
        themselves, bring the host mirrors and the parameter versions along."""
        for p in params:
            self.state[p]["step"] += 1
            torch.autograd.graph.increment_version(p)

    def state_dict(self):
        sd = super().state_dict()
        sd["state"] = {k: {kk: vv for kk, vv in v.items() if kk != "step_dev"} for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._state_epoch += 1  # new exp_avg / exp_avg_sq / step_dev tensors: graphs holding the old addresses are stale
        for p, st in self.state.items():
            st["step"] = torch.as_tensor(st["step"]).detach().to("cpu", torch.float32).reshape(())
            st["step_dev"] = torch.full((), int(st["step"]), dtype=torch.int32, device=p.device)
