"""Fused multi-tensor Adam on the HIP kernel `mg_adam_step` -- one launch per optimizer step.

Semantics and state layout are torch.optim.Adam's (amsgrad=False, weight_decay=0, maximize=False), which is what the
reference uses (/root/reference/music_gan/train.py:64-70,175,214,262-272): per-parameter `step`, `exp_avg`, `exp_avg_sq`;
parameters whose .grad is None are skipped (their step count does not advance); param groups added later start at step 0.
`state_dict()` is loadable by torch.optim.Adam and vice versa.
"""
from __future__ import annotations

import ctypes
import math
from typing import List

import torch

from . import _lib
from ._lib import AdamTensor, check


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self.grad_scale = 1.0  # multiplied into every gradient inside the kernel (data-parallel averaging)

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
            recs: List[AdamTensor] = []
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
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                step = float(st["step"])
                bc1 = 1.0 - beta1 ** step
                bc2 = 1.0 - beta2 ** step
                recs.append(AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                       st["exp_avg_sq"].data_ptr(), p.numel(), lr / bc1, math.sqrt(bc2)))
                touched.append(p)
                device = p.device
            if not recs:
                continue
            arr = (AdamTensor * len(recs))(*recs)  # host records; the library hands them to the kernel by value
            with torch.cuda.device(device):
                stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                check(lib.mg_adam_step(ctypes.cast(arr, ctypes.c_void_p), len(recs), beta1, beta2, eps,
                                       float(self.grad_scale), stream), "mg_adam_step")
            for p in touched:  # the kernel wrote behind autograd's back: bump versions so packed-weight caches refresh
                torch.autograd.graph.increment_version(p)
        return loss
