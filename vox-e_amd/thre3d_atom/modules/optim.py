"""Fused Adam for the voxel-grid tensors (new in this build).

`torch.optim.Adam(betas=(0.9, 0.999))` is what the reference's trainers construct
(modules/sds_trainer.py:200-203, modules/trainers.py:247-255).  `VoxeAdam` is an Optimizer with the same
state layout (`step`, `exp_avg`, `exp_avg_sq`) and update rule whose `step()` is ONE streaming HIP kernel
per tensor (voxe_adam_step: 7 * n * 4 bytes), so LR schedulers and checkpoint code that expect a
torch Optimizer keep working."""
from typing import Iterable

import torch

from voxe_hip import ops as _ops


class VoxeAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                state = self.state[p]
                if not state:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                state["step"] += 1
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                _ops.adam_step_(p.data, grad, state["exp_avg"], state["exp_avg_sq"], state["step"],
                                lr=group["lr"], beta1=beta1, beta2=beta2, eps=group["eps"])
                torch.autograd.graph.increment_version(p)
        return loss
