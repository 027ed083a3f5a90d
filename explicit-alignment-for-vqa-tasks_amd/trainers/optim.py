"""Fused AdamW over the mapper's flat parameter buffer.

``torch.optim.AdamW(params, lr=config.train.lr)`` with torch defaults (betas 0.9/0.999, eps 1e-8,
weight_decay 0.01) is what the reference configures (src/trainers/clipcap_exector.py:79-81); here the
whole update is one HBM-bound kernel (``eavqa_adamw``) over the contiguous master / grad / moment
buffers, which also refreshes the bf16 shadow used by the next forward.
"""
from __future__ import annotations

import torch

from .. import ops


class FusedAdamW:
    def __init__(self, flat, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        self.flat = flat
        self.param_groups = [dict(lr=lr, initial_lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.m = torch.zeros_like(flat.master)
        self.v = torch.zeros_like(flat.master)
        self.step_count = 0

    def step(self, grad_scale: float = 1.0) -> None:
        g = self.param_groups[0]
        self.step_count += 1
        shadow = None if self.flat.shadow is self.flat.master else self.flat.shadow
        ops.adamw(self.flat.master, self.flat.grad, self.m, self.v, self.step_count, g["lr"], g["betas"][0], g["betas"][1],
                  g["eps"], g["weight_decay"], grad_scale, shadow=shadow)
        self.flat.mark_shadow_fresh()

    def zero_grad(self, set_to_none: bool = True) -> None:
        """No memset: the next backward overwrites the flat gradient instead of accumulating."""
        self.flat.grad_live = False

    def state_dict(self):
        return dict(step=self.step_count, m=self.m, v=self.v, param_groups=self.param_groups)

    def load_state_dict(self, sd) -> None:
        self.step_count = int(sd["step"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.param_groups = sd["param_groups"]


class ConstantScheduleWithWarmup:
    """``get_constant_schedule_with_warmup`` (the reference's ``"scheduler": "none"`` branch,
    clipcap_exector.py:102-110): lr * min(1, step / warmup)."""

    def __init__(self, optimizer: FusedAdamW, num_warmup_steps: int = 0):
        self.opt, self.warmup, self.n = optimizer, num_warmup_steps, 0
        self._apply()

    def _factor(self) -> float:
        return 1.0 if self.warmup <= 0 or self.n >= self.warmup else float(self.n) / float(max(1, self.warmup))

    def _apply(self) -> None:
        for g in self.opt.param_groups:
            g["lr"] = g["initial_lr"] * self._factor()

    def step(self) -> None:
        self.n += 1
        self._apply()

    def resume(self, global_step: int) -> None:
        """Continue from ``global_step`` optimiser steps (the reference passes ``last_epoch=self.global_step`` when it
        builds the scheduler, clipcap_exector.py:96-124)."""
        self.n = int(global_step)
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.opt.param_groups]


class LinearScheduleWithWarmup(ConstantScheduleWithWarmup):
    """``get_linear_schedule_with_warmup`` (``"scheduler": "linear"``, clipcap_exector.py:83-92): linear warm-up to lr,
    then linear decay to 0 at ``num_training_steps``."""

    def __init__(self, optimizer: FusedAdamW, num_warmup_steps: int, num_training_steps: int):
        self.total = max(1, num_training_steps)
        super().__init__(optimizer, num_warmup_steps)

    def _factor(self) -> float:
        if self.n < self.warmup:
            return float(self.n) / float(max(1, self.warmup))
        return max(0.0, float(self.total - self.n) / float(max(1, self.total - self.warmup)))


class CosineAnnealing(ConstantScheduleWithWarmup):
    """``optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=epochs, eta_min=1e-5)`` (``"scheduler": "cosine"``,
    clipcap_exector.py:93-101), closed form."""

    def __init__(self, optimizer: FusedAdamW, t_max: int, eta_min: float = 1e-5):
        import math
        self._math, self.t_max, self.eta_min = math, max(1, t_max), eta_min
        super().__init__(optimizer, 0)

    def _apply(self) -> None:
        for g in self.opt.param_groups:
            base = g["initial_lr"]
            g["lr"] = self.eta_min + (base - self.eta_min) * (1 + self._math.cos(self._math.pi * self.n / self.t_max)) / 2
