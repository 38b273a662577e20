"""Native optimiser step: clip_grad_norm_(1.0) + AdamW (train_art.py:221-222, 401-405) as two HIP kernels over
the model's flat fp32 parameter / gradient buffers, plus the per-epoch cosine schedule (train_art.py:409, 494)."""
from __future__ import annotations

import math

import torch


class HipAdamW:
    def __init__(self, model, lr: float = 1e-4, weight_decay: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_grad_norm: float = 1.0):
        self.model, self.base_lr, self.lr = model, lr, lr
        self.weight_decay, self.betas, self.eps, self.max_grad_norm = weight_decay, betas, eps, max_grad_norm
        self.t = 0
        self.m = None
        self.v = None

    def _ensure(self):
        fp = self.model._flat
        if self.m is None or self.m.device != fp.flat.device or self.m.numel() != fp.total:
            self.m = torch.zeros_like(fp.flat)
            self.v = torch.zeros_like(fp.flat)

    def begin_step(self, eng, seed: int, grad_scale: float = 1.0):
        """Publishes this step's scalars (dropout seed, lr, bias corrections) to the device-resident state."""
        self.t += 1
        eng.set_state(seed=seed, lr=self.lr, step=self.t, grad_scale=grad_scale, beta1=self.betas[0], beta2=self.betas[1])

    def step(self, eng):
        self._ensure()
        eng.optimizer_step(self.m, self.v, self.max_grad_norm, self.betas, self.eps, self.weight_decay)

    def set_epoch(self, epoch: int, t_max: int):
        """CosineAnnealingLR(T_max=t_max, eta_min=0) evaluated at `epoch` (closed form)."""
        self.lr = self.base_lr * (1 + math.cos(math.pi * epoch / t_max)) / 2

    def state_dict(self):
        return {"t": self.t, "lr": self.lr, "base_lr": self.base_lr, "m": self.m, "v": self.v}

    def load_state_dict(self, sd):
        self.t, self.lr, self.base_lr = sd["t"], sd["lr"], sd["base_lr"]
        self.m, self.v = sd["m"], sd["v"]
