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
        self._eng = None
        from . import ops
        self._op_handle = ops.register_owner(self)

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
        """clip_grad_norm_(max_grad_norm) + AdamW through the registered operator eyegaze::clip_adamw_step (ops.py)."""
        self._ensure()
        self._eng = eng
        fp = self.model._flat
        torch.ops.eyegaze.clip_adamw_step(fp.flat, fp.grad, self.m, self.v, self._op_handle)

    def _native_step(self, flat_params, flat_grads, exp_avg, exp_avg_sq):
        fp = self.model._flat
        if flat_params.data_ptr() != fp.flat.data_ptr() or flat_grads.data_ptr() != fp.grad.data_ptr():
            raise RuntimeError("eyegaze::clip_adamw_step works on the module's own flat parameter / gradient buffers")
        self._eng.optimizer_step(exp_avg, exp_avg_sq, self.max_grad_norm, self.betas, self.eps, self.weight_decay)

    def set_epoch(self, epoch: int, t_max: int):
        """CosineAnnealingLR(T_max=t_max, eta_min=0) evaluated at `epoch` (closed form)."""
        self.lr = self.base_lr * (1 + math.cos(math.pi * epoch / t_max)) / 2
        self.epoch, self.t_max = epoch, t_max

    def scheduler_state_dict(self):
        """The fields torch.optim.lr_scheduler.CosineAnnealingLR.state_dict() carries (train_art.py:480-491 stores them in the
        periodic checkpoints), so `CosineAnnealingLR.load_state_dict` accepts it."""
        ep = getattr(self, "epoch", 0)
        return {"T_max": getattr(self, "t_max", 1), "eta_min": 0.0, "base_lrs": [self.base_lr], "last_epoch": ep,
                "_step_count": ep + 1, "_last_lr": [self.lr]}

    def state_dict(self):
        """torch.optim.AdamW's layout (`state` keyed by parameter index in named_parameters() order, one `param_groups`
        entry), so a checkpoint written here loads into the reference's optimizer and the other way round."""
        self._ensure()
        fp = self.model._flat
        state = {}
        for i, (n, p) in enumerate(zip(fp.names, fp.params)):
            o = fp.offsets[n]
            state[i] = {"step": torch.tensor(float(self.t)), "exp_avg": self.m[o:o + p.numel()].view(p.shape).clone(),
                        "exp_avg_sq": self.v[o:o + p.numel()].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": True, "initial_lr": self.base_lr, "params": list(range(len(fp.names)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if "param_groups" not in sd:          # round-1 layout {t, lr, base_lr, m, v}
            self.t, self.lr, self.base_lr = sd["t"], sd["lr"], sd["base_lr"]
            self.m, self.v = sd["m"], sd["v"]
            return
        self._ensure()
        fp = self.model._flat
        g = sd["param_groups"][0]
        self.lr, self.base_lr = g["lr"], g.get("initial_lr", g["lr"])
        self.betas, self.eps, self.weight_decay = tuple(g["betas"]), g["eps"], g["weight_decay"]
        self.m.zero_()
        self.v.zero_()
        steps = [0.0]
        for i, (n, p) in enumerate(zip(fp.names, fp.params)):
            st = sd["state"].get(i)
            if st is None:                    # torch keeps no state for parameters that never received a gradient
                continue
            o = fp.offsets[n]
            self.m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            steps.append(float(st["step"]))
        self.t = int(max(steps))
