"""
TEST INFRASTRUCTURE — CPU restatement of one step of the reference's multimodal logit-fusion loop
(4_Experiments/scripts/train_multimodal_fuzzy_fusion.py: model forward :140-179, losses :436-460, clip + AdamW with
per-group learning rates :464-470 / :727-737, LambdaLR warm-up + cosine per step :197-214), with the ViT replaced by the in-tree
2-D CNN (3_Models/backbones/dual_eeg_transformer.py:70-86) exactly as eyegaze_multimodal_amd/image_encoder.py defines the
image branch.  Only tests/ may import this.  parity: the EEG branch and the fuzzy gate are pinned by the reference-generated
fixtures (tests/golden); the loop itself cannot be run from the reference here (it imports timm / torchvision, absent), so the
STEP is "parity unpinned" beyond those pinned parts -- this restatement is plain torch (autograd, torch.optim.AdamW, LambdaLR).
"""
from __future__ import annotations

import copy
import math

import torch
import torch.nn.functional as F

from oracle import dual_eeg_oracle as O
from oracle import fuzzy_oracle as FO


def image_logits(enc_cpu, img1, img2):
    """per-player CNN features, pair head (image_encoder.py); torch's own Conv2d / MaxPool / AdaptiveAvgPool / Linear"""
    def f(x):
        return enc_cpu.cnn.proj(enc_cpu.cnn.spec_conv(x[:, None]).flatten(1))
    return enc_cpu.head(torch.cat([f(img1), f(img2)], -1))


def synthetic_gaze_state(module, seed: int):
    """deterministic synthetic weights for the image branch (`cnn.*`, `head.*`), independent of torch's default-init RNG stream;
    used by oracle/make_golden_mm_loop.py on the reference side and by the tests on this side"""
    import numpy as np
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in module.state_dict().items():
        fan = max(1, int(np.prod(v.shape[1:]))) if v.dim() > 1 else 1
        sd[k] = (torch.randn(v.shape, generator=g) * (0.5 / np.sqrt(fan) if v.dim() > 1 else 0.02)).to(v.dtype)
    return sd


class Stepper:
    def __init__(self, gaze_cpu, eeg_cfg, eeg_sd, fusion_sd, mode, encoder_lr, fusion_lr, weight_decay, max_grad_norm, lams,
                 treg, warmup_steps, total_steps):
        self.gaze = copy.deepcopy(gaze_cpu).eval()
        self.cfg = eeg_cfg
        self.eeg = {k: v.clone().requires_grad_(True) for k, v in eeg_sd.items() if v.dtype.is_floating_point and k != "spectrogram_generator.window"}
        self.buf = {k: v.clone() for k, v in eeg_sd.items() if k not in self.eeg}
        self.fus = {k: v.clone().requires_grad_(True) for k, v in fusion_sd.items() if k != "c_reliable"}
        self.mode, self.lams, self.treg, self.max_norm = mode, lams, treg, max_grad_norm
        groups = [{"params": list(self.gaze.parameters()), "lr": encoder_lr}, {"params": list(self.eeg.values()), "lr": encoder_lr},
                  {"params": list(self.fus.values()), "lr": fusion_lr}]
        self.opt = torch.optim.AdamW(groups, weight_decay=weight_decay)

        def lam(step):
            if step < warmup_steps:
                return float(step) / float(max(1, warmup_steps))
            prog = float(step - warmup_steps) / float(max(1, total_steps - warmup_steps))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lam)

    def step(self, img1, img2, eeg1, eeg2, labels):
        self.opt.zero_grad()
        z_img = image_logits(self.gaze, img1, img2)
        z_eeg = O.forward(eeg1, eeg2, {**self.eeg, **self.buf}, self.cfg, labels)["logits"]
        li, le, lr_ = self.lams
        parts = {}
        loss, fused, alpha = FO.fusion_loop_loss(z_img, z_eeg, labels, self.fus, self.mode, li, le, lr_, *self.treg, parts=parts)
        loss.backward()
        params = [p for g in self.opt.param_groups for p in g["params"]]
        grads = {"gaze": torch.cat([p.grad.reshape(-1) for p in self.gaze.parameters()]),
                 "eeg": {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in self.eeg.items()},
                 "fusion": {k: p.grad.clone() for k, p in self.fus.items()}}
        norm = torch.nn.utils.clip_grad_norm_(params, self.max_norm) if self.max_norm else None
        lrs = [g["lr"] for g in self.opt.param_groups]
        self.opt.step()
        self.sched.step()
        return dict(loss=loss.detach(), fused=fused.detach(), alpha=alpha.detach(), z_img=z_img.detach(), z_eeg=z_eeg.detach(),
                    grads=grads, norm=norm, lrs=lrs, **parts)
