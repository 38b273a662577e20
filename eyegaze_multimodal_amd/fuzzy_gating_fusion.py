"""Drop-in `FuzzyGatingFusion` (reference 3_Models/fusion/fuzzy_gating_fusion.py:23-427): same constructor,
parameter names / initial values and forward signature; the forward runs as one HIP kernel (eg_fuzzy_gate_fwd).
Forward only in this version (inference / evaluation of the logit-level fusion); training its 12 scalars is a
"next" item."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from ._lib import call, ptr


def _inv_softplus(x: float) -> float:
    return math.log(math.expm1(x))


class FuzzyGatingFusion(nn.Module):
    VALID_MODES = ("full", "no_temperature", "no_fuzzification", "fixed_weights")

    def __init__(self, num_classes: int = 3, mode: str = "full", eps_temp: float = 0.1, eps_log: float = 1e-8,
                 eps_div: float = 1e-8):
        super().__init__()
        if mode not in self.VALID_MODES:
            raise ValueError(f"Invalid mode '{mode}'. Must be one of {self.VALID_MODES}")
        self.num_classes, self.mode = num_classes, mode
        self.eps_temp, self.eps_log, self.eps_div = eps_temp, eps_log, eps_div
        self.max_entropy = math.log(num_classes)
        self.tau_img = nn.Parameter(torch.tensor(_inv_softplus(1.5 - eps_temp)))
        self.tau_eeg = nn.Parameter(torch.tensor(_inv_softplus(1.0 - eps_temp)))
        self.register_buffer("c_reliable", torch.tensor(0.0))
        c0 = self.max_entropy * 0.8
        self.c_unreliable_img = nn.Parameter(torch.tensor(c0))
        self.c_unreliable_eeg = nn.Parameter(torch.tensor(c0))
        ls = math.log(self.max_entropy * 0.3)
        self.log_sigma_reliable_img = nn.Parameter(torch.tensor(ls))
        self.log_sigma_reliable_eeg = nn.Parameter(torch.tensor(ls))
        self.log_sigma_unreliable_img = nn.Parameter(torch.tensor(ls))
        self.log_sigma_unreliable_eeg = nn.Parameter(torch.tensor(ls))
        self.beta = nn.Parameter(torch.tensor([math.log(0.8 / 0.2), math.log(0.2 / 0.8), math.log(0.6 / 0.4), 0.0]))

    def _packed(self, device) -> torch.Tensor:
        s = [self.tau_img, self.tau_eeg, self.c_unreliable_img, self.c_unreliable_eeg, self.log_sigma_reliable_img,
             self.log_sigma_reliable_eeg, self.log_sigma_unreliable_img, self.log_sigma_unreliable_eeg]
        return torch.cat([torch.stack([p.detach().float() for p in s]), self.beta.detach().float()]).to(device).contiguous()

    @torch.no_grad()
    def forward(self, img_logits: torch.Tensor, eeg_logits: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, Dict]:
        if not img_logits.is_cuda:
            raise L.EgError("FuzzyGatingFusion (HIP) needs device tensors; there is no CPU fallback")
        zi, ze = img_logits.detach().float().contiguous(), eeg_logits.detach().float().contiguous()
        B, K = zi.shape
        prm = self._packed(zi.device)
        fused, alpha = torch.empty_like(zi), torch.empty(B, device=zi.device)
        call("eg_fuzzy_gate_fwd", ptr(zi), ptr(ze), ptr(prm), ptr(fused), ptr(alpha), B, K, self.VALID_MODES.index(self.mode),
             self.eps_temp, self.eps_log, self.eps_div, torch.cuda.current_stream(zi.device).cuda_stream)
        self._keep = (zi, ze, prm)
        return fused, alpha, {}
