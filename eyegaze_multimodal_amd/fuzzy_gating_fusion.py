"""Drop-in `FuzzyGatingFusion` (reference 3_Models/fusion/fuzzy_gating_fusion.py:23-427): same constructor,
parameter names / initial values and forward signature.  Forward and backward are one HIP kernel each
(eg_fuzzy_gate_fwd / eg_fuzzy_gate_bwd) behind a torch.autograd.Function, so the module trains inside the reference's
multimodal step (4_Experiments/scripts/train_multimodal_fuzzy_fusion.py:432-470) unchanged: gradients reach both logit sets
and the 12 scalars."""
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
        return torch.cat([torch.stack([p.float() for p in s]), self.beta.float()]).to(device).contiguous()

    @property
    def temp_img(self) -> torch.Tensor:      # :120-123
        return torch.nn.functional.softplus(self.tau_img) + self.eps_temp

    @property
    def temp_eeg(self) -> torch.Tensor:      # :125-128
        return torch.nn.functional.softplus(self.tau_eeg) + self.eps_temp

    def compute_temperature_regularization(self, t_min: float = 0.5, t_max: float = 5.0) -> torch.Tensor:
        """:392-419 — four ReLUs on two scalars (host-side glue on the parameters themselves)."""
        relu = torch.nn.functional.relu
        Ti, Te = self.temp_img, self.temp_eeg
        return relu(Ti - t_max) + relu(t_min - Ti) + relu(Te - t_max) + relu(t_min - Te)

    def forward(self, img_logits: torch.Tensor, eeg_logits: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, Dict]:
        if not img_logits.is_cuda:
            raise L.EgError("FuzzyGatingFusion (HIP) needs device tensors; there is no CPU fallback")
        if img_logits.shape != eeg_logits.shape or img_logits.dim() != 2:
            raise L.EgError(f"expected two [B, K] logit tensors, got {tuple(img_logits.shape)} / {tuple(eeg_logits.shape)}")
        mode = self.VALID_MODES.index(self.mode)
        fused, alpha = _FuzzyFn.apply(img_logits.float(), eeg_logits.float(), self._packed(img_logits.device), mode,
                                      self.eps_temp, self.eps_log, self.eps_div)
        temps = ((self.temp_img.detach(), self.temp_eeg.detach()) if self.mode in ("full", "no_fuzzification")
                 else (torch.ones(1, device=img_logits.device), torch.ones(1, device=img_logits.device)))
        aux = {"temperatures": {"img": temps[0], "eeg": temps[1]}}      # what the train loop reads (:443-444)
        return fused, alpha, aux


class _FuzzyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, zi, ze, prm, mode, eps_temp, eps_log, eps_div):
        zi, ze, prm = zi.detach().contiguous(), ze.detach().contiguous(), prm.detach().contiguous()
        B, K = zi.shape
        fused, alpha = torch.empty_like(zi), torch.empty(B, device=zi.device)
        call("eg_fuzzy_gate_fwd", ptr(zi), ptr(ze), ptr(prm), ptr(fused), ptr(alpha), B, K, mode, eps_temp, eps_log, eps_div,
             torch.cuda.current_stream(zi.device).cuda_stream)
        ctx.save_for_backward(zi, ze, prm)
        ctx.cfg = (mode, eps_temp, eps_log, eps_div)
        return fused, alpha

    @staticmethod
    def backward(ctx, dfused, dalpha):
        zi, ze, prm = ctx.saved_tensors
        mode, eps_temp, eps_log, eps_div = ctx.cfg
        B, K = zi.shape
        dfused = dfused.float().contiguous()
        dalpha = dalpha.float().contiguous() if dalpha is not None else None
        dzi, dze = torch.empty_like(zi), torch.empty_like(ze)
        nblk = (B + 127) // 128
        part = torch.empty(nblk, 12, device=zi.device)
        dprm = torch.empty(12, device=zi.device)
        st = torch.cuda.current_stream(zi.device).cuda_stream
        call("eg_fuzzy_gate_bwd", ptr(zi), ptr(ze), ptr(prm), ptr(dfused), ptr(dalpha), ptr(dzi), ptr(dze), ptr(part), B, K, mode,
             eps_temp, eps_log, eps_div, st)
        call("eg_reduce_partials", ptr(part), ptr(dprm), 12, nblk, 12, 0, st)
        return dzi, dze, dprm, None, None, None, None
