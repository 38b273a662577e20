"""
TEST INFRASTRUCTURE — gradient golden vectors for FuzzyGatingFusion.  Runs ONLY in the build container.
Runs the REFERENCE module (3_Models/fusion/fuzzy_gating_fusion.py) under autograd, in all four modes, with perturbed
parameters, on (i) a generic upstream gradient (CE on the fused logits + a term in alpha) and (ii) the loss the reference's
multimodal loop composes (4_Experiments/scripts/train_multimodal_fuzzy_fusion.py:436-460: CE(fused) + 0.3 CE(img/T_img) +
0.3 CE(eeg/T_eeg) + 0.1 temperature regulariser, temperatures detached in the auxiliary terms).  Data only.

Usage:  python oracle/make_golden_fuzzy_grad.py     -> tests/golden/fuzzy_grad.npz
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle.make_golden import REF, _load  # noqa: E402


def main():
    fz = _load("fuzzy_gating_fusion", REF / "3_Models" / "fusion" / "fuzzy_gating_fusion.py")
    g = torch.Generator().manual_seed(23)
    B, K = 24, 3
    zi0 = torch.randn(B, K, generator=g) * 2
    ze0 = torch.randn(B, K, generator=g) * 2
    zi0[1] = torch.tensor([9.0, 0, 0]); ze0[1] = torch.tensor([0.1, 0.0, -0.1])      # confident image, flat EEG
    zi0[2] = torch.tensor([0.0, 0.05, 0]); ze0[2] = torch.tensor([0, 0, 8.0])        # the opposite
    labels = torch.arange(B) % K
    blob = {"z_img": zi0.numpy(), "z_eeg": ze0.numpy(), "labels": labels.numpy()}
    for mode in fz.FuzzyGatingFusion.VALID_MODES:
        m = fz.FuzzyGatingFusion(num_classes=K, mode=mode)
        with torch.no_grad():                      # move the parameters off their initial values
            pg = torch.Generator().manual_seed(5)
            for p in m.parameters():
                p.add_(0.3 * torch.randn(p.shape, generator=pg))
            m.tau_img.fill_(3.5); m.tau_eeg.fill_(-1.2)       # T_img = 3.63 , T_eeg = 0.36: regulariser inactive / active
        for n, p in m.state_dict().items():
            blob[f"{mode}/state/{n}"] = p.detach().numpy().copy()
        for variant in ("generic", "loop"):
            zi, ze = zi0.clone().requires_grad_(True), ze0.clone().requires_grad_(True)
            m.zero_grad()
            fused, alpha, aux = m(zi, ze)
            if variant == "generic":
                loss = F.cross_entropy(fused, labels) + 0.3 * (alpha ** 2).sum()
            else:
                T_i, T_e = aux["temperatures"]["img"], aux["temperatures"]["eeg"]
                loss = (F.cross_entropy(fused, labels) + 0.3 * F.cross_entropy(zi / T_i, labels)
                        + 0.3 * F.cross_entropy(ze / T_e, labels) + 0.1 * m.compute_temperature_regularization(0.5, 5.0))
            loss.backward()
            pre = f"{mode}/{variant}/"
            blob[pre + "fused"], blob[pre + "alpha"], blob[pre + "loss"] = fused.detach().numpy(), alpha.detach().numpy(), loss.detach().numpy()
            blob[pre + "d_img"], blob[pre + "d_eeg"] = zi.grad.numpy().copy(), ze.grad.numpy().copy()
            for n, p in m.named_parameters():
                blob[pre + "d_" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    out = REPO / "tests" / "golden" / "fuzzy_grad.npz"
    np.savez_compressed(out, **blob)
    print("wrote", out, out.stat().st_size)


if __name__ == "__main__":
    main()
