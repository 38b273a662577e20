"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of FuzzyGatingFusion.forward
(reference 3_Models/fusion/fuzzy_gating_fusion.py:297-390; helpers :132-295), pinned by tests/golden/fuzzy_gating.npz
(generated from the reference by oracle/make_golden.py) and by the reference's own known answers
(:430-538: alpha = 0.5000 / 0.7907 / 0.2102, T_img = 1.5, T_eeg = 1.0)."""
import math

import numpy as np


def softplus(x):
    return np.log1p(np.exp(-abs(x))) + max(x, 0.0)


def fuzzy_forward(z_img, z_eeg, p, mode="full", eps_temp=0.1, eps_log=1e-8, eps_div=1e-8):
    """z_*: [B, K] float arrays; p: dict of the module's scalar parameters (state_dict names)."""
    z_img, z_eeg = np.asarray(z_img, np.float32), np.asarray(z_eeg, np.float32)  # fp32 like the reference
    K = z_img.shape[1]
    if mode in ("no_temperature", "fixed_weights"):            # :323-327
        zi, ze = z_img, z_eeg
    else:                                                      # :328-332
        zi = z_img / np.float32(softplus(float(p["tau_img"])) + eps_temp)
        ze = z_eeg / np.float32(softplus(float(p["tau_eeg"])) + eps_temp)

    def entropy(z):                                            # :132-146
        e = np.exp(z - z.max(-1, keepdims=True))
        pr = e / e.sum(-1, keepdims=True)
        return -(pr * np.log(pr + np.float32(eps_log))).sum(-1)
    Hi, He = entropy(zi), entropy(ze)
    if mode == "fixed_weights":                                # :338-340
        alpha = np.full(z_img.shape[0], 0.5)
    elif mode == "no_fuzzification":                           # :264-295
        hmax = np.float32(math.log(K))
        ci = np.clip(np.float32(1) - Hi / (hmax + np.float32(eps_div)), 0, None)
        ce = np.clip(np.float32(1) - He / (hmax + np.float32(eps_div)), 0, None)
        alpha = np.clip(ci / (ci + ce + np.float32(eps_div)), 0, 1)
    else:
        def mu(x, c, ls):                                      # :148-168
            s = math.exp(float(ls))
            return np.exp(-((x - c) ** 2) / (2 * s * s + eps_div))
        ir, iu = mu(Hi, 0.0, p["log_sigma_reliable_img"]), mu(Hi, float(p["c_unreliable_img"]), p["log_sigma_unreliable_img"])
        er, eu = mu(He, 0.0, p["log_sigma_reliable_eeg"]), mu(He, float(p["c_unreliable_eeg"]), p["log_sigma_unreliable_eeg"])
        w = np.stack([ir * eu, iu * er, ir * er, iu * eu], -1)  # :228-233
        theta = 1 / (1 + np.exp(-np.asarray(p["beta"], np.float64)))
        alpha = np.clip((w * theta).sum(-1) / (w.sum(-1) + eps_div), 0, 1)  # :256-262
    fused = alpha[:, None] * zi + (1 - alpha[:, None]) * ze    # :386-388
    return fused, alpha


# ------------------------------------------------------------------------------------------------------------------
# differentiable restatement (torch fp32, autograd) used to pin gradients: tests/golden/fuzzy_grad.npz
# ------------------------------------------------------------------------------------------------------------------
PARAM_NAMES = ["tau_img", "tau_eeg", "c_unreliable_img", "c_unreliable_eeg", "log_sigma_reliable_img", "log_sigma_reliable_eeg",
               "log_sigma_unreliable_img", "log_sigma_unreliable_eeg", "beta"]


def fuzzy_forward_torch(z_img, z_eeg, p, mode="full", eps_temp=0.1, eps_log=1e-8, eps_div=1e-8):
    """Same math as fuzzy_forward on torch tensors (p: name -> tensor, may require grad).  Returns fused, alpha, (T_img, T_eeg)."""
    import torch
    import torch.nn.functional as F
    K = z_img.shape[1]
    if mode in ("no_temperature", "fixed_weights"):
        Ti = Te = torch.ones(())
        zi, ze = z_img, z_eeg
    else:
        Ti, Te = F.softplus(p["tau_img"]) + eps_temp, F.softplus(p["tau_eeg"]) + eps_temp
        zi, ze = z_img / Ti, z_eeg / Te

    def entropy(z):
        pr = torch.softmax(z, -1)
        return -(pr * torch.log(pr + eps_log)).sum(-1)
    Hi, He = entropy(zi), entropy(ze)
    if mode == "fixed_weights":
        alpha = torch.full((z_img.shape[0],), 0.5)
    elif mode == "no_fuzzification":
        hmax = math.log(K)
        ci = torch.clamp(1.0 - Hi / (hmax + eps_div), min=0.0)
        ce = torch.clamp(1.0 - He / (hmax + eps_div), min=0.0)
        alpha = torch.clamp(ci / (ci + ce + eps_div), 0.0, 1.0)
    else:
        def mu(x, c, ls):
            return torch.exp(-((x - c) ** 2) / (2 * torch.exp(ls) ** 2 + eps_div))
        ir, iu = mu(Hi, 0.0, p["log_sigma_reliable_img"]), mu(Hi, p["c_unreliable_img"], p["log_sigma_unreliable_img"])
        er, eu = mu(He, 0.0, p["log_sigma_reliable_eeg"]), mu(He, p["c_unreliable_eeg"], p["log_sigma_unreliable_eeg"])
        w = torch.stack([ir * eu, iu * er, ir * er, iu * eu], -1)
        alpha = torch.clamp((w * torch.sigmoid(p["beta"])).sum(-1) / (w.sum(-1) + eps_div), 0.0, 1.0)
    fused = alpha[:, None] * zi + (1 - alpha[:, None]) * ze
    return fused, alpha, (Ti, Te)


def temperature_regularization(p, t_min=0.5, t_max=5.0, eps_temp=0.1):
    """fuzzy_gating_fusion.py:392-419"""
    import torch.nn.functional as F
    Ti, Te = F.softplus(p["tau_img"]) + eps_temp, F.softplus(p["tau_eeg"]) + eps_temp
    return F.relu(Ti - t_max) + F.relu(t_min - Ti) + F.relu(Te - t_max) + F.relu(t_min - Te)


def fusion_loop_loss(z_img, z_eeg, labels, p, mode, lam_img=0.3, lam_eeg=0.3, lam_reg=0.1, t_min=0.5, t_max=5.0, parts=None):
    """The loss of the reference's multimodal step (train_multimodal_fuzzy_fusion.py:436-460); auxiliary terms use the
    DETACHED temperatures (aux_info['temperatures'] holds .detach()ed tensors, fuzzy_gating_fusion.py:334)."""
    import torch.nn.functional as F
    fused, alpha, (Ti, Te) = fuzzy_forward_torch(z_img, z_eeg, p, mode)
    ce, ai, ae = F.cross_entropy(fused, labels), F.cross_entropy(z_img / Ti.detach(), labels), F.cross_entropy(z_eeg / Te.detach(), labels)
    reg = temperature_regularization(p, t_min, t_max)
    loss = ce + lam_img * ai + lam_eeg * ae + lam_reg * reg
    if parts is not None:           # the four terms the reference loop logs per step (:512-516)
        parts.update(loss_ce=ce.detach(), loss_aux_img=ai.detach(), loss_aux_eeg=ae.detach(), loss_reg=reg.detach())
    return loss, fused, alpha
