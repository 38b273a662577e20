"""Multimodal logit-fusion training loop on the MI355X engines — the counterpart of the reference's
4_Experiments/scripts/train_multimodal_fuzzy_fusion.py (model :106-179, train_one_epoch :395-543, optimizer / scheduler /
GradScaler :727-753) for BASELINE configs[4] ("three-modality ... image (2-D CNN backbone) ... fp16").

  image branch   GazeCNNEncoder (image_encoder.py): the in-tree 2-D CNN on each player's one-channel image + a pair head.
                 (The reference's timm ViT needs a package and downloaded weights that do not exist offline.)
  EEG branch     the HIP DualEEGTransformer (any ablation flags; configs[4] = + spectrogram tokens).
  fusion         FuzzyGatingFusion (HIP forward / backward kernels, fuzzy_gating_fusion.py).

One step (train_one_epoch :427-472), native, without an autograd graph over the encoders:
  forward of both encoders -> fused logits, alpha -> loss = CE(fused) + l_img CE(z_img / T_img) + l_eeg CE(z_eeg / T_eeg)
  + l_reg L_reg(T) on the [B, K] logits (a tiny autograd graph over 2 x [B, K] + 12 scalars) -> the gradients w.r.t. the two
  logit sets enter the encoders' HIP backwards -> ONE global clip over all three parameter sets -> AdamW with per-group learning
  rates (encoders / fusion, :727-737) -> per-STEP linear warm-up + cosine schedule (:197-214).
fp16 (config training.fp16, :752-753): dynamic loss scaling with GradScaler's semantics, entirely on the device -- the loss
gradient is multiplied by the device-resident scale, eg_clip_coef un-scales and flags non-finite norms, every AdamW kernel skips
on the flag, eg_scaler_update backs off / grows.  The three parameter sets share ONE eg_step_state (one scale, one flag, one
clip coefficient), exactly as one GradScaler + one clip_grad_norm_(model.parameters()) do in the reference.

Data parallel (BASELINE configs[4] is an 8-GPU configuration; the reference itself is single-process): one process per GPU,
rank r trains on samples r::world of the global batch; the three flat gradient buffers are summed over ranks by
ddp.MultimodalReducers on one side stream (fusion scalars first, the image branch under the EEG backward, the EEG encoder's
buckets as its backward releases them), 1/world is folded into eg_clip_coef / eg_adamw_group, and the overflow flag is made
collective before any AdamW kernel reads it, so an fp16 overflow on ONE rank skips the step on EVERY rank.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from ._lib import call, ptr
from .dual_eeg_transformer import DualEEGTransformer
from .engine import FlatParams
from .fuzzy_gating_fusion import FuzzyGatingFusion
from .image_encoder import GazeCNNEncoder


def warmup_cosine_factor(step: int, warmup_steps: int, total_steps: int) -> float:
    """lr_lambda of get_linear_warmup_cosine_scheduler (train_multimodal_fuzzy_fusion.py:207-212)"""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    progress = float(step - warmup_steps) / float(max(1, total_steps - warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


class MultimodalFusionModel(nn.Module):
    """gaze_encoder / eeg_encoder / fusion attribute names as in the reference (:119-126); freeze flags likewise."""

    def __init__(self, gaze_encoder: GazeCNNEncoder, eeg_encoder: DualEEGTransformer, fusion_module: FuzzyGatingFusion,
                 freeze_gaze: bool = False, freeze_eeg: bool = False):
        super().__init__()
        self.gaze_encoder, self.eeg_encoder, self.fusion = gaze_encoder, eeg_encoder, fusion_module
        self.freeze_gaze, self.freeze_eeg = freeze_gaze, freeze_eeg
        for flag, enc in ((freeze_gaze, gaze_encoder), (freeze_eeg, eeg_encoder)):
            if flag:
                for p in enc.parameters():
                    p.requires_grad = False


class MultimodalTrainer:
    def __init__(self, model: MultimodalFusionModel, device, *, encoder_lr: float = 1e-4, fusion_lr: float = 1e-3,
                 weight_decay: float = 0.01, max_grad_norm: Optional[float] = 1.0, lambda_aux_img: float = 0.3,
                 lambda_aux_eeg: float = 0.3, lambda_reg: float = 0.1, temp_reg_min: float = 0.5, temp_reg_max: float = 5.0,
                 warmup_steps: int = 0, total_steps: int = 1, betas=(0.9, 0.999), eps: float = 1e-8, seed: int = 0,
                 group=None, force_dist: bool = False):
        """group / force_dist: data-parallel exchange over torch.distributed (active when a process group is initialised and has
        more than one rank; force_dist runs the collective path with a single rank as a rehearsal)."""
        self.model, self.device = model.to(device), torch.device(device)
        self.group, self.force_dist, self.red = group, force_dist, None
        self.encoder_lr, self.fusion_lr, self.wd, self.max_norm = encoder_lr, fusion_lr, weight_decay, max_grad_norm
        self.lams = (lambda_aux_img, lambda_aux_eeg, lambda_reg)
        self.treg = (temp_reg_min, temp_reg_max)
        self.warmup_steps, self.total_steps, self.betas, self.eps, self.seed = warmup_steps, total_steps, betas, eps, seed
        self.step_no = 0
        self.fus = FlatParams(model.fusion)
        self.fus.ensure(self.device)
        self.state = {}        # per parameter set: exp_avg, exp_avg_sq
        self.sqpart = None
        self.eng = None
        # fusion + losses + their gradients on the [B, K] logits as four HIP launches (eg_fuzzy_gate_fwd, eg_fusion_loop_loss,
        # eg_fuzzy_gate_bwd, a reduce) instead of a torch autograd graph of ~70 tiny kernels (0.8 ms of idle GPU per step at B = 256)
        self.fused_loss = os.environ.get("EYEGAZE_MM_FUSED_LOSS", "1") != "0"
        fz = model.fusion
        names = ["tau_img", "tau_eeg", "c_unreliable_img", "c_unreliable_eeg", "log_sigma_reliable_img", "log_sigma_reliable_eeg",
                 "log_sigma_unreliable_img", "log_sigma_unreliable_eeg"]       # FuzzyGatingFusion._packed's order
        idx = [self.fus.offsets[n] for n in names] + [self.fus.offsets["beta"] + i for i in range(4)]
        self._prm_idx = torch.tensor(idx, dtype=torch.int64, device=self.device)
        self._lw = {}          # per (B, K): work tensors of the fused loss path

    # ------------------------------------------------------------------------------------------
    def _engines(self, B, T, F_, W_):
        m = self.model
        eeg = m.eeg_encoder.engine(B, T, self.device)
        img = m.gaze_encoder.engine(B, F_, W_, self.device, state_dev=eeg.state_dev)
        if img.state_dev.data_ptr() != eeg.state_dev.data_ptr():
            img.state_dev = eeg.state_dev
        img.scaler_on = eeg.scaler_on          # one scale / one overflow flag for the whole model
        return eeg, img

    def _reducers(self, eeg, img):
        """The data-parallel exchange, built on first use (None in a single-process run): parameters of all three sets are
        broadcast from rank 0 once, so the replicas start identical whatever each rank's initialisation was."""
        import torch.distributed as dist
        if self.red is None and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force_dist):
            from .ddp import MultimodalReducers
            m = self.model
            ecfg = m.eeg_encoder.cfg
            self.red = MultimodalReducers(None if m.freeze_eeg else eeg.fp, None if m.freeze_gaze else img.fp, self.fus,
                                          ecfg.num_layers, ecfg.use_cross_attention, self.group, self.force_dist)
            self.red.broadcast(eeg.fp.flat, img.fp.flat, self.fus.flat)
        return self.red

    def _moments(self, name, fp):
        if name not in self.state:
            self.state[name] = (torch.zeros_like(fp.flat), torch.zeros_like(fp.flat))
        return self.state[name]

    def current_lr_factor(self) -> float:
        return warmup_cosine_factor(self.step_no, self.warmup_steps, self.total_steps)

    def train_step(self, img1, img2, eeg1, eeg2, labels, dropout: bool = True) -> Dict[str, torch.Tensor]:
        """dropout=False runs the step with every dropout site inactive (deterministic parity checks)"""
        m = self.model
        m.train()
        B, _, T = eeg1.shape
        F_, W_ = img1.shape[-2], img1.shape[-1]
        eeg, img = self._engines(B, T, F_, W_)
        red = self._reducers(eeg, img)
        lr = self.encoder_lr * self.current_lr_factor()           # scheduler.step() follows optimizer.step(): step k uses lambda(k)
        t = self.step_no + 1
        eeg.set_state(seed=self.seed * 7919 + t, lr=lr, step=t, beta1=self.betas[0], beta2=self.betas[1],
                      grad_scale=(red.grad_scale if red else 1.0))
        if self.fused_loss:
            return self._train_step_fused(m, eeg, img, red, img1, img2, eeg1, eeg2, labels, dropout)
        # ---- forward of both encoders (HIP) ----
        z_img = img.forward(img1.contiguous().float(), img2.contiguous().float(), train=dropout).detach().clone().requires_grad_(True)
        eeg.forward(eeg1.contiguous().float(), eeg2.contiguous().float(), labels, train=dropout)
        z_eeg = eeg.a["logits"].detach().clone().requires_grad_(True)
        # ---- fusion + losses on the [B, K] logits (:436-460) ----
        for p in m.fusion.parameters():
            p.grad = None
        fused, alpha, aux = m.fusion(z_img, z_eeg)
        T_img, T_eeg = aux["temperatures"]["img"], aux["temperatures"]["eeg"]
        loss_ce = F.cross_entropy(fused, labels)
        loss_aux_img = F.cross_entropy(z_img / T_img, labels)
        loss_aux_eeg = F.cross_entropy(z_eeg / T_eeg, labels)
        loss_reg = m.fusion.compute_temperature_regularization(*self.treg)
        li, le, lr_ = self.lams
        loss = loss_ce + li * loss_aux_img + le * loss_aux_eeg + lr_ * loss_reg
        scale = eeg.loss_scale_dev if eeg.scaler_on else None     # scaler.scale(loss).backward() (:462)
        (loss * scale if scale is not None else loss).sum().backward()
        fus = self.fus
        for n, p in zip(fus.names, fus.params):
            o = fus.offsets[n]
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            fus.grad[o:o + p.numel()].copy_(g.reshape(-1))
        if red:
            red.on_fusion()
        # ---- encoders' backwards from the (scaled) logit gradients ----
        if not m.freeze_gaze:
            img.backward(z_img.grad)
            if red:
                red.on_gaze()                 # the image branch's gradients travel under the EEG encoder's backward
        if not m.freeze_eeg:
            eeg.backward(glogits=z_eeg.grad, prescaled=True, on_segment=(red.on_eeg_segment if red else None))
        if red:
            red.finish()
        # ---- scaler.unscale_ + clip_grad_norm_(model.parameters()) + scaler.step + scaler.update (:464-472) ----
        self._optimizer_step(eeg, img)
        self.step_no += 1
        return {"loss": loss.detach(), "loss_ce": loss_ce.detach(), "loss_aux_img": loss_aux_img.detach(),
                "loss_aux_eeg": loss_aux_eeg.detach(), "loss_reg": loss_reg.detach(), "alpha": alpha.detach(),
                "fused_logits": fused.detach()}

    def _train_step_fused(self, m, eeg, img, red, img1, img2, eeg1, eeg2, labels, dropout):
        """The same step with the `[B, K]` part -- fusion, the four loss terms and every gradient on the logits and the 12 fusion
        scalars -- as HIP launches (M:436-462); equal to the autograd form to fp32 rounding (tests/test_gpu_multimodal.py)."""
        fz, fus, dev = m.fusion, self.fus, self.device
        st = eeg._cur_stream()
        z_img = img.forward(img1.contiguous().float(), img2.contiguous().float(), train=dropout)
        eeg.forward(eeg1.contiguous().float(), eeg2.contiguous().float(), labels, train=dropout)
        z_eeg = eeg.a["logits"]
        B, K = z_eeg.shape
        w = self._lw.get((B, K))
        if w is None:
            f32 = dict(device=dev, dtype=torch.float32)
            w = dict(fused=torch.empty(B, K, **f32), alpha=torch.empty(B, **f32), losses=torch.zeros(5, **f32),
                     dfused=torch.empty(B, K, **f32), dai=torch.empty(B, K, **f32), dae=torch.empty(B, K, **f32),
                     dzi=torch.empty(B, K, **f32), dze=torch.empty(B, K, **f32), dtau=torch.zeros(2, **f32),
                     part=torch.empty((B + 127) // 128, 12, **f32), dprm=torch.zeros(12, **f32))
            self._lw[(B, K)] = w
        mode = fz.VALID_MODES.index(fz.mode)
        prm = fus.flat.index_select(0, self._prm_idx)
        lab = labels.to(torch.int64).contiguous()
        li, le, lr_ = self.lams
        call("eg_fuzzy_gate_fwd", ptr(z_img), ptr(z_eeg), ptr(prm), ptr(w["fused"]), ptr(w["alpha"]), B, K, mode, fz.eps_temp,
             fz.eps_log, fz.eps_div, st)
        call("eg_fusion_loop_loss", ptr(w["fused"]), ptr(z_img), ptr(z_eeg), ptr(lab), ptr(prm), ptr(w["losses"]), ptr(w["dfused"]),
             ptr(w["dai"]), ptr(w["dae"]), ptr(w["dtau"]), B, K, mode, fz.eps_temp, li, le, lr_, self.treg[0], self.treg[1],
             eeg.st_ptr if eeg.scaler_on else 0, st)
        call("eg_fuzzy_gate_bwd", ptr(z_img), ptr(z_eeg), ptr(prm), ptr(w["dfused"]), 0, ptr(w["dzi"]), ptr(w["dze"]), ptr(w["part"]),
             B, K, mode, fz.eps_temp, fz.eps_log, fz.eps_div, st)
        call("eg_reduce_partials", ptr(w["part"]), ptr(w["dprm"]), 12, w["part"].shape[0], 12, 0, st)
        w["dzi"] += w["dai"]
        w["dze"] += w["dae"]
        w["dprm"][:2] += w["dtau"]
        fus.grad.index_copy_(0, self._prm_idx, w["dprm"])        # (the flat buffer's 16-B padding slots stay zero)
        if red:
            red.on_fusion()
        if not m.freeze_gaze:
            img.backward(w["dzi"])
            if red:
                red.on_gaze()
        if not m.freeze_eeg:
            eeg.backward(glogits=w["dze"], prescaled=True, on_segment=(red.on_eeg_segment if red else None))
        if red:
            red.finish()
        self._optimizer_step(eeg, img)
        self.step_no += 1
        ls = w["losses"].clone()
        return {"loss": ls[0], "loss_ce": ls[1], "loss_aux_img": ls[2], "loss_aux_eeg": ls[3], "loss_reg": ls[4],
                "alpha": w["alpha"].clone(), "fused_logits": w["fused"].clone()}

    def _optimizer_step(self, eeg, img):
        m, st = self.model, eeg._cur_stream()
        sets = []
        if not m.freeze_eeg:
            sets.append(("eeg", eeg.fp, 1.0))
        if not m.freeze_gaze:
            sets.append(("gaze", img.fp, 1.0))
        sets.append(("fusion", self.fus, self.fusion_lr / self.encoder_lr))
        nblk = 512
        if self.sqpart is None:
            self.sqpart = torch.zeros(3 * nblk, device=self.device)
        self.sqpart.zero_()
        for i, (_, fp, _) in enumerate(sets):
            call("eg_grad_sqnorm", ptr(fp.grad), fp.total, ptr(self.sqpart) + 4 * i * nblk, nblk, st)
        call("eg_clip_coef", ptr(self.sqpart), 3 * nblk, float(self.max_norm or 0.0), eeg.st_ptr, st)
        if self.red:
            self.red.sync_flag(eeg.state_dev)
        for name, fp, mult in sets:
            mm, vv = self._moments(name, fp)
            call("eg_adamw_group", ptr(fp.flat), ptr(fp.grad), ptr(mm), ptr(vv), fp.total, self.betas[0], self.betas[1], self.eps,
                 self.wd, float(mult), eeg.st_ptr, st)
        c = eeg.scaler_cfg
        call("eg_scaler_update", eeg.st_ptr, c["growth"], c["backoff"], c["growth_interval"], st)

    @torch.no_grad()
    def evaluate(self, batches) -> Dict[str, float]:
        """validate() of the reference (:546-640): fused-logit argmax, macro metrics, mean alpha"""
        from .train_art import macro_metrics
        m = self.model
        m.eval()
        preds, labs, alphas, tot, n = [], [], [], 0.0, 0
        for img1, img2, eeg1, eeg2, labels in batches:
            B, _, T = eeg1.shape
            eeg, img = self._engines(B, T, img1.shape[-2], img1.shape[-1])
            z_img = img.forward(img1.contiguous().float(), img2.contiguous().float(), train=False).clone()
            eeg.forward(eeg1.contiguous().float(), eeg2.contiguous().float(), labels, train=False)
            fused, alpha, _ = m.fusion(z_img, eeg.a["logits"].clone())
            tot += float(F.cross_entropy(fused, labels))
            n += 1
            preds.append(fused.argmax(-1).cpu().numpy())
            labs.append(labels.cpu().numpy())
            alphas.append(alpha.cpu().numpy())
        yp, yt = np.concatenate(preds), np.concatenate(labs)
        out = {k.replace("eval/", ""): v for k, v in macro_metrics(yt, yp).items()}
        out.update(loss=tot / max(n, 1), alpha_mean=float(np.concatenate(alphas).mean()), alpha_std=float(np.concatenate(alphas).std()))
        return out


def synth_multimodal(n: int, C: int, T: int, F_: int, W_: int, num_classes: int, seed: int):
    """class-conditional synthetic pairs: EEG windows as data.synth_windows; images = a class-specific blob + noise"""
    from .data import synth_windows
    x1, x2, y = synth_windows(n, C, T, num_classes, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    yy, xx = torch.meshgrid(torch.arange(F_, dtype=torch.float32), torch.arange(W_, dtype=torch.float32), indexing="ij")
    imgs = []
    for who in range(2):
        cy = (y.float() + 0.5) / num_classes * F_
        cx = torch.full_like(cy, W_ / 2.0) + (1 - 2 * who) * 1.5
        blob = torch.exp(-(((yy[None] - cy[:, None, None]) / (F_ / 8.0)) ** 2 + ((xx[None] - cx[:, None, None]) / (W_ / 4.0)) ** 2))
        imgs.append(blob + 0.3 * torch.randn(n, F_, W_, generator=g))
    return imgs[0], imgs[1], x1, x2, y


def build_from_config(config: Dict, device) -> MultimodalTrainer:
    """config sections as multimodal_fuzzy_fusion.yaml: gaze_encoder / eeg_encoder / fusion / training / data"""
    t, fz, ee = config["training"], config.get("fusion", {}), config["eeg_encoder"]
    dtype = "fp16" if t.get("fp16", False) else t.get("compute_dtype", "bf16")
    ncls = config["data"].get("num_classes", 3)
    torch.manual_seed(config.get("system", {}).get("seed", 42))
    eeg = DualEEGTransformer(in_channels=ee["in_channels"], num_classes=ncls, d_model=ee.get("d_model", 256),
                             num_layers=ee.get("num_layers", 6), num_heads=ee.get("num_heads", 8), d_ff=ee.get("d_ff", 1024),
                             dropout=ee.get("dropout", 0.1), max_len=config["data"]["window_size"] // 4,
                             use_spectrogram=ee.get("use_spectrogram", True), use_ibs=ee.get("use_ibs", False),
                             use_cross_attention=ee.get("use_cross_attention", True), compute_dtype=dtype)
    gaze = GazeCNNEncoder(num_classes=ncls, d_model=config.get("gaze_encoder", {}).get("d_model", 256), compute_dtype=dtype)
    fusion = FuzzyGatingFusion(num_classes=ncls, mode=fz.get("mode", "full"), eps_temp=fz.get("eps_temp", 0.1))
    model = MultimodalFusionModel(gaze, eeg, fusion, freeze_gaze=config.get("gaze_encoder", {}).get("freeze", False),
                                  freeze_eeg=ee.get("freeze", False))
    spe = t.get("steps_per_epoch", 1)
    return MultimodalTrainer(model, device, encoder_lr=t["encoder_learning_rate"], fusion_lr=t["fusion_learning_rate"],
                             weight_decay=t["weight_decay"], max_grad_norm=t.get("max_grad_norm"),
                             lambda_aux_img=t.get("lambda_aux_img", 0.3), lambda_aux_eeg=t.get("lambda_aux_eeg", 0.3),
                             lambda_reg=t.get("lambda_reg", 0.1), temp_reg_min=fz.get("temp_reg_min", 0.5),
                             temp_reg_max=fz.get("temp_reg_max", 5.0), warmup_steps=t.get("warmup_epochs", 0) * spe,
                             total_steps=t["epochs"] * spe, seed=config.get("system", {}).get("seed", 42))
