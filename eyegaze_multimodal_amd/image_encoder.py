"""ViT-free image branch of the multimodal fusion model (BASELINE configs[4] "image (2-D CNN backbone)").

The reference's gaze encoder is a timm ViT with downloaded weights (3_Models/backbones/late_fusion_vit.py:106-110) -- neither
exists offline.  The only in-tree 2-D CNN is SpectrogramTokenGenerator's (3_Models/backbones/dual_eeg_transformer.py:70-86):
Conv2d(1->32,3x3)+ReLU+MaxPool2, Conv2d(32->64,3x3)+ReLU, AdaptiveAvgPool(4,4), Linear(1024->2d)+ReLU+Dropout(0.1),
Linear(2d->d).  `GazeCNNEncoder` applies exactly that CNN (same HIP kernels, shared code: tokens.spec_cnn_*) to each player's
one-channel image [B, F, W] and classifies the pair with Linear(2d -> num_classes) on [f(img1) | f(img2)] -- the late-fusion
shape of the reference's gaze model (late_fusion_vit.py: per-player features, then a head on the pair).

Every arithmetic op is a kernel of libeyegaze_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import tokens
from ._lib import EG_BF16, EG_F16, EG_F32, call, ptr, rowmap
from .engine import Engine, FlatParams


class _CNN(nn.Module):
    def __init__(self, d_model: int):
        super().__init__()
        self.spec_conv = nn.Sequential(nn.Conv2d(1, 32, 3, padding=1), nn.ReLU(), nn.MaxPool2d(2),
                                       nn.Conv2d(32, 64, 3, padding=1), nn.ReLU(), nn.AdaptiveAvgPool2d((4, 4)))
        self.proj = nn.Sequential(nn.Linear(64 * 16, 2 * d_model), nn.ReLU(), nn.Dropout(0.1), nn.Linear(2 * d_model, d_model))


class GazeCNNEncoder(nn.Module):
    """img1, img2: f32 [B, F, W] (or [B, 1, F, W]) on a HIP device -> logits f32 [B, num_classes]."""

    def __init__(self, num_classes: int = 3, d_model: int = 256, compute_dtype: Optional[str] = None):
        super().__init__()
        from .dual_eeg_transformer import _resolve_dtype
        self.cnn = _CNN(d_model)
        self.head = nn.Linear(2 * d_model, num_classes)
        self.d_model, self.num_classes = d_model, num_classes
        self._dtype = _resolve_dtype(compute_dtype)
        self._flat = FlatParams(self)
        self._engines = {}

    def engine(self, B: int, F: int, W: int, device, state_dev: Optional[torch.Tensor] = None) -> "ImageEngine":
        L.lib()
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self._flat.ensure(device)
        key = (B, F, W, str(device), self._dtype)
        if key not in self._engines:
            self._engines[key] = ImageEngine(self, B, F, W, device, self._dtype, state_dev)
        return self._engines[key]


class ImageEngine(Engine):
    """Workspace + kernel sequencing of GazeCNNEncoder for a fixed (B, F, W); reuses Engine's GEMM / weight-gradient /
    staging / optimiser plumbing, and may share the eg_step_state of another engine (`state_dev`) so that ONE global clip
    coefficient, ONE loss scale and ONE overflow flag govern the whole multimodal model (GradScaler + clip_grad_norm_ over
    model.parameters(), train_multimodal_fuzzy_fusion.py:462-472)."""

    def __init__(self, model: GazeCNNEncoder, B: int, F: int, W: int, device, dtype: int, state_dev=None):
        if F % 8 or (W // 2) % 4 or W < 8:
            raise L.EgError(f"image {F}x{W} unsupported: F % 8 == 0 and floor(W/2) % 4 == 0 required")
        self.model, self.B, self.device, self.dtype = model, B, device, dtype
        self.cfg = SimpleNamespace(d_model=model.d_model, num_classes=model.num_classes)
        self.tdtype = {EG_BF16: torch.bfloat16, EG_F16: torch.float16, EG_F32: torch.float32}[dtype]
        self.es = 4 if dtype == EG_F32 else 2
        self.bk = 32 if dtype == EG_F32 else 64
        self.fp = model._flat
        self.stream, self.probes, self.probe_all = 0, {}, None
        self.cus = torch.cuda.get_device_properties(device).multi_processor_count if torch.device(device).type == "cuda" else 256
        self._shared_state = state_dev
        self._recording, self._plan = False, []
        self.scaler_on = False
        self.scaler_cfg = dict(init_scale=65536.0, growth=2.0, backoff=0.5, growth_interval=2000)
        self.a, self.w, self.g = {}, {}, {}
        d, nc = model.d_model, model.num_classes
        tokens.spec_cnn_alloc(self, 2 * B, F, W)
        self.a["feat"] = self._t(2 * B, d)                     # image order (b, player): a row pair IS the head's input row
        self.a["logits"] = self._t(B, nc, dtype=torch.float32)
        self.a["sloss"] = self._t(B, dtype=torch.float32)
        self.a["loss"] = self._t(1, dtype=torch.float32)
        if state_dev is None:
            self.state_dev = torch.zeros(L.STATE_WORDS, dtype=torch.int32, device=device)
            self.set_state(seed=0, lr=0.0, step=1, reset_scaler=2)
        else:
            self.state_dev = state_dev
        self.train_p01 = 0.0

    def _alloc_bwd(self):
        if self.g:
            return
        d, B = self.cfg.d_model, self.B
        g = self.g
        tokens.spec_cnn_alloc_bwd(self)
        g["dfeat"] = self._t(2 * B, d)
        g["dlogits"] = self._t(B, self.cfg.num_classes, dtype=torch.float32)
        self.tn_cap = 8 * 1024 * 1024
        g["partial"] = self._t(self.tn_cap, dtype=torch.float32)
        g["cspart"] = self._t(512 * 1024, dtype=torch.float32)

    def _pack_body(self):
        tokens.spec_cnn_pack(self, "cnn.")

    def forward(self, img1: torch.Tensor, img2: torch.Tensor, train: bool):
        B, sp = self.B, self.sp
        for x in (img1, img2):
            if x.dtype != torch.float32 or x.numel() != B * sp["F"] * sp["nfr"]:
                raise L.EgError(f"images must be f32 [{B}, {sp['F']}, {sp['nfr']}], got {tuple(x.shape)} {x.dtype}")
        self.stream = self._cur_stream()
        img = self.a["spimg"].view(B, 2, sp["F"], sp["nfr"])
        img[:, 0].copy_(img1.reshape(B, sp["F"], sp["nfr"]))
        img[:, 1].copy_(img2.reshape(B, sp["F"], sp["nfr"]))
        self.train_p01 = 0.1 if train else 0.0
        self.pack_params()
        d = self.cfg.d_model
        tokens.spec_cnn_forward(self, "cnn.", self.train_p01, None, ptr(self.a["feat"]), rowmap(d), rowmap(d), 0)
        call("eg_classifier_ce_fwd", ptr(self.a["feat"]), self.fp.p_ptr("head.weight"), self.fp.p_ptr("head.bias"), 0,
             ptr(self.a["logits"]), ptr(self.a["sloss"]), ptr(self.a["loss"]), B, 2 * d, self.cfg.num_classes, self.dtype, self.stream)
        return self.a["logits"]

    def backward(self, glogits: torch.Tensor):
        """glogits: f32 [B, num_classes], ALREADY multiplied by the loss scale when one is in use."""
        self._alloc_bwd()
        self.stream = self._cur_stream()
        B, d, g, fp = self.B, self.cfg.d_model, self.g, self.fp
        glogits = glogits.contiguous().float()
        call("eg_classifier_ce_bwd", ptr(self.a["feat"]), fp.p_ptr("head.weight"), ptr(self.a["logits"]), 0, 0, ptr(glogits),
             ptr(g["dlogits"]), ptr(g["dfeat"]), fp.g_ptr("head.weight"), fp.g_ptr("head.bias"), B, 2 * d, self.cfg.num_classes,
             0, 1.0, self.dtype, self.stream)
        sc01 = 1.0 / (1.0 - self.train_p01) if self.train_p01 > 0 else 1.0
        tokens.spec_cnn_backward(self, "cnn.", ptr(g["dfeat"]), rowmap(d), sc01)
