"""Drop-in `DualEEGTransformer` for MI355X.

Mirrors the reference module's interface (3_Models/backbones/dual_eeg_transformer.py:995-1021 ctor,
:1110-1253 forward, :1255-1371 loss helpers) and its state_dict key set / parameter registration order
(SURVEY.md §8b), so checkpoints and the reference's train loop work unchanged.  The sub-modules below only
HOLD parameters (standard torch layers are used so that default initialisation under a given
torch.manual_seed consumes the RNG exactly like the reference); their forward() is never called.
All arithmetic runs in libeyegaze_hip.so through `engine.Engine`.  No CPU fallback.
"""
from __future__ import annotations

import os
import weakref
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib as L
from .engine import Engine, FlatParams


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise L.EgError("parameter holder: computation happens in the HIP engine")


class _TemporalConv(_Holder):  # D:138-161
    def __init__(self, cin, d, k, s, layers):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv1d(cin, d, k, s, padding=k // 2)] +
                                   [nn.Conv1d(d, d, k, s, padding=k // 2) for _ in range(layers - 1)])


class _Spectrogram(_Holder):  # D:47-86
    def __init__(self, d, n_fft, hop, fs, bins):
        super().__init__()
        self.n_fft, self.hop_length, self.sampling_rate, self.freq_bins = n_fft, hop, fs, bins
        self.register_buffer("window", torch.hann_window(n_fft))
        self.spec_conv = nn.Sequential(nn.Conv2d(1, 32, 3, padding=1), nn.ReLU(), nn.MaxPool2d(2),
                                       nn.Conv2d(32, 64, 3, padding=1), nn.ReLU(), nn.AdaptiveAvgPool2d((4, 4)))
        self.proj = nn.Sequential(nn.Linear(1024, 2 * d), nn.ReLU(), nn.Dropout(0.1), nn.Linear(2 * d, d))


class _IBSMatrixGenerator(nn.Module):  # D:488-525, forward D:760-819 (no parameters)
    """Callable like the reference's generator: `gen(eeg1, eeg2) -> f32 [B, 6, n_feat, C, C]`, computed by the HIP
    kernels of the owning model's engine.  Inside the model's forward it is invoked only when hooks are registered
    (the analysis contract of 5_Metrics/eeg_metrics.py:195-205, 335-343); the hot path skips the extra copies."""

    def __init__(self, cin, fs, feature_type):
        super().__init__()
        self.in_channels, self.sampling_rate, self.feature_type = cin, fs, feature_type
        self.feature_indices = {"phase": [0, 1, 2, 5], "amplitude": [3, 4, 6]}.get(feature_type, list(range(7)))
        self.num_features = len(self.feature_indices)
        self._pending = None
        self._owner = None  # weakref to the DualEEGTransformer (not a registered sub-module)

    def forward(self, eeg1, eeg2):
        if self._pending is not None:
            out, self._pending = self._pending, None
            return out
        owner = self._owner() if self._owner is not None else None
        if owner is None:
            raise L.EgError("ibs_matrix_generator is detached from its model")
        if not eeg1.is_cuda:
            raise L.EgError("ibs_matrix_generator (HIP) needs device tensors; there is no CPU fallback")
        from . import tokens
        eeg1, eeg2 = eeg1.contiguous().float(), eeg2.contiguous().float()
        eng = owner.engine(eeg1.shape[0], eeg1.shape[2], eeg1.device)
        return tokens.ibs_matrices(owner, eng, eeg1, eeg2).clone()


class _IBSTokenizer(_Holder):  # D:837-877
    def __init__(self, cin, d, inorm, nfeat):
        super().__init__()
        self.num_tokens = 6 * nfeat
        if inorm:
            self.instance_norm = nn.InstanceNorm1d(cin * cin, affine=True)
        self.bottleneck = nn.Sequential(nn.Linear(cin * cin, 64), nn.GELU(), nn.Dropout(0.1), nn.Linear(64, d))
        self.type_embedding = nn.Parameter(torch.randn(1, self.num_tokens, d))
        nn.init.normal_(self.type_embedding, std=0.02)


class _IBSScalar(_Holder):  # D:189-222
    def __init__(self, d):
        super().__init__()
        self.proj = nn.Sequential(nn.Linear(28, 2 * d), nn.ReLU(), nn.Dropout(0.1), nn.Linear(2 * d, d))


class _PosEmbed(_Holder):  # A:100-107 (learned)
    def __init__(self, max_len, d):
        super().__init__()
        self.pos_embed = nn.Embedding(max_len, d)


class _MHA(_Holder):  # A:170-180
    def __init__(self, d, p):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(d, d) for _ in range(4))
        self.dropout = nn.Dropout(p)


class _FFN(_Holder):  # A:249-253
    def __init__(self, d, f, p):
        super().__init__()
        self.linear1 = nn.Linear(d, f)
        self.dropout = nn.Dropout(p)
        self.linear2 = nn.Linear(f, d)


class _Block(_Holder):  # A:279-286
    def __init__(self, d, f, p):
        super().__init__()
        self.mha = _MHA(d, p)
        self.drop1 = nn.Dropout(p)
        self.ln1 = nn.LayerNorm(d, eps=1e-5)
        self.ffn = _FFN(d, f, p)
        self.drop2 = nn.Dropout(p)
        self.ln2 = nn.LayerNorm(d, eps=1e-5)


class _Encoder(_Holder):  # A:303-306
    def __init__(self, d, n, f, p):
        super().__init__()
        self.layers = nn.ModuleList([_Block(d, f, p) for _ in range(n)])
        self.norm = nn.LayerNorm(d, eps=1e-5)


class _CrossAttn(_Holder):  # D:949-953
    def __init__(self, d, p):
        super().__init__()
        self.cross_attn = _MHA(d, p)
        self.norm = nn.LayerNorm(d)
        self.dropout = nn.Dropout(p)


class _SymFusion(_Holder):  # D:919-923
    def __init__(self, d):
        super().__init__()
        self.proj = nn.Linear(3 * d, d)


class _Cfg:
    """kwargs of the reference ctor plus derived counts (D:1035-1043, D:1198-1202)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.num_ibs_features = {"all": 7, "phase": 4, "amplitude": 3}.get(self.ibs_feature_type, 7)
        self.num_ibs_tokens = (6 * self.num_ibs_features if self.use_robust_ibs else 1) if self.use_ibs else 0


def _resolve_dtype(compute_dtype: Optional[str]) -> int:
    name = (compute_dtype or os.environ.get("EYEGAZE_DTYPE", "bf16")).lower()
    if name in ("bf16", "bfloat16"):
        return L.EG_BF16
    if name in ("f32", "fp32", "float32"):
        return L.EG_F32
    if name in ("fp16", "f16", "float16", "half"):
        return L.EG_F16
    raise ValueError(f"compute_dtype must be 'bf16', 'fp16' or 'f32', got {name!r}")


class _AuxLoss(torch.autograd.Function):
    """loss, d loss / d tokens from one of the eg_aux_* kernels; backward = upstream scalar x the stored gradients."""

    @staticmethod
    def forward(ctx, kind, temperature, labels, *toks):
        if not toks[0].is_cuda:
            raise L.EgError("auxiliary losses (HIP) need device tensors; there is no CPU fallback")
        toks = [t.detach().float().contiguous() for t in toks]
        B, D = toks[0].shape
        dev = toks[0].device
        st = torch.cuda.current_stream(dev).cuda_stream
        loss = torch.empty(1, device=dev)
        grads = [torch.empty_like(t) for t in toks]
        if kind == "sym":
            L.call("eg_aux_symmetry", L.ptr(toks[0]), L.ptr(toks[1]), L.ptr(loss), L.ptr(grads[0]), L.ptr(grads[1]), B, D, st)
        elif kind == "infonce":
            work = torch.empty(3 * B * D + 4 * B + 2 * B * B, device=dev)
            L.call("eg_aux_infonce", L.ptr(toks[0]), L.ptr(toks[1]), L.ptr(toks[2]), temperature, L.ptr(loss), L.ptr(grads[0]),
                   L.ptr(grads[1]), L.ptr(grads[2]), L.ptr(work), B, D, st)
        else:
            lab = labels.to(device=dev, dtype=torch.int64).contiguous()
            work = torch.empty(B * D + 5 * B + B * B + 4, device=dev)
            L.call("eg_aux_supcon", L.ptr(toks[0]), L.ptr(lab), temperature, L.ptr(loss), L.ptr(grads[0]), L.ptr(work), B, D, st)
        ctx.save_for_backward(*grads)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        return (None, None, None, *[g * t for t in ctx.saved_tensors])


class DualEEGTransformer(nn.Module):
    """Dual-stream window classifier, HIP engine behind the reference's module interface."""

    def __init__(self, in_channels: int = 62, num_classes: int = 3, d_model: int = 256, num_layers: int = 6,
                 num_heads: int = 8, d_ff: int = 1024, dropout: float = 0.1, max_len: int = 2048,
                 conv_kernel_size: int = 25, conv_stride: int = 4, conv_layers: int = 2, sampling_rate: int = 256,
                 use_spectrogram: bool = True, spec_n_fft: int = 128, spec_hop_length: int = 64, spec_freq_bins: int = 64,
                 use_robust_ibs: bool = True, use_ibs: bool = True, use_cross_attention: bool = True,
                 ibs_instance_norm: bool = True, ibs_feature_type: str = "all", compute_dtype: Optional[str] = None):
        super().__init__()
        self.cfg = _Cfg(in_channels=in_channels, num_classes=num_classes, d_model=d_model, num_layers=num_layers,
                        num_heads=num_heads, d_ff=d_ff, dropout=dropout, max_len=max_len, conv_kernel_size=conv_kernel_size,
                        conv_stride=conv_stride, conv_layers=conv_layers, sampling_rate=sampling_rate,
                        use_spectrogram=use_spectrogram, spec_n_fft=spec_n_fft, spec_hop_length=spec_hop_length,
                        spec_freq_bins=spec_freq_bins, use_robust_ibs=use_robust_ibs, use_ibs=use_ibs,
                        use_cross_attention=use_cross_attention, ibs_instance_norm=ibs_instance_norm,
                        ibs_feature_type=ibs_feature_type)
        c = self.cfg
        # attributes the reference exposes (D:1024-1043)
        self.d_model, self.in_channels, self.num_classes = d_model, in_channels, num_classes
        self.use_spectrogram, self.use_robust_ibs, self.use_ibs = use_spectrogram, use_robust_ibs, use_ibs
        self.use_cross_attention, self.ibs_feature_type = use_cross_attention, ibs_feature_type
        self.num_ibs_features, self.num_ibs_tokens = c.num_ibs_features, c.num_ibs_tokens
        # registration order = reference order (D:1046-1105)
        self.temporal_conv = _TemporalConv(in_channels, d_model, conv_kernel_size, conv_stride, conv_layers)
        if use_spectrogram:
            self.spectrogram_generator = _Spectrogram(d_model, spec_n_fft, spec_hop_length, sampling_rate, spec_freq_bins)
        if use_ibs:
            if use_robust_ibs:
                self.ibs_matrix_generator = _IBSMatrixGenerator(in_channels, sampling_rate, ibs_feature_type)
                self.ibs_matrix_generator._owner = weakref.ref(self)
                self.ibs_tokenizer = _IBSTokenizer(in_channels, d_model, ibs_instance_norm, c.num_ibs_features)
            else:
                self.ibs_generator = _IBSScalar(d_model)
            self.ibs_classifier = nn.Sequential(nn.Linear(d_model, d_model // 2), nn.ReLU(), nn.Dropout(0.3),
                                                nn.Linear(d_model // 2, num_classes))
        self.cls_token = nn.Parameter(torch.randn(1, 1, d_model))
        self.pos_embed = _PosEmbed(max_len, d_model)
        self.encoder = _Encoder(d_model, num_layers, d_ff, dropout)
        if use_cross_attention:
            self.cross_attn = _CrossAttn(d_model, dropout)
        self.symmetric_fusion = _SymFusion(d_model)
        self.classifier = nn.Sequential(nn.Linear(3 * d_model, d_model), nn.ReLU(), nn.Dropout(dropout),
                                        nn.Linear(d_model, num_classes))
        self.dropout = nn.Dropout(dropout)
        # engine state (not part of state_dict)
        self._dtype = _resolve_dtype(compute_dtype)
        self._flat = FlatParams(self)
        self._engines: Dict[tuple, Engine] = {}
        self._step_states: Dict[str, torch.Tensor] = {}
        self._state_ready: Dict[str, bool] = {}
        self._fwd_count = 0
        self._seed_base = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
        from . import ops
        self._op_handle = ops.register_owner(self)     # the registered operators (ops.py) find this module by handle

    # ------------------------------------------------------------------------------------------
    def engine(self, B: int, T: int, device: torch.device) -> Engine:
        L.lib()  # raises when the HIP library is missing
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:      # "cuda" and "cuda:0" must name ONE engine and ONE step state
            device = torch.device("cuda", torch.cuda.current_device())
        self._flat.ensure(device)
        key = (B, T, str(device), self._dtype)
        eng = self._engines.get(key)
        if eng is None:
            if len(self._engines) >= 4:  # bound the workspace held for rarely used shapes
                self._engines.pop(next(iter(self._engines)))
            eng = Engine(self, B, T, device, self._dtype, state_dev=self._state_for(device))
            if not self._state_ready.get(str(device)):
                # first engine on this device: it owns the initialisation of the shared step state (loss scaling on for fp16)
                eng.set_state(seed=0, lr=0.0, step=1, grad_scale=1.0, reset_scaler=(1 if eng.scaler_on else 2),
                              init_scale=eng.scaler_cfg["init_scale"])
                self._state_ready[str(device)] = True
            self._engines[key] = eng
        return eng

    def _state_for(self, device: torch.device) -> torch.Tensor:
        """ONE eg_step_state per (model, device), shared by the engines of every batch shape: the tail batch of an epoch steps
        the same AdamW count (bias corrections under fp16 come from the device's own opt_steps), the same loss scale and the
        same overflow history as the full batches (round-2 ADVICE: each engine used to carry a private state)."""
        key = str(device)
        st = self._step_states.get(key)
        if st is None:
            st = torch.zeros(L.STATE_WORDS, dtype=torch.int32, device=device)
            self._step_states[key] = st
            self._state_ready[key] = False
        return st

    def _run_forward(self, eng: Engine, eeg1, eeg2, labels, train: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        train = self.training if train is None else train
        self._fwd_count += 1
        if train:
            eng.set_state(seed=self._seed_base * 1000003 + self._fwd_count, lr=0.0, step=1)
        eng.forward(eeg1, eeg2, labels, train=train)
        a = eng.a
        out = {"logits": a["logits"].clone(), "cls1": a["cls1"].clone(), "cls2": a["cls2"].clone()}
        if self.cfg.use_ibs:
            out["ibs_logits"] = a["ibs_logits"].clone()
            out["ibs_token"] = a["ibs_pool_f"].clone()
        if labels is not None:
            out["loss_ce"] = a["loss"].clone().reshape(())
            if self.cfg.use_ibs:
                out["loss_ibs_cls"] = a["ibs_loss"].clone().reshape(())
        return out

    def _run_backward(self, shape, fwd_id: int, gouts: Dict[str, Optional[torch.Tensor]]):
        """autograd formula of eyegaze::dual_eeg_forward: the HIP backward over the engine's saved activations; returns one
        gradient per parameter (views of a copy of the flat gradient buffer, named_parameters() order)."""
        B, T, device = shape
        eng = self.engine(B, T, device)
        if self._fwd_count != fwd_id:
            raise L.EgError("backward() called after a newer forward(): the engine keeps one step's activations")
        g = {k: (v.contiguous().float() if (v is not None and v.numel() > 0) else None) for k, v in gouts.items()}
        one = lambda t: None if t is None else t.reshape(1)
        eng.backward(gloss=one(g.get("loss_ce")), gloss_ibs=one(g.get("loss_ibs_cls")), glogits=g.get("logits"),
                     gcls1=g.get("cls1"), gcls2=g.get("cls2"), gibs_logits=g.get("ibs_logits"), gibs_token=g.get("ibs_token"))
        from . import tokens
        tokens.fire_spec_backward_hooks(self, eng)
        fp = self._flat
        if eng.scaler_on:
            # fp16: the HIP backward carried every gradient at loss_scale x its value (so fp16 intermediates stay representable).
            # That scale is internal to the engine: what autograd hands to p.grad is the TRUE gradient, so the reference's own
            # loop (loss.backward(); clip_grad_norm_; torch AdamW, train_art.py:178-222) -- or an outer GradScaler -- sees
            # ordinary magnitudes; non-finite entries stay non-finite (inf / scale = inf) and are visible to the caller, and
            # the internal scale backs off / grows exactly as in the native step (eg_clip_coef flags, eg_scaler_update adapts).
            flat = fp.grad / eng.loss_scale_dev
            eng.check_overflow_and_update_scaler()
        else:
            flat = fp.grad.clone()
        return [flat[fp.offsets[n]:fp.offsets[n] + p.numel()].view(p.shape) for n, p in zip(fp.names, fp.params)]

    def forward(self, eeg1: torch.Tensor, eeg2: torch.Tensor, labels: Optional[torch.Tensor] = None) -> dict:
        """eeg1, eeg2: f32 [B, C, T] on a HIP device; labels: i64 [B] or None.  Returns the reference's dict
        (D:1232-1253): logits, cls1, cls2 (+ ibs_logits, ibs_token) (+ loss, loss_ce, loss_ibs_cls)."""
        if not eeg1.is_cuda:
            raise L.EgError("DualEEGTransformer (HIP) needs device tensors; there is no CPU fallback "
                            "(the CPU restatement lives in oracle/ and is test infrastructure)")
        if eeg1.shape != eeg2.shape or eeg1.dim() != 3 or eeg1.shape[1] != self.cfg.in_channels:
            raise L.EgError(f"expected two [B, {self.cfg.in_channels}, T] windows, got {tuple(eeg1.shape)} / {tuple(eeg2.shape)}")
        if labels is not None:
            labels = labels.to(device=eeg1.device, dtype=torch.int64).contiguous()
        eeg1, eeg2 = eeg1.contiguous().float(), eeg2.contiguous().float()
        eng = self.engine(eeg1.shape[0], eeg1.shape[2], eeg1.device)
        # The whole forward is ONE registered operator, eyegaze::dual_eeg_forward (ops.py); its autograd formula is the HIP
        # backward.  Analysis code (Grad-CAM) freezes the parameters and marks the INPUTS instead: the HIP path has no gradient
        # w.r.t. the raw windows (the reference never trains through them either), their .grad stays None, but the backward
        # still runs and feeds the hooks.
        self.engine(eeg1.shape[0], eeg1.shape[2], eeg1.device)       # (re)flattens the parameters before they are handed over
        from . import ops
        vals = torch.ops.eyegaze.dual_eeg_forward(eeg1, eeg2, labels, list(self._flat.params), self._op_handle, self.training)
        out = {k: v for k, v in zip(ops.OUTPUT_KEYS, vals) if v.numel() > 0}
        if labels is not None:
            out["loss"] = out["loss_ce"]
        return out

    # ------------------------------------------------------------------------------------------
    # auxiliary losses (D:1255-1371): HIP kernels (csrc/aux.hip) that return the loss together with its gradient w.r.t. the
    # [B, d] tokens; autograd scales that gradient by the upstream scalar and hands it to the HIP backward through
    # cls1 / cls2 / ibs_token.  Off by default in the reference config (dual_eeg_transformer.yaml:96-101).
    # ------------------------------------------------------------------------------------------
    def compute_symmetry_loss(self, cls1, cls2):
        return _AuxLoss.apply("sym", 0.0, None, cls1, cls2)

    def compute_ibs_alignment_loss(self, ibs_token, cls1, cls2, temperature: float = 0.07):
        return _AuxLoss.apply("infonce", float(temperature), None, ibs_token, cls1, cls2)

    def compute_ibs_contrastive_loss(self, ibs_tokens, labels, temperature: float = 0.07):
        return _AuxLoss.apply("supcon", float(temperature), labels, ibs_tokens)

    # ------------------------------------------------------------------------------------------
    # hooks the engine calls for the optional token families (spectrogram / synchrony tokens)
    # ------------------------------------------------------------------------------------------
    def _pack_extra(self, eng: Engine):
        from . import tokens
        tokens.pack(self, eng)

    def _extra_tokens_fwd(self, eng: Engine, eeg1, eeg2, train: bool):
        from . import tokens
        tokens.forward(self, eng, eeg1, eeg2, train)

    def _extra_tokens_bwd(self, eng: Engine, dseq):
        from . import tokens
        tokens.backward(self, eng, dseq)
