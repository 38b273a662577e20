"""Step engine: sequences the C-ABI kernels (include/eyegaze_hip.h) for one forward / backward / optimiser
step of the dual-stream window classifier.  torch is used for device memory, streams and (in ddp.py)
torch.distributed only; every arithmetic op on the path is a HIP kernel from libeyegaze_hip.so.

Data layout in HBM (NB = 2B windows: stream 1 = samples [0,B), stream 2 = [B,2B); M = NB*S token rows):
  xt      [NB, Tp, Cp]      channel-last, zero-padded input windows (k//2 in front)           compute dtype
  h0pad   [NB, R0, d]       conv-0 output, k//2 zero rows in front/behind (R0 = U*stride)      compute dtype
  seq/x_l [M, d]            token rows [CLS | IBS | spec | temporal], row-major                compute dtype
  qkv_l   [M, 3d]  ctx_l [M, d]  r1/r2 (pre-LN sums) [M, d]  hff_l [M, d_ff]                   compute dtype
  lse_l   [NB, H, S]  LN stats [M, 2]  logits / losses                                          fp32
  parameters, gradients, AdamW moments: ONE flat fp32 buffer each (16-B aligned segments, registration order)
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional

import torch

from . import _lib as L
from ._lib import EG_BF16, EG_F16, EG_F32, GemmDesc, GemmTNDesc, StepState, call, ptr, rowmap

# dropout site ids (any fixed numbering works: the mask depends on (seed, site, element index))
SITE_CONV0, SITE_CONV1, SITE_SPEC, SITE_IBSTOK, SITE_IBSGEN, SITE_CLS, SITE_IBSCLS = 1, 2, 3, 4, 5, 6, 7


def scramble_seed(seed: int) -> int:
    """splitmix64 finalizer: the 64-bit word whose halves eg_step_state.seed_lo / seed_hi carry to the dropout hash."""
    m = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def _reduce_blocks(n: int, splits: int) -> int:
    """workgroups eg_reduce_table spends on one entry (include/eyegaze_hip.h: EG_REDUCE_WIDE_SPLITS = 8)"""
    cols = n // 4
    return (cols + 255) // 256 if splits <= 8 else (cols + 7) // 8


def _layer_sites(l: int):
    b = 16 + 8 * l
    return dict(attn=b, drop1=b + 1, ffn_a=b + 2, ffn_b=b + 3, drop2=b + 4)


def _align(n: int, a: int) -> int:
    return (n + a - 1) // a * a


class FlatParams:
    """All parameters of a module as views of one flat fp32 buffer (+ a flat gradient buffer).
    Segment offsets are multiples of 4 floats so every kernel can use 16-B accesses."""

    def __init__(self, module: torch.nn.Module):
        self.module = module
        self.names: List[str] = []
        self.params: List[torch.nn.Parameter] = []
        self.offsets: Dict[str, int] = {}
        off = 0
        for n, p in module.named_parameters():
            self.names.append(n)
            self.params.append(p)
            self.offsets[n] = off
            off += _align(p.numel(), 4)
        self.total = off
        self.flat: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None

    def ensure(self, device: torch.device):
        """(Re)flattens when the module was moved / re-created since the last call."""
        ok = self.flat is not None and self.flat.device == device
        if ok:
            base = self.flat.data_ptr()
            for n, p in zip(self.names, self.params):
                if p.data_ptr() != base + 4 * self.offsets[n] or p.dtype != torch.float32:
                    ok = False
                    break
        if ok:
            return
        flat = torch.zeros(self.total, device=device, dtype=torch.float32)
        for n, p in zip(self.names, self.params):
            o = self.offsets[n]
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            p.data = flat[o:o + p.numel()].view(p.shape)
        self.flat = flat
        self.grad = torch.zeros(self.total, device=device, dtype=torch.float32)

    def p_ptr(self, name: str) -> int:
        return self.flat.data_ptr() + 4 * self.offsets[name]

    def g_ptr(self, name: str) -> int:
        return self.grad.data_ptr() + 4 * self.offsets[name]

    def grad_view(self, name: str, p: torch.nn.Parameter) -> torch.Tensor:
        o = self.offsets[name]
        return self.grad[o:o + p.numel()].view(p.shape)

    def has(self, name: str) -> bool:
        return name in self.offsets


class Engine:
    """Workspace + kernel sequencing for a fixed (B, T) shape."""
    LN_PARTIAL_BLOCKS = 2048      # rows of [2, d] the LayerNorm-backward scratch partial buffer holds (g["lnpart"])

    def __init__(self, model, B: int, T: int, device: torch.device, dtype: int, state_dev: Optional[torch.Tensor] = None):
        """state_dev: the model's shared eg_step_state.  Every engine of one model (one per batch shape: the ragged tail batch
        of an epoch gets its own workspace) must step ONE optimiser count, ONE loss scale and ONE overflow history; an engine
        given no state creates (and initialises) its own."""
        cfg = model.cfg
        self._shared_state = state_dev
        self.model, self.cfg, self.B, self.T, self.device, self.dtype = model, cfg, B, T, device, dtype
        self.tdtype = {EG_BF16: torch.bfloat16, EG_F16: torch.float16, EG_F32: torch.float32}[dtype]
        self.es = 4 if dtype == EG_F32 else 2
        self.bk = 32 if dtype == EG_F32 else 64
        # fp16 has 5 exponent bits: gradients are carried at loss_scale x their value (torch.cuda.amp.GradScaler semantics,
        # train_multimodal_fuzzy_fusion.py:435-472); the scale, the overflow flag and the step counter live in eg_step_state
        self.scaler_on = dtype == EG_F16 and os.environ.get("EYEGAZE_LOSS_SCALING", "1") != "0"
        self.scaler_cfg = dict(init_scale=65536.0, growth=2.0, backoff=0.5, growth_interval=2000)
        d, H = cfg.d_model, cfg.num_heads
        if d % H != 0 or d // H != 32:
            raise L.EgError(f"HIP attention core needs d_model/num_heads == 32 (got {d}/{H})")
        if cfg.conv_layers != 2:
            raise L.EgError("HIP path implements the reference's 2-layer temporal conv front-end (conv_layers=2)")
        if d % 64 != 0 or cfg.d_ff % 64 != 0:
            raise L.EgError("d_model and d_ff must be multiples of 64")
        k, s = cfg.conv_kernel_size, cfg.conv_stride
        pad = k // 2
        self.k, self.s, self.pad = k, s, pad
        self.NB = 2 * B
        self.C = cfg.in_channels
        self.Cp = _align(self.C, 8)
        self.T1 = (T + 2 * pad - k) // s + 1
        self.T2 = (self.T1 + 2 * pad - k) // s + 1
        self.J = (k + s - 1) // s
        self.K0 = _align(k * self.Cp, self.bk)                    # conv-0 GEMM depth (zero-padded)
        self.Tp = _align(max(T + 2 * pad, s * (self.T1 - 1) + self.K0 // self.Cp + 1), 8)
        self.U = (self.T1 + 2 * pad + s - 1) // s                   # backward-data rows per phase
        self.R0 = self.U * s                                        # padded conv-0 output rows per window
        self.RY = self.T2 + 2 * (self.J - 1)                        # padded dY rows per window (backward-data)
        self.n_ibs = cfg.num_ibs_tokens
        self.n_spec = self.C if cfg.use_spectrogram else 0
        self.off = 1 + self.n_ibs + self.n_spec
        self.S = self.off + self.T2
        if self.S > cfg.max_len:
            raise L.EgError(f"sequence length {self.S} exceeds max_len {cfg.max_len} of the positional table")
        if self.S > 160:
            raise L.EgError(f"sequence length {self.S} exceeds the attention core's limit of 160")
        self.M = self.NB * self.S
        self._want_attn_block = os.environ.get("EYEGAZE_ATTN_BLOCK", "1") != "0"
        self.fp: FlatParams = model._flat
        self.stream = 0
        self.probes = {}   # tag -> (start_event, end_event) recorded around that launch
        self.cus = torch.cuda.get_device_properties(device).multi_processor_count if device.type == "cuda" else 256
        # attention half of an encoder layer (q|k|v projection, attention core, out-proj + dropout + residual) as ONE launch with a
        # workgroup per window (csrc/attnblock.hip): 16-bit compute dtypes, d_model == 256, 8 heads, S <= 80 (set after S is known)
        self.attn_block = False
        # feed-forward pair as one launch (csrc/ffn.hip): 16-bit compute dtypes, d_model == 256, d_ff a multiple of 128
        self.fuse_ffn = (dtype != EG_F32 and cfg.d_model == 256 and cfg.d_ff % 128 == 0
                         and os.environ.get("EYEGAZE_FFN", "1") != "0")
        # LayerNorm backward: a block walks 8 rows per trip with TWO trips in flight (112 registers: four 256-thread blocks per CU).
        # All blocks must be resident at once -- a late block starts when an early one ends and doubles the launch -- so at most
        # 4 blocks per CU: 33 280 rows = 1024 blocks x 4.06 trips (round 2: 1040 x 4 with one trip in flight)
        env_nb = os.environ.get("EYEGAZE_LN_BLOCKS")
        trips = max(1, self.M // 8192)
        self.LN_BLOCKS = int(env_nb) if env_nb else min(self.LN_PARTIAL_BLOCKS, 4 * self.cus,
                                                        max(1, (self.M + 8 * trips - 1) // (8 * trips)))
        if not 1 <= self.LN_BLOCKS <= self.LN_PARTIAL_BLOCKS:
            # round 2: a sweep at 2080 blocks stored past the 2048-row partial buffer (GPU memory access fault); refuse, never clamp
            raise L.EgError(f"EYEGAZE_LN_BLOCKS={env_nb} is outside [1, {self.LN_PARTIAL_BLOCKS}] (rows of the LayerNorm-backward "
                            "partial buffer)")
        self.ln_nblk_cap = max(self.LN_BLOCKS, (self.M + 63) // 64)
        self.probe_all = None  # list of (start, end, flops) for every gemm_nt launch when bench.py enables it
        self.attn_block = bool(self._want_attn_block and dtype != EG_F32
                               and L.lib().eg_attn_block_ok(self.S, cfg.d_model, cfg.num_heads, dtype))
        # the two LayerNorms of an encoder layer as the tail of the launches that complete their input rows (eg_attn_block_fwd ->
        # norm1, eg_ffn_chain forward -> norm2) instead of two launches that re-read them; the statistics are summed in another order
        # than eg_layernorm_fwd's, so the step agrees with EYEGAZE_LN_FUSE=0 to rounding, not bit for bit
        self.ln_fuse = os.environ.get("EYEGAZE_LN_FUSE", "1") != "0"
        # norm1's backward and out_proj's backward-data product as one launch over 80-row tiles (eg_ln_bwd_proj): bit-identical to the
        # two launches (the gain / bias partials are grouped by tile instead of by LayerNorm block: equal to fp32 rounding)
        self.ln_proj = (dtype != EG_F32 and cfg.d_model == 256 and os.environ.get("EYEGAZE_LN_PROJ", "1") != "0")
        self.ln_proj_blocks = L.lib().eg_ln_bwd_proj_blocks(self.M) if self.ln_proj else 0
        self._alloc()
        self.packed_version = -1
        self._recording = False
        self._plan = []

    # ------------------------------------------------------------------------------------------
    def _t(self, *shape, dtype=None):
        return torch.zeros(*shape, device=self.device, dtype=dtype or self.tdtype)

    def _alloc(self):
        cfg, d, F, L_ = self.cfg, self.cfg.d_model, self.cfg.d_ff, self.cfg.num_layers
        NB, M, S, H, B = self.NB, self.M, self.S, self.cfg.num_heads, self.B
        f32 = torch.float32
        w = {}
        # packed parameters (compute dtype)
        w["conv0"] = self._t(d, self.K0)
        w["conv1"] = self._t(d, self.k * d)
        w["conv1T"] = self._t(self.s, d, self.J * d)
        w["pos"] = self._t(cfg.max_len, d)
        for l in list(range(L_)) + (["x"] if cfg.use_cross_attention else []):
            w[f"qkv{l}"] = self._t(3 * d, d)
            w[f"qkvT{l}"] = self._t(d, 3 * d)
            w[f"o{l}"] = self._t(d, d)
            w[f"oT{l}"] = self._t(d, d)
            w[f"bqkv{l}"] = self._t(3 * d, dtype=f32)
            if self.ln_proj and l != "x":             # out_proj^T in MFMA-fragment order (eg_pack_table mode 6) for eg_ln_bwd_proj
                w[f"oTf{l}"] = self._t(d * d)
            if self.attn_block and l != "x":          # eg_attn_block_fwd's fragment-ordered q|k|v and out-proj weights
                w[f"wqkvb{l}"] = self._t(3 * d * d)
                w[f"wob{l}"] = self._t(d * d)
            if l != "x":
                w[f"w1{l}"] = self._t(F, d)
                w[f"w1T{l}"] = self._t(d, F)
                w[f"w2{l}"] = self._t(d, F)
                w[f"w2T{l}"] = self._t(F, d)
                if self.fuse_ffn:       # the same four matrices in eg_ffn_chain's MFMA-fragment order (eg_pack_table modes 3-6)
                    for nm in ("w1f", "w2f", "w2Tf", "w1Tf"):
                        w[f"{nm}{l}"] = self._t(F * d)
        w["sf"] = self._t(d, 3 * d)
        w["sfT"] = self._t(3 * d, d)
        w["c0"] = self._t(d, 3 * d)
        w["c0T"] = self._t(3 * d, d)
        if cfg.use_ibs:
            w["i0"] = self._t(d // 2, d)
            w["i0T"] = self._t(d, _align(d // 2, self.bk))
        self.w = w
        a = {}
        a["xt"] = self._t(NB, self.Tp, self.Cp)
        a["h0pad"] = self._t(NB, self.R0, d)
        a["h1"] = self._t(NB * self.T2, d)
        a["x0"] = self._t(M, d)
        for l in range(L_):
            a[f"qkv{l}"] = self._t(M, 3 * d)
            a[f"lse{l}"] = self._t(NB, H, S, dtype=f32)
            a[f"ctx{l}"] = self._t(M, d)
            a[f"r1_{l}"] = self._t(M, d)
            a[f"st1_{l}"] = self._t(M, 2, dtype=f32)
            a[f"y1_{l}"] = self._t(M, d)
            a[f"hff{l}"] = self._t(M, F)
            if self.fuse_ffn:           # ReLU / dropout gate of the hidden rows, one bit per element (forward -> backward)
                a[f"gbits{l}"] = torch.zeros(L.gate_bits_bytes(M, F) // 8, device=self.device, dtype=torch.int64)
            a[f"r2_{l}"] = self._t(M, d)
            a[f"st2_{l}"] = self._t(M, 2, dtype=f32)
            a[f"x{l + 1}"] = self._t(M, d)
        a["stf"] = self._t(M, 2, dtype=f32)
        a["zn"] = self._t(M, d)
        if cfg.use_cross_attention:
            a["qkvx"] = self._t(M, 3 * d)
            a["lsex"] = self._t(NB, H, S, dtype=f32)
            a["ctxx"] = self._t(M, d)
            a["rx"] = self._t(M, d)
            a["stx"] = self._t(M, 2, dtype=f32)
            a["zc"] = self._t(M, d)
        a["cls1"] = self._t(B, d, dtype=f32)
        a["cls2"] = self._t(B, d, dtype=f32)
        a["comb"] = self._t(B, 3 * d)
        a["zf"] = self._t(B, 3 * d)
        a["hcl"] = self._t(B, d)
        a["logits"] = self._t(B, cfg.num_classes, dtype=f32)
        a["sloss"] = self._t(B, dtype=f32)
        a["loss"] = self._t(1, dtype=f32)
        if cfg.use_ibs:
            a["ibs_pool_f"] = self._t(B, d, dtype=f32)
            a["ibs_pool"] = self._t(B, d)
            a["hib"] = self._t(B, d // 2)
            a["ibs_logits"] = self._t(B, cfg.num_classes, dtype=f32)
            a["ibs_sloss"] = self._t(B, dtype=f32)
            a["ibs_loss"] = self._t(1, dtype=f32)
        self.a = a
        # backward temporaries (allocated lazily on the first backward)
        self.g: Dict[str, torch.Tensor] = {}
        # device-resident step state (eg_step_state); the host publishes a step's scalars as kernel arguments
        if self._shared_state is not None:      # initialised once, by whoever created it (DualEEGTransformer._state_for)
            self.state_dev = self._shared_state
        else:
            self.state_dev = torch.zeros(L.STATE_WORDS, dtype=torch.int32, device=self.device)
            self.set_state(seed=0, lr=0.0, step=1, grad_scale=1.0, reset_scaler=(1 if self.scaler_on else 2),
                           init_scale=self.scaler_cfg["init_scale"])

    def _alloc_bwd(self):
        if self.g:
            return
        d, F, M, B, NB = self.cfg.d_model, self.cfg.d_ff, self.M, self.B, self.NB
        g = {}
        for n in ("dzA", "dzB", "dr", "drm", "dctx", "dy1"):
            g[n] = self._t(M, d)
        g["dqkv"] = self._t(M, 3 * d)
        g["dh"] = self._t(M, F)
        g["dzf"] = self._t(B, 3 * d)
        g["dcomb"] = self._t(B, 3 * d)
        g["dhcl"] = self._t(B, d)
        g["dlogits"] = self._t(B, self.cfg.num_classes, dtype=torch.float32)
        if self.cfg.use_ibs:
            g["dhib"] = self._t(B, d // 2)
            g["dibs_pool"] = self._t(B, d)
            g["dibs_logits"] = self._t(B, self.cfg.num_classes, dtype=torch.float32)
        g["dy1pad"] = self._t(NB, self.RY, d)
        g["dh0pad"] = self._t(NB, self.R0, d)
        g["possum"] = self._t(self.S, d, dtype=torch.float32)
        g["one"] = torch.ones(1, device=self.device, dtype=torch.float32)
        # TN split-K partials: sized for the largest product (conv-1 weights / FFN)
        self.tn_cap = 24 * 1024 * 1024  # floats (96 MB)
        g["partial"] = self._t(self.tn_cap, dtype=torch.float32)
        g["lnpart"] = self._t(self.LN_PARTIAL_BLOCKS * 2 * max(d, 8), dtype=torch.float32)
        g["cspart"] = self._t(512 * max(3 * d, F), dtype=torch.float32)
        self.g = g

    # ------------------------------------------------------------------------------------------
    def set_state(self, seed: int, lr: float, step: int, grad_scale: float = 1.0, beta1=0.9, beta2=0.999,
                  reset_scaler: int = 0, init_scale: float = 65536.0, use_dev_t: bool = False):
        """Publishes this step's host scalars.  They travel as kernel ARGUMENTS of a one-thread launch on the current
        stream (eg_set_step_state), so the host may run any number of steps ahead: a queued step can never observe a later
        step's seed / lr / bias corrections (a pinned staging buffer re-used per step could be overwritten before its copy ran).
        reset_scaler: 0 keep the device's loss-scaling words, 1 enable dynamic loss scaling at init_scale, 2 disable."""
        seed = scramble_seed(seed)     # consecutive step seeds must not share their low / high words (common.h: eg_hash)
        # with loss scaling a step may be skipped on the device: the device's own count of taken steps feeds the bias corrections
        use_dev_t = bool(use_dev_t) or self.scaler_on
        call("eg_set_step_state", self.state_dev.data_ptr(), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, float(lr),
             1.0 - beta1 ** step, 1.0 - beta2 ** step, float(grad_scale), int(reset_scaler), float(init_scale),
             int(use_dev_t), self._cur_stream())

    def reset_scaler(self, init_scale: float = 65536.0, growth: float = 2.0, backoff: float = 0.5, growth_interval: int = 2000):
        """(re)starts dynamic loss scaling with GradScaler's parameters (fp16 engines only)"""
        self.scaler_cfg = dict(init_scale=init_scale, growth=growth, backoff=backoff, growth_interval=growth_interval)
        self.scaler_on = True
        self.set_state(seed=0, lr=0.0, step=1, reset_scaler=1, init_scale=init_scale)

    def check_overflow_and_update_scaler(self):
        """autograd path at fp16: flags a non-finite gradient norm (eg_step_state.found_inf) and adapts the internal loss scale,
        as the native optimiser step does between eg_clip_coef and eg_scaler_update -- without touching parameters."""
        self._alloc_bwd()
        self.stream = self._cur_stream()
        nblk = 1024
        if "sqpart" not in self.g:
            self.g["sqpart"] = self._t(nblk, dtype=torch.float32)
        call("eg_grad_sqnorm", ptr(self.fp.grad), self.fp.total, ptr(self.g["sqpart"]), nblk, self.stream)
        call("eg_clip_coef", ptr(self.g["sqpart"]), nblk, 0.0, self.st_ptr, self.stream)
        c = self.scaler_cfg
        call("eg_scaler_update", self.st_ptr, c["growth"], c["backoff"], c["growth_interval"], self.stream)

    @property
    def loss_scale_dev(self) -> torch.Tensor:
        """the device-resident loss scale as a 1-element fp32 view of eg_step_state (word 8)"""
        return self.state_dev.view(torch.float32)[8:9]

    def read_state(self) -> StepState:
        host = self.state_dev.cpu()
        st = StepState()
        C.memmove(C.addressof(st), host.data_ptr(), C.sizeof(st))
        return st

    @property
    def st_ptr(self) -> int:
        return self.state_dev.data_ptr()

    def _cur_stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    # ------------------------------------------------------------------------------------------
    # thin wrappers
    # ------------------------------------------------------------------------------------------
    def gemm(self, A, W, Cout, M, N, K, *, a=None, c=None, r=None, p=None, ldw=None, bias=0, residual=0, gate=0,
             out_pre=0, act=0, drop1=(0.0, 0), drop2=(0.0, 0), gate_scale=1.0, tag=None, seg=(0, 0)):
        probe = self.probes.get(tag) if tag else None
        dsc = self._gemm_desc(A, W, Cout, M, N, K, a, c, r, p, ldw, bias, residual, gate, out_pre, act, drop1, drop2,
                              gate_scale, seg)
        if self.probe_all is not None:      # bench.py: HIP events around EVERY eg_gemm_nt launch of the timed region
            probe = self._probe_pair()
            self.probe_all.append((probe[0], probe[1], 2.0 * M * N * K,
                                   self._gemm_bytes(M, N, K, a, seg, residual, gate, out_pre, Cout), (M, N, K),
                                   L.lib().eg_gemm_nt_route(C.byref(dsc))))
        if probe:
            probe[0].record(torch.cuda.current_stream(self.device))
        call("eg_gemm_nt", C.byref(dsc), self.stream)
        if probe:
            probe[1].record(torch.cuda.current_stream(self.device))

    def qkv_proj(self, x, l):
        """q|k|v = x W^T + b (A:203-205), fused over the three projections (the row-stream GEMM at K = 256)"""
        M, d = self.M, self.cfg.d_model
        self.gemm(ptr(x), ptr(self.w[f"qkv{l}"]), ptr(self.a[f"qkv{l}"]), M, 3 * d, d, bias=ptr(self.w[f"bqkv{l}"]))

    def attn_block_fwd(self, x, l, p, sites, ln=None):
        """eg_attn_block_fwd (csrc/attnblock.hip): A:202-213 + the residual of A:292-293 for encoder layer l in one launch"""
        a, w, fp, d = self.a, self.w, self.fp, self.cfg.d_model
        dsc = L.AttnBlockDesc()
        dsc.x, dsc.wqkv_frag, dsc.wo_frag = ptr(x), ptr(w[f"wqkvb{l}"]), ptr(w[f"wob{l}"])
        dsc.bqkv, dsc.bo = ptr(w[f"bqkv{l}"]), fp.p_ptr(f"encoder.layers.{l}.mha.out_proj.bias")
        dsc.qkv, dsc.ctx, dsc.lse, dsc.r1 = ptr(a[f"qkv{l}"]), ptr(a[f"ctx{l}"]), ptr(a[f"lse{l}"]), ptr(a[f"r1_{l}"])
        dsc.state = self.st_ptr
        dsc.NB, dsc.S, dsc.d_model, dsc.num_heads, dsc.dtype = self.NB, self.S, d, self.cfg.num_heads, self.dtype
        dsc.attn_drop_p, dsc.attn_drop_site = p, sites["attn"]
        dsc.out_drop_p, dsc.out_drop_site = p, sites["drop1"]
        if ln is not None:                  # (gain name, output rows, statistics): norm1 in the same launch
            dsc.ln_gamma, dsc.ln_beta = fp.p_ptr(ln[0] + ".weight"), fp.p_ptr(ln[0] + ".bias")
            dsc.ln_out, dsc.ln_stats = ptr(ln[1]), ptr(ln[2])
        probe = None
        if self.probe_all is not None:      # bench.py: timed like the eg_gemm_nt launches, as its own kernel (route 8)
            probe = self._probe_pair()
            M, S, H = self.M, self.S, self.cfg.num_heads
            flops = 2.0 * M * d * 3 * d + 2.0 * M * d * d + 4.0 * self.NB * H * S * S * (d // H)
            nbytes = self.es * (M * d * 3 + M * 3 * d + 4 * d * d) + 4 * (self.NB * H * S + 4 * d)    # x, ctx, r1 | qkv | weights | lse, biases
            if ln is not None:
                nbytes += self.es * M * d + 8 * M + 8 * d                                             # norm1 rows, statistics, gain / bias
            self.probe_all.append((probe[0], probe[1], flops, float(nbytes), (M, 3 * d, d), 8))
            probe[0].record(torch.cuda.current_stream(self.device))
        call("eg_attn_block_fwd", C.byref(dsc), self.stream)
        if probe:
            probe[1].record(torch.cuda.current_stream(self.device))

    def _probe_pair(self):
        """A (start, end) event pair for a timed launch: from bench.py's pre-created pool when there is one -- creating a
        hipEvent costs the host far more than recording one, and fresh events inside the timed region made short runs host-bound."""
        pool = getattr(self, "probe_pool", None)
        if pool:
            return pool.pop()
        return (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def ffn(self, A, W1f, W2f, H, Cout, M, F, *, bias1=0, bias2=0, act1=0, residual=0, gate=0, bits_out=0, bits_in=0,
            drop_h=(0.0, 0), drop_c1=(0.0, 0), drop_c2=(0.0, 0), gate_scale=1.0, ln=None):
        """eg_ffn_chain: H = epi1(A W1^T), C = epi2(H W2^T) in one launch (weights in fragment order)."""
        d = self.cfg.d_model
        dsc = L.FfnDesc()
        dsc.A, dsc.W1, dsc.W2, dsc.H, dsc.C = A, W1f, W2f, H, Cout
        dsc.bias1, dsc.bias2, dsc.gate, dsc.residual = bias1 or None, bias2 or None, gate or None, residual or None
        dsc.gate_bits_out, dsc.gate_bits_in = bits_out or None, bits_in or None
        dsc.state = self.st_ptr
        dsc.lda, dsc.ldh, dsc.ldc, dsc.ldg, dsc.ldr = d, F, d, F, d
        dsc.M, dsc.F, dsc.act1, dsc.dtype = M, F, act1, self.dtype
        dsc.drop_h_p, dsc.drop_h_site = drop_h
        dsc.drop_c1_p, dsc.drop_c1_site = drop_c1
        dsc.drop_c2_p, dsc.drop_c2_site = drop_c2
        dsc.gate_scale = gate_scale
        if ln is not None:                  # (gain name, output rows, statistics): norm2 in the same launch
            dsc.ln_gamma, dsc.ln_beta = self.fp.p_ptr(ln[0] + ".weight"), self.fp.p_ptr(ln[0] + ".bias")
            dsc.ln_out, dsc.ln_stats = ptr(ln[1]), ptr(ln[2])
        probe = None
        if self.probe_all is not None:      # bench.py: timed like the eg_gemm_nt launches, as its own kernel (route 4)
            probe = self._probe_pair()
            es = self.es
            nbytes = es * (M * d * (2 + (1 if residual and residual != A else 0)) + M * F * (1 + (1 if gate else 0)) + 2 * F * d) \
                + (M * F // 8 if (bits_out or bits_in) else 0) + 4 * (F + d) + ((es * M * d + 8 * M + 8 * d) if ln is not None else 0)
            self.probe_all.append((probe[0], probe[1], 4.0 * M * F * d, float(nbytes), (M, F, d), 4))
            probe[0].record(torch.cuda.current_stream(self.device))
        call("eg_ffn_chain", C.byref(dsc), self.stream)
        if probe:
            probe[1].record(torch.cuda.current_stream(self.device))

    def _probs_hook(self, drop_module, qkv, lse, kv_shift):
        """Analysis contract (5_Metrics/eeg_metrics.py:433-452): a forward hook on an attention-dropout module receives
        the probabilities [B, H, S, S] as its input, once per stream / direction in the reference's call order.  Only
        runs when such a hook is registered; the hook's return value does not feed back into the HIP path."""
        if not (drop_module._forward_hooks or drop_module._forward_pre_hooks):
            return
        NB, B, S, H = self.NB, self.B, self.S, self.cfg.num_heads
        probs = torch.empty(NB, H, S, S, device=self.device, dtype=torch.float32)
        call("eg_attention_probs", ptr(qkv), ptr(lse), ptr(probs), NB, S, H, kv_shift, self.dtype, self.stream)
        was = drop_module.training
        drop_module.training = False      # the module call is only the hook carrier: identity, no torch RNG use
        try:
            drop_module(probs[:B])
            drop_module(probs[B:])
        finally:
            drop_module.training = was

    def _gemm_bytes(self, M, N, K, a, seg, residual, gate, out_pre, Cout=1) -> float:
        """Algorithmic HBM bytes of one gemm_nt launch: every distinct operand element read once, every output element
        written once (overlapping conv rows count once; the weights count once)."""
        es = self.es
        if a is not None and a.rows_per_group > 0 and not seg[0]:
            groups = M // a.rows_per_group
            a_elems = groups * ((a.rows_per_group - 1) * min(a.row_stride, K) + K)
        elif seg[0]:                       # segmented rows (spectrogram conv): K elements per row drawn from K/seg_len runs
            a_elems = M * K if a is None else min(M * K, M * max(a.row_stride, 1) + K)
        else:
            a_elems = M * K
        outs = (1 if Cout else 0) + (1 if out_pre else 0)
        ins = (1 if residual else 0) + (1 if gate else 0)
        return float(es * (a_elems + N * K + (outs + ins) * M * N) + 4 * N)

    def _gemm_desc(self, A, W, Cout, M, N, K, a, c, r, p, ldw, bias, residual, gate, out_pre, act, drop1, drop2, gate_scale,
                   seg=(0, 0)):
        dsc = GemmDesc()
        dsc.a_seg_len, dsc.a_seg_stride = seg
        dsc.A, dsc.W, dsc.C = A, W, Cout or None
        dsc.bias, dsc.residual, dsc.gate, dsc.out_pre = bias or None, residual or None, gate or None, out_pre or None
        dsc.state = self.st_ptr
        dsc.a = a or rowmap(K)
        dsc.c = c or rowmap(N)
        dsc.r = r or dsc.c
        dsc.p = p or dsc.c
        dsc.M, dsc.N, dsc.K, dsc.ldw = M, N, K, ldw or K
        dsc.act, dsc.dtype = act, self.dtype
        dsc.drop1_p, dsc.drop1_site = drop1
        dsc.drop2_p, dsc.drop2_site = drop2
        dsc.gate_scale = gate_scale
        return dsc

    def wgrad(self, dY, X, out_w, M, N, K, *, y=None, x=None, out_b=0, conv=None, split_out=None, x_tile_stride=0,
              conv2d=None, linear=None):
        """dW = dY^T X (+ db = colsum dY).
        linear=[prefix, ...]: the product feeds N/len(linear) rows to each `prefix.weight` / `prefix.bias`; when those
        parameters are laid out back to back in the flat buffer (they are: registration order) the bias sums are fused
        into the GEMM launch and ONE reduce writes every weight and bias gradient.
        conv / conv2d: tap-major partials are un-permuted into the parameter layout; out_b via a column-sum launch."""
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        fused = False
        if linear is not None:
            P = N // len(linear)
            base = self.fp.offsets[linear[0] + ".weight"]
            fused = P % 4 == 0 and all(self.fp.offsets[n + ".weight"] == base + i * (P * K + P) and
                                       self.fp.offsets[n + ".bias"] == base + i * (P * K + P) + P * K
                                       for i, n in enumerate(linear))
        slab = N * K + (N if fused else 0)
        # one resident round: 256 CUs x 3 workgroups; rounding the split count UP would leave a nearly empty second round
        splits = max(1, min((M + 127) // 128, max(1, 768 // tiles), self.tn_cap // slab))
        # big products on 16-bit operands: 256 x 256 tiles, one 512-thread workgroup per CU (conv-1: 25 tiles x 10 row splits)
        big = (self.dtype != EG_F32 and N % 256 == 0 and K % 256 == 0 and not x_tile_stride and M * N * K >= (1 << 34)
               and os.environ.get("EYEGAZE_TN256", "1") != "0")
        if big:
            t256 = (N // 256) * (K // 256)
            splits = max(1, min((M + 63) // 64, max(1, self.cus // t256), self.tn_cap // slab))
        dsc = GemmTNDesc()
        dsc.tile = 256 if big else 128
        dsc.dY, dsc.X, dsc.partial = dY, X, ptr(self.g["partial"])
        dsc.y = y or rowmap(N)
        dsc.x = x or rowmap(K)
        dsc.M, dsc.N, dsc.K, dsc.splits, dsc.dtype = M, N, K, splits, self.dtype
        dsc.x_tile_stride = x_tile_stride
        if fused:
            dsc.part_rows, dsc.has_bias = N // len(linear), 1
        call("eg_gemm_tn", C.byref(dsc), self.stream)
        pp = ptr(self.g["partial"])
        if fused:
            call("eg_reduce_partials", pp, self.fp.g_ptr(linear[0] + ".weight"), slab, splits, slab, 0, self.stream)
            return
        if linear is not None:  # non-contiguous parameters: per-parameter reduces + a column-sum launch
            P = N // len(linear)
            split_out = [(self.fp.g_ptr(n + ".weight"), i * P, P) for i, n in enumerate(linear)]
            out_b = [(self.fp.g_ptr(n + ".bias"), i * P, P) for i, n in enumerate(linear)]
        if conv2d is not None:
            call("eg_unpack_conv2d_wgrad", pp, out_w, splits, conv2d[0], conv2d[1], self.stream)
        elif conv is not None:
            cin, kk, cp = conv
            call("eg_unpack_conv_wgrad", pp, out_w, splits, N, cin, kk, cp, K, self.stream)
        elif split_out is not None:
            for gp, row0, rows in split_out:
                call("eg_reduce_partials", pp + 4 * row0 * K, gp, rows * K, splits, N * K, 0, self.stream)
        else:
            call("eg_reduce_partials", pp, out_w, N * K, splits, N * K, 0, self.stream)
        if out_b:
            nblk = min(512, (M + 63) // 64)
            call("eg_colsum", dY, dsc.y, M, N, ptr(self.g["cspart"]), nblk, self.dtype, self.stream)
            if isinstance(out_b, (list, tuple)):
                for gp, col0, cols in out_b:
                    call("eg_reduce_partials", ptr(self.g["cspart"]) + 4 * col0, gp, cols, nblk, N, 0, self.stream)
            else:
                call("eg_reduce_partials", ptr(self.g["cspart"]), out_b, N, nblk, N, 0, self.stream)

    # ------------------------------------------------------------------------------------------
    # grouped weight gradients of the encoder: ONE launch for all 4*L products, ONE reduce launch
    # ------------------------------------------------------------------------------------------
    GROUP_SPLITS = int(os.environ.get("EYEGAZE_GROUP_SPLITS", "5"))
    GROUP_MIN_ROWS = 4096   # below this the per-product launches (many splits) are the better shape

    def _wgrad_group_plan(self):
        if getattr(self, "_wg_plan", "unset") != "unset" and getattr(self, "_wg_key", None) == self.fp.grad.data_ptr():
            return self._wg_plan
        cfg, d, F, M, fp = self.cfg, self.cfg.d_model, self.cfg.d_ff, self.M, self.fp
        self._wg_key = fp.grad.data_ptr()
        self._wg_plan = None
        if M < self.GROUP_MIN_ROWS:
            return None
        probs = []
        for l in range(cfg.num_layers):
            pre = f"encoder.layers.{l}."
            probs += [([pre + "mha.out_proj"], f"dYo{l}", f"ctx{l}", d, d, d),
                      ([pre + "mha.q_proj", pre + "mha.k_proj", pre + "mha.v_proj"], f"dqkv{l}", f"x{l}", 3 * d, d, 3 * d),
                      ([pre + "ffn.linear2"], f"dYf{l}", f"hff{l}", d, F, d),
                      ([pre + "ffn.linear1"], f"dh{l}", f"y1_{l}", F, d, F)]
        def packed(names, N, K):                 # (weight, bias) pairs must sit back to back in the flat buffer
            P = N // len(names)
            base = fp.offsets[names[0] + ".weight"]
            return all(fp.offsets[n + ".weight"] == base + i * (P * K + P) and fp.offsets[n + ".bias"] == base + i * (P * K + P) + P * K
                       for i, n in enumerate(names))
        if not all(packed(names, N, K) for names, _, _, N, K, _ in probs):
            return None
        # the two weight gradients of the cross-attention block ride in the same launch (their own dY buffers, as the layers have)
        cx = "cross_attn.cross_attn."
        cross = [([cx + "out_proj"], "dYo_x", "ctxx", d, d, d),
                 ([cx + "q_proj", cx + "k_proj", cx + "v_proj"], "dqkv_x", "zn", 3 * d, d, 3 * d)]
        self._wg_cross = (cfg.use_cross_attention and os.environ.get("EYEGAZE_WGRAD_CROSS", "1") != "0"
                          and all(packed(names, N, K) for names, _, _, N, K, _ in cross))
        ncross = 0
        if self._wg_cross:
            probs += cross
            ncross = len(cross)
        g = self.g
        for l in range(cfg.num_layers):
            g[f"dYo{l}"], g[f"dYf{l}"] = self._t(M, d), self._t(M, d)
            g[f"dqkv{l}"], g[f"dh{l}"] = self._t(M, 3 * d), self._t(M, F)
        if self._wg_cross:
            g["dYo_x"], g["dqkv_x"] = self._t(M, d), self._t(M, 3 * d)
        # 16-bit dtypes: 256 x 256 tiles, one 512-thread workgroup per CU (72 tiles x 3 row splits = 216 blocks, one round);
        # otherwise 128 x 128 tiles, three 256-thread workgroups per CU (288 tiles x 5 splits)
        big = (self.dtype != EG_F32 and os.environ.get("EYEGAZE_TN256", "1") != "0"
               and all(N % 256 == 0 and K % 256 == 0 for _, _, _, N, K, _ in probs))
        splits = int(os.environ.get("EYEGAZE_GROUP_SPLITS256", "3")) if big else self.GROUP_SPLITS
        tile = 256 if big else 128
        total = sum(N * K + N for _, _, _, N, K, _ in probs)
        # data-parallel runs cut the launch in two pieces; with 256 x 256 tiles a piece has ~40 tiles, so it takes twice the
        # row splits to fill the chip (240 / 216 blocks) -- with the whole launch's 3 splits each piece ran on 45 % of the CUs
        # and the pair cost 0.42 ms more than the single launch
        Lr_ = cfg.num_layers
        piece_tiles = ((Lr_ - Lr_ // 2) * 12 + (4 if self._wg_cross else 0)) if big else 0
        splits_p = max(splits, self.cus // piece_tiles) if big and piece_tiles else splits
        smax = max(splits, splits_p)
        g["wg_partial"] = self._t(smax * total, dtype=torch.float32)
        ln_names = [f"encoder.layers.{l}.{n}" for l in range(cfg.num_layers) for n in ("ln1", "ln2")]
        if not all(fp.offsets[n + ".bias"] == fp.offsets[n + ".weight"] + d for n in ln_names):
            return None
        g["lnpart_all"] = self._t(len(ln_names) * self.ln_nblk_cap * 2 * d, dtype=torch.float32)
        self._ln_slot = {n: i for i, n in enumerate(ln_names)}
        ln_of = {}
        for i, n in enumerate(ln_names):
            ln_of.setdefault(int(n.split(".")[2]), []).append((i, n))
        dev = lambda arr: torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        offs, off = [], 0
        for names, dyn, xn, N, K, ldy in probs:
            offs.append(off)
            off += smax * (N * K + N)

        def tables(layers, with_cross=False, nsplit=None):
            """TN problem table + reduce table (weights, biases and the deferred LayerNorm gain / bias partials) of `layers`
            (+ the cross-attention block's two products); block ranges are relative to the tables' own launches.  Which launch
            a product rides in does not change its result AT EQUAL nsplit (same row split, same ordered sum); the data-parallel
            pieces run with more row splits than the single launch (splits_p vs splits), so their gradients differ from the
            single launch's by fp32 summation order (~2e-6 relative), each arrangement deterministic in itself."""
            nsplit = nsplit or splits
            sel = [4 * l + j for l in layers for j in range(4)]
            if with_cross:
                sel += [4 * cfg.num_layers + j for j in range(ncross)]
            lns = [e for l in layers for e in ln_of[l]]
            tp = (L.TNProblem * len(sel))()
            rt = (L.ReduceEntry * (len(sel) + len(lns)))()
            blk, rblk = 0, 0
            for e, r, pi in zip(tp, rt, sel):
                names, dyn, xn, N, K, ldy = probs[pi]
                slab = N * K + N
                base = ptr(g["wg_partial"]) + 4 * offs[pi]
                e.dY, e.X, e.partial = ptr(g[dyn]), ptr(self.a[xn]), base
                e.ldy, e.ldx, e.N, e.K, e.part_rows, e.has_bias, e.blk0 = ldy, K, N, K, N // len(names), 1, blk
                blk += ((N + tile - 1) // tile) * ((K + tile - 1) // tile) * nsplit
                r.partial, r.out, r.n, r.stride, r.splits, r.blk0 = base, fp.g_ptr(names[0] + ".weight"), slab, slab, nsplit, rblk
                rblk += _reduce_blocks(slab, nsplit)
            for k, (i, n) in enumerate(lns):   # deferred LayerNorm gain / bias partials ride in the same reduce launch
                r = rt[len(sel) + k]
                r.partial, r.out = ptr(g["lnpart_all"]) + 4 * i * self.ln_nblk_cap * 2 * d, fp.g_ptr(n + ".weight")
                r.n, r.stride, r.splits, r.blk0 = 2 * d, 2 * d, (self.ln_proj_blocks if (self.ln_proj and n.endswith(".ln1")) else self.LN_BLOCKS), rblk
                rblk += _reduce_blocks(2 * d, r.splits)
            return dict(tp=dev(tp), rt=dev(rt), n=len(sel), nr=len(sel) + len(lns), blocks=blk, rblocks=rblk, layers=list(layers),
                        splits=nsplit)

        Lr = cfg.num_layers
        whole = tables(range(Lr), with_cross=True)
        # data parallel: two pieces, so the gradient buckets of layers L-1 .. L/2 start their all-reduce while layers L/2-1 .. 0
        # are still in backward (one piece would hold every encoder bucket back until backward has finished)
        h = Lr // 2
        pieces = ([tables(range(h, Lr), with_cross=True, nsplit=splits_p), tables(range(0, h), nsplit=splits_p)]
                  if Lr >= 2 else [whole])
        self._wg_plan = dict(whole, splits=splits, pieces=pieces, split_layer=h,
                             entry="eg_gemm_tn_grouped256" if big else "eg_gemm_tn_grouped")
        return self._wg_plan

    def _wgrad_group_launch(self, piece=None):
        """piece: None = every encoder layer in one launch; else one of `_wg_plan['pieces']` (data-parallel runs)."""
        pl = self._wg_plan if piece is None else piece
        call(self._wg_plan["entry"], ptr(pl["tp"]), pl["n"], pl["blocks"], self.M, pl["splits"], self.dtype, self.stream)
        call("eg_reduce_table", ptr(pl["rt"]), pl["nr"], pl["rblocks"], self.stream)

    def ln_fwd(self, x, gname, y, stats):
        call("eg_layernorm_fwd", ptr(x), self.fp.p_ptr(gname + ".weight"), self.fp.p_ptr(gname + ".bias"), ptr(y),
             ptr(stats), self.M, self.cfg.d_model, self.dtype, self.stream)

    def ln_bwd_proj(self, dy, x, stats, gname, wfrag, dx, dx_drop, dC, d1=(0.0, 0), slot=None):
        """eg_ln_bwd_proj: LayerNorm backward + the backward-data product dC = dx_drop W^T in one launch (csrc/lnproj.hip)"""
        d = self.cfg.d_model
        nblk = self.ln_proj_blocks
        lp = ptr(self.g["lnpart"]) if slot is None else ptr(self.g["lnpart_all"]) + 4 * slot * self.ln_nblk_cap * 2 * d
        cap = self.LN_PARTIAL_BLOCKS if slot is None else self.ln_nblk_cap
        dsc = L.LnBwdProjDesc()
        dsc.dy, dsc.x, dsc.stats, dsc.gamma, dsc.W_frag = ptr(dy), ptr(x), ptr(stats), self.fp.p_ptr(gname + ".weight"), ptr(wfrag)
        dsc.dx, dsc.dx_drop, dsc.dC, dsc.partial, dsc.state = ptr(dx), ptr(dx_drop), ptr(dC), lp, self.st_ptr
        dsc.M, dsc.d_model, dsc.dtype, dsc.partial_capacity_blocks = self.M, d, self.dtype, cap
        dsc.drop1_p, dsc.drop1_site = d1
        call("eg_ln_bwd_proj", C.byref(dsc), self.stream)
        if slot is not None:
            return
        if self.fp.offsets[gname + ".bias"] == self.fp.offsets[gname + ".weight"] + d:
            call("eg_reduce_partials", lp, self.fp.g_ptr(gname + ".weight"), 2 * d, nblk, 2 * d, 0, self.stream)
        else:
            call("eg_reduce_partials", lp, self.fp.g_ptr(gname + ".weight"), d, nblk, 2 * d, 0, self.stream)
            call("eg_reduce_partials", lp + 4 * d, self.fp.g_ptr(gname + ".bias"), d, nblk, 2 * d, 0, self.stream)

    def ln_bwd(self, dy, x, stats, gname, dx, dx_drop=None, d1=(0.0, 0), d2=(0.0, 0), slot=None):
        """slot: index into the deferred gain/bias partial buffer (reduced by the grouped reduce at the end of backward)"""
        d = self.cfg.d_model
        nblk = self.LN_BLOCKS
        lp = ptr(self.g["lnpart"]) if slot is None else ptr(self.g["lnpart_all"]) + 4 * slot * self.ln_nblk_cap * 2 * d
        cap = self.LN_PARTIAL_BLOCKS if slot is None else self.ln_nblk_cap      # rows of [2, d] behind `lp`
        call("eg_layernorm_bwd", ptr(dy), ptr(x), ptr(stats), self.fp.p_ptr(gname + ".weight"), ptr(dx), ptr(dx_drop),
             lp, nblk, cap, self.M, d, self.dtype, d1[0], d1[1], d2[0], d2[1], self.st_ptr, self.stream)
        if slot is not None:
            return
        if self.fp.offsets[gname + ".bias"] == self.fp.offsets[gname + ".weight"] + d:   # (gain | bias) back to back
            call("eg_reduce_partials", lp, self.fp.g_ptr(gname + ".weight"), 2 * d, nblk, 2 * d, 0, self.stream)
        else:
            call("eg_reduce_partials", lp, self.fp.g_ptr(gname + ".weight"), d, nblk, 2 * d, 0, self.stream)
            call("eg_reduce_partials", lp + 4 * d, self.fp.g_ptr(gname + ".bias"), d, nblk, 2 * d, 0, self.stream)

    # ------------------------------------------------------------------------------------------
    # parameter staging
    # ------------------------------------------------------------------------------------------
    # table-driven staging: cast / transpose entries are recorded once and replayed as ONE launch per step
    def p_cast(self, src, dst, n):
        if self._recording:
            self._plan.append((src, dst, 1, n, 0, 0))

    def p_copy(self, src, dst, n):
        if self._recording:
            self._plan.append((src, dst, 1, n, 0, 2))

    def p_transpose(self, src, dst, R, Cc, ldd):
        if self._recording:
            self._plan.append((src, dst, R, Cc, ldd, 1))

    def p_frag(self, src, dst, R, Cc, mode, part=0):
        """fragment order of the fp32 parameter src [R, Cc]: eg_ffn_chain's (eg_pack_table modes 3-6) or eg_attn_block_fwd's
        (mode 7 with part = 0 / 1 / 2 for q / k / v_proj, mode 8 for out_proj)."""
        if self._recording:
            self._plan.append((src, dst, R, Cc, part, mode))

    def pack_params(self):
        key = (self.fp.flat.data_ptr(), self.stream)
        if getattr(self, "_plan_key", None) != key[0]:
            self._plan, self._recording = [], True
            self._pack_body()
            self._recording = False
            ents = (L.PackEntry * len(self._plan))()
            blk = 0
            for e, (src, dst, R, Cc, ldd, mode) in zip(ents, self._plan):
                nb = (((R + 31) // 32) * ((Cc + 31) // 32) if mode == 1 else (R * Cc) // 2048 if mode >= 3 else
                      (R * Cc + 1023) // 1024)
                e.src, e.dst, e.rows, e.cols, e.ldd, e.mode, e.blk0, e.nblk = src, dst, R, Cc, ldd, mode, blk, nb
                blk += nb
            raw = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8)
            self._plan_dev = raw.to(self.device)
            self._plan_n, self._plan_blocks, self._plan_key = len(self._plan), blk, key[0]
        else:
            self._pack_body()
        call("eg_pack_table", ptr(self._plan_dev), self._plan_n, self._plan_blocks, self.dtype, self.stream)

    def _pack_body(self):
        cfg, d, F, fp, w, dt, st = self.cfg, self.cfg.d_model, self.cfg.d_ff, self.fp, self.w, self.dtype, self.stream
        call("eg_pack_conv_weight", fp.p_ptr("temporal_conv.convs.0.weight"), ptr(w["conv0"]), d, self.C, self.k, self.Cp,
             self.K0, dt, st)
        call("eg_pack_conv_weight", fp.p_ptr("temporal_conv.convs.1.weight"), ptr(w["conv1"]), d, d, self.k, d, self.k * d,
             dt, st)
        call("eg_pack_convT_weight", fp.p_ptr("temporal_conv.convs.1.weight"), ptr(w["conv1T"]), d, d, self.k, self.s, dt, st)
        self.p_cast(fp.p_ptr("pos_embed.pos_embed.weight"), ptr(w["pos"]), cfg.max_len * d)

        def attn_pack(pre, l):
            for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                self.p_cast(fp.p_ptr(f"{pre}{n}.weight"), ptr(w[f"qkv{l}"]) + i * d * d * self.es, d * d)
                self.p_transpose(fp.p_ptr(f"{pre}{n}.weight"), ptr(w[f"qkvT{l}"]) + i * d * self.es, d, d, 3 * d)
                self.p_copy(fp.p_ptr(f"{pre}{n}.bias"), ptr(w[f"bqkv{l}"]) + 4 * i * d, d)
            self.p_cast(fp.p_ptr(f"{pre}out_proj.weight"), ptr(w[f"o{l}"]), d * d)
            self.p_transpose(fp.p_ptr(f"{pre}out_proj.weight"), ptr(w[f"oT{l}"]), d, d, d)
            if self.ln_proj and l != "x":
                self.p_frag(fp.p_ptr(f"{pre}out_proj.weight"), ptr(w[f"oTf{l}"]), d, d, 6)
            if self.attn_block and l != "x":
                for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                    self.p_frag(fp.p_ptr(f"{pre}{n}.weight"), ptr(w[f"wqkvb{l}"]), d, d, 7, part=i)
                self.p_frag(fp.p_ptr(f"{pre}out_proj.weight"), ptr(w[f"wob{l}"]), d, d, 8)

        for l in range(cfg.num_layers):
            pre = f"encoder.layers.{l}."
            attn_pack(pre + "mha.", l)
            self.p_cast(fp.p_ptr(pre + "ffn.linear1.weight"), ptr(w[f"w1{l}"]), F * d)
            self.p_transpose(fp.p_ptr(pre + "ffn.linear1.weight"), ptr(w[f"w1T{l}"]), F, d, F)
            self.p_cast(fp.p_ptr(pre + "ffn.linear2.weight"), ptr(w[f"w2{l}"]), d * F)
            self.p_transpose(fp.p_ptr(pre + "ffn.linear2.weight"), ptr(w[f"w2T{l}"]), d, F, d)
            if self.fuse_ffn:
                self.p_frag(fp.p_ptr(pre + "ffn.linear1.weight"), ptr(w[f"w1f{l}"]), F, d, 3)     # forward product 1
                self.p_frag(fp.p_ptr(pre + "ffn.linear2.weight"), ptr(w[f"w2f{l}"]), d, F, 5)     # forward product 2
                self.p_frag(fp.p_ptr(pre + "ffn.linear2.weight"), ptr(w[f"w2Tf{l}"]), d, F, 4)    # backward product 1 = linear2^T
                self.p_frag(fp.p_ptr(pre + "ffn.linear1.weight"), ptr(w[f"w1Tf{l}"]), F, d, 6)    # backward product 2 = linear1^T
        if cfg.use_cross_attention:
            attn_pack("cross_attn.cross_attn.", "x")
        self.p_cast(fp.p_ptr("symmetric_fusion.proj.weight"), ptr(w["sf"]), 3 * d * d)
        self.p_transpose(fp.p_ptr("symmetric_fusion.proj.weight"), ptr(w["sfT"]), d, 3 * d, d)
        self.p_cast(fp.p_ptr("classifier.0.weight"), ptr(w["c0"]), 3 * d * d)
        self.p_transpose(fp.p_ptr("classifier.0.weight"), ptr(w["c0T"]), d, 3 * d, d)
        if cfg.use_ibs:
            self.p_cast(fp.p_ptr("ibs_classifier.0.weight"), ptr(w["i0"]), (d // 2) * d)
            self.p_transpose(fp.p_ptr("ibs_classifier.0.weight"), ptr(w["i0T"]), d // 2, d, w["i0T"].shape[1])
        self.model._pack_extra(self)

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def forward(self, eeg1: torch.Tensor, eeg2: torch.Tensor, labels: Optional[torch.Tensor], train: bool):
        cfg, d, F, H = self.cfg, self.cfg.d_model, self.cfg.d_ff, self.cfg.num_heads
        B, NB, M, S, a, w, fp, es = self.B, self.NB, self.M, self.S, self.a, self.w, self.fp, self.es
        self.stream = self._cur_stream()
        st = self.stream
        p = cfg.dropout if train else 0.0
        p01 = 0.1 if train else 0.0
        self.train_flags = (p, p01, train)
        self.pack_params()
        for i, x in enumerate((eeg1, eeg2)):
            if x.dtype != torch.float32 or not x.is_contiguous() or tuple(x.shape) != (B, self.C, self.T):
                raise L.EgError(f"input windows must be contiguous f32 [{B},{self.C},{self.T}], got {tuple(x.shape)} {x.dtype}")
            call("eg_window_pack", ptr(x), ptr(a["xt"]) + i * B * self.Tp * self.Cp * es, B, self.C, self.T, self.Cp,
                 self.pad, self.Tp, self.dtype, st)
        # K1: conv0 as a GEMM over overlapping channel-last rows (D:154,171)
        self.gemm(ptr(a["xt"]), ptr(w["conv0"]), ptr(a["h0pad"]) + self.pad * d * es, NB * self.T1, d, self.K0,
                  a=rowmap(self.s * self.Cp, self.Tp * self.Cp, self.T1), c=rowmap(d, self.R0 * d, self.T1),
                  bias=fp.p_ptr("temporal_conv.convs.0.bias"), act=L.ACT_RELU, drop1=(p01, SITE_CONV0))
        # K2: conv1, epilogue writes token rows [off:] of the sequence with the positional rows added (D:158,171,174; A:120-126)
        self.gemm(ptr(a["h0pad"]), ptr(w["conv1"]), ptr(a["x0"]) + self.off * d * es, NB * self.T2, d, self.k * d,
                  a=rowmap(self.s * d, self.R0 * d, self.T2), c=rowmap(d, S * d, self.T2),
                  r=rowmap(d, 0, self.T2), p=rowmap(d), bias=fp.p_ptr("temporal_conv.convs.1.bias"), act=L.ACT_RELU,
                  drop1=(p01, SITE_CONV1), residual=ptr(w["pos"]) + self.off * d * es, out_pre=ptr(a["h1"]), tag="conv1_fwd")
        # CLS rows (D:1157) + pos row 0
        call("eg_rows_bcast_f32", fp.p_ptr("cls_token"), fp.p_ptr("pos_embed.pos_embed.weight"), ptr(a["x0"]), NB, S, d, 1,
             0, 1, self.dtype, st)
        self.model._extra_tokens_fwd(self, eeg1, eeg2, train)
        # encoder (A:292-295, 326-328), both streams batched (Siamese weights)
        for l in range(cfg.num_layers):
            pre, sites = f"encoder.layers.{l}.", _layer_sites(l)
            x = a[f"x{l}"]
            if self.attn_block:     # q|k|v projection + attention core + out-proj / dropout / residual: one launch, a workgroup per window
                self.attn_block_fwd(x, l, p, sites, ln=((pre + "ln1", a[f"y1_{l}"], a[f"st1_{l}"]) if self.ln_fuse else None))
                self._probs_hook(self.model.encoder.layers[l].mha.dropout, a[f"qkv{l}"], a[f"lse{l}"], 0)
            else:
                self.qkv_proj(x, l)
                call("eg_attention_fwd", ptr(a[f"qkv{l}"]), ptr(a[f"ctx{l}"]), ptr(a[f"lse{l}"]), NB, S, H, 0, self.dtype, p,
                     sites["attn"], self.st_ptr, st)
                self._probs_hook(self.model.encoder.layers[l].mha.dropout, a[f"qkv{l}"], a[f"lse{l}"], 0)
                self.gemm(ptr(a[f"ctx{l}"]), ptr(w[f"o{l}"]), ptr(a[f"r1_{l}"]), M, d, d, bias=fp.p_ptr(pre + "mha.out_proj.bias"),
                          drop1=(p, sites["drop1"]), residual=ptr(x))
            if not (self.attn_block and self.ln_fuse):
                self.ln_fwd(a[f"r1_{l}"], pre + "ln1", a[f"y1_{l}"], a[f"st1_{l}"])
            if self.fuse_ffn:       # linear1 -> ReLU -> dropout -> linear2 -> dropout x2 -> + residual in one launch (A:272, A:294)
                self.ffn(ptr(a[f"y1_{l}"]), ptr(w[f"w1f{l}"]), ptr(w[f"w2f{l}"]), ptr(a[f"hff{l}"]), ptr(a[f"r2_{l}"]), M, F,
                         bias1=fp.p_ptr(pre + "ffn.linear1.bias"), bias2=fp.p_ptr(pre + "ffn.linear2.bias"), act1=L.ACT_RELU,
                         residual=ptr(a[f"y1_{l}"]), drop_h=(p, sites["ffn_a"]), drop_c1=(p, sites["ffn_b"]),
                         drop_c2=(p, sites["drop2"]), bits_out=ptr(a[f"gbits{l}"]),
                         ln=((pre + "ln2", a[f"x{l + 1}"], a[f"st2_{l}"]) if self.ln_fuse else None))
            else:
                self.gemm(ptr(a[f"y1_{l}"]), ptr(w[f"w1{l}"]), ptr(a[f"hff{l}"]), M, F, d, bias=fp.p_ptr(pre + "ffn.linear1.bias"),
                          act=L.ACT_RELU, drop1=(p, sites["ffn_a"]))
                self.gemm(ptr(a[f"hff{l}"]), ptr(w[f"w2{l}"]), ptr(a[f"r2_{l}"]), M, d, F, bias=fp.p_ptr(pre + "ffn.linear2.bias"),
                          drop1=(p, sites["ffn_b"]), drop2=(p, sites["drop2"]), residual=ptr(a[f"y1_{l}"]))
            if not (self.fuse_ffn and self.ln_fuse):
                self.ln_fwd(a[f"r2_{l}"], pre + "ln2", a[f"x{l + 1}"], a[f"st2_{l}"])
        Lr = cfg.num_layers
        self.ln_fwd(a[f"x{Lr}"], "encoder.norm", a["zn"], a["stf"])
        z = a["zn"]
        if cfg.use_cross_attention:
            # D:966-974: both directions in one launch each (kv_shift = B pairs window b with b+B)
            xs = _layer_sites(Lr)
            self.qkv_proj(z, "x")
            call("eg_attention_fwd", ptr(a["qkvx"]), ptr(a["ctxx"]), ptr(a["lsex"]), NB, S, H, B, self.dtype, p, xs["attn"],
                 self.st_ptr, st)
            self._probs_hook(self.model.cross_attn.cross_attn.dropout, a["qkvx"], a["lsex"], B)
            self.gemm(ptr(a["ctxx"]), ptr(w["ox"]), ptr(a["rx"]), M, d, d, bias=fp.p_ptr("cross_attn.cross_attn.out_proj.bias"),
                      drop1=(p, xs["drop1"]), residual=ptr(z))
            call("eg_layernorm_fwd", ptr(a["rx"]), fp.p_ptr("cross_attn.norm.weight"), fp.p_ptr("cross_attn.norm.bias"),
                 ptr(a["zc"]), ptr(a["stx"]), M, d, self.dtype, st)
            z = a["zc"]
        self.z_final = z
        # heads (D:1193-1213)
        call("eg_pool_fuse_fwd", ptr(z), ptr(a["cls1"]), ptr(a["cls2"]), ptr(a["comb"]), ptr(a["zf"]),
             ptr(a.get("ibs_pool_f")), ptr(a.get("ibs_pool")), B, S, d, self.off, self.n_ibs, 1, self.dtype, st)
        self.gemm(ptr(a["comb"]), ptr(w["sf"]), ptr(a["zf"]), B, d, 3 * d, c=rowmap(3 * d),
                  bias=fp.p_ptr("symmetric_fusion.proj.bias"))
        self.gemm(ptr(a["zf"]), ptr(w["c0"]), ptr(a["hcl"]), B, d, 3 * d, bias=fp.p_ptr("classifier.0.bias"),
                  act=L.ACT_RELU, drop1=(p, SITE_CLS))
        lab = ptr(labels) if labels is not None else 0
        call("eg_classifier_ce_fwd", ptr(a["hcl"]), fp.p_ptr("classifier.3.weight"), fp.p_ptr("classifier.3.bias"), lab,
             ptr(a["logits"]), ptr(a["sloss"]), ptr(a["loss"]), B, d, cfg.num_classes, self.dtype, st)
        if cfg.use_ibs:
            p3 = 0.3 if train else 0.0
            self.gemm(ptr(a["ibs_pool"]), ptr(w["i0"]), ptr(a["hib"]), B, d // 2, d, bias=fp.p_ptr("ibs_classifier.0.bias"),
                      act=L.ACT_RELU, drop1=(p3, SITE_IBSCLS))
            call("eg_classifier_ce_fwd", ptr(a["hib"]), fp.p_ptr("ibs_classifier.3.weight"), fp.p_ptr("ibs_classifier.3.bias"),
                 lab, ptr(a["ibs_logits"]), ptr(a["ibs_sloss"]), ptr(a["ibs_loss"]), B, d // 2, cfg.num_classes, self.dtype, st)
        self.labels = labels

    # ------------------------------------------------------------------------------------------
    # backward.  g* arguments are optional fp32 device tensors (gradients of the module's outputs);
    # gloss / gloss_ibs are 1-element fp32 device tensors (d total / d loss_ce, d total / d loss_ibs_cls).
    # Gradients land in the flat gradient buffer (overwritten, not accumulated).
    # ------------------------------------------------------------------------------------------
    def backward(self, gloss=None, gloss_ibs=None, glogits=None, gcls1=None, gcls2=None, gibs_logits=None,
                 gibs_token=None, on_segment=None, prescaled: bool = False):
        self._alloc_bwd()
        cfg, d, F, H = self.cfg, self.cfg.d_model, self.cfg.d_ff, self.cfg.num_heads
        B, NB, M, S, a, w, fp, g, es = self.B, self.NB, self.M, self.S, self.a, self.w, self.fp, self.g, self.es
        self.stream = self._cur_stream()
        st = self.stream
        p, p01, train = self.train_flags
        sc = 1.0 / (1.0 - p) if p > 0 else 1.0
        sc01 = 1.0 / (1.0 - p01) if p01 > 0 else 1.0
        lab = ptr(self.labels) if self.labels is not None else 0
        seg = on_segment or (lambda name: None)
        if self.scaler_on and not prescaled:
            # every gradient entering the backward is multiplied by the device-resident loss scale (no host sync); the
            # optimiser kernels divide it out again (eg_clip_coef / eg_adamw)
            ls = self.loss_scale_dev
            sc_ = lambda t_: None if t_ is None else t_ * ls
            gloss, gloss_ibs, glogits, gcls1, gcls2 = sc_(gloss), sc_(gloss_ibs), sc_(glogits), sc_(gcls1), sc_(gcls2)
            gibs_logits, gibs_token = sc_(gibs_logits), sc_(gibs_token)
        # ---- heads ----
        call("eg_classifier_ce_bwd", ptr(a["hcl"]), fp.p_ptr("classifier.3.weight"), ptr(a["logits"]), lab, ptr(gloss),
             ptr(glogits), ptr(g["dlogits"]), ptr(g["dhcl"]), fp.g_ptr("classifier.3.weight"), fp.g_ptr("classifier.3.bias"),
             B, d, cfg.num_classes, 1, sc, self.dtype, st)
        self.gemm(ptr(g["dhcl"]), ptr(w["c0T"]), ptr(g["dzf"]), B, 3 * d, d)
        self.wgrad(ptr(g["dhcl"]), ptr(a["zf"]), 0, B, d, 3 * d, linear=["classifier.0"])
        self.gemm(ptr(g["dzf"]), ptr(w["sfT"]), ptr(g["dcomb"]), B, 3 * d, d, a=rowmap(3 * d))
        self.wgrad(ptr(g["dzf"]), ptr(a["comb"]), 0, B, d, 3 * d, y=rowmap(3 * d), linear=["symmetric_fusion.proj"])
        dibs = None
        if cfg.use_ibs:
            sc3 = 1.0 / 0.7 if train else 1.0
            call("eg_classifier_ce_bwd", ptr(a["hib"]), fp.p_ptr("ibs_classifier.3.weight"), ptr(a["ibs_logits"]), lab,
                 ptr(gloss_ibs), ptr(gibs_logits), ptr(g["dibs_logits"]), ptr(g["dhib"]), fp.g_ptr("ibs_classifier.3.weight"),
                 fp.g_ptr("ibs_classifier.3.bias"), B, d // 2, cfg.num_classes, 1, sc3, self.dtype, st)
            Kp = w["i0T"].shape[1]
            if Kp != d // 2:
                raise L.EgError("ibs_classifier hidden width must be a multiple of the GEMM K-tile")
            self.gemm(ptr(g["dhib"]), ptr(w["i0T"]), ptr(g["dibs_pool"]), B, d, d // 2)
            self.wgrad(ptr(g["dhib"]), ptr(a["ibs_pool"]), 0, B, d // 2, d, linear=["ibs_classifier.0"])
            dibs = g["dibs_pool"]
        z = self.z_final
        dz = g["dzA"]
        call("eg_pool_fuse_bwd", ptr(z), ptr(g["dcomb"]), ptr(g["dzf"]), ptr(gcls1), ptr(gcls2), ptr(dibs), ptr(gibs_token),
             ptr(dz), B, S, d, self.off, self.n_ibs, 1, self.dtype, st)
        seg("heads")
        Lr = cfg.num_layers
        other = g["dzB"]

        grouped = self._wgrad_group_plan() is not None
        # with a gradient reducer listening (data parallel) the grouped launch is cut in two pieces (see _wgrad_group_plan);
        # EYEGAZE_WGRAD_PIECES=1 forces the cut without a reducer (bit-identity tests), =0 forbids it
        pcs = os.environ.get("EYEGAZE_WGRAD_PIECES", "")
        pieced = grouped and len(self._wg_plan["pieces"]) == 2 and pcs != "0" and (on_segment is not None or pcs == "1")

        def attn_block_bwd(pre, l, x_in, dr, drm, kv_shift, site_attn, dx_out, dqkv, defer, dctx_done=False):
            """dr: grad of the pre-LN sum (residual path), drm: same, masked by the branch dropout."""
            names = [pre + n for n in ("q_proj", "k_proj", "v_proj")]
            if not defer:
                self.wgrad(ptr(drm), ptr(a[f"ctx{l}"]), 0, M, d, d, linear=[pre + "out_proj"])
            if not dctx_done:       # (eg_ln_bwd_proj has written dctx together with dr / drm)
                self.gemm(ptr(drm), ptr(w[f"oT{l}"]), ptr(g["dctx"]), M, d, d)
            call("eg_attention_bwd", ptr(a[f"qkv{l}"]), ptr(a[f"ctx{l}"]), ptr(g["dctx"]), ptr(a[f"lse{l}"]), ptr(dqkv),
                 NB, S, H, kv_shift, self.dtype, p, site_attn, self.st_ptr, st)
            if not defer:
                self.wgrad(ptr(dqkv), ptr(x_in), 0, M, 3 * d, d, linear=names)
            self.gemm(ptr(dqkv), ptr(w[f"qkvT{l}"]), ptr(dx_out), M, d, 3 * d, residual=ptr(dr))

        has_drop = p > 0
        gx = False
        if cfg.use_cross_attention:
            xs = _layer_sites(Lr)
            gx = grouped and self._wg_cross                     # its weight gradients ride in the grouped launch
            drm = g["dYo_x"] if gx else (g["drm"] if has_drop else g["dr"])
            if has_drop:
                self.ln_bwd(dz, a["rx"], a["stx"], "cross_attn.norm", g["dr"], drm, d1=(p, xs["drop1"]))
                drx = g["dr"]
            else:
                self.ln_bwd(dz, a["rx"], a["stx"], "cross_attn.norm", drm, None)
                drx = drm
            attn_block_bwd("cross_attn.cross_attn.", "x", a["zn"], drx, drm, B, xs["attn"], other,
                           g["dqkv_x"] if gx else g["dqkv"], gx)
            dz, other = other, dz
            if not gx:
                seg("cross")
        # final encoder norm (A:328)
        self.ln_bwd(dz, a[f"x{Lr}"], a["stf"], "encoder.norm", other)
        dz, other = other, dz
        seg("encoder.norm")
        for l in reversed(range(Lr)):
            pre, sites = f"encoder.layers.{l}.", _layer_sites(l)
            # with the grouped weight-gradient launch every layer keeps its own dY operands until the end of backward
            dYf = g[f"dYf{l}"] if grouped else (g["drm"] if has_drop else g["dr"])
            dYo = g[f"dYo{l}"] if grouped else (g["drm"] if has_drop else g["dr"])
            dh = g[f"dh{l}"] if grouped else g["dh"]
            dqkv = g[f"dqkv{l}"] if grouped else g["dqkv"]
            s2 = self._ln_slot[pre + "ln2"] if grouped else None
            s1 = self._ln_slot[pre + "ln1"] if grouped else None
            if has_drop:
                self.ln_bwd(dz, a[f"r2_{l}"], a[f"st2_{l}"], pre + "ln2", g["dr"], dYf, d1=(p, sites["ffn_b"]),
                            d2=(p, sites["drop2"]), slot=s2)
            else:
                self.ln_bwd(dz, a[f"r2_{l}"], a[f"st2_{l}"], pre + "ln2", dYf, None, slot=s2)
            dr = g["dr"] if has_drop else dYf
            if not grouped:
                self.wgrad(ptr(dYf), ptr(a[f"hff{l}"]), 0, M, d, F, linear=[pre + "ffn.linear2"])
            if self.fuse_ffn:
                # dH = gate(dY W2) (stored: the weight gradients read it) and dy1 = dH W1 + dr in one launch; the gate is the
                # bit image the forward launch of this layer left
                self.ffn(ptr(dYf), ptr(w[f"w2Tf{l}"]), ptr(w[f"w1Tf{l}"]), ptr(dh), ptr(g["dy1"]), M, F, residual=ptr(dr),
                         bits_in=ptr(a[f"gbits{l}"]), gate_scale=sc)
            else:
                self.gemm(ptr(dYf), ptr(w[f"w2T{l}"]), ptr(dh), M, F, d, gate=ptr(a[f"hff{l}"]), gate_scale=sc)
            if not grouped:
                self.wgrad(ptr(dh), ptr(a[f"y1_{l}"]), 0, M, F, d, linear=[pre + "ffn.linear1"])
            if not self.fuse_ffn:
                self.gemm(ptr(dh), ptr(w[f"w1T{l}"]), ptr(g["dy1"]), M, d, F, residual=ptr(dr))
            if self.ln_proj:        # norm1 backward + out_proj backward-data in one launch
                self.ln_bwd_proj(g["dy1"], a[f"r1_{l}"], a[f"st1_{l}"], pre + "ln1", w[f"oTf{l}"], g["dr"], dYo, g["dctx"],
                                 d1=(p, sites["drop1"]) if has_drop else (0.0, 0), slot=s1)
                dr = g["dr"]
            else:
                if has_drop:
                    self.ln_bwd(g["dy1"], a[f"r1_{l}"], a[f"st1_{l}"], pre + "ln1", g["dr"], dYo, d1=(p, sites["drop1"]), slot=s1)
                else:
                    self.ln_bwd(g["dy1"], a[f"r1_{l}"], a[f"st1_{l}"], pre + "ln1", dYo, None, slot=s1)
                dr = g["dr"] if has_drop else dYo
            attn_block_bwd(pre + "mha.", l, a[f"x{l}"], dr, dYo, 0, sites["attn"], other, dqkv, grouped, dctx_done=self.ln_proj)
            dz, other = other, dz
            if not grouped:
                seg(f"layer{l}")
            elif pieced and l == self._wg_plan["split_layer"]:
                # upper half of the encoder: its weight gradients are complete -> reduce them now, hand the buckets to the
                # all-reduce while the lower layers' backward-data chain keeps the compute stream busy
                self._wgrad_group_launch(self._wg_plan["pieces"][0])
                if gx:
                    seg("cross")
                for ll in reversed(self._wg_plan["pieces"][0]["layers"]):
                    seg(f"layer{ll}")
        if grouped:
            if pieced:
                self._wgrad_group_launch(self._wg_plan["pieces"][1])
                for ll in reversed(self._wg_plan["pieces"][1]["layers"]):
                    seg(f"layer{ll}")
            else:
                self._wgrad_group_launch()
                if gx:
                    seg("cross")
                for l in reversed(range(Lr)):
                    seg(f"layer{l}")
        dseq = dz
        # positional table / cls token (A:120-126, D:1157)
        call("eg_batch_rowsum", ptr(dseq), fp.g_ptr("pos_embed.pos_embed.weight"), NB, S, d, S, self.dtype, st)
        call("eg_cast", fp.g_ptr("pos_embed.pos_embed.weight"), fp.g_ptr("cls_token"), d, EG_F32, st)
        self.model._extra_tokens_bwd(self, dseq)
        seg("tokens")           # positions, token generators, the extra heads: everything registered between conv-1 and the encoder
        # conv1 backward: dY = dseq[:, off:, :] * relu/dropout gate
        ymap = rowmap(d, self.RY * d, self.T2)
        call("eg_rows_gather_gate", ptr(dseq), ptr(a["h1"]), ptr(g["dy1pad"]) + (self.J - 1) * d * es, ymap, NB, S, d,
             self.T2, self.off, 0, sc01, self.dtype, st)
        dy1 = ptr(g["dy1pad"]) + (self.J - 1) * d * es
        self.wgrad(dy1, ptr(a["h0pad"]), fp.g_ptr("temporal_conv.convs.1.weight"), NB * self.T2, d, self.k * d, y=ymap,
                   x=rowmap(self.s * d, self.R0 * d, self.T2), out_b=fp.g_ptr("temporal_conv.convs.1.bias"),
                   conv=(d, self.k, d))
        seg("conv1")            # 6.5 MB of the front end's 7 MB: reduces under the backward-data phases and conv-0's gradient
        for ph in range(self.s):
            self.gemm(ptr(g["dy1pad"]), ptr(w["conv1T"]) + ph * d * self.J * d * es, ptr(g["dh0pad"]) + ph * d * es,
                      NB * self.U, d, self.J * d, a=rowmap(d, self.RY * d, self.U), c=rowmap(self.s * d, self.R0 * d, self.U),
                      gate=ptr(a["h0pad"]) + ph * d * es, gate_scale=sc01)
        h0map = rowmap(d, self.R0 * d, self.T1)
        self.wgrad(ptr(g["dh0pad"]) + self.pad * d * es, ptr(a["xt"]), fp.g_ptr("temporal_conv.convs.0.weight"),
                   NB * self.T1, d, self.K0, y=h0map, x=rowmap(self.s * self.Cp, self.Tp * self.Cp, self.T1),
                   out_b=fp.g_ptr("temporal_conv.convs.0.bias"), conv=(self.C, self.k, self.Cp))
        seg("frontend")

    # ------------------------------------------------------------------------------------------
    # optimiser: clip_grad_norm_(max_norm) + AdamW on the flat buffers (T:221-222)
    # ------------------------------------------------------------------------------------------
    def optimizer_step(self, m: torch.Tensor, v: torch.Tensor, max_norm: float = 1.0, betas=(0.9, 0.999), eps=1e-8,
                       weight_decay=0.01):
        self._alloc_bwd()
        self.stream = self._cur_stream()
        nblk = 1024
        if "sqpart" not in self.g:
            self.g["sqpart"] = self._t(nblk, dtype=torch.float32)
        call("eg_grad_sqnorm", ptr(self.fp.grad), self.fp.total, ptr(self.g["sqpart"]), nblk, self.stream)
        call("eg_clip_coef", ptr(self.g["sqpart"]), nblk, max_norm, self.st_ptr, self.stream)
        call("eg_adamw", ptr(self.fp.flat), ptr(self.fp.grad), ptr(m), ptr(v), self.fp.total, betas[0], betas[1], eps,
             weight_decay, self.st_ptr, self.stream)
        if self.scaler_on:      # GradScaler.update(): back off after an overflow, grow after growth_interval clean steps
            c = self.scaler_cfg
            call("eg_scaler_update", self.st_ptr, c["growth"], c["backoff"], c["growth_interval"], self.stream)
