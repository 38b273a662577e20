"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not the product path.

CPU restatement (plain PyTorch fp32 / numpy, functional style, no nn.Module) of the reference's
dual-stream window classifier hot path.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this file; the product path
(`eyegaze_multimodal_amd`) never does and fails loudly if its HIP library is missing.

Parity pinning: the reference publishes no golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, generated in the build container by
`oracle/make_golden.py` (imports /root/reference by file path) and committed under `tests/golden/`.
`tests/test_oracle_golden.py` checks every function below against those fixtures.

Every function cites the reference lines it restates (paths relative to the reference root):
  D = 3_Models/backbones/dual_eeg_transformer.py,  A = 3_Models/backbones/art.py,
  T = 4_Experiments/scripts/train_art.py
All arithmetic is third-party torch ATen on CPU (version unpinned in the reference; here 2.10).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------
# configuration mirror of DualEEGTransformer.__init__ kwargs  (D:995-1021)
# --------------------------------------------------------------------------------------


@dataclass
class ModelCfg:
    in_channels: int = 62
    num_classes: int = 3
    d_model: int = 256
    num_layers: int = 6
    num_heads: int = 8
    d_ff: int = 1024
    dropout: float = 0.1
    max_len: int = 2048
    conv_kernel_size: int = 25
    conv_stride: int = 4
    conv_layers: int = 2
    sampling_rate: int = 256
    use_spectrogram: bool = True
    spec_n_fft: int = 128
    spec_hop_length: int = 64
    spec_freq_bins: int = 64
    use_robust_ibs: bool = True
    use_ibs: bool = True
    use_cross_attention: bool = True
    ibs_instance_norm: bool = True
    ibs_feature_type: str = "all"

    # derived (D:1035-1043)
    @property
    def num_ibs_features(self) -> int:
        return {"all": 7, "phase": 4, "amplitude": 3}.get(self.ibs_feature_type, 7)

    @property
    def num_ibs_tokens(self) -> int:
        if not self.use_ibs:
            return 0
        return 6 * self.num_ibs_features if self.use_robust_ibs else 1

    @property
    def feature_indices(self) -> List[int]:
        # D:515-525
        if self.ibs_feature_type == "phase":
            return [0, 1, 2, 5]
        if self.ibs_feature_type == "amplitude":
            return [3, 4, 6]
        return list(range(7))

    @property
    def pool_offset(self) -> int:
        # D:1198-1202
        off = 1 + self.num_ibs_tokens
        if self.use_spectrogram:
            off += self.in_channels
        return off


# band tables: robust generator D:500-509, scalar generator D:201-206
ROBUST_BANDS: List[Tuple[float, float]] = [(0.5, 45), (0.5, 4), (4, 8), (8, 13), (13, 30), (30, 45)]
SCALAR_BANDS: List[Tuple[float, float]] = [(4, 8), (8, 13), (13, 30), (30, 45)]


# --------------------------------------------------------------------------------------
# a4  TemporalConvFrontend  (D:138-175)
# --------------------------------------------------------------------------------------

# Train-mode dropout.  By default torch's own generator (as in the reference).  Tests may install DROPOUT_OVERRIDE(t, p, site)
# to inject masks (e.g. the HIP path's counter-hash masks, so that train mode can be compared exactly); `site` names the call:
# ("conv", i) ("attn", prefix) ("drop1", l) ("ffn_a", l) ("ffn_b", l) ("drop2", l) ("xdrop1",) ("cls",) ("ibscls",) ("spec",)
# ("ibstok",) ("ibsgen",), and CTX["stream"] says which of the two streams (or cross-attention directions) is being computed.
DROPOUT_OVERRIDE = None
CTX = {"stream": 0}
SPEC_CONV2_TAPS = None   # tests may set this to a list: spectrogram_tokens appends its second conv's output (pre-ReLU)
# Storage-format model.  The HIP path keeps activations in its compute dtype (bf16 / fp16) between kernels and accumulates in
# fp32 inside them.  STORAGE_ROUND(t, site), when installed, is applied at exactly the points where the HIP path writes an
# activation to HBM (xt, h0pad, x0, q|k|v, attention probabilities as the PV operand, ctx, r1, y1, hff, r2, x_l,
# zn, the cross block's rx / zc, comb, zf, hcl; the token generators' stored stages), so that `fp32 arithmetic + bf16 storage`
# can be measured against the fp32 reference: tests/bf16_floor.py.  None (default) = the reference's fp32 everywhere.
STORAGE_ROUND = None


def _st(t: Tensor, site: str) -> Tensor:
    return t if STORAGE_ROUND is None else STORAGE_ROUND(t, site)


def _dropout(t: Tensor, p: float, site) -> Tensor:
    if p <= 0:
        return t
    if DROPOUT_OVERRIDE is not None:
        return DROPOUT_OVERRIDE(t, p, site)
    return F.dropout(t, p, True)


def temporal_conv(x: Tensor, sd: Dict[str, Tensor], cfg: ModelCfg, p_drop: float = 0.0) -> Tensor:
    """[B,C,T] -> [B,T~,d]: (conv1d k,stride,pad=k//2 -> ReLU -> dropout(0.1 in train)) x L, then
    channel-last.  D:170-174.  p_drop=0 restates eval mode."""
    h = _st(x, "xt")
    for i in range(cfg.conv_layers):
        w = sd[f"temporal_conv.convs.{i}.weight"]
        b = sd[f"temporal_conv.convs.{i}.bias"]
        h = F.conv1d(h, w, b, stride=cfg.conv_stride, padding=cfg.conv_kernel_size // 2)
        h = torch.relu(h)
        h = _dropout(h, p_drop, ("conv", i))
        if i + 1 < cfg.conv_layers:          # (the last layer's rows are stored once, with the positional rows added: "x0")
            h = _st(h, "h0pad")
    return h.transpose(1, 2)


# --------------------------------------------------------------------------------------
# a5  SpectrogramTokenGenerator  (D:40-135)
# --------------------------------------------------------------------------------------

def stft_logmag(x: Tensor, n_fft: int, hop: int, freq_bins: int, window: Optional[Tensor] = None) -> Tensor:
    """[B,C,T] -> [B*C, freq_bins, n_frames] log-magnitude STFT.  Restates torch.stft(center=True,
    reflect pad, onesided, periodic hann) + abs + [:freq_bins] + log(.+1e-8)  (D:98-118) as
    explicit framing + rFFT so that it is an independent statement of the algorithm."""
    B, C, T = x.shape
    if window is None:
        window = torch.hann_window(n_fft, dtype=x.dtype)  # periodic, D:66
    flat = x.reshape(B * C, 1, T)
    padded = F.pad(flat, (n_fft // 2, n_fft // 2), mode="reflect").squeeze(1)
    frames = padded.unfold(-1, n_fft, hop)  # [BC, n_frames, n_fft]
    spec = torch.fft.rfft(frames * window, dim=-1)  # [BC, n_frames, n_fft/2+1]
    mag = spec.abs()[..., :freq_bins].transpose(1, 2)  # [BC, F, n_frames]
    return torch.log(mag + 1e-8)


def spectrogram_tokens(x: Tensor, sd: Dict[str, Tensor], cfg: ModelCfg, p_drop: float = 0.0,
                       return_logmag: bool = False):
    """[B,C,T] -> [B,C,d]  (D:88-135): STFT image -> Conv2d(1,32,3,p1) ReLU MaxPool2 -> Conv2d(32,64,3,p1)
    ReLU AdaptiveAvgPool(4,4) -> flatten -> Linear(1024,2d) ReLU Dropout(.1) Linear(2d,d)."""
    B, C, T = x.shape
    pre = "spectrogram_generator."
    lm = stft_logmag(x, cfg.spec_n_fft, cfg.spec_hop_length, cfg.spec_freq_bins, sd.get(pre + "window"))
    img = lm.unsqueeze(1)
    h = F.conv2d(img, sd[pre + "spec_conv.0.weight"], sd[pre + "spec_conv.0.bias"], padding=1)
    h = _st(F.max_pool2d(torch.relu(h), 2), "sp_p1")
    h = _st(F.conv2d(h, sd[pre + "spec_conv.3.weight"], sd[pre + "spec_conv.3.bias"], padding=1), "sp_out2")
    if SPEC_CONV2_TAPS is not None:          # what a hook on spec_conv[3] sees (Grad-CAM, eeg_metrics.py:742-764)
        if h.requires_grad:
            h.retain_grad()
        SPEC_CONV2_TAPS.append(h)
    h = _st(F.adaptive_avg_pool2d(torch.relu(h), (4, 4)).flatten(1), "sp_pooled")
    h = torch.relu(F.linear(h, sd[pre + "proj.0.weight"], sd[pre + "proj.0.bias"]))
    if p_drop > 0:
        h = _dropout(h, p_drop, ("spec",))
    h = _st(h, "sp_hp0")
    tok = F.linear(h, sd[pre + "proj.3.weight"], sd[pre + "proj.3.bias"]).reshape(B, C, cfg.d_model)
    return (tok, lm) if return_logmag else tok


# --------------------------------------------------------------------------------------
# a6  IBSConnectivityMatrixGenerator  (D:473-819) -- vectorised restatement of the C x C loops
# --------------------------------------------------------------------------------------

def band_mask(T: int, fs: float, lo: float, hi: float) -> Tensor:
    """rFFT-bin mask, both edges inclusive (D:548-552)."""
    freqs = torch.fft.rfftfreq(T, d=1.0 / fs)
    return ((freqs >= lo) & (freqs <= hi)).float()


def bandpass(x: Tensor, fs: float, lo: float, hi: float) -> Tensor:
    """FFT-mask band-pass (D:527-560 and 224-257)."""
    T = x.shape[-1]
    return torch.fft.irfft(torch.fft.rfft(x, dim=-1) * band_mask(T, fs, lo, hi), n=T, dim=-1)


def hilbert_phase(x: Tensor) -> Tensor:
    """Instantaneous phase via FFT Hilbert transform (D:562-591 and 292-322)."""
    T = x.shape[-1]
    h = torch.zeros(T)
    if T % 2 == 0:
        h[0] = h[T // 2] = 1
        h[1:T // 2] = 2
    else:
        h[0] = 1
        h[1:(T + 1) // 2] = 2
    return torch.angle(torch.fft.ifft(torch.fft.fft(x, dim=-1) * h, dim=-1))


def _zscore_unbiased(v: Tensor) -> Tensor:
    """(v-mean)/(std_unbiased+1e-8) along the last axis (D:707-708, 751-752)."""
    return (v - v.mean(-1, keepdim=True)) / (v.std(-1, keepdim=True) + 1e-8)


def ibs_band_matrices(e1: Tensor, e2: Tensor) -> Tensor:
    """For one band: band-limited [B,C,T] x2 -> [B,7,C,C] in the reference's feature order
    [PLV, PLI, wPLI, Coherence, Power_Corr, Phase_Diff, Time_Corr] (D:796-812)."""
    B, C, T = e1.shape
    p1, p2 = e1 ** 2, e2 ** 2                       # D:789-790
    ph1, ph2 = hilbert_phase(e1), hilbert_phase(e2)  # D:793-794
    d = ph1[:, :, None, :] - ph2[:, None, :, :]      # [B,C,C,T] raw (unwrapped) difference, D:606
    plv = torch.exp(1j * d).mean(-1).abs()                                   # D:607-608
    sgn = torch.sign(d)
    pli = sgn.mean(-1).abs()                                                 # D:627
    w = (p1[:, :, None, :] + p2[:, None, :, :]) / 2                          # D:653
    w = w / (w.sum(-1, keepdim=True) + 1e-8)                                 # D:654
    wpli = (sgn * w).sum(-1).abs()                                           # D:655-656
    f1, f2 = torch.fft.rfft(e1, dim=-1), torch.fft.rfft(e2, dim=-1)          # D:672-673
    pxy = f1[:, :, None, :] * f2[:, None, :, :].conj()                       # D:678
    pxx = (f1 * f1.conj()).real[:, :, None, :]                               # D:681
    pyy = (f2 * f2.conj()).real[:, None, :, :]                               # D:682
    coh = ((pxy.abs() ** 2) / (pxx * pyy + 1e-8)).mean(-1)                   # D:685-686
    z1, z2 = _zscore_unbiased(p1), _zscore_unbiased(p2)                      # D:707-708
    pcorr = torch.einsum("bit,bjt->bij", z1, z2) / T                         # D:711
    pdiff = d.abs().mean(-1)                                                 # D:729-730
    s1, s2 = _zscore_unbiased(e1), _zscore_unbiased(e2)                      # D:751-752
    tcorr = torch.einsum("bit,bjt->bij", s1, s2) / T                         # D:755
    return torch.stack([plv, pli, wpli, coh, pcorr, pdiff, tcorr], dim=1).float()


def ibs_connectivity(eeg1: Tensor, eeg2: Tensor, cfg: ModelCfg) -> Tensor:
    """[B,C,T]x2 -> [B,6,num_features,C,C] fp32 (D:760-819)."""
    out = []
    for lo, hi in ROBUST_BANDS:
        e1 = bandpass(eeg1, cfg.sampling_rate, lo, hi)
        e2 = bandpass(eeg2, cfg.sampling_rate, lo, hi)
        out.append(ibs_band_matrices(e1, e2))
    full = torch.stack(out, dim=1)  # [B,6,7,C,C]
    return full[:, :, cfg.feature_indices]


# --------------------------------------------------------------------------------------
# a7  RobustIBSTokenizer  (D:822-911)
# --------------------------------------------------------------------------------------

def ibs_tokenize(conn: Tensor, sd: Dict[str, Tensor], cfg: ModelCfg, p_drop: float = 0.0) -> Tensor:
    """[B,6,F,C,C] -> [B,6F,d].  InstanceNorm1d(C*C, affine) is applied to [B, C*C, Ntok], i.e. it
    normalises each matrix entry over the TOKEN axis (biased var, eps 1e-5) (D:897-901)."""
    B, nb, nf, C1, C2 = conn.shape
    x = conn.reshape(B, nb * nf, C1 * C2)
    pre = "ibs_tokenizer."
    if cfg.ibs_instance_norm:
        mu = x.mean(1, keepdim=True)
        var = x.var(1, unbiased=False, keepdim=True)
        x = (x - mu) / torch.sqrt(var + 1e-5)
        x = x * sd[pre + "instance_norm.weight"] + sd[pre + "instance_norm.bias"]
    x = _st(x, "ib_in")
    h = F.gelu(_st(F.linear(x, sd[pre + "bottleneck.0.weight"], sd[pre + "bottleneck.0.bias"]), "ib_u"))
    if p_drop > 0:
        h = _dropout(h, p_drop, ("ibstok",))
    h = _st(h, "ib_h")
    h = F.linear(h, sd[pre + "bottleneck.3.weight"], sd[pre + "bottleneck.3.bias"])
    return h + sd[pre + "type_embedding"]


# --------------------------------------------------------------------------------------
# a8  IBSTokenGenerator, scalar mode  (D:178-470)
# --------------------------------------------------------------------------------------

def ibs_scalar_features(eeg1: Tensor, eeg2: Tensor, cfg: ModelCfg) -> Tensor:
    """[B,C,T]x2 -> [B,28]: 4 bands x [plv,pli,wpli,coh,power_corr,phase_diff,time_corr] (D:436-461)."""
    feats = []
    for lo, hi in SCALAR_BANDS:
        e1 = bandpass(eeg1, cfg.sampling_rate, lo, hi)
        e2 = bandpass(eeg2, cfg.sampling_rate, lo, hi)
        p1, p2 = e1 ** 2, e2 ** 2
        ph1, ph2 = hilbert_phase(e1), hilbert_phase(e2)
        d = ph1 - ph2
        plv = torch.exp(1j * d).mean(dim=(1, 2)).abs()                       # D:267-270
        pli = torch.sign(d).mean(dim=(1, 2)).abs()                           # D:334-335
        w = (p1 + p2) / 2
        w = w / (w.sum(dim=(1, 2), keepdim=True) + 1e-8)
        wpli = (torch.sign(d) * w).sum(dim=(1, 2)).abs()                     # D:355-363
        f1, f2 = torch.fft.rfft(e1, dim=2), torch.fft.rfft(e2, dim=2)
        pxy = (f1 * f2.conj()).mean(1)
        pxx = (f1 * f1.conj()).mean(1).real
        pyy = (f2 * f2.conj()).mean(1).real
        coh = ((pxy.abs() ** 2) / (pxx * pyy + 1e-8)).mean(1)                # D:378-392
        pc = (_zscore_unbiased(p1.flatten(1)) * _zscore_unbiased(p2.flatten(1))).mean(1)  # D:281-289
        pd = d.mean(dim=(1, 2)).abs()                                        # D:455
        tc = (_zscore_unbiased(e1.mean(1)) * _zscore_unbiased(e2.mean(1))).mean(1)        # D:406-414
        feats += [plv, pli, wpli, coh, pc, pd, tc]
    return torch.stack(feats, dim=1).float()


def ibs_scalar_token(eeg1: Tensor, eeg2: Tensor, sd: Dict[str, Tensor], cfg: ModelCfg, p_drop: float = 0.0) -> Tensor:
    """[B,C,T]x2 -> [B,d]: features -> Linear(28,2d) ReLU Dropout(.1) Linear(2d,d) (D:213-218, 464)."""
    f = ibs_scalar_features(eeg1, eeg2, cfg)
    h = torch.relu(F.linear(f, sd["ibs_generator.proj.0.weight"], sd["ibs_generator.proj.0.bias"]))
    if p_drop > 0:
        h = _dropout(h, p_drop, ("ibsgen",))
    return F.linear(h, sd["ibs_generator.proj.3.weight"], sd["ibs_generator.proj.3.bias"])


# --------------------------------------------------------------------------------------
# a9/a10  positional embedding + post-LN encoder  (A:116-126, A:202-213, A:272, A:292-295, A:326-328)
# --------------------------------------------------------------------------------------

def mha(q_in: Tensor, kv_in: Tensor, sd: Dict[str, Tensor], pre: str, H: int, p_attn: float = 0.0,
        return_probs: bool = False):
    """Multi-head scaled dot-product attention with separate q/k/v/out projections (A:202-213)."""
    B, Tq, D = q_in.shape
    dk = D // H
    q = _st(F.linear(q_in, sd[pre + "q_proj.weight"], sd[pre + "q_proj.bias"]), "qkv").view(B, -1, H, dk).transpose(1, 2)
    k = _st(F.linear(kv_in, sd[pre + "k_proj.weight"], sd[pre + "k_proj.bias"]), "qkv").view(B, -1, H, dk).transpose(1, 2)
    v = _st(F.linear(kv_in, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"]), "qkv").view(B, -1, H, dk).transpose(1, 2)
    s = (q @ k.transpose(-2, -1)) / math.sqrt(dk)
    p = torch.softmax(s, dim=-1)
    pd = _st(_dropout(p, p_attn, ("attn", pre)), "probs")
    ctx = _st((pd @ v).transpose(1, 2).reshape(B, Tq, D), "ctx")
    out = F.linear(ctx, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])
    return (out, p) if return_probs else out


def _ln(x: Tensor, sd: Dict[str, Tensor], pre: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[pre + "weight"], sd[pre + "bias"], 1e-5)


def encoder(x: Tensor, sd: Dict[str, Tensor], cfg: ModelCfg, p_drop: float = 0.0) -> Tensor:
    """6 x post-LN blocks + final LN (A:292-295, 326-328)."""
    for l in range(cfg.num_layers):
        pre = f"encoder.layers.{l}."
        h = mha(x, x, sd, pre + "mha.", cfg.num_heads, p_drop)
        x = _st(_ln(_st(x + _dropout(h, p_drop, ("drop1", l)), "r1"), sd, pre + "ln1."), "y1")
        h = torch.relu(F.linear(x, sd[pre + "ffn.linear1.weight"], sd[pre + "ffn.linear1.bias"]))
        h = _dropout(F.linear(_st(_dropout(h, p_drop, ("ffn_a", l)), "hff"), sd[pre + "ffn.linear2.weight"], sd[pre + "ffn.linear2.bias"]),
                     p_drop, ("ffn_b", l))
        x = _st(_ln(_st(x + _dropout(h, p_drop, ("drop2", l)), "r2"), sd, pre + "ln2."), "x")
    return _st(_ln(x, sd, "encoder.norm."), "zn")


def cross_brain_attention(z1: Tensor, z2: Tensor, sd: Dict[str, Tensor], cfg: ModelCfg, p_drop: float = 0.0,
                          probs: Optional[list] = None):
    """Both directions share one MHA and one LN and read the pre-update z1, z2 (D:966-974)."""
    CTX["stream"] = 0
    c1, p1 = mha(z1, z2, sd, "cross_attn.cross_attn.", cfg.num_heads, p_drop, return_probs=True)
    o1 = _st(_ln(_st(z1 + _dropout(c1, p_drop, ("xdrop1",)), "rx"), sd, "cross_attn.norm."), "zc")
    CTX["stream"] = 1
    c2, p2 = mha(z2, z1, sd, "cross_attn.cross_attn.", cfg.num_heads, p_drop, return_probs=True)
    o2 = _st(_ln(_st(z2 + _dropout(c2, p_drop, ("xdrop1",)), "rx"), sd, "cross_attn.norm."), "zc")
    CTX["stream"] = 0
    if probs is not None:
        probs.extend([p1, p2])
    return o1, o2


# --------------------------------------------------------------------------------------
# a3/a12  full forward  (D:1110-1253)
# --------------------------------------------------------------------------------------

def forward(eeg1: Tensor, eeg2: Tensor, sd: Dict[str, Tensor], cfg: ModelCfg, labels: Optional[Tensor] = None,
            train: bool = False, stages: Optional[dict] = None, conn_edit=None) -> Dict[str, Tensor]:
    """Restates DualEEGTransformer.forward.  `stages`, when given, receives the intermediates the
    golden fixtures pin (temporal tokens, connectivity, ibs/spec tokens, encoder and cross-attn outputs).
    `conn_edit(conn) -> conn` stands in for a forward hook on the matrix generator (eeg_metrics.py:335-343)."""
    B = eeg1.shape[0]
    p = cfg.dropout if train else 0.0
    p01 = 0.1 if train else 0.0  # hard-coded sites D:161, D:84, D:866, D:216
    CTX["stream"] = 0
    h1 = temporal_conv(eeg1, sd, cfg, p01)
    CTX["stream"] = 1
    h2 = temporal_conv(eeg2, sd, cfg, p01)
    CTX["stream"] = 0
    ibs_tokens = None
    if cfg.use_ibs:
        if cfg.use_robust_ibs:
            conn = ibs_connectivity(eeg1, eeg2, cfg)
            if conn_edit is not None:
                conn = conn_edit(conn)
            ibs_tokens = ibs_tokenize(conn, sd, cfg, p01)
            if stages is not None:
                stages["connectivity"] = conn
        else:
            ibs_tokens = ibs_scalar_token(eeg1, eeg2, sd, cfg, p01).unsqueeze(1)
    comps1 = [sd["cls_token"].expand(B, -1, -1)]
    comps2 = [sd["cls_token"].expand(B, -1, -1)]
    if ibs_tokens is not None:
        comps1.append(ibs_tokens)
        comps2.append(ibs_tokens)
    if cfg.use_spectrogram:
        s1 = spectrogram_tokens(eeg1, sd, cfg, p01)
        s2 = spectrogram_tokens(eeg2, sd, cfg, p01)
        comps1.append(s1)
        comps2.append(s2)
        if stages is not None:
            stages["spec1"], stages["spec2"] = s1, s2
    comps1.append(h1)
    comps2.append(h2)
    seq1, seq2 = torch.cat(comps1, 1), torch.cat(comps2, 1)
    S = seq1.shape[1]
    pos = sd["pos_embed.pos_embed.weight"][:S]          # A:120-126 (learned)
    CTX["stream"] = 0
    z1 = encoder(_st(seq1 + pos, "x0"), sd, cfg, p)
    CTX["stream"] = 1
    z2 = encoder(_st(seq2 + pos, "x0"), sd, cfg, p)
    CTX["stream"] = 0
    if stages is not None:
        stages.update(h1=h1, h2=h2, z1=z1, z2=z2)
        if ibs_tokens is not None:
            stages["ibs_tokens"] = ibs_tokens
    if cfg.use_cross_attention:
        xp = [] if stages is not None else None
        z1c, z2c = cross_brain_attention(z1, z2, sd, cfg, p, probs=xp)
        if stages is not None:
            stages["xattn_probs"] = torch.stack(xp)          # [direction, B, H, S, S]: what a dropout-module hook sees
    else:
        z1c, z2c = z1, z2
    if stages is not None:
        stages.update(z1c=z1c, z2c=z2c)
    cls1, cls2 = z1c[:, 0], z2c[:, 0]
    off = cfg.pool_offset
    mp1, mp2 = z1c[:, off:].mean(1), z2c[:, off:].mean(1)
    comb = _st(torch.cat([cls1 + cls2, cls1 * cls2, (cls1 - cls2).abs()], -1), "comb")   # D:933-938
    f_pair = F.linear(comb, sd["symmetric_fusion.proj.weight"], sd["symmetric_fusion.proj.bias"])
    zf = _st(torch.cat([f_pair, mp1, mp2], -1), "zf")                                 # D:1212
    hcl = torch.relu(F.linear(zf, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    hcl = _st(_dropout(hcl, p, ("cls",)), "hcl")
    logits = F.linear(hcl, sd["classifier.3.weight"], sd["classifier.3.bias"])
    out = {"logits": logits, "cls1": cls1, "cls2": cls2}
    if cfg.use_ibs:
        if cfg.use_robust_ibs:
            pooled = z1c[:, 1:1 + cfg.num_ibs_tokens].mean(1)                         # D:1222-1223
        else:
            pooled = z1c[:, 1]                                                        # D:1228
        hi = torch.relu(F.linear(pooled, sd["ibs_classifier.0.weight"], sd["ibs_classifier.0.bias"]))
        if train:
            hi = _dropout(hi, 0.3, ("ibscls",))                                       # D:1077
        out["ibs_logits"] = F.linear(hi, sd["ibs_classifier.3.weight"], sd["ibs_classifier.3.bias"])
        out["ibs_token"] = pooled
    if labels is not None:
        out["loss_ce"] = F.cross_entropy(logits, labels)                              # D:1244
        out["loss"] = out["loss_ce"]
        if cfg.use_ibs:
            out["loss_ibs_cls"] = F.cross_entropy(out["ibs_logits"], labels)          # D:1250
    return out


# --------------------------------------------------------------------------------------
# a13  auxiliary losses  (D:1255-1371)
# --------------------------------------------------------------------------------------

def symmetry_loss(cls1: Tensor, cls2: Tensor) -> Tensor:
    return ((cls1 - cls2) ** 2).mean()                                                # D:1260


def ibs_alignment_loss(ibs: Tensor, cls1: Tensor, cls2: Tensor, temperature: float = 0.07) -> Tensor:
    """InfoNCE over [B,2B] similarities with the diagonal of the first B as positives (D:1284-1302)."""
    n = lambda t: t / t.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    sim = n(ibs) @ torch.cat([n(cls1), n(cls2)], 0).T / temperature
    return F.cross_entropy(sim, torch.arange(ibs.shape[0]))


def ibs_contrastive_loss(ibs: Tensor, labels: Tensor, temperature: float = 0.07) -> Tensor:
    """Supervised contrastive loss; 0 when no sample has a same-class partner (D:1329-1369)."""
    B = ibs.shape[0]
    z = ibs / ibs.norm(dim=1, keepdim=True).clamp_min(1e-12)
    e = torch.exp(z @ z.T / temperature)
    eye = torch.eye(B, dtype=torch.bool)
    pos = (labels[:, None] == labels[None, :]).float().masked_fill(eye, 0)
    has = pos.sum(1) > 0
    if has.sum() == 0:
        return torch.tensor(0.0)
    num = (e * pos).sum(1)
    den = e.masked_fill(eye, 0).sum(1)
    loss = -torch.log(num / (den + 1e-8) + 1e-8)
    return loss[has].mean()


# --------------------------------------------------------------------------------------
# a14  optimiser step: clip_grad_norm_(1.0) + AdamW  (T:221-222, T:401-405)
# --------------------------------------------------------------------------------------

def clip_and_adamw(params: Dict[str, Tensor], grads: Dict[str, Tensor], state: Dict[str, Dict[str, Tensor]],
                   step: int, lr: float = 1e-4, wd: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-8,
                   max_norm: float = 1.0) -> float:
    """In-place restatement of torch.nn.utils.clip_grad_norm_ followed by torch.optim.AdamW.step
    (decoupled weight decay, bias-corrected moments).  Returns the pre-clip global L2 norm.  Parameters whose gradient is
    None are skipped, as torch does (e.g. ibs_classifier when its loss term is off)."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values() if g is not None))
    coef = min(1.0, max_norm / (total + 1e-6))
    b1, b2 = betas
    for k, p in params.items():
        if grads[k] is None:
            continue
        g = grads[k] * coef
        st = state.setdefault(k, {"m": torch.zeros_like(p), "v": torch.zeros_like(p)})
        p.mul_(1 - lr * wd)
        st["m"].mul_(b1).add_(g, alpha=1 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (st["v"].sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
        p.addcdiv_(st["m"], denom, value=-lr / (1 - b1 ** step))
    return total


def cosine_lr(base_lr: float, epoch: int, t_max: int) -> float:
    """CosineAnnealingLR(eta_min=0) closed form, stepped per epoch (T:409, T:494)."""
    return base_lr * (1 + math.cos(math.pi * epoch / t_max)) / 2


# --------------------------------------------------------------------------------------
# a2  argmax + macro metrics  (T:289, T:299-304)
# --------------------------------------------------------------------------------------

def macro_metrics(y_true: np.ndarray, y_pred: np.ndarray) -> Dict[str, float]:
    """accuracy + macro precision/recall/F1 with zero_division=0 over the labels present in
    y_true U y_pred (sklearn semantics used at T:301-304)."""
    labels = np.union1d(y_true, y_pred)
    P, R, Fs = [], [], []
    for c in labels:
        tp = float(np.sum((y_pred == c) & (y_true == c)))
        fp = float(np.sum((y_pred == c) & (y_true != c)))
        fn = float(np.sum((y_pred != c) & (y_true == c)))
        p = tp / (tp + fp) if tp + fp > 0 else 0.0
        r = tp / (tp + fn) if tp + fn > 0 else 0.0
        P.append(p)
        R.append(r)
        Fs.append(2 * p * r / (p + r) if p + r > 0 else 0.0)
    return {"accuracy": float(np.mean(y_true == y_pred)), "precision": float(np.mean(P)),
            "recall": float(np.mean(R)), "f1": float(np.mean(Fs))}


# --------------------------------------------------------------------------------------
# a16  FuzzyGatingFusion forward  (3_Models/fusion/fuzzy_gating_fusion.py:297-390)
# --------------------------------------------------------------------------------------
# restated in oracle/fuzzy_oracle.py (kept separate: config-5-only, "next" row of SURVEY §8f)


# --------------------------------------------------------------------------------------
# helpers shared by tests / bench (not part of the reference): deterministic weights and inputs
# --------------------------------------------------------------------------------------

def state_shapes(cfg: ModelCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    """state_dict keys and shapes in the reference's registration order (D:1046-1105)."""
    d, C, k = cfg.d_model, cfg.in_channels, cfg.conv_kernel_size
    out: List[Tuple[str, Tuple[int, ...]]] = []
    out.append(("cls_token", (1, 1, d)))
    for i in range(cfg.conv_layers):
        out.append((f"temporal_conv.convs.{i}.weight", (d, C if i == 0 else d, k)))
        out.append((f"temporal_conv.convs.{i}.bias", (d,)))
    if cfg.use_spectrogram:
        p = "spectrogram_generator."
        out += [(p + "window", (cfg.spec_n_fft,)),
                (p + "spec_conv.0.weight", (32, 1, 3, 3)), (p + "spec_conv.0.bias", (32,)),
                (p + "spec_conv.3.weight", (64, 32, 3, 3)), (p + "spec_conv.3.bias", (64,)),
                (p + "proj.0.weight", (2 * d, 1024)), (p + "proj.0.bias", (2 * d,)),
                (p + "proj.3.weight", (d, 2 * d)), (p + "proj.3.bias", (d,))]
    if cfg.use_ibs:
        if cfg.use_robust_ibs:
            p = "ibs_tokenizer."
            out.append((p + "type_embedding", (1, cfg.num_ibs_tokens, d)))
            if cfg.ibs_instance_norm:
                out += [(p + "instance_norm.weight", (C * C,)), (p + "instance_norm.bias", (C * C,))]
            out += [(p + "bottleneck.0.weight", (64, C * C)), (p + "bottleneck.0.bias", (64,)),
                    (p + "bottleneck.3.weight", (d, 64)), (p + "bottleneck.3.bias", (d,))]
        else:
            p = "ibs_generator.proj."
            out += [(p + "0.weight", (2 * d, 28)), (p + "0.bias", (2 * d,)),
                    (p + "3.weight", (d, 2 * d)), (p + "3.bias", (d,))]
        p = "ibs_classifier."
        out += [(p + "0.weight", (d // 2, d)), (p + "0.bias", (d // 2,)),
                (p + "3.weight", (cfg.num_classes, d // 2)), (p + "3.bias", (cfg.num_classes,))]
    out.append(("pos_embed.pos_embed.weight", (cfg.max_len, d)))
    for l in range(cfg.num_layers):
        p = f"encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out += [(p + f"mha.{n}.weight", (d, d)), (p + f"mha.{n}.bias", (d,))]
        out += [(p + "ln1.weight", (d,)), (p + "ln1.bias", (d,)),
                (p + "ffn.linear1.weight", (cfg.d_ff, d)), (p + "ffn.linear1.bias", (cfg.d_ff,)),
                (p + "ffn.linear2.weight", (d, cfg.d_ff)), (p + "ffn.linear2.bias", (d,)),
                (p + "ln2.weight", (d,)), (p + "ln2.bias", (d,))]
    out += [("encoder.norm.weight", (d,)), ("encoder.norm.bias", (d,))]
    if cfg.use_cross_attention:
        p = "cross_attn.cross_attn."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out += [(p + f"{n}.weight", (d, d)), (p + f"{n}.bias", (d,))]
        out += [("cross_attn.norm.weight", (d,)), ("cross_attn.norm.bias", (d,))]
    out += [("symmetric_fusion.proj.weight", (d, 3 * d)), ("symmetric_fusion.proj.bias", (d,)),
            ("classifier.0.weight", (d, 3 * d)), ("classifier.0.bias", (d,)),
            ("classifier.3.weight", (cfg.num_classes, d)), ("classifier.3.bias", (cfg.num_classes,))]
    return out


def synthetic_state_dict(cfg: ModelCfg, seed: int) -> Dict[str, Tensor]:
    """Deterministic, reference-independent weights (numpy PCG64, keys in registration order):
    matrices ~ N(0, 1/fan_in), norm gains 1+0.1N, biases/embeddings 0.05N / 0.02N.  Used by
    make_golden.py to load the SAME weights into the reference, so full-size fixtures need not
    carry a 28 MB state_dict."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, Tensor] = {}
    for key, shape in state_shapes(cfg):
        if key.endswith("window"):
            sd[key] = torch.hann_window(shape[0])
            continue
        z = rng.standard_normal(shape).astype(np.float32)
        if key.endswith(".weight") and len(shape) >= 2 and "pos_embed" not in key:
            fan_in = int(np.prod(shape[1:]))
            v = z / math.sqrt(fan_in)
        elif key.endswith(".weight") and ("ln" in key or "norm" in key):
            v = 1.0 + 0.1 * z
        elif key in ("cls_token",) or "type_embedding" in key or "pos_embed" in key:
            v = 0.5 * z if key == "cls_token" else 0.02 * z
        else:
            v = 0.05 * z
        sd[key] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def zscore_window(x: np.ndarray) -> np.ndarray:
    """Per-window GLOBAL z-score with population std (1_Data/processed/dual_eeg_dataset.py:201-202)."""
    return ((x - x.mean()) / (x.std() + 1e-8)).astype(np.float32)


def preprocess_window(x: np.ndarray) -> np.ndarray:
    """`enable_preprocessing=True` branch (1_Data/processed/dual_eeg_dataset.py:142-168): the band-pass step there is a
    TODO that does nothing; common-average reference, then per-channel z-score with population std + 1e-8."""
    x = x - x.mean(axis=0, keepdims=True)
    return ((x - x.mean(axis=1, keepdims=True)) / (x.std(axis=1, keepdims=True) + 1e-8)).astype(np.float32)


def enumerate_windows(lengths, window_size: int, stride: int):
    """(item index, start, end) in the order DualEEGDataset._prepare_windows emits them (dual_eeg_dataset.py:62-120);
    `lengths[i]` = min(len(player1), len(player2)) of item i, or None when a file is missing."""
    out = []
    for idx, n in enumerate(lengths):
        if n is None or n < window_size:
            continue
        for w in range((n - window_size) // stride + 1):
            if w * stride + window_size <= n:
                out.append((idx, w * stride, w * stride + window_size))
    return out
