"""Optional token families in front of the temporal tokens: spectrogram tokens (D:40-135) and inter-stream
synchrony (IBS) tokens, matrix form (D:473-911) or scalar form (D:178-470).  Called by the engine through the
module hooks; every op is a C-ABI kernel (signal.hip, spec.hip, gemm.hip).  D = dual_eeg_transformer.py."""
from __future__ import annotations

import os
import ctypes as C

import torch

from . import _lib as L
from ._lib import EG_F32, call, ptr, rowmap
from .engine import SITE_IBSGEN, SITE_IBSTOK, SITE_SPEC, Engine, _align

CONV2_FWD_FLAT = os.environ.get("EYEGAZE_CONV2_FLAT", "1") != "0"           # 0: conv-2 forward / backward-data as segmented-row eg_gemm_nt
CONV2_WGRAD_FLAT = os.environ.get("EYEGAZE_CONV2_WGRAD_FLAT", "1") != "0"   # 0: the conv-2 weight gradient as an im2col eg_gemm_tn
ROBUST_BANDS = [(0.5, 45.0), (0.5, 4.0), (4.0, 8.0), (8.0, 13.0), (13.0, 30.0), (30.0, 45.0)]  # D:500-507
SCALAR_BANDS = [(4.0, 8.0), (8.0, 13.0), (13.0, 30.0), (30.0, 45.0)]                           # D:201-206


def _bands(eng: Engine, bands):
    key = ("bands", len(bands))
    if key not in eng.w:
        lo = (C.c_float * len(bands))(*[b[0] for b in bands])
        hi = (C.c_float * len(bands))(*[b[1] for b in bands])
        eng.w[key] = (lo, hi)
    lo, hi = eng.w[key]
    return C.addressof(lo), C.addressof(hi), len(bands)


def _nbin(eng: Engine, bands) -> int:
    return min(eng.T // 2 + 1, int(max(b[1] for b in bands) * eng.T / eng.cfg.sampling_rate) + 1)


# ------------------------------------------------------------------------------------------------
def _alloc(model, eng: Engine):
    if getattr(eng, "_tok_alloc", False):
        return
    cfg, d, B, NB, Cn, T = eng.cfg, eng.cfg.d_model, eng.B, eng.NB, eng.C, eng.T
    f32 = torch.float32
    a, w = eng.a, eng.w
    if cfg.use_spectrogram:
        spec_cnn_alloc(eng, NB * Cn, cfg.spec_freq_bins, 1 + T // cfg.spec_hop_length)
    if cfg.use_ibs:
        bands = ROBUST_BANDS if cfg.use_robust_ibs else SCALAR_BANDS
        nb, nsig = len(bands), NB * Cn
        eng.ib = dict(nb=nb, nsig=nsig, nbin=_nbin(eng, bands), bands=bands)
        a["ib_xb"] = eng._t(nb, nsig, T, dtype=f32)
        a["ib_ph"] = eng._t(nb, nsig, T, dtype=f32)
        a["ib_stats"] = eng._t(nb, nsig, 4, dtype=f32)
        a["ib_spec"] = eng._t(nsig, eng.ib["nbin"], 2, dtype=f32)
        if cfg.use_robust_ibs:
            ntok, E = cfg.num_ibs_tokens, Cn * Cn
            if E % eng.bk != 0:
                raise L.EgError(f"IBS tokenizer needs in_channels^2 (= {E}) to be a multiple of {eng.bk}")
            idx = {"phase": [0, 1, 2, 5], "amplitude": [3, 4, 6]}.get(cfg.ibs_feature_type, list(range(7)))
            a["ib_fidx"] = torch.tensor(idx, dtype=torch.int32, device=eng.device)
            a["ib_conn"] = eng._t(B, nb, 7, Cn, Cn, dtype=f32)
            a["ib_in"] = eng._t(B * ntok, E)
            a["ib_xhat"] = eng._t(B * ntok, E, dtype=f32)
            a["ib_u"] = eng._t(B * ntok, 64)
            a["ib_h"] = eng._t(B * ntok, 64)
            w["ib_add"] = eng._t(1, 1 + ntok, d)
            w["ib0"] = eng._t(64, E)
            w["ib0T"] = eng._t(E, 64)
            w["ib3"] = eng._t(d, 64)
            w["ib3T"] = eng._t(64, d)
        else:
            a["ig_feat"] = eng._t(B, 64, dtype=f32)
            a["ig_featp"] = eng._t(B, 64)
            a["ig_h"] = eng._t(B, 2 * d)
            w["ig0"] = eng._t(2 * d, 64)
            w["ig3"] = eng._t(d, 2 * d)
            w["ig3T"] = eng._t(2 * d, d)
    eng._tok_alloc = True


def _alloc_bwd(model, eng: Engine):
    if getattr(eng, "_tok_alloc_bwd", False):
        return
    cfg, d, B = eng.cfg, eng.cfg.d_model, eng.B
    g = eng.g
    if cfg.use_spectrogram:
        sp = eng.sp
        spec_cnn_alloc_bwd(eng)
    if cfg.use_ibs:
        if cfg.use_robust_ibs:
            ntok, E = cfg.num_ibs_tokens, eng.C * eng.C
            g["ib_dtok"] = eng._t(B * ntok, d)
            g["ib_dh"] = eng._t(B * ntok, 64)
            g["ib_du"] = eng._t(B * ntok, 64)
            g["ib_dxn"] = eng._t(B * ntok, E)
            g["ib_affpart"] = eng._t(256, 2, E, dtype=torch.float32)
        else:
            g["ig_dtok"] = eng._t(B, d)
            g["ig_dh"] = eng._t(B, 2 * d)
    eng._tok_alloc_bwd = True


# ------------------------------------------------------------------------------------------------
def pack(model, eng: Engine):
    cfg, d, fp, w, dt, st = eng.cfg, eng.cfg.d_model, eng.fp, eng.w, eng.dtype, eng.stream
    if not (cfg.use_spectrogram or cfg.use_ibs):
        return
    _alloc(model, eng)
    if cfg.use_spectrogram:
        pre = "spectrogram_generator."
        spec_cnn_pack(eng, pre)
    if cfg.use_ibs and cfg.use_robust_ibs:
        pre, ntok, E = "ibs_tokenizer.", cfg.num_ibs_tokens, eng.C * eng.C
        eng.p_cast(fp.p_ptr(pre + "bottleneck.0.weight"), ptr(w["ib0"]), 64 * E)
        eng.p_transpose(fp.p_ptr(pre + "bottleneck.0.weight"), ptr(w["ib0T"]), 64, E, 64)
        eng.p_cast(fp.p_ptr(pre + "bottleneck.3.weight"), ptr(w["ib3"]), d * 64)
        eng.p_transpose(fp.p_ptr(pre + "bottleneck.3.weight"), ptr(w["ib3T"]), d, 64, d)
        # additive row table of the token GEMM: type_embedding[i] + pos[1 + i]   (D:909, A:120-126)
        call("eg_rows_bcast_f32", fp.p_ptr(pre + "type_embedding"), fp.p_ptr("pos_embed.pos_embed.weight"), ptr(w["ib_add"]), 1,
             1 + ntok, d, ntok, 1, 1, dt, st)
    elif cfg.use_ibs:
        pre = "ibs_generator.proj."
        call("eg_pack_conv_weight", fp.p_ptr(pre + "0.weight"), ptr(w["ig0"]), 2 * d, 28, 1, 28, 64, dt, st)
        eng.p_cast(fp.p_ptr(pre + "3.weight"), ptr(w["ig3"]), d * 2 * d)
        eng.p_transpose(fp.p_ptr(pre + "3.weight"), ptr(w["ig3T"]), d, 2 * d, d)


# ------------------------------------------------------------------------------------------------
def ibs_analytic(eng: Engine, eeg1, eeg2):
    """Band-passed signals, Hilbert phases and per-signal statistics of all 2B*C signals (D:527-591)."""
    ib, a = eng.ib, eng.a
    lo, hi, nb = _bands(eng, ib["bands"])
    xcat = torch.cat([eeg1, eeg2], 0)  # [NB, C, T] player-1 windows then player-2 windows (device copy, plumbing)
    eng._keep = xcat
    call("eg_ibs_analytic", ptr(xcat), ptr(a["ib_xb"]), ptr(a["ib_ph"]), ptr(a["ib_stats"]), ptr(a["ib_spec"]), ib["nsig"], eng.T,
         float(eng.cfg.sampling_rate), ib["nbin"], lo, hi, nb, eng.stream)


def ibs_pairs(eng: Engine):
    """[B, 6, 7, C, C] connectivity matrices from the analytic signals (D:593-819)."""
    ib, a = eng.ib, eng.a
    lo, hi, nb = _bands(eng, ib["bands"])
    call("eg_ibs_pairs", ptr(a["ib_xb"]), ptr(a["ib_ph"]), ptr(a["ib_stats"]), ptr(a["ib_spec"]), ptr(a["ib_conn"]), eng.B, eng.C,
         eng.T, float(eng.cfg.sampling_rate), ib["nbin"], lo, hi, nb, eng.stream)


def ibs_matrices(model, eng: Engine, eeg1, eeg2) -> torch.Tensor:
    """Stand-alone call of the matrix generator (what `model.ibs_matrix_generator(eeg1, eeg2)` returns, D:760-819)."""
    _alloc(model, eng)
    ibs_analytic(eng, eeg1, eeg2)
    ibs_pairs(eng)
    return eng.a["ib_conn"].index_select(2, eng.a["ib_fidx"].long())


def forward(model, eng: Engine, eeg1, eeg2, train: bool):
    cfg = eng.cfg
    if not (cfg.use_spectrogram or cfg.use_ibs):
        return
    d, B, NB, Cn, T, S, a, w, fp, es, st, dt = cfg.d_model, eng.B, eng.NB, eng.C, eng.T, eng.S, eng.a, eng.w, eng.fp, eng.es, eng.stream, eng.dtype
    p01 = 0.1 if train else 0.0
    n_ibs = eng.n_ibs
    if cfg.use_ibs:
        ib = eng.ib
        lo, hi, nb = _bands(eng, ib["bands"])
        fs = float(cfg.sampling_rate)
        ibs_analytic(eng, eeg1, eeg2)
        if cfg.use_robust_ibs:
            ntok, E = cfg.num_ibs_tokens, Cn * Cn
            ibs_pairs(eng)
            gen = model.ibs_matrix_generator
            if gen._forward_hooks or gen._forward_pre_hooks:
                # analysis contract (5_Metrics/eeg_metrics.py:195-205, 335-343): a forward hook on `ibs_matrix_generator`
                # sees the [B, 6, n_feat, C, C] matrices and may edit them in place or return a replacement
                idx = a["ib_fidx"].long()
                gen._pending = a["ib_conn"].index_select(2, idx)
                out = gen(eeg1, eeg2)
                a["ib_conn"].index_copy_(2, idx, out.to(device=eng.device, dtype=torch.float32))
            pre = "ibs_tokenizer."
            inorm = cfg.ibs_instance_norm
            call("eg_ibs_inorm", ptr(a["ib_conn"]), ptr(a["ib_fidx"]), fp.p_ptr(pre + "instance_norm.weight") if inorm else 0,
                 fp.p_ptr(pre + "instance_norm.bias") if inorm else 0, ptr(a["ib_in"]), ptr(a["ib_xhat"]), B, nb,
                 cfg.num_ibs_features, E, 1 if inorm else 0, dt, st)
            eng.gemm(ptr(a["ib_in"]), ptr(w["ib0"]), ptr(a["ib_u"]), B * ntok, 64, E, bias=fp.p_ptr(pre + "bottleneck.0.bias"))
            call("eg_gelu_fwd", ptr(a["ib_u"]), ptr(a["ib_h"]), B * ntok * 64, dt, p01, SITE_IBSTOK, eng.st_ptr, st)
            eng.gemm(ptr(a["ib_h"]), ptr(w["ib3"]), ptr(a["x0"]) + d * es, B * ntok, d, 64, c=rowmap(d, S * d, ntok),
                     r=rowmap(d, 0, ntok), bias=fp.p_ptr(pre + "bottleneck.3.bias"), residual=ptr(w["ib_add"]) + d * es)
        else:
            call("eg_ibs_scalar", ptr(a["ib_xb"]), ptr(a["ib_ph"]), ptr(a["ib_spec"]), ptr(a["ig_feat"]), B, Cn, T, fs, ib["nbin"],
                 lo, hi, nb, 0, nb, 64, st)
            call("eg_cast", ptr(a["ig_feat"]), ptr(a["ig_featp"]), B * 64, dt, st)
            pre = "ibs_generator.proj."
            eng.gemm(ptr(a["ig_featp"]), ptr(w["ig0"]), ptr(a["ig_h"]), B, 2 * d, 64, bias=fp.p_ptr(pre + "0.bias"), act=L.ACT_RELU,
                     drop1=(p01, SITE_IBSGEN))
            eng.gemm(ptr(a["ig_h"]), ptr(w["ig3"]), ptr(a["x0"]) + d * es, B, d, 2 * d, c=rowmap(S * d), r=rowmap(0),
                     bias=fp.p_ptr(pre + "3.bias"), residual=ptr(w["pos"]) + d * es)
        # both streams carry the same synchrony tokens (D:1163-1165)
        call("eg_rows_copy", ptr(a["x0"]), S, d, n_ibs, 1, 0, B, B, dt, st)
    if cfg.use_spectrogram:
        sp, pre = eng.sp, "spectrogram_generator."
        F, nfr, Hp, Wp, nimg = sp["F"], sp["nfr"], sp["Hp"], sp["Wp"], sp["nimg"]
        win = model.spectrogram_generator.window
        for i, x in enumerate((eeg1, eeg2)):
            call("eg_stft_logmag", ptr(x), ptr(win), ptr(a["spimg"]) + i * B * Cn * F * nfr * 4, B * Cn, T, cfg.spec_n_fft,
                 cfg.spec_hop_length, F, st)
        off = (1 + n_ibs) * d * es
        spec_cnn_forward(eng, pre, p01, model.spectrogram_generator.spec_conv[3], ptr(a["x0"]) + off, rowmap(d, S * d, Cn),
                         rowmap(d, 0, Cn), ptr(w["pos"]) + off)


def spec_cnn_forward(eng, pre, p01, conv2, out_ptr, c_map, r_map, residual_ptr):
    """The reference's 2-D CNN over one-channel images (D:70-86, D:123-130): Conv2d(1->32,3x3)+ReLU+MaxPool2 (one kernel),
    Conv2d(32->64,3x3)+ReLU as a segmented-row GEMM, AdaptiveAvgPool(4,4), Linear(1024->2d)+ReLU+Dropout, Linear(2d->d) whose
    epilogue writes rows addressed by c_map at out_ptr (+ residual rows r_map).  Images are eng.a["spimg"] [nimg, F, nfr] f32;
    parameters are `pre`spec_conv.{0,3}.* / `pre`proj.{0,3}.*.  Shared by the spectrogram tokens of DualEEGTransformer and by the
    image branch of the multimodal fusion model (image_encoder.py)."""
    a, w, fp, dt, st, sp, d = eng.a, eng.w, eng.fp, eng.dtype, eng.stream, eng.sp, eng.cfg.d_model
    F, nfr, Hp, Wp, nimg = sp["F"], sp["nfr"], sp["Hp"], sp["Wp"], sp["nimg"]
    call("eg_spec_conv1_fwd", ptr(a["spimg"]), fp.p_ptr(pre + "spec_conv.0.weight"), fp.p_ptr(pre + "spec_conv.0.bias"),
         ptr(a["sp_p1"]), nimg, F, nfr, dt, st)
    row = (Wp + 4) * 32
    if dt != L.EG_F32 and CONV2_FWD_FLAT and 2 * (Wp + 4) + 2 <= 64:
        # 16-bit operands: nine row offsets of one LDS image of p1 (csrc/spec.hip) instead of a segmented-row product that re-reads
        # every pixel twelve times through L2
        Q = nimg * (Hp + 2) * (Wp + 4)
        call("eg_conv2d_flat", ptr(a["sp_p1"]), ptr(w["spc2"]), fp.p_ptr(pre + "spec_conv.3.bias"), ptr(a["sp_out2"]), Q,
             Q + 4 * (Wp + 4), Wp + 4, Wp, 32, 64, L.ACT_RELU, 2 * eng.cus, dt, st)
    else:
        eng.gemm(ptr(a["sp_p1"]), ptr(w["spc2"]), ptr(a["sp_out2"]), sp["rows"], 64, 384, a=rowmap(32, row, Wp), seg=(128, row),
                 bias=fp.p_ptr(pre + "spec_conv.3.bias"), act=L.ACT_RELU)
    if conv2 is not None and conv2._forward_hooks:
        # Grad-CAM contract (5_Metrics/eeg_metrics.py:742-764): a forward hook on spec_conv[3] sees that layer's output
        # (before the ReLU the production GEMM fuses), once per stream, as [B*C, 64, H', W']
        tmp = torch.empty_like(a["sp_out2"])
        eng.gemm(ptr(a["sp_p1"]), ptr(w["spc2"]), ptr(tmp), sp["rows"], 64, 384, a=rowmap(32, row, Wp), seg=(128, row),
                 bias=fp.p_ptr(pre + "spec_conv.3.bias"))
        act = tmp.view(nimg, Hp + 2, Wp, 64)[:, :Hp].permute(0, 3, 1, 2).float()
        for half in (act[: nimg // 2], act[nimg // 2:]):
            for hook in list(conv2._forward_hooks.values()):
                hook(conv2, (None,), half.contiguous())
    call("eg_spec_avgpool_fwd", ptr(a["sp_out2"]), ptr(a["sp_pooled"]), nimg, Hp, Wp, dt, st)
    eng.gemm(ptr(a["sp_pooled"]), ptr(w["spp0"]), ptr(a["sp_hp0"]), nimg, 2 * d, 1024, bias=fp.p_ptr(pre + "proj.0.bias"),
             act=L.ACT_RELU, drop1=(p01, SITE_SPEC))
    eng.gemm(ptr(a["sp_hp0"]), ptr(w["spp3"]), out_ptr, nimg, d, 2 * d, c=c_map, r=r_map, bias=fp.p_ptr(pre + "proj.3.bias"),
             residual=residual_ptr)


def spec_cnn_pack(eng, pre):
    """compute-dtype copies of the CNN's GEMM weights (run once per optimiser step, from the fp32 masters)"""
    fp, w, dt, st, d = eng.fp, eng.w, eng.dtype, eng.stream, eng.cfg.d_model
    call("eg_pack_conv2d_weight", fp.p_ptr(pre + "spec_conv.3.weight"), ptr(w["spc2"]), 64, 32, 0, dt, st)
    call("eg_pack_conv2d_weight", fp.p_ptr(pre + "spec_conv.3.weight"), ptr(w["spc2T"]), 64, 32, 1, dt, st)
    eng.p_cast(fp.p_ptr(pre + "proj.0.weight"), ptr(w["spp0"]), 2 * d * 1024)
    eng.p_transpose(fp.p_ptr(pre + "proj.0.weight"), ptr(w["spp0T"]), 2 * d, 1024, 2 * d)
    eng.p_cast(fp.p_ptr(pre + "proj.3.weight"), ptr(w["spp3"]), d * 2 * d)
    eng.p_transpose(fp.p_ptr(pre + "proj.3.weight"), ptr(w["spp3T"]), d, 2 * d, d)


def spec_cnn_alloc(eng, nimg, F, nfr):
    """workspaces of spec_cnn_forward / spec_cnn_backward for nimg images of F x nfr"""
    d, f32 = eng.cfg.d_model, torch.float32
    a, w = eng.a, eng.w
    Hp, Wp = F // 2, nfr // 2
    eng.sp = dict(F=F, nfr=nfr, Hp=Hp, Wp=Wp, nimg=nimg, rows=nimg * (Hp + 2) * Wp)
    w["spc2"] = eng._t(64, 384)
    w["spc2T"] = eng._t(32, 768)
    w["spp0"] = eng._t(2 * d, 1024)
    w["spp0T"] = eng._t(1024, 2 * d)
    w["spp3"] = eng._t(d, 2 * d)
    w["spp3T"] = eng._t(2 * d, d)
    a["spimg"] = eng._t(nimg, F, nfr, dtype=f32)
    a["sp_p1"] = eng._t(nimg * (Hp + 2) * (Wp + 4) * 32 + 4 * (Wp + 4) * 32)   # + slack rows read by the unused tail rows
    a["sp_out2"] = eng._t(nimg * (Hp + 2) * Wp, 64)
    a["sp_pooled"] = eng._t(nimg, 1024)
    a["sp_hp0"] = eng._t(nimg, 2 * d)


def spec_cnn_alloc_bwd(eng):
    sp, d, g = eng.sp, eng.cfg.d_model, eng.g
    g["sp_d2"] = eng._t(sp["nimg"] * (sp["Hp"] + 2) * (sp["Wp"] + 4) * 64 + 4 * (sp["Wp"] + 4) * 64)
    g["sp_dp1"] = eng._t(sp["rows"], 32)
    g["sp_dpooled"] = eng._t(sp["nimg"], 1024)
    g["sp_dhp0"] = eng._t(sp["nimg"], 2 * d)
    g["sp_part"] = eng._t(sp["nimg"], 320, dtype=torch.float32)


def spec_cnn_backward(eng, pre, dy_ptr, dmap, sc01):
    """backward of spec_cnn_forward from the gradient rows dy (addressed by dmap) of its output"""
    a, w, g, fp, dt, st, sp, d, es = eng.a, eng.w, eng.g, eng.fp, eng.dtype, eng.stream, eng.sp, eng.cfg.d_model, eng.es
    F, nfr, Hp, Wp, nimg, rows = sp["F"], sp["nfr"], sp["Hp"], sp["Wp"], sp["nimg"], sp["rows"]
    eng.wgrad(dy_ptr, ptr(a["sp_hp0"]), 0, nimg, d, 2 * d, y=dmap, linear=[pre + "proj.3"])
    eng.gemm(dy_ptr, ptr(w["spp3T"]), ptr(g["sp_dhp0"]), nimg, 2 * d, d, a=dmap, gate=ptr(a["sp_hp0"]), gate_scale=sc01)
    eng.wgrad(ptr(g["sp_dhp0"]), ptr(a["sp_pooled"]), 0, nimg, 2 * d, 1024, linear=[pre + "proj.0"])
    eng.gemm(ptr(g["sp_dhp0"]), ptr(w["spp0T"]), ptr(g["sp_dpooled"]), nimg, 1024, 2 * d)
    call("eg_spec_avgpool_bwd", ptr(a["sp_out2"]), ptr(g["sp_dpooled"]), ptr(g["sp_d2"]), nimg, Hp, Wp, dt, st)
    row32, row64 = (Wp + 4) * 32, (Wp + 4) * 64
    if dt != L.EG_F32 and CONV2_WGRAD_FLAT and 2 * (Wp + 4) + 2 <= 64:   # (the kernel's LDS halo holds two image rows + 2 pixels)
        # 16-bit operands: the flat correlation (csrc/spec.hip) reads every activation / gradient byte once; the im2col product below
        # fetched p1 twelve times and d2 three times through L2 (0.93 ms of the C = 32 step)
        Q = nimg * (Hp + 2) * (Wp + 4)
        splits = L.lib().eg_conv2d_wgrad_flat_splits(Q, min(2 * eng.cus, eng.tn_cap // (64 * 384)))
        # (the bias gradient rides in the same launch: column sums of the gradient rows as one more MFMA product; cspart holds 512 x 768)
        call("eg_conv2d_wgrad_flat", ptr(g["sp_d2"]), ptr(a["sp_p1"]), ptr(g["partial"]), ptr(g["cspart"]), Q, Q + 4 * (Wp + 4), Wp + 4,
             splits, dt, st)
        call("eg_unpack_conv2d_wgrad", ptr(g["partial"]), fp.g_ptr(pre + "spec_conv.3.weight"), splits, 64, 32, st)
        call("eg_reduce_partials", ptr(g["cspart"]), fp.g_ptr(pre + "spec_conv.3.bias"), 64, splits, 64, 0, st)
    else:
        eng.wgrad(ptr(g["sp_d2"]) + (row64 + 64) * es, ptr(a["sp_p1"]), fp.g_ptr(pre + "spec_conv.3.weight"), rows, 64, 384,
                  y=rowmap(64, row64, Wp), x=rowmap(32, row32, Wp), x_tile_stride=row32, conv2d=(64, 32),
                  out_b=fp.g_ptr(pre + "spec_conv.3.bias"))
    if dt != L.EG_F32 and CONV2_FWD_FLAT and 2 * (Wp + 4) + 2 <= 64:
        Q = nimg * (Hp + 2) * (Wp + 4)
        call("eg_conv2d_flat", ptr(g["sp_d2"]), ptr(w["spc2T"]), 0, ptr(g["sp_dp1"]), Q, Q + 4 * (Wp + 4), Wp + 4, Wp, 64, 32, L.ACT_NONE,
             2 * eng.cus, dt, st)
    else:
        eng.gemm(ptr(g["sp_d2"]), ptr(w["spc2T"]), ptr(g["sp_dp1"]), rows, 32, 768, a=rowmap(64, row64, Wp), seg=(256, row64))
    call("eg_spec_conv1_bwd", ptr(a["spimg"]), fp.p_ptr(pre + "spec_conv.0.weight"), fp.p_ptr(pre + "spec_conv.0.bias"),
         ptr(g["sp_dp1"]), ptr(g["sp_part"]), nimg, F, nfr, dt, st)
    call("eg_reduce_partials", ptr(g["sp_part"]), fp.g_ptr(pre + "spec_conv.0.weight"), 288, nimg, 320, 0, st)
    call("eg_reduce_partials", ptr(g["sp_part"]) + 288 * 4, fp.g_ptr(pre + "spec_conv.0.bias"), 32, nimg, 320, 0, st)


# ------------------------------------------------------------------------------------------------
def backward(model, eng: Engine, dseq):
    cfg = eng.cfg
    if not (cfg.use_spectrogram or cfg.use_ibs):
        return
    _alloc_bwd(model, eng)
    d, B, NB, Cn, S, a, w, g, fp, es, st, dt = cfg.d_model, eng.B, eng.NB, eng.C, eng.S, eng.a, eng.w, eng.g, eng.fp, eng.es, eng.stream, eng.dtype
    p, p01, train = eng.train_flags
    sc01 = 1.0 / (1.0 - p01) if p01 > 0 else 1.0
    n_ibs = eng.n_ibs
    if cfg.use_ibs and cfg.use_robust_ibs:
        pre, ntok, E = "ibs_tokenizer.", cfg.num_ibs_tokens, Cn * Cn
        M = B * ntok
        # tokens are shared by both streams: their gradient is the sum of the two sequences' rows
        call("eg_rows_gather_gate", ptr(dseq), 0, ptr(g["ib_dtok"]), rowmap(d), B, S, d, ntok, 1, B, 1.0, dt, st)
        call("eg_batch_rowsum", ptr(g["ib_dtok"]), fp.g_ptr(pre + "type_embedding"), B, ntok, d, ntok, dt, st)
        eng.wgrad(ptr(g["ib_dtok"]), ptr(a["ib_h"]), 0, M, d, 64, linear=[pre + "bottleneck.3"])
        eng.gemm(ptr(g["ib_dtok"]), ptr(w["ib3T"]), ptr(g["ib_dh"]), M, 64, d)
        call("eg_gelu_bwd", ptr(a["ib_u"]), ptr(g["ib_dh"]), ptr(g["ib_du"]), M * 64, dt, p01, SITE_IBSTOK, eng.st_ptr, st)
        eng.wgrad(ptr(g["ib_du"]), ptr(a["ib_in"]), 0, M, 64, E, linear=[pre + "bottleneck.0"])
        if cfg.ibs_instance_norm:
            eng.gemm(ptr(g["ib_du"]), ptr(w["ib0T"]), ptr(g["ib_dxn"]), M, E, 64)
            nsp = min(256, (M + 3) // 4)
            call("eg_affine_grad", ptr(g["ib_dxn"]), ptr(a["ib_xhat"]), ptr(g["ib_affpart"]), nsp, M, E, dt, st)
            call("eg_reduce_partials", ptr(g["ib_affpart"]), fp.g_ptr(pre + "instance_norm.weight"), E, nsp, 2 * E, 0, st)
            call("eg_reduce_partials", ptr(g["ib_affpart"]) + 4 * E, fp.g_ptr(pre + "instance_norm.bias"), E, nsp, 2 * E, 0, st)
    elif cfg.use_ibs:
        pre = "ibs_generator.proj."
        call("eg_rows_gather_gate", ptr(dseq), 0, ptr(g["ig_dtok"]), rowmap(d), B, S, d, 1, 1, B, 1.0, dt, st)
        eng.wgrad(ptr(g["ig_dtok"]), ptr(a["ig_h"]), 0, B, d, 2 * d, linear=[pre + "3"])
        eng.gemm(ptr(g["ig_dtok"]), ptr(w["ig3T"]), ptr(g["ig_dh"]), B, 2 * d, d, gate=ptr(a["ig_h"]), gate_scale=sc01)
        eng.wgrad(ptr(g["ig_dh"]), ptr(a["ig_featp"]), fp.g_ptr(pre + "0.weight"), B, 2 * d, 64, out_b=fp.g_ptr(pre + "0.bias"),
                  conv=(28, 1, 28))
    if cfg.use_spectrogram:
        sp, pre = eng.sp, "spectrogram_generator."
        F, nfr, Hp, Wp, nimg, rows = sp["F"], sp["nfr"], sp["Hp"], sp["Wp"], sp["nimg"], sp["rows"]
        off = (1 + n_ibs) * d * es
        spec_cnn_backward(eng, pre, ptr(dseq) + off, rowmap(d, S * d, Cn), sc01)


def fire_spec_backward_hooks(model, eng: Engine):
    """Full backward hooks on spectrogram_generator.spec_conv[3] (Grad-CAM, eeg_metrics.py:758-764) receive
    grad_output = d score / d (that layer's output), stream 2 first (autograd runs the second call's backward first)."""
    if not eng.cfg.use_spectrogram:
        return
    conv2 = model.spectrogram_generator.spec_conv[3]
    hooks = list(getattr(conv2, "_backward_hooks", {}).values())
    if not hooks:
        return
    sp = eng.sp
    Hp, Wp, nimg = sp["Hp"], sp["Wp"], sp["nimg"]
    d2 = eng.g["sp_d2"][: nimg * (Hp + 2) * (Wp + 4) * 64].view(nimg, Hp + 2, Wp + 4, 64)
    grad = d2[:, 1: Hp + 1, 1: Wp + 1].permute(0, 3, 1, 2).float()
    for half in (grad[nimg // 2:], grad[: nimg // 2]):
        for hook in hooks:
            hook(conv2, (None,), (half.contiguous(),))
