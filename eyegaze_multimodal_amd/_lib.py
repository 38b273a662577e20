"""ctypes binding of libeyegaze_hip.so (C ABI declared in include/eyegaze_hip.h).

The library is mandatory: importing this module without it raises.  There is NO CPU fallback anywhere in
the package — the product path is the HIP path or nothing.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libeyegaze_hip.so"

EG_F32, EG_BF16, EG_F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
ABI_VERSION = 5


class EgError(RuntimeError):
    pass


class RowMap(C.Structure):
    _fields_ = [("row_stride", C.c_int64), ("group_stride", C.c_int64), ("rows_per_group", C.c_int32), ("_pad", C.c_int32)]


def rowmap(row_stride: int, group_stride: int = 0, rows_per_group: int = 0) -> RowMap:
    return RowMap(int(row_stride), int(group_stride), int(rows_per_group), 0)


class StepState(C.Structure):
    _fields_ = [("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32), ("lr", C.c_float), ("bias_corr1", C.c_float),
                ("bias_corr2", C.c_float), ("grad_scale", C.c_float), ("clip_coef", C.c_float), ("grad_norm", C.c_float),
                ("loss_scale", C.c_float), ("found_inf", C.c_uint32), ("good_steps", C.c_uint32), ("opt_steps", C.c_uint32),
                ("use_dev_t", C.c_uint32), ("skipped", C.c_uint32), ("scaler_on", C.c_uint32), ("_pad", C.c_uint32)]


STATE_WORDS = C.sizeof(StepState) // 4   # 16


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
                ("gate", C.c_void_p), ("out_pre", C.c_void_p), ("state", C.c_void_p),
                ("a", RowMap), ("c", RowMap), ("r", RowMap), ("p", RowMap),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("ldw", C.c_int32),
                ("act", C.c_int32), ("dtype", C.c_int32),
                ("drop1_p", C.c_float), ("drop2_p", C.c_float), ("drop1_site", C.c_uint32), ("drop2_site", C.c_uint32),
                ("gate_scale", C.c_float), ("a_seg_len", C.c_int32), ("a_seg_stride", C.c_int64)]


class FfnDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W1", C.c_void_p), ("W2", C.c_void_p), ("H", C.c_void_p), ("C", C.c_void_p),
                ("bias1", C.c_void_p), ("bias2", C.c_void_p), ("gate", C.c_void_p), ("gate_bits_out", C.c_void_p),
                ("gate_bits_in", C.c_void_p), ("residual", C.c_void_p),
                ("state", C.c_void_p),
                ("lda", C.c_int64), ("ldh", C.c_int64), ("ldc", C.c_int64), ("ldg", C.c_int64), ("ldr", C.c_int64),
                ("M", C.c_int32), ("F", C.c_int32), ("act1", C.c_int32), ("dtype", C.c_int32),
                ("drop_h_p", C.c_float), ("drop_c1_p", C.c_float), ("drop_c2_p", C.c_float),
                ("drop_h_site", C.c_uint32), ("drop_c1_site", C.c_uint32), ("drop_c2_site", C.c_uint32),
                ("gate_scale", C.c_float),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_out", C.c_void_p), ("ln_stats", C.c_void_p)]


class LnBwdProjDesc(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("stats", C.c_void_p), ("gamma", C.c_void_p), ("W_frag", C.c_void_p),
                ("dx", C.c_void_p), ("dx_drop", C.c_void_p), ("dC", C.c_void_p), ("partial", C.c_void_p), ("state", C.c_void_p),
                ("M", C.c_int32), ("d_model", C.c_int32), ("dtype", C.c_int32), ("partial_capacity_blocks", C.c_int32),
                ("drop1_p", C.c_float), ("drop2_p", C.c_float), ("drop1_site", C.c_uint32), ("drop2_site", C.c_uint32)]


class AttnBlockDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("wqkv_frag", C.c_void_p), ("wo_frag", C.c_void_p), ("bqkv", C.c_void_p), ("bo", C.c_void_p),
                ("qkv", C.c_void_p), ("ctx", C.c_void_p), ("lse", C.c_void_p), ("r1", C.c_void_p), ("state", C.c_void_p),
                ("NB", C.c_int32), ("S", C.c_int32), ("d_model", C.c_int32), ("num_heads", C.c_int32), ("dtype", C.c_int32),
                ("attn_drop_p", C.c_float), ("out_drop_p", C.c_float), ("attn_drop_site", C.c_uint32), ("out_drop_site", C.c_uint32),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_out", C.c_void_p), ("ln_stats", C.c_void_p)]


class PackEntry(C.Structure):
    _fields_ = [("src", C.c_uint64), ("dst", C.c_uint64), ("rows", C.c_int32), ("cols", C.c_int32), ("ldd", C.c_int32),
                ("mode", C.c_int32), ("blk0", C.c_int32), ("nblk", C.c_int32)]


class TNProblem(C.Structure):
    _fields_ = [("dY", C.c_uint64), ("X", C.c_uint64), ("partial", C.c_uint64), ("ldy", C.c_int64), ("ldx", C.c_int64),
                ("N", C.c_int32), ("K", C.c_int32), ("part_rows", C.c_int32), ("has_bias", C.c_int32), ("blk0", C.c_int32),
                ("_pad", C.c_int32)]


class ReduceEntry(C.Structure):
    _fields_ = [("partial", C.c_uint64), ("out", C.c_uint64), ("n", C.c_int64), ("stride", C.c_int64),
                ("splits", C.c_int32), ("blk0", C.c_int32)]


class GemmTNDesc(C.Structure):
    _fields_ = [("dY", C.c_void_p), ("X", C.c_void_p), ("partial", C.c_void_p), ("y", RowMap), ("x", RowMap),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("splits", C.c_int32), ("dtype", C.c_int32),
                ("x_tile_stride", C.c_int64), ("part_rows", C.c_int32), ("has_bias", C.c_int32), ("tile", C.c_int32)]


_P, _I, _L, _F, _U = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32

# name -> argtypes; every function returns int (0 = ok).  Mirrors include/eyegaze_hip.h exactly.
SIGNATURES = {
    "eg_device_info": [C.POINTER(C.c_int), C.c_char_p, _I],
    "eg_window_pack": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "eg_pack_table": [_P, _I, _I, _I, _P],
    "eg_cast": [_P, _P, _L, _I, _P],
    "eg_transpose_cast": [_P, _P, _I, _I, _I, _I, _P],
    "eg_pack_conv_weight": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "eg_pack_convT_weight": [_P, _P, _I, _I, _I, _I, _I, _P],
    "eg_gemm_nt": [C.POINTER(GemmDesc), _P],
    "eg_gemm_nt_route": [C.POINTER(GemmDesc)],
    "eg_ffn_chain": [C.POINTER(FfnDesc), _P],
    "eg_ln_bwd_proj": [C.POINTER(LnBwdProjDesc), _P],
    "eg_ln_bwd_proj_blocks": [_I],
    "eg_attn_block_fwd": [C.POINTER(AttnBlockDesc), _P],
    "eg_attn_block_ok": [_I, _I, _I, _I],
    "eg_gemm_tn": [C.POINTER(GemmTNDesc), _P],
    "eg_reduce_partials": [_P, _P, _L, _I, _L, _I, _P],
    "eg_gemm_tn_grouped": [_P, _I, _I, _I, _I, _I, _P],
    "eg_gemm_tn_grouped256": [_P, _I, _I, _I, _I, _I, _P],
    "eg_reduce_table": [_P, _I, _I, _P],
    "eg_unpack_conv_wgrad": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "eg_colsum": [_P, RowMap, _I, _I, _P, _I, _I, _P],
    "eg_layernorm_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "eg_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _U, _F, _U, _P, _P],
    "eg_attention_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _U, _P, _P],
    "eg_attention_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _U, _P, _P],
    "eg_window_normalize": [_P, _P, _P, _I, _I, _I, _I, _P],
    "eg_attention_probs": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "eg_rows_bcast_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "eg_rows_copy": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "eg_pool_fuse_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "eg_pool_fuse_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "eg_classifier_ce_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eg_classifier_ce_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "eg_batch_rowsum": [_P, _P, _I, _I, _I, _I, _I, _P],
    "eg_rows_gather_gate": [_P, _P, _P, RowMap, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "eg_grad_sqnorm": [_P, _L, _P, _I, _P],
    "eg_clip_coef": [_P, _I, _F, _P, _P],
    "eg_adamw": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P, _P],
    "eg_fill_f32": [_P, _L, _F, _P],
    "eg_adamw_group": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P, _P],
    "eg_set_step_state": [_P, _U, _U, _F, _F, _F, _F, _I, _F, _I, _P],
    "eg_scaler_update": [_P, _F, _F, _I, _P],
    "eg_aux_symmetry": [_P, _P, _P, _P, _P, _I, _I, _P],
    "eg_aux_infonce": [_P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _P],
    "eg_aux_supcon": [_P, _P, _F, _P, _P, _P, _I, _I, _P],
    "eg_fuzzy_gate_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P],
    "eg_fusion_loop_loss": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F, _F, _F, _P, _P],
    "eg_fuzzy_gate_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P],
    "eg_ibs_analytic": [_P, _P, _P, _P, _P, _I, _I, _F, _I, _P, _P, _I, _P],
    "eg_ibs_pairs": [_P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P, _P, _I, _P],
    "eg_ibs_scalar": [_P, _P, _P, _P, _I, _I, _I, _F, _I, _P, _P, _I, _I, _I, _I, _P],
    "eg_affine_grad": [_P, _P, _P, _I, _I, _I, _I, _P],
    "eg_ibs_inorm": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "eg_gelu_fwd": [_P, _P, _L, _I, _F, _U, _P, _P],
    "eg_gelu_bwd": [_P, _P, _P, _L, _I, _F, _U, _P, _P],
    "eg_stft_logmag": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "eg_spec_conv1_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eg_spec_conv1_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eg_spec_avgpool_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "eg_spec_avgpool_bwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "eg_pack_conv2d_weight": [_P, _P, _I, _I, _I, _I, _P],
    "eg_unpack_conv2d_wgrad": [_P, _P, _I, _I, _I, _P],
    "eg_conv2d_wgrad_flat": [_P, _P, _P, _P, _L, _L, _I, _I, _I, _P],
    "eg_conv2d_wgrad_flat_splits": [_L, _I],
    "eg_conv2d_flat": [_P, _P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _P],
}
_lib = None


def lib() -> C.CDLL:
    """Loads the library once; raises EgError when it is missing or its ABI differs (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise EgError(f"{LIB_PATH} is missing: build it with `python -m eyegaze_multimodal_amd.build` "
                      "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    l = C.CDLL(str(LIB_PATH))
    l.eg_abi_version.restype = C.c_int
    if l.eg_abi_version() != ABI_VERSION:
        raise EgError(f"ABI mismatch: library {l.eg_abi_version()} vs binding {ABI_VERSION}")
    l.eg_last_error.restype = C.c_char_p
    for name, args in SIGNATURES.items():
        fn = getattr(l, name, None)
        if fn is None:
            continue  # optional (not yet built) entry points are reported by exported_symbols()
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = l
    return l


def gate_bits_bytes(M: int, F: int) -> int:
    """Size of the gate bit image eg_ffn_chain writes / reads for an [M, F] hidden tensor (eg_ffn_gate_bits_bytes)."""
    l = lib()
    l.eg_ffn_gate_bits_bytes.restype = C.c_int64
    l.eg_ffn_gate_bits_bytes.argtypes = [C.c_int, C.c_int]
    return int(l.eg_ffn_gate_bits_bytes(M, F))


def exported_symbols():
    l = lib()
    return {n: hasattr(l, n) for n in list(SIGNATURES) + ["eg_abi_version", "eg_last_error"]}


CALLS = 0  # number of C-ABI calls issued so far (graph.py uses it to skip empty capture segments)


def call(name: str, *args):
    global CALLS
    CALLS += 1
    l = lib()
    rc = getattr(l, name)(*args)
    if rc != 0:
        raise EgError(f"{name}: {l.eg_last_error().decode()}")


def ptr(t) -> int:
    """Device (or host) address of a torch tensor / None."""
    return 0 if t is None else t.data_ptr()
