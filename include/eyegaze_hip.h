/*
 * eyegaze_hip.h — C ABI of libeyegaze_hip.so: the MI355X (gfx950) kernels behind the reference's
 * dual-stream window classifier hot path
 *     run_experiments.py -> 4_Experiments/scripts/train_art.py::train_epoch/evaluate
 *       -> 3_Models/backbones/dual_eeg_transformer.py::DualEEGTransformer.forward (+ art.py encoder)
 *       -> loss -> backward -> clip -> AdamW.
 *
 * The reference has no FFI of its own (it is pure PyTorch); its seam is the nn.Module interface
 * (dual_eeg_transformer.py:995-1021 ctor, :1110-1253 forward).  A maintainer binds this library with
 * ctypes from a drop-in `DualEEGTransformer` (see INTEGRATION.md and
 * eyegaze_multimodal_amd/dual_eeg_transformer.py).  Each entry point below names the reference lines
 * whose arithmetic it replaces ("D:" = dual_eeg_transformer.py, "A:" = art.py, "T:" = train_art.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed or owned by a torch tensor) unless marked host;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work is stream-ordered,
 *     nothing synchronises, nothing allocates;
 *   - dtype: EG_F32 (exact fp32 path, v_mfma_f32_16x16x4_f32) or EG_BF16 (bf16 storage, fp32 accumulate,
 *     v_mfma_f32_16x16x32_bf16).  Parameters, gradients, optimiser state, LayerNorm statistics,
 *     soft-max statistics and losses are always fp32;
 *   - return value: 0 on success, non-zero on a rejected call (eg_last_error() gives the text).
 *     Shape/argument checks happen on the host BEFORE any launch.
 */
#ifndef EYEGAZE_HIP_H
#define EYEGAZE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EG_ABI_VERSION 5
enum { EG_F32 = 0, EG_BF16 = 1, EG_F16 = 2 };
enum { EG_ACT_NONE = 0, EG_ACT_RELU = 1, EG_ACT_GELU = 2 };

int eg_abi_version(void);
const char* eg_last_error(void);
/* fills cu_count / arch name of the current device (host pointers) */
int eg_device_info(int* cu_count, char* arch, int arch_len);

/* Device-resident per-step scalars.  Kernels read them from memory (not from kernel arguments) so a
 * captured hipGraph of one training step replays correctly with a new seed / learning rate. */
typedef struct eg_step_state {
  uint32_t seed_lo, seed_hi; /* dropout seed of this step */
  float lr;                  /* learning rate (cosine schedule, T:409/494) */
  float bias_corr1;          /* 1 - beta1^t */
  float bias_corr2;          /* 1 - beta2^t */
  float grad_scale;          /* multiplies gradients before clipping (1/world_size for a summed all-reduce) */
  float clip_coef;           /* written by eg_clip_coef: min(1, max_norm/(norm+1e-6)) (T:221) */
  float grad_norm;           /* written by eg_clip_coef: global L2 norm before clipping (after un-scaling) */
  /* dynamic loss scaling (fp16 path; torch.cuda.amp.GradScaler semantics, train_multimodal_fuzzy_fusion.py:435-472).
   * These four words live on the device across steps; eg_set_step_state leaves them alone unless asked to reset. */
  float loss_scale;          /* the loss gradient fed to backward is multiplied by this; eg_clip_coef divides it out */
  uint32_t found_inf;        /* written by eg_clip_coef: 1 when the gradient norm is not finite -> eg_adamw skips the step */
  uint32_t good_steps;       /* consecutive finite steps since the last change of loss_scale */
  uint32_t opt_steps;        /* optimiser steps actually taken (skipped steps do not count); t of the bias corrections
                                when use_dev_t != 0 */
  uint32_t use_dev_t;        /* 0: bias_corr1/2 come from the host; 1: eg_adamw derives them from opt_steps + 1 */
  uint32_t skipped;          /* total skipped steps */
  uint32_t scaler_on;        /* 0: loss_scale is 1 and non-finite gradients are NOT intercepted (the reference's fp32 loop) */
  uint32_t _pad;
} eg_step_state;
/* Publishes the host-side scalars of a step.  They travel as KERNEL ARGUMENTS (copied at launch), so a host that runs
 * many steps ahead of the device can never overwrite a value a queued step still has to read.  reset_scaler: 0 leaves the
 * loss-scaling words alone; 1 enables scaling (loss_scale = init_scale, scaler_on = 1); 2 disables it (loss_scale = 1,
 * scaler_on = 0); 1 and 2 also zero found_inf / good_steps / opt_steps / skipped. */
int eg_set_step_state(eg_step_state* state, uint32_t seed_lo, uint32_t seed_hi, float lr, float bias_corr1,
                      float bias_corr2, float grad_scale, int reset_scaler, float init_scale, int use_dev_t, void* stream);

/* Grouped row addressing: row r of a logical [M, *] matrix starts at element
 *   (r / rows_per_group) * group_stride + (r % rows_per_group) * row_stride
 * rows_per_group == 0 means plain rows (r * row_stride).  This is how the strided 1-D convolutions read
 * overlapping windows of a channel-last, zero-padded signal as GEMM rows without an im2col copy. */
typedef struct eg_rowmap {
  int64_t row_stride;
  int64_t group_stride;
  int32_t rows_per_group;
  int32_t _pad;
} eg_rowmap;

/* ---------------------------------------------------------------------------------------------
 * Input staging — D:1127-1128 feed, 1_Data/processed/dual_eeg_dataset.py:236-248 layout [B,C,T] f32
 * x [NB, C, T] f32  ->  xt [NB, Tp, Cp] (dtype), channel-last, `pad_front` zero time steps in front,
 * zeros behind up to Tp, channels C..Cp-1 zero.  Coalesced reads along T, LDS transpose.
 * ------------------------------------------------------------------------------------------- */
int eg_window_pack(const float* x, void* xt, int NB, int C, int T, int Cp, int pad_front, int Tp, int dtype,
                   void* stream);
/* data-path normalisation (1_Data/processed/dual_eeg_dataset.py:142-168 when enable_preprocessing, :201-202 otherwise):
 *   raw [N, 2, C, T] f32 (player-1 window, player-2 window) -> eeg1, eeg2 [N, C, T] f32
 *   mode 0: per-window global z-score, population std + 1e-8
 *   mode 1: common-average reference, then per-channel z-score, population std + 1e-8 */
int eg_window_normalize(const float* raw, float* eeg1, float* eeg2, int N, int C, int T, int mode, void* stream);

/* Parameter staging (fp32 master -> compute dtype copies), run once per optimiser step.
 *   eg_cast:             n contiguous elements
 *   eg_transpose_cast:   src [R, Cc] f32 -> dst[c*ldd + r]         (weights for backward-data GEMMs)
 *   eg_pack_conv_weight: w [N, Cin, k] f32 -> dst [N, Kp], dst[n][tap*Cp + c] = w[n][c][tap], zero padded
 *                        (D:154,158 Conv1d weights in the tap-major order the channel-last GEMM rows need)
 *   eg_pack_convT_weight: backward-data weights of the stride-s conv, one matrix per output phase p:
 *                        dst[p][c][j*N + n] = w[n][c][s*(J-1-j) + p] (0 where the tap is > k-1), J = ceil(k/s)
 */
/* eg_pack_table: all of the above casts / transposes of one model in ONE launch.  `table` is a DEVICE array; entry i owns
 * blocks [blk0, blk0 + nblk): mode 0 cast (1024 elements per block), 1 transpose-cast (one 32x32 tile per block),
 * 2 fp32 copy (fused bias vectors); 3-6 eg_ffn_chain's MFMA-fragment order of a 16-bit weight (2048 elements per block; the
 * source is fp32 [rows, cols]): 3 role 1 = src [F, 256], 4 role 1 = src^T (src [256, F]), 5 role 2 = src [256, F],
 * 6 role 2 = src^T (src [F, 256]); see eg_ffn_desc; 7 / 8 eg_attn_block_fwd's fragment order of a [256, 256] projection weight
 * (7: q / k / v_proj with ldd = 0 / 1 / 2, 8: out_proj; 2048 elements per block).  src / dst are absolute device addresses. */
typedef struct eg_pack_entry {
  uint64_t src, dst;
  int32_t rows, cols, ldd, mode, blk0, nblk;
} eg_pack_entry;
int eg_pack_table(const eg_pack_entry* table, int nentries, int total_blocks, int dtype, void* stream);
int eg_cast(const float* src, void* dst, int64_t n, int dtype, void* stream);
int eg_transpose_cast(const float* src, void* dst, int R, int Cc, int ldd, int dtype, void* stream);
int eg_pack_conv_weight(const float* w, void* dst, int N, int Cin, int k, int Cp, int Kp, int dtype, void* stream);
int eg_pack_convT_weight(const float* w, void* dst, int N, int Cin, int k, int stride, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * eg_gemm_nt — the MFMA workhorse.  Y = epilogue(A[M,K] * W[N,K]^T)
 *   replaces: Conv1d+ReLU+Dropout (D:154-171) via grouped rows, q/k/v/out projections (A:203-213),
 *   FeedForward linears (A:272), head linears (D:939, D:1100-1105, D:1074-1079), tokenizer / spectrogram
 *   projections (D:81-86, D:863-868) and every backward-data product.
 *   epilogue, in this order:  v = acc + bias[n]; v = act(v); v = gate>0 ? v*gate_scale : 0; v = drop1(v);
 *   v = drop2(v); if (out_pre) out_pre = v; v += residual; C = v
 *   K % (128/sizeof(elem)) == 0, N % 8 == 0; rows beyond M are neither read past nor written.
 * ------------------------------------------------------------------------------------------- */
typedef struct eg_gemm_desc {
  const void* A;        /* activations, rows addressed by `a` */
  const void* W;        /* [N, ldw] row-major, K contiguous */
  void* C;              /* rows addressed by `c` */
  const float* bias;    /* [N] fp32 or NULL */
  const void* residual; /* rows addressed by `r`, same dtype, or NULL */
  const void* gate;     /* rows addressed by `c`; output zeroed where gate <= 0 (ReLU backward), or NULL */
  void* out_pre;        /* rows addressed by `p`; value before the residual add, or NULL */
  const eg_step_state* state; /* dropout seed; may be NULL when both p are 0 */
  eg_rowmap a, c, r, p; /* p addresses out_pre */
  int32_t M, N, K, ldw;
  int32_t act;
  int32_t dtype;
  float drop1_p, drop2_p;
  uint32_t drop1_site, drop2_site;
  float gate_scale; /* 1/(1-p) of the dropout that followed the gated ReLU in the forward pass, else 1 */
  int32_t a_seg_len;    /* 0, or: an A row is K/a_seg_len segments of a_seg_len contiguous elements ... */
  int64_t a_seg_stride; /* ... a_seg_stride elements apart (rows of a 2-D convolution window, D:74) */
} eg_gemm_desc;
int eg_gemm_nt(const eg_gemm_desc* d, void* stream);
/* Which kernel eg_gemm_nt launches for `d` (no launch; a measurement aid so that per-launch timings can be attributed):
 * EG_ROUTE_TILED gemm_nt_kernel (128x128 tile), EG_ROUTE_WIDE gemm_nt_wide_kernel (160x256, N == 256),
 * EG_ROUTE_ROWSTREAM rs_gemm_kernel (K == 256, register-stationary weights). */
enum { EG_ROUTE_TILED = 0, EG_ROUTE_WIDE = 1, EG_ROUTE_ROWSTREAM = 2 };
int eg_gemm_nt_route(const eg_gemm_desc* d);

/* ---------------------------------------------------------------------------------------------
 * eg_ffn_chain — the two products of the position-wise feed-forward block (A:264-272) in one launch, d_model == 256:
 *   H[M,F]   = drop_h(gate(act1(A[M,256] * W1[F,256]^T + bias1)))       stored (backward and the weight gradients read it)
 *   C[M,256] = drop_c2(drop_c1(H * W2[256,F]^T + bias2)) + residual
 *   forward  (A:272 linear1 -> ReLU -> dropout -> linear2, then A:294 dropout + residual): A = residual = LayerNorm-1 rows
 *   backward-data of the same block: A = dY, W1 = linear2^T, gate = the saved H rows (value zeroed where gate <= 0, else
 *   scaled by gate_scale), W2 = linear1^T, residual = the gradient arriving on the skip path.
 *   Bit-identical to the two eg_gemm_nt launches it replaces (same MFMA chains, epilogue order and dropout indices m*N + n);
 *   the hidden rows cross HBM once (the stored H) instead of three times.  16-bit dtypes, F % 128 == 0, strides in elements.
 * ------------------------------------------------------------------------------------------- */
typedef struct eg_ffn_desc {
  const void* A;        /* [M, 256], row stride lda */
  const void* W1;       /* [F, 256] in fragment order: eg_pack_table mode 3 (or 4 from the transposed parameter) */
  const void* W2;       /* [256, F] in fragment order: eg_pack_table mode 5 (or 6) */
  void* H;              /* [M, F], row stride ldh */
  void* C;              /* [M, 256], row stride ldc */
  const float* bias1;   /* [F] or NULL */
  const float* bias2;   /* [256] or NULL */
  const void* gate;     /* [M, F], row stride ldg, or NULL */
  /* The same gate as one bit per element ("stored H value > 0"), written by a launch with gate_bits_out set and read by a
   * later launch with gate_bits_in set (then `gate` is not read).  Opaque, in the kernel's own lane order: valid only between
   * launches with equal M and F; size eg_ffn_gate_bits_bytes(M, F). */
  void* gate_bits_out;
  const void* gate_bits_in;
  const void* residual; /* [M, 256], row stride ldr, or NULL; may alias A (then it is taken from the on-chip A tile) */
  const eg_step_state* state;
  int64_t lda, ldh, ldc, ldg, ldr;
  int32_t M, F, act1, dtype;
  float drop_h_p, drop_c1_p, drop_c2_p;
  uint32_t drop_h_site, drop_c1_site, drop_c2_site;
  float gate_scale;
  /* optional (forward): the layer's second LayerNorm (A:295, norm2) on the completed C rows, in the same launch:
   * ln_out = LayerNorm(C) * gamma + beta, ln_stats as eg_layernorm_fwd's.  NULL ln_out = off; needs ldc == 256. */
  const float* ln_gamma;
  const float* ln_beta;
  void* ln_out;         /* out [M, 256] contiguous */
  float* ln_stats;      /* out [M, 2] or NULL */
} eg_ffn_desc;
int eg_ffn_chain(const eg_ffn_desc* d, void* stream);
int64_t eg_ffn_gate_bits_bytes(int M, int F);

/* ---------------------------------------------------------------------------------------------
 * eg_ln_bwd_proj — LayerNorm backward (d_model = 256) and the backward-data product that consumes it, one launch over 80-row tiles:
 *   dx = LayerNorm backward of dy (what eg_layernorm_bwd writes, bit for bit), dx_drop = dropout1(dropout2(dx)) (ditto),
 *   dC = dx_drop * W^T with W[256, 256] pre-packed in fragment order (eg_pack_table mode 5 / 6; bit-identical to eg_gemm_nt),
 *   partial[block][512] = the block's gain | bias gradient sums (eg_ln_bwd_proj_blocks(M) rows; reduce with eg_reduce_partials).
 *   Replaces autograd's backward of `norm1` (A:293) + of `out_proj` (A:213) of an encoder layer.  16-bit dtypes.
 * ------------------------------------------------------------------------------------------- */
typedef struct eg_ln_bwd_proj_desc {
  const void* dy;       /* [M, 256] gradient of the LayerNorm output */
  const void* x;        /* [M, 256] the LayerNorm input */
  const float* stats;   /* [M, 2] mean, rstd of the forward */
  const float* gamma;   /* [256] */
  const void* W_frag;   /* 256 x 256 elements in fragment order */
  void* dx;             /* out [M, 256] */
  void* dx_drop;        /* out [M, 256] */
  void* dC;             /* out [M, 256] */
  float* partial;       /* out [blocks, 512] */
  const eg_step_state* state;
  int32_t M, d_model, dtype, partial_capacity_blocks;
  float drop1_p, drop2_p;
  uint32_t drop1_site, drop2_site;
} eg_ln_bwd_proj_desc;
int eg_ln_bwd_proj(const eg_ln_bwd_proj_desc* d, void* stream);
int eg_ln_bwd_proj_blocks(int M);   /* rows of `partial` a launch over M rows writes */

/* ---------------------------------------------------------------------------------------------
 * eg_attn_block_fwd — the attention half of a post-LN encoder layer in ONE launch, a workgroup per WINDOW (S <= 80 rows):
 *   q|k|v = x Wqkv^T + b (A:203-205) -> softmax(q k^T / sqrt(32)), attention dropout, P v per head (A:206-212) ->
 *   r1 = x + dropout(ctx Wo^T + bo) (A:213, A:292-293's residual; the LayerNorm stays eg_layernorm_fwd).
 *   Stored: qkv [NB*S, 768], lse [NB, 8, S], ctx [NB*S, 256] (what eg_attention_bwd and the weight gradients read) and r1.
 *   Bit-identical to eg_gemm_nt (q|k|v) -> eg_attention_fwd (kv_shift 0) -> eg_gemm_nt (out-proj, drop1 = out_drop, residual x).
 *   16-bit dtypes, d_model == 256, 8 heads; rows contiguous (stride 256 / 768).  Weights in fragment order: wqkv_frag by three
 *   eg_pack_table entries of mode 7 (ldd = 0, 1, 2 for q_proj, k_proj, v_proj, same dst), wo_frag by one entry of mode 8.
 * ------------------------------------------------------------------------------------------- */
typedef struct eg_attn_block_desc {
  const void* x;          /* [NB*S, 256] */
  const void* wqkv_frag;  /* 768 x 256 elements, eg_pack_table mode 7 */
  const void* wo_frag;    /* 256 x 256 elements, eg_pack_table mode 8 */
  const float* bqkv;      /* [768] = q_proj.bias | k_proj.bias | v_proj.bias */
  const float* bo;        /* [256] */
  void* qkv;              /* out [NB*S, 768] */
  void* ctx;              /* out [NB*S, 256] */
  float* lse;             /* out [NB, 8, S] */
  void* r1;               /* out [NB*S, 256] */
  const eg_step_state* state;
  int32_t NB, S, d_model, num_heads, dtype;
  float attn_drop_p, out_drop_p;
  uint32_t attn_drop_site, out_drop_site;
  /* optional: the layer's first LayerNorm (A:293, norm1 of the post-LN layer) on the completed rows, in the same launch:
   * ln_out = LayerNorm(r1) * gamma + beta, ln_stats[2 m] = mean, [2 m + 1] = rstd (what eg_layernorm_fwd writes).  NULL ln_out = off. */
  const float* ln_gamma;
  const float* ln_beta;
  void* ln_out;           /* out [NB*S, 256] */
  float* ln_stats;        /* out [NB*S, 2] or NULL */
} eg_attn_block_desc;
int eg_attn_block_fwd(const eg_attn_block_desc* d, void* stream);
int eg_attn_block_ok(int S, int d_model, int num_heads, int dtype);   /* 1 when eg_attn_block_fwd serves this geometry */

/* ---------------------------------------------------------------------------------------------
 * eg_gemm_tn — weight-gradient product  dW[N,K] = sum_m dY[m,n] * X[m,k]  (fp32 result)
 *   replaces autograd's grad_weight of every Linear / Conv1d above.  Reads both operands row-major
 *   (rows = the reduction index) and transposes on the LDS read (ds_read_b64_tr_b16).  The reduction over
 *   M is split over `splits` workgroups; partial slabs [splits, slab] fp32 go to `partial` and
 *   eg_reduce_partials sums them (deterministic, no atomics).  With has_bias the column sums of dY (the bias
 *   gradient) are produced by the same launch; slab = (N/part_rows) * (part_rows*K + part_rows).
 * ------------------------------------------------------------------------------------------- */
typedef struct eg_gemm_tn_desc {
  const void* dY; /* [M, N] rows addressed by `y` */
  const void* X;  /* [M, K] rows addressed by `x` */
  float* partial; /* [splits, N, K] fp32 workspace */
  eg_rowmap y, x;
  int32_t M, N, K, splits;
  int32_t dtype;
  int64_t x_tile_stride; /* 0/128 = contiguous X rows; else elements between consecutive 128-column tiles */
  int32_t part_rows;     /* 0 = N.  One split's slab is N/part_rows parts of [part_rows x K | part_rows bias sums]: */
  int32_t has_bias;      /* the layout of consecutive (weight, bias) parameters, so ONE eg_reduce_partials writes both */
  int32_t tile;          /* 0 / 128: 128 x 128 tiles (256 threads); 256: 256 x 256 tiles (512 threads; 16-bit dtypes, N and K
                            multiples of 256, contiguous X rows): half the operand traffic per output, same partial slabs */
} eg_gemm_tn_desc;
int eg_gemm_tn(const eg_gemm_tn_desc* d, void* stream);
/* Grouped form: every weight-gradient product of the encoder in ONE launch (plain row-major operands, common
 * reduction length M, common split count).  `probs` is a DEVICE array; problem i owns blocks [blk0, blk0 + tiles*splits)
 * and writes its [splits, slab] partial slabs at `partial`.  eg_reduce_table sums many slab sets in one launch
 * (n and stride must be multiples of 4 floats). */
typedef struct eg_tn_problem {
  uint64_t dY, X, partial;
  int64_t ldy, ldx;
  int32_t N, K, part_rows, has_bias, blk0, _pad;
} eg_tn_problem;
/* entry i owns blocks [blk0, blk0 + nblk): nblk = ceil(n/4 / 256) when splits <= EG_REDUCE_WIDE_SPLITS (one float4 column
 * per thread, splits walked in order), else ceil(n/4 / 8) (8 float4 columns x 32 split lanes, for many short slabs) */
#define EG_REDUCE_WIDE_SPLITS 8
typedef struct eg_reduce_entry {
  uint64_t partial, out;
  int64_t n, stride;
  int32_t splits, blk0;
} eg_reduce_entry;
int eg_gemm_tn_grouped(const eg_tn_problem* probs, int nprob, int total_blocks, int M, int splits, int dtype, void* stream);
/* The same table served by 256 x 256 tiles (16-bit dtypes; every N and K a multiple of 256; blk0 counts
 * (N/256) * (K/256) * splits blocks per problem): half the operand traffic per output, bit-identical partial slabs. */
int eg_gemm_tn_grouped256(const eg_tn_problem* probs, int nprob, int total_blocks, int M, int splits, int dtype, void* stream);
int eg_reduce_table(const eg_reduce_entry* table, int nentries, int total_blocks, void* stream);
/* out[i] = (accumulate ? out[i] : 0) + sum_s partial[s*split_stride + i]; used for dW, db, LayerNorm dgamma/dbeta.  `partial` is
 * scratch: with splits >= 2048 over a short vector the sum runs in two stages and stage 1 overwrites rows of `partial`. */
int eg_reduce_partials(float* partial, float* out, int64_t n, int splits, int64_t split_stride, int accumulate,
                       void* stream);
/* conv weight gradient back to the parameter layout: dW[n][c][tap] = sum_s partial[s][n][tap*Cp + c] */
int eg_unpack_conv_wgrad(const float* partial, float* dW, int splits, int N, int Cin, int k, int Cp, int Kp,
                         void* stream);
/* column sums: out_partial[blk, n] = sum over the block's rows of Y[m, n]; rows addressed by `y` */
int eg_colsum(const void* Y, eg_rowmap y, int M, int N, float* partial, int nblk, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm(eps=1e-5) over the last dim  — A:283,286,293,295,306,328; D:952,968,972
 *   fwd: y = (x-mean)*rstd*gamma+beta, stats [M,2] = (mean, rstd)
 *   bwd: dx from dy; per-block partial dgamma/dbeta [nblk, 2, D] (sum with eg_reduce_partials); `partial` holds
 *        partial_capacity_blocks such rows and nblk beyond that is rejected on the host before any launch;
 *        optional dx_drop = dropout-masked dx (the gradient entering the branch whose output was dropped
 *        before the residual add: drop1/drop2 of A:293,295 and the FeedForward's own final dropout A:272)
 * ------------------------------------------------------------------------------------------- */
int eg_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int M, int D,
                     int dtype, void* stream);
int eg_layernorm_bwd(const void* dy, const void* x, const float* stats, const float* gamma, void* dx, void* dx_drop,
                     float* partial, int nblk, int partial_capacity_blocks, int M, int D, int dtype, float drop1_p,
                     uint32_t drop1_site, float drop2_p, uint32_t drop2_site, const eg_step_state* state, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-head attention core on a fused [NB*S, 3*D] q|k|v buffer — A:206-212; cross form D:967,971
 *   head h of sample b attends over the keys/values of sample (b + kv_shift) mod NB
 *   (kv_shift = 0: Siamese self-attention; kv_shift = NB/2: CrossBrainAttention, both directions at once).
 *   d_k must be 32.  lse [NB, H, S] = log-sum-exp of the scaled scores (for the backward recompute).
 *   attention-probability dropout (A:210) uses site `drop_site`.
 * ------------------------------------------------------------------------------------------- */
int eg_attention_fwd(const void* qkv, void* ctx, float* lse, int NB, int S, int H, int kv_shift, int dtype,
                     float drop_p, uint32_t drop_site, const eg_step_state* state, void* stream);
int eg_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv, int NB, int S,
                     int H, int kv_shift, int dtype, float drop_p, uint32_t drop_site, const eg_step_state* state,
                     void* stream);
/* probs[NB, H, S, S] (fp32) = softmax rows recomputed from qkv and the forward's lse: what a forward hook on the
 * attention-dropout module receives as input (5_Metrics/eeg_metrics.py:433-452).  Analysis only. */
int eg_attention_probs(const void* qkv, const float* lse, float* probs, int NB, int S, int H, int kv_shift, int dtype,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sequence assembly and heads (all [B, d]-sized, latency-bound single launches)
 *   eg_rows_bcast_f32  seq[b, off+r, :] = src[b % src_nb, r, :] + pos[off+r, :]   (cls_token expand + pos, D:1157,1178)
 *   eg_rows_copy       seq[b_dst0+b, off+r, :] = seq[b_src0+b, off+r, :]          (shared IBS tokens, D:1163-1165)
 *   eg_pool_fuse_fwd   cls slices, mean pools, symmetric-fusion operands          (D:1193-1212, D:933-938, D:1222-1223)
 *   eg_classifier_ce_* last Linear(K -> ncls<=16) fused with cross-entropy         (D:1104, D:1078, D:1244, D:1250)
 *   eg_batch_rowsum    out[s,:] = sum_b dseq[b,s,:]  (pos_embed / cls_token gradients)
 *   eg_rows_gather_gate dst rows = (src[b, off+r] (+ src[b+pair_shift, off+r])) * relu-gate * gate_scale
 * ------------------------------------------------------------------------------------------- */
int eg_rows_bcast_f32(const float* src, const float* pos, void* seq, int NB, int S, int D, int R, int off, int src_nb,
                      int dtype, void* stream);
int eg_rows_copy(void* seq, int S, int D, int R, int off, int b_src0, int b_dst0, int nb, int dtype, void* stream);
int eg_pool_fuse_fwd(const void* z, float* cls1, float* cls2, void* comb, void* zf, float* ibs_pool_f, void* ibs_pool,
                     int B, int S, int D, int off, int n_ibs, int ibs_first, int dtype, void* stream);
int eg_pool_fuse_bwd(const void* z, const void* dcomb, const void* dzf, const float* gcls1, const float* gcls2,
                     const void* dibs_pool, const float* gibs_pool, void* dz, int B, int S, int D, int off, int n_ibs,
                     int ibs_first, int dtype, void* stream);
int eg_classifier_ce_fwd(const void* h, const float* W, const float* bias, const int64_t* labels, float* logits,
                         float* sample_loss, float* loss, int B, int K, int ncls, int dtype, void* stream);
int eg_classifier_ce_bwd(const void* h, const float* W, const float* logits, const int64_t* labels, const float* gloss,
                         const float* glogits, float* dlogits, void* dh, float* dW, float* db, int B, int K, int ncls,
                         int use_gate, float gate_scale, int dtype, void* stream);
int eg_batch_rowsum(const void* dseq, float* out, int NB, int S, int D, int rows, int dtype, void* stream);
int eg_rows_gather_gate(const void* src, const void* gate, void* dst, eg_rowmap dmap, int nb, int S, int D, int R,
                        int off, int pair_shift, float gate_scale, int dtype, void* stream);

/* Batch-level auxiliary losses (3_Models/backbones/dual_eeg_transformer.py:1255-1371), fp32, each with the gradient of the
 * loss w.r.t. the [B, D] tokens it reads (upstream gradient 1; the caller scales).  B <= 1024.
 *   eg_aux_symmetry  F.mse_loss(cls1, cls2)                                                                    :1255-1260
 *   eg_aux_infonce   cross_entropy(normalize(ibs) normalize(cat[cls1, cls2])^T / temperature, arange(B))       :1262-1304
 *                    work: 3*B*D + 4*B + 2*B*B floats
 *   eg_aux_supcon    supervised contrastive loss over normalize(ibs) with the reference's 1e-8 guards, mean over the rows
 *                    that have a same-label partner, 0 (and zero gradient) if none has                          :1306-1371
 *                    work: B*D + 5*B + B*B + 4 floats */
int eg_aux_symmetry(const float* cls1, const float* cls2, float* loss, float* d_cls1, float* d_cls2, int B, int D, void* stream);
int eg_aux_infonce(const float* ibs, const float* cls1, const float* cls2, float temperature, float* loss, float* d_ibs,
                   float* d_cls1, float* d_cls2, float* work, int B, int D, void* stream);
int eg_aux_supcon(const float* ibs, const int64_t* labels, float temperature, float* loss, float* d_ibs, float* work, int B,
                  int D, void* stream);

/* FuzzyGatingFusion.forward — 3_Models/fusion/fuzzy_gating_fusion.py:297-390 (config 5's logit-level fusion).
 * params = [tau_img, tau_eeg, c_unreliable_img, c_unreliable_eeg, log_sigma_reliable_img, log_sigma_reliable_eeg,
 *           log_sigma_unreliable_img, log_sigma_unreliable_eeg, beta[4]] (12 device floats);
 * mode 0 full, 1 no_temperature, 2 no_fuzzification, 3 fixed_weights.
 * eg_fuzzy_gate_bwd: what autograd computes through that forward — dz_img, dz_eeg [B,K] and the 12 parameter gradients as
 *   per-workgroup partials partial[ceil(B/128)][12] (sum them in block order with eg_reduce_partials), from dfused [B,K] and
 *   dalpha [B] (or NULL).  torch.clamp semantics: the gradient passes where min <= x <= max. */
/* The loss of the multimodal logit-fusion step and its gradients on the [B, K] logits in one launch --
 * 4_Experiments/scripts/train_multimodal_fuzzy_fusion.py:436-460: total = CE(fused) + lambda_aux_img CE(z_img / T_img) +
 * lambda_aux_eeg CE(z_eeg / T_eeg) + lambda_reg R(T), temperatures detached in the auxiliary terms (fuzzy_gating_fusion.py:334),
 * R the temperature regulariser (:392-419).  losses[5] = total, ce, aux_img, aux_eeg, reg; dfused = d total / d fused (the upstream
 * gradient of eg_fuzzy_gate_bwd); daux_img / daux_eeg = the auxiliary terms' direct gradients on z_img / z_eeg; dtau[2] = the
 * regulariser's gradients on tau_img, tau_eeg.  Every gradient is multiplied by state->loss_scale when the scaler is on. */
int eg_fusion_loop_loss(const float* fused, const float* z_img, const float* z_eeg, const int64_t* labels, const float* params,
                        float* losses, float* dfused, float* daux_img, float* daux_eeg, float* dtau, int B, int K, int mode,
                        float eps_temp, float lambda_aux_img, float lambda_aux_eeg, float lambda_reg, float t_min, float t_max,
                        const eg_step_state* state, void* stream);
int eg_fuzzy_gate_fwd(const float* z_img, const float* z_eeg, const float* params, float* fused, float* alpha, int B,
                      int K, int mode, float eps_temp, float eps_log, float eps_div, void* stream);
int eg_fuzzy_gate_bwd(const float* z_img, const float* z_eeg, const float* params, const float* dfused, const float* dalpha,
                      float* dz_img, float* dz_eeg, float* partial, int B, int K, int mode, float eps_temp, float eps_log,
                      float eps_div, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser over flat fp32 buffers — clip_grad_norm_(1.0) + AdamW (T:221-222, T:401-405)
 *   eg_grad_sqnorm: partial[blk] = sum of squares;  eg_clip_coef: state->grad_norm / clip_coef (no host sync)
 *   eg_adamw: p *= 1-lr*wd; m,v update with g*grad_scale*clip_coef/loss_scale; p -= lr/bc1 * m/(sqrt(v)/sqrt(bc2)+eps);
 *             the whole step is skipped when state->scaler_on && state->found_inf (GradScaler.step)
 * ------------------------------------------------------------------------------------------- */
int eg_grad_sqnorm(const float* g, int64_t n, float* partial, int nblk, void* stream);
int eg_clip_coef(const float* partial, int nblk, float max_norm, eg_step_state* state, void* stream);
int eg_adamw(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
             float weight_decay, const eg_step_state* state, void* stream);
int eg_fill_f32(float* p, int64_t n, float value, void* stream);
/* per-group form of eg_adamw for optimisers with parameter groups (train_multimodal_fuzzy_fusion.py:395-432: one learning
 * rate / weight decay per group): lr = state->lr * lr_mult. */
int eg_adamw_group(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                   float weight_decay, float lr_mult, const eg_step_state* state, void* stream);
/* GradScaler.update() on the device (train_multimodal_fuzzy_fusion.py:471-472): after the optimiser kernels of a step.
 *   found_inf: loss_scale *= backoff, good_steps = 0, skipped += 1
 *   else:      opt_steps += 1, good_steps += 1; good_steps == growth_interval -> loss_scale *= growth, good_steps = 0
 * With scaler_on == 0 only opt_steps advances. */
int eg_scaler_update(eg_step_state* state, float growth, float backoff, int growth_interval, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Inter-stream synchrony ("IBS") features — D:473-819 (connectivity matrices), D:178-470 (scalar variant)
 *   eg_ibs_analytic: per signal x[T] (one window channel): rFFT, then for every band the FFT-mask band-pass
 *       (D:527-560) and FFT Hilbert transform (D:562-591) -> xb, phase [nbands, nsig, T]; stats [nbands, nsig, 4] =
 *       (mean, 1/(std_unbiased+1e-8)) of xb and of xb^2 (D:707-708, 751-752); spec [nsig, nbin] complex bins
 *   eg_ibs_pairs:  conn [B, nbands, 7, C, C] = [PLV, PLI, wPLI, Coherence, Power_Corr, Phase_Diff, Time_Corr]
 *       (D:593-758) for signals ordered [player1 windows | player2 windows] x C channels
 *   eg_ibs_scalar: feats [B, 7*nout_bands] global features of bands band0.. (D:436-461)
 *   eg_ibs_inorm:  token rows [B, nbands*nfeat, C*C] with InstanceNorm1d over the token axis (D:897-901)
 *   eg_gelu_fwd/bwd: GELU(erf)+dropout of the tokenizer bottleneck (D:865-866)
 * eg_stft_logmag — D:98-118: reflect-padded, hann-windowed n_fft-point DFT magnitudes, first F bins, log(.+1e-8)
 *   x [nsig, T] -> img [nsig, F, 1 + T/hop] fp32
 * ------------------------------------------------------------------------------------------- */
int eg_ibs_analytic(const float* x, float* xb, float* phase, float* stats, float* spec, int nsig, int T, float fs,
                    int nbin, const float* band_lo, const float* band_hi, int nbands, void* stream);
int eg_ibs_pairs(const float* xb, const float* phase, const float* stats, const float* spec, float* conn, int B, int C,
                 int T, float fs, int nbin, const float* band_lo, const float* band_hi, int nbands, void* stream);
int eg_ibs_scalar(const float* xb, const float* phase, const float* spec, float* feats, int B, int C, int T, float fs,
                  int nbin, const float* band_lo, const float* band_hi, int nbands, int band0, int nout_bands, int ld,
                  void* stream);
int eg_ibs_inorm(const float* conn, const int* fidx, const float* gamma, const float* beta, void* out, float* xhat, int B,
                 int nbands, int nfeat, int E, int use_norm, int dtype, void* stream);
/* InstanceNorm affine gradients: dgamma[e] = sum_m dy[m,e]*xhat[m,e], dbeta[e] = sum_m dy[m,e], as nsplit row-split
 * partials partial[sp][0][e] (gamma) | partial[sp][1][e] (beta); sum them in order with eg_reduce_partials */
int eg_affine_grad(const void* dy, const float* xhat, float* partial, int nsplit, int M, int E, int dtype, void* stream);
int eg_gelu_fwd(const void* u, void* h, int64_t n, int dtype, float drop_p, uint32_t drop_site, const eg_step_state* state,
                void* stream);
int eg_gelu_bwd(const void* u, const void* dh, void* du, int64_t n, int dtype, float drop_p, uint32_t drop_site,
                const eg_step_state* state, void* stream);
int eg_stft_logmag(const float* x, const float* window, float* img, int nsig, int T, int n_fft, int hop, int F,
                   void* stream);
/* band_lo / band_hi above are HOST arrays of nbands floats (<= 8 bands); every other pointer is a device pointer. */

/* ---------------------------------------------------------------------------------------------
 * 2-D CNN over the STFT image — D:70-77, 124-127.  Conv2d(1,32,3,p1)+ReLU+MaxPool2 is a direct kernel
 * (K = 9 is too shallow for MFMA); Conv2d(32,64,3,p1)+ReLU runs through eg_gemm_nt on segmented channel-last rows
 * (a_seg_len = 4*32); AdaptiveAvgPool2d(4,4) emits PyTorch's (c, py, px) flatten order.
 *   p1 [nimg, Hp+2, Wp+4, 32] zero-padded, out2 [nimg, Hp+2, Wp, 64], d2 [nimg, Hp+2, Wp+4, 64], dp1 [nimg, Hp+2, Wp, 32]
 *   eg_spec_conv1_bwd writes partial [nimg, 320] = (dW[32][9] | db[32]) for eg_reduce_partials
 *   eg_pack_conv2d_weight: dst[n][(ky*4+kx)*Cin + c] = w[n][c][ky][kx]; transposed: dst[c][(ky*4+kx)*N + n] = w[n][c][2-ky][2-kx]
 * ------------------------------------------------------------------------------------------- */
int eg_spec_conv1_fwd(const float* img, const float* w, const float* bias, void* p1, int nimg, int F, int nfr, int dtype,
                      void* stream);
int eg_spec_conv1_bwd(const float* img, const float* w, const float* bias, const void* dp1, float* partial, int nimg,
                      int F, int nfr, int dtype, void* stream);
int eg_spec_avgpool_fwd(const void* out2, void* pooled, int nimg, int Hp, int Wp, int dtype, void* stream);
int eg_spec_avgpool_bwd(const void* out2, const void* dpooled, void* d2, int nimg, int Hp, int Wp, int dtype, void* stream);
int eg_pack_conv2d_weight(const float* w, void* dst, int N, int Cin, int transposed, int dtype, void* stream);
int eg_unpack_conv2d_wgrad(float* partial /* scratch: reduced in place when splits > 64 */, float* dW, int splits, int N, int Cin, void* stream);
/* Weight gradient of Conv2d(32,64,3,p1) (autograd's grad_weight of D:74) as a flat correlation over the padded pixel index q
 * (16-bit operands): partial[split][n][(ky*4+kx)*32 + c] = sum over the split's q of d2[q + rowpx + 1][n] * p1[q + ky*rowpx + kx][c],
 * Q = nimg*(Hp+2)*rowpx pixels, rowpx = Wp + 4.  The pads of d2 must be zero (eg_spec_avgpool_bwd writes interiors only); p1 must hold
 * p1_rows >= Q + 2*rowpx + 2 readable pixel rows.  The slabs are the ones eg_unpack_conv2d_wgrad reduces (pad taps kx = 3 are not
 * written).  bias_partial (optional): per split the column sums of the gradient rows = autograd's grad_bias, for eg_reduce_partials.
 * eg_conv2d_wgrad_flat_splits: the largest split count <= want whose 256-pixel-aligned splits all own pixels. */
int eg_conv2d_wgrad_flat(const void* d2, const void* p1, float* partial, float* bias_partial /* [splits][64] or NULL */, long long Q,
                         long long p1_rows, int rowpx, int splits, int dtype, void* stream);
int eg_conv2d_wgrad_flat_splits(long long Q, int want);
/* Conv2d(32,64,3,p1) forward (D:74, + the ReLU of D:75 when act = EG_ACT_RELU; cin 32, nout 64, W = eg_pack_conv2d_weight(transposed 0))
 * and its backward-data (autograd's grad_input: in = d2, cin 64, nout 32, W = the transposed packing, out = dp1) on the same flat
 * pixel index, 16-bit operands:
 *   out[g*Wp + x][n] = act(bias[n] + sum_{ky,kx,c} in[q + ky*rowpx + kx][c] * W[n][(ky*4+kx)*cin + c]), q = g*rowpx + x, x < Wp,
 * for all Q = nimg*(Hp+2)*rowpx pixel slots; `in` must hold in_rows >= Q + 2*rowpx + 2 readable pixel rows; `splits` = workgroups wanted. */
int eg_conv2d_flat(const void* in, const void* W, const float* bias, void* out, long long Q, long long in_rows, int rowpx, int Wp,
                   int cin, int nout, int act, int splits, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EYEGAZE_HIP_H */
