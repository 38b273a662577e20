// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of the dual-stream window classifier.
// Wave = 64 lanes everywhere.  No CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/eyegaze_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

typedef uint16_t bf16_t;  // storage type for bf16 tensors
// storage type for fp16 tensors: a distinct C++ type, so that overloads / templates pick the IEEE-half conversions and the
// f16 MFMA (v_mfma_f32_16x16x32_f16) wherever the bf16 path picks its own
struct f16_t { uint16_t bits; };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define EG_WAVE 64

// ---------------------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------------------
extern thread_local char eg_err_buf[512];
int eg_fail(const char* fmt, ...);
#define EG_CHECK(cond, ...)            \
  do {                                 \
    if (!(cond)) return eg_fail(__VA_ARGS__); \
  } while (0)
#define EG_LAUNCH_CHECK(name)                                                      \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) return eg_fail("%s launch: %s", name, hipGetErrorString(e__)); \
  } while (0)

// cross-file host helpers (not part of the C ABI)
// gemm.hip: first stage of a long split reduction, IN PLACE (groups of consecutive slabs summed into each group's first slab)
int eg_reduce_groups_inplace(float* partial, long long n, int splits, long long stride, int groups, int* group_out, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// bf16 <-> f32
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // round-to-nearest-even via the hardware convert (NaN stays NaN)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
// both halves in ONE v_cvt_pk_bf16_f32 (written as two scalar converts the compiler emits one convert per value plus a shift and an or)
typedef __attribute__((ext_vector_type(2))) __bf16 eg_bf16x2;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, eg_bf16x2));
}
__device__ __forceinline__ float h2f(uint16_t bits) { return (float)__builtin_bit_cast(_Float16, bits); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }   // round-to-nearest-even
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, f16x2));
}

// 16-bit operand traits: MFMA fragment type, the 16x16x32 MFMA, packing and rounding of the storage type
template <typename T> struct H16;
template <> struct H16<bf16_t> {
  typedef bf16x8 frag;
  __device__ static __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack2bf(lo, hi); }
  __device__ static __forceinline__ float ld(bf16_t v) { return bf2f(v); }
  __device__ static __forceinline__ float round(float v) { return bf2f(f2bf(v)); }
};
template <> struct H16<f16_t> {
  typedef f16x8 frag;
  __device__ static __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack2h(lo, hi); }
  __device__ static __forceinline__ float ld(f16_t v) { return h2f(v.bits); }
  __device__ static __forceinline__ float round(float v) { return h2f(f2h(v)); }
};
// rounding through the storage type (identity for fp32): what a stored element reads back as
template <typename T> __device__ __forceinline__ float round_store(float v) { return H16<T>::round(v); }
template <> __device__ __forceinline__ float round_store<float>(float v) { return v; }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPer16B = 4;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kPer16B = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};
template <> struct Elem<f16_t> {
  static constexpr int kPer16B = 8;
  __device__ static __forceinline__ float ld(const f16_t* p) { return h2f(p->bits); }
  __device__ static __forceinline__ void st(f16_t* p, float v) { p->bits = f2h(v); }
};

// load / store 8 consecutive elements as floats (pointer must be 16-B aligned for bf16, 32-B region for f32)
__device__ __forceinline__ void load8(const float* p, float v[8]) {
  f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void load8(const bf16_t* p, float v[8]) {
  u32x4 a = *(const u32x4*)p;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(a[i] << 16);
    v[2 * i + 1] = __uint_as_float(a[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void load8(const f16_t* p, float v[8]) {
  u32x4 a = *(const u32x4*)p;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t w = a[i];     // (a bit_cast applied directly to a vector element reads element 0 of the vector)
    const f32x2 t = __builtin_convertvector(__builtin_bit_cast(f16x2, w), f32x2);
    v[2 * i] = t[0];
    v[2 * i + 1] = t[1];
  }
}
__device__ __forceinline__ void store8(f16_t* p, const float v[8]) {
  u32x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = pack2h(v[2 * i], v[2 * i + 1]);
  *(u32x4*)p = a;
}
__device__ __forceinline__ void load4(const f16_t* p, float v[4]) {
  u32x2 a = *(const u32x2*)p;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t w = a[i];
    const f32x2 t = __builtin_convertvector(__builtin_bit_cast(f16x2, w), f32x2);
    v[2 * i] = t[0];
    v[2 * i + 1] = t[1];
  }
}
__device__ __forceinline__ void store4(f16_t* p, const float v[4]) {
  u32x2 a;
  a[0] = pack2h(v[0], v[1]);
  a[1] = pack2h(v[2], v[3]);
  *(u32x2*)p = a;
}
__device__ __forceinline__ void store8(float* p, const float v[8]) {
  f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  *(f32x4*)p = a;
  *(f32x4*)(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16_t* p, const float v[8]) {
  u32x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = pack2bf(v[2 * i], v[2 * i + 1]);
  *(u32x4*)p = a;
}
__device__ __forceinline__ void load4(const float* p, float v[4]) {
  f32x4 a = *(const f32x4*)p;
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
}
__device__ __forceinline__ void load4(const bf16_t* p, float v[4]) {
  u32x2 a = *(const u32x2*)p;
  v[0] = __uint_as_float(a[0] << 16); v[1] = __uint_as_float(a[0] & 0xffff0000u);
  v[2] = __uint_as_float(a[1] << 16); v[3] = __uint_as_float(a[1] & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float v[4]) {
  f32x4 a = {v[0], v[1], v[2], v[3]};
  *(f32x4*)p = a;
}
__device__ __forceinline__ void store4(bf16_t* p, const float v[4]) {
  u32x2 a;
  a[0] = pack2bf(v[0], v[1]);
  a[1] = pack2bf(v[2], v[3]);
  *(u32x2*)p = a;
}

// ---------------------------------------------------------------------------------------------
// counter-based dropout mask, identical in forward and backward because it depends only on
// (seed, site, element index).  One 32-bit hash serves TWO consecutive elements (its 16-bit halves):
// keep <=> half >= round(p * 2^16).  32-bit integer multiplies are quarter rate on CDNA, so halving the
// hash count matters (the hash is ~60 % of the attention backward's VALU time).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t eg_hash(uint32_t seed_lo, uint32_t seed_hi, uint32_t site, uint32_t idx) {
  // Per-site words (wave-uniform: scalar ALU, hoisted out of the element loops): every dropout site of a step gets its own
  // pair (a, b), so two sites -- or two steps, the host passes a scrambled 64-bit seed (engine.scramble_seed) -- are never
  // XOR-permutations or constant offsets of one fixed pattern (with a plain `idx ^ site ^ seed` every row kept the same
  // NUMBER of elements at every step).
  const uint32_t k = site * 0x9E3779B9u;
  uint32_t a = (seed_lo ^ k) * 0x85EBCA6Bu; a ^= a >> 15;
  uint32_t b = (seed_hi + k) * 0xC2B2AE35u; b ^= b >> 13;
  // per element: "lowbias32" finalizer (two multiplies) with b folded in between the rounds
  uint32_t x = idx ^ a;
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x += b;
  x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
struct DropCfg {
  uint32_t thresh;  // round(p * 2^16); 0 = disabled
  float scale;      // 1/(1-p)
  uint32_t site;
};
// single element (index idx): uses half (idx & 1) of the pair hash
__device__ __forceinline__ float eg_dropout(float v, const DropCfg& d, uint32_t seed_lo, uint32_t seed_hi, uint32_t idx) {
  if (d.thresh == 0) return v;
  const uint32_t h = eg_hash(seed_lo, seed_hi, d.site, idx >> 1);
  const uint32_t half = (idx & 1u) ? (h >> 16) : (h & 0xFFFFu);
  return half >= d.thresh ? v * d.scale : 0.0f;
}
// N consecutive elements starting at an EVEN index: N/2 hashes
template <int N>
__device__ __forceinline__ void eg_dropout_run(float (&v)[N], const DropCfg& d, uint32_t seed_lo, uint32_t seed_hi,
                                               uint32_t idx0) {
  if (d.thresh == 0) return;
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    const uint32_t h = eg_hash(seed_lo, seed_hi, d.site, (idx0 + j) >> 1);
    v[j] = (h & 0xFFFFu) >= d.thresh ? v[j] * d.scale : 0.0f;
    v[j + 1] = (h >> 16) >= d.thresh ? v[j + 1] * d.scale : 0.0f;
  }
}
static inline DropCfg make_drop(float p, uint32_t site) {
  DropCfg d;
  d.thresh = p > 0.f ? (uint32_t)(p * 65536.0f + 0.5f) : 0u;
  d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  d.site = site;
  return d;
}

// ---------------------------------------------------------------------------------------------
// wave reductions (64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over d_model = 256 as the tail of a row-complete epilogue (eg_attn_block_fwd -> ln1, eg_ffn_chain forward -> ln2):
// four waves own the 256 columns of an 80-row tile (wave wn: columns 64 wn .. +63), and after the final epilogue lane
// (er = lane / 4, ec = lane % 4) holds, per 16-row tile i, the 16 STORED values vv[i][0..15] of row 16 i + er, columns
// 64 wn + 16 ec ..  Two-pass statistics as layernorm_fwd256_kernel (mean, then sum of squared deviations; eps 1e-5), the row sums
// gathered across the quad by two shuffles and across the waves through `red` (LDS, 2 x 4 x 80 floats): two workgroup barriers per
// launch instead of a launch of its own that re-reads the rows.  The summation ORDER differs from the stand-alone kernel's, so the
// outputs agree with it to rounding (tests/test_gpu_lnfuse.py), not bit for bit.  Rows >= nvalid are skipped (vv must be 0 there).
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void eg_epilogue_layernorm256(const float (&vv)[5][16], float* red, int wn, int lane, int nvalid, size_t row0,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         T* __restrict__ out, float* __restrict__ stats) {
  const int er = lane >> 2, ec = lane & 3;
  const int n = 64 * wn + 16 * ec;
  float* const rs = red;                 // [4][80] row sums per wave
  float* const rq = red + 4 * 80;        // [4][80] sums of squared deviations
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += vv[i][j];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (ec == 0) rs[wn * 80 + 16 * i + er] = s;
  }
  __syncthreads();
  float mean[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int r = 16 * i + er;
    mean[i] = (((rs[r] + rs[80 + r]) + rs[160 + r]) + rs[240 + r]) * (1.0f / 256.f);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float d = vv[i][j] - mean[i]; q += d * d; }
    q += __shfl_xor(q, 1, 64);
    q += __shfl_xor(q, 2, 64);
    if (ec == 0) rq[wn * 80 + r] = q;
  }
  __syncthreads();
  float g[16], bt[16];
  load8(gamma + n, g); load8(gamma + n + 8, g + 8);
  load8(beta + n, bt); load8(beta + n + 8, bt + 8);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int r = 16 * i + er;
    if (r < nvalid) {
      const float rstd = rsqrtf((((rq[r] + rq[80 + r]) + rq[160 + r]) + rq[240 + r]) * (1.0f / 256.f) + 1e-5f);
      float o[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) o[j] = (vv[i][j] - mean[i]) * rstd * g[j] + bt[j];
      T* po = out + (row0 + r) * 256 + n;
      store8(po, o);
      store8(po + 8, o + 8);
      if (stats && wn == 0 && ec == 0) { stats[2 * (row0 + r)] = mean[i]; stats[2 * (row0 + r) + 1] = rstd; }
    }
  }
}

// LayerNorm backward of one row of 256, 8 elements per lane over a HALF-wave (32 lanes): shared by layernorm_bwd256_kernel and
// ln_bwd_proj_kernel so that both produce the same bits -- every product / sum is an explicit round-to-nearest intrinsic (left to the
// compiler, `dv * g - c1` became an fma in one kernel and a rounded product minus c1 in the other: one bf16 step apart now and then).
__device__ __forceinline__ float eg_half_sum32(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void eg_ln_bwd_row8(const float (&xv)[8], const float (&dv)[8], const float (&g)[8], float mean, float rstd,
                                               float (&dg)[8], float (&db)[8], float (&o)[8]) {
  float xh[8], dxh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    xh[e] = __fmul_rn(__fsub_rn(xv[e], mean), rstd);
    dg[e] = __fmaf_rn(dv[e], xh[e], dg[e]);
    db[e] = __fadd_rn(db[e], dv[e]);
    dxh[e] = __fmul_rn(dv[e], g[e]);
    s1 = __fadd_rn(s1, dxh[e]);
    s2 = __fmaf_rn(dxh[e], xh[e], s2);
  }
  const float c1 = __fmul_rn(eg_half_sum32(s1), 1.0f / 256.f), c2 = __fmul_rn(eg_half_sum32(s2), 1.0f / 256.f);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = __fmul_rn(rstd, __fmaf_rn(-xh[e], c2, __fsub_rn(dxh[e], c1)));
}

// XCD-aware bijective remap of a linear block id (8 XCDs, round-robin dispatch): blocks that are
// neighbours after the remap share an XCD (and its L2).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// grouped row addressing: row r of a logical [M, *] matrix lives at
//   base + (r / rows_per_group) * group_stride + (r % rows_per_group) * row_stride   (elements)
struct RowMap {
  long long row_stride;
  long long group_stride;
  int rows_per_group;  // 0 => plain (r * row_stride)
};
__device__ __forceinline__ long long row_off(const RowMap& m, int r) {
  if (m.rows_per_group <= 0) return (long long)r * m.row_stride;
  int g = r / m.rows_per_group;
  return (long long)g * m.group_stride + (long long)(r - g * m.rows_per_group) * m.row_stride;
}
static inline RowMap to_rowmap(const eg_rowmap& m) {
  RowMap r;
  r.row_stride = m.row_stride;
  r.group_stride = m.group_stride;
  r.rows_per_group = m.rows_per_group;
  return r;
}
