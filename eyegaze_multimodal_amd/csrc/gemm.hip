// MFMA GEMMs for gfx950.
//   gemm_nt_kernel : C[M,N] = epilogue(A[M,K] * W[N,K]^T)   (forward linears, strided-conv rows, backward-data)
//   gemm_tn_kernel : P[s][N,K] = sum_{m in split s} dY[m,n] * X[m,k]   (weight gradients, split over M)
// 128x128 output tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16 tiles.
//   bf16: v_mfma_f32_16x16x32_bf16, K-tile 64      f32: v_mfma_f32_16x16x4_f32 (exact fmaf chain), K-tile 32
// LDS rows are 128 B wide with a 16-B-chunk XOR swizzle so ds_read_b128 fragment reads are conflict-free.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int CT_PITCH = 132;                      // fp32 epilogue tile pitch (floats): 128 + 4
constexpr int NT_LDS_BYTES = 64 * CT_PITCH * 4;    // 33792 B: one 32 KiB staging tile / half of the fp32 epilogue image

template <typename T>
struct GemmNT {
  const T* A; const T* W; T* C; const float* bias; const T* residual; const T* gate; T* out_pre;
  const eg_step_state* st;
  RowMap a, c, r, pm;
  int M, N, K, ldw, act, tiles_n, nblocks;
  DropCfg d1, d2;
  float gate_scale;
  int seg_tiles;              // K-tiles per A-row segment (0 = one contiguous row)
  long long seg_stride_bytes; // distance between segments of an A row
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == EG_ACT_RELU) return fmaxf(v, 0.f);
  if (act == EG_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  return v;
}

template <typename T>
__device__ __forceinline__ void mma_ktile(const char* bufA, const char* bufW, int wm, int wn, int lane, f32x4 (&acc)[4][4]) {
  typedef typename H16<T>::frag frag;          // 16-bit operands (bf16 / fp16): K-tile 64, 2 MFMA k-steps
  const int l15 = lane & 15, g = lane >> 4, sw = lane & 7;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ch = ((kk * 4 + g) ^ sw) << 4;
    frag xf[4], wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xf[i] = *(const frag*)(bufA + (wm * 64 + i * 16 + l15) * 128 + ch);
      wf[i] = *(const frag*)(bufW + (wn * 64 + i * 16 + l15) * 128 + ch);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = H16<T>::mfma(wf[ni], xf[mi], acc[ni][mi]);
  }
}

template <>
__device__ __forceinline__ void mma_ktile<float>(const char* bufA, const char* bufW, int wm, int wn, int lane,
                                                 f32x4 (&acc)[4][4]) {
  const int l15 = lane & 15, g = lane >> 4, sw = lane & 7;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int ch = ((s ^ sw) << 4) + g * 4;
    float xf[4], wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xf[i] = *(const float*)(bufA + (wm * 64 + i * 16 + l15) * 128 + ch);
      wf[i] = *(const float*)(bufW + (wn * 64 + i * 16 + l15) * 128 + ch);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
  }
}

// the activation is a template parameter: with a run-time `act` every element carries the branch tree of erff() inline
template <int ACT>
__device__ __forceinline__ float apply_act_t(float v) {
  if (ACT == EG_ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == EG_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  return v;
}

// NARROW: products with N <= 64 (the spectrogram / image convolutions' 64 output channels).  On the 128 x 128 tile the two waves of
// columns 64 .. 127 multiply padding; here the tile is 256 rows x 64 columns -- wave (wm, wn) owns rows 128 wn + 64 wm .. + 63 of it --
// so every MFMA is useful (conv-2 at C = 32: 930 + 593 us per step on the square tile).  Same k-ordered chains, same epilogue.
template <typename T, int ACT, bool NARROW = false>
__global__ __launch_bounds__(256, 3) void gemm_nt_kernel(GemmNT<T> p) {
  // 3 workgroups per CU: ONE 32 KiB LDS staging tile (the next K-tile waits in registers) and an epilogue that
  // passes the two 64-row halves of the tile through a 33 KiB fp32 LDS image one after the other.  The small-K
  // products of this model are latency-bound, so resident waves (12 per CU) matter more than barrier count.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = xcd_remap(blockIdx.x, p.nblocks);
  const int m0 = NARROW ? bid * 256 : (bid / p.tiles_n) * BM, n0 = NARROW ? 0 : (bid % p.tiles_n) * BN;
  constexpr int NA = NARROW ? 8 : 4, NW = NARROW ? 2 : 4;       // 16-B pieces per thread of the A / W tile
  constexpr int WOFF = NARROW ? 32768 : 16384;                   // the W tile behind the A tile

  // staging: each thread moves 4 x 16 B of the A tile and 4 x 16 B of the W tile per K-tile
  const int crow = tid >> 3, cch = tid & 7;
  const char* ap[NA];
  const char* wp[NW];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int m = min(m0 + crow + 32 * i, p.M - 1);
    ap[i] = (const char*)(p.A + row_off(p.a, m)) + cch * 16;
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int n = min(n0 + crow + 32 * i, p.N - 1);
    wp[i] = (const char*)(p.W + (long long)n * p.ldw) + cch * 16;
  }
  const int soff = crow * 128 + ((cch ^ (crow & 7)) << 4);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / (128 / (int)sizeof(T));
  // A rows may consist of several equally long segments (2-D convolution windows): K-tile kt starts at koff(kt)
  auto koff = [&](int kt) -> size_t {
    if (p.seg_tiles <= 0) return (size_t)kt * 128;
    const int sg = kt / p.seg_tiles;
    return (size_t)sg * (size_t)p.seg_stride_bytes + (size_t)(kt - sg * p.seg_tiles) * 128;
  };
  u32x4 ra[NA], rw[NW];
#pragma unroll
  for (int i = 0; i < NA; ++i) ra[i] = *(const u32x4*)(ap[i]);
#pragma unroll
  for (int i = 0; i < NW; ++i) rw[i] = *(const u32x4*)(wp[i]);
  // the epilogue's residual (or gate) rows are requested early so their latency hides under the K loop / under the
  // other half's epilogue: 4 x 16 B per thread and half (16-bit types only)
  constexpr bool kPre = sizeof(T) == 2;
  const T* const eop = p.residual ? p.residual : p.gate;
  const RowMap& emap = p.residual ? p.r : p.c;
  const int en = n0 + (NARROW ? (tid & 7) : (tid & 15)) * 8;
  u32x4 pre[4];
  auto prefetch_epi = [&](int half) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + half * 64 + (NARROW ? (tid >> 3) + 32 * i : (tid >> 4) + 16 * i);
      pre[i] = (u32x4){0u, 0u, 0u, 0u};
      if (kPre && eop && (!NARROW || i < 2) && m < p.M && en < p.N) pre[i] = *(const u32x4*)(eop + row_off(emap, m) + en);
    }
  };
  prefetch_epi(0);
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int i = 0; i < NA; ++i) *(u32x4*)(smem + soff + i * 4096) = ra[i];
#pragma unroll
    for (int i = 0; i < NW; ++i) *(u32x4*)(smem + WOFF + soff + i * 4096) = rw[i];
    __syncthreads();
    if (kt + 1 < nk) {
      const size_t ka = koff(kt + 1);
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = *(const u32x4*)(ap[i] + ka);
#pragma unroll
      for (int i = 0; i < NW; ++i) rw[i] = *(const u32x4*)(wp[i] + (size_t)(kt + 1) * 128);
    }
    if (NARROW) mma_ktile<T>(smem, smem + WOFF, 2 * wn + wm, 0, lane, acc);      // row block 2 wn + wm, column block 0
    else mma_ktile<T>(smem, smem + WOFF, wm, wn, lane, acc);
    __syncthreads();
  }

  // epilogue: accumulators -> fp32 LDS image (64 rows at a time) -> row-wise 8-element chunks with fused
  // bias / act / gate / dropout / second output / residual
  float* ct = (float*)smem;
  uint32_t seed_lo = 0, seed_hi = 0;
  if (p.d1.thresh | p.d2.thresh) {
    seed_lo = p.st->seed_lo;
    seed_hi = p.st->seed_hi;
  }
  const int ch = NARROW ? (tid & 7) : (tid & 15);
  const int n = n0 + ch * 8;
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = 0.f;
  if (p.bias && n < p.N) load8(p.bias + n, bv);
#pragma unroll
  for (int half = 0; half < (NARROW ? 4 : 2); ++half) {            // 64 rows of the tile per pass through the fp32 image
    if ((NARROW ? 2 * wn + wm : wm) == half) {
      const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          *(f32x4*)(ct + (mi * 16 + l15) * CT_PITCH + (NARROW ? 0 : wn * 64) + ni * 16 + 4 * g) = acc[ni][mi];
    }
    __syncthreads();
    u32x4 cur[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = pre[i];
    if (half + 1 < (NARROW ? 4 : 2)) prefetch_epi(half + 1);
    if (n < p.N) {
#pragma unroll
      for (int i = 0; i < (NARROW ? 2 : 4); ++i) {
        const int row = NARROW ? (tid >> 3) + 32 * i : (tid >> 4) + 16 * i;
        const int m = m0 + half * 64 + row;
        if (m >= p.M) continue;
        float v[8];
        load8(ct + row * CT_PITCH + ch * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = apply_act_t<ACT>(v[j] + bv[j]);
        const long long coff = row_off(p.c, m) + n;
        if (p.gate) {
          float gv[8];
          if (kPre && !p.residual) load8((const T*)&cur[i], gv);
          else load8(p.gate + coff, gv);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gv[j] > 0.f ? v[j] * p.gate_scale : 0.f;
        }
        if (p.d1.thresh | p.d2.thresh) {
          const uint32_t idx = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;  // even: N % 8 == 0, n % 8 == 0
          eg_dropout_run<8>(v, p.d1, seed_lo, seed_hi, idx);
          eg_dropout_run<8>(v, p.d2, seed_lo, seed_hi, idx);
        }
        if (p.out_pre) store8(p.out_pre + row_off(p.pm, m) + n, v);
        if (p.residual) {
          float rv[8];
          if (kPre) load8((const T*)&cur[i], rv);
          else load8(p.residual + row_off(p.r, m) + n, rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += rv[j];
        }
        store8(p.C + coff, v);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// TN (weight gradient).  LDS tiles hold [32 reduction rows][128 columns]; bf16 fragments are gathered
// with ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group, delivered column-major).
// chunk swizzle f(r) = 2*((r&3) | ((r>>1)&4)): the 8 rows one half-wave touches land on 8 distinct
// 32-B slots of the 256-B bank row, so the transposed reads are conflict-free.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct GemmTN {
  const T* dY; const T* X; float* partial;
  RowMap y, x;
  int M, N, K, splits, rows_per_split, tiles_k, tiles_nk;
  long long x_tile_stride;  // elements between consecutive 128-column tiles of an X row (128 = contiguous)
  // output slab of one split: N/part_rows parts of [part_rows x K weights | part_rows bias sums (if has_bias)]
  int part_rows, has_bias;
  long long part_size, slab;
};

__device__ __forceinline__ int tn_swz(int r) { return (((r & 3) | ((r >> 1) & 4)) << 1); }

template <typename T> struct TNCfg;
template <> struct TNCfg<bf16_t> { static constexpr int ROWB = 256; static constexpr int CHUNKS = 16; static constexpr int STAGE_ROWS = 64; };
template <> struct TNCfg<f16_t> { static constexpr int ROWB = 256; static constexpr int CHUNKS = 16; static constexpr int STAGE_ROWS = 64; };
template <> struct TNCfg<float> { static constexpr int ROWB = 512; static constexpr int CHUNKS = 32; static constexpr int STAGE_ROWS = 32; };

template <typename T>
__device__ __forceinline__ void tn_mma(const char* bufY, const char* bufX, int wn, int wk, int lane,
                                               f32x4 (&acc)[4][4]) {
  // lane (g = lane>>4, q = (lane&15)>>2, p = lane&3) addresses row 8g+4h+q, columns base+4p..4p+3
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  typedef typename H16<T>::frag frag;
  frag yf[4], xf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s16x4 lo[2], hi[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = 8 * g + 4 * h + q;
      const int f = tn_swz(r);
      const int cy = (wn * 64 + i * 16 + 4 * pp);  // column (elements)
      const int cx = (wk * 64 + i * 16 + 4 * pp);
      const int ay = r * 256 + ((((cy >> 3) ^ f)) << 4) + ((cy >> 2) & 1) * 8;
      const int ax = r * 256 + ((((cx >> 3) ^ f)) << 4) + ((cx >> 2) & 1) * 8;
      lo[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bufY + ay));
      hi[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bufX + ax));
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 ty = {lo[0][0], lo[0][1], lo[0][2], lo[0][3], lo[1][0], lo[1][1], lo[1][2], lo[1][3]};
    s16x8 tx = {hi[0][0], hi[0][1], hi[0][2], hi[0][3], hi[1][0], hi[1][1], hi[1][2], hi[1][3]};
    yf[i] = __builtin_bit_cast(frag, ty);
    xf[i] = __builtin_bit_cast(frag, tx);
  }
#pragma unroll
  for (int ki = 0; ki < 4; ++ki)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      acc[ki][ni] = H16<T>::mfma(xf[ki], yf[ni], acc[ki][ni]);
}

template <>
__device__ __forceinline__ void tn_mma<float>(const char* bufY, const char* bufX, int wn, int wk, int lane,
                                              f32x4 (&acc)[4][4]) {
  const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int r = 4 * s + g;
    const int f = tn_swz(r);
    float yf[4], xf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cy = wn * 64 + i * 16 + l15, cx = wk * 64 + i * 16 + l15;
      yf[i] = *(const float*)(bufY + r * 512 + (((cy >> 2) ^ f) << 4) + (cy & 3) * 4);
      xf[i] = *(const float*)(bufX + r * 512 + (((cx >> 2) ^ f) << 4) + (cx & 3) * 4);
    }
#pragma unroll
    for (int ki = 0; ki < 4; ++ki)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[ki], yf[ni], acc[ki][ni], 0, 0, 0);
  }
}

// column sums of the dY tile (rows 16*half .. +15, column `col`) straight from its swizzled LDS image
template <typename T>
__device__ __forceinline__ float tn_tile_colsum(const char* bufY, int col, int half) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = half * 16 + i;
    s += H16<T>::ld(*(const T*)(bufY + r * 256 + (((col >> 3) ^ tn_swz(r)) << 4) + (col & 7) * 2));
  }
  return s;
}
template <>
__device__ __forceinline__ float tn_tile_colsum<float>(const char* bufY, int col, int half) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = half * 16 + i;
    s += *(const float*)(bufY + r * 512 + (((col >> 2) ^ tn_swz(r)) << 4) + (col & 3) * 4);
  }
  return s;
}

template <typename T>
__device__ __forceinline__ void tn_body(const GemmTN<T>& p, const int split, const int t, char* smem) {
  // 64 reduction rows per stage in ONE LDS buffer (the next stage waits in registers): 32 KiB (bf16) per workgroup,
  // three workgroups per CU, 32 KiB of loads in flight per workgroup.
  constexpr int ROWB = TNCfg<T>::ROWB;          // bytes per LDS tile row (128 columns)
  constexpr int CHUNKS = TNCfg<T>::CHUNKS;      // 16-B chunks per row
  constexpr int EPC = 16 / (int)sizeof(T);      // elements per chunk
  constexpr int RS = TNCfg<T>::STAGE_ROWS;      // reduction rows per stage (64 bf16 / 32 f32)
  constexpr int TILEB = RS * ROWB;              // one operand tile
  constexpr int PER_THREAD = RS * CHUNKS / 256; // chunks per thread per operand
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave >> 1, wn = wave & 1;
  const int n0 = (t / p.tiles_k) * 128, k0 = (t % p.tiles_k) * 128;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);
  const long long xk0 = (long long)(t % p.tiles_k) * p.x_tile_stride;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging map: chunk id c = tid + 256*i -> row = c / CHUNKS, chunk-in-row = c % CHUNKS
  u32x4 ry[PER_THREAD], rx[PER_THREAD];
  auto load_tile = [&](int mt) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int c = tid + 256 * i;
      const int row = c / CHUNKS, ch = c % CHUNKS;
      const int m = mt + row;
      const bool rv = m < mend;
      u32x4 z = {0u, 0u, 0u, 0u};
      ry[i] = z;
      rx[i] = z;
      if (rv && n0 + ch * EPC < p.N) ry[i] = *(const u32x4*)(p.dY + row_off(p.y, m) + n0 + ch * EPC);
      if (rv && k0 + ch * EPC < p.K) rx[i] = *(const u32x4*)(p.X + row_off(p.x, m) + xk0 + ch * EPC);
    }
  };
  auto store_tile = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int c = tid + 256 * i;
      const int row = c / CHUNKS, ch = c % CHUNKS;
      const int off = row * ROWB + ((ch ^ tn_swz(row)) << 4);
      *(u32x4*)(buf + off) = ry[i];
      *(u32x4*)(buf + TILEB + off) = rx[i];
    }
  };
  const int nt = (mend - mbeg + RS - 1) / RS;
  const bool do_bias = p.has_bias && (t % p.tiles_k) == 0;  // the k-tile-0 blocks see every dY row exactly once
  float bsum = 0.f;
  if (nt > 0) load_tile(mbeg);
  for (int it = 0; it < nt; ++it) {
    store_tile(smem);
    __syncthreads();
    if (it + 1 < nt) load_tile(mbeg + (it + 1) * RS);
#pragma unroll
    for (int sub = 0; sub < RS / 32; ++sub) {
      tn_mma<T>(smem + sub * 32 * ROWB, smem + TILEB + sub * 32 * ROWB, wn, wk, lane, acc);
      if (do_bias) bsum += tn_tile_colsum<T>(smem + sub * 32 * ROWB, tid & 127, tid >> 7);
    }
    __syncthreads();
  }
  // D[i = k][j = n]: lane holds 4 consecutive k (rows 4g+r) for column n = lane&15
  float* out = p.partial + (size_t)split * p.slab;
  const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ki = 0; ki < 4; ++ki)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + l15;
      const int k = k0 + wk * 64 + ki * 16 + 4 * g;
      if (n < p.N && k < p.K) {
        const int part = n / p.part_rows;
        *(f32x4*)(out + (size_t)part * p.part_size + (size_t)(n - part * p.part_rows) * p.K + k) = acc[ki][ni];
      }
    }
  if (do_bias) {
    float* red = (float*)smem;  // the staging buffer is free after the loop's final barrier
    red[tid] = bsum;
    __syncthreads();
    if (tid < 128) {
      const int n = n0 + tid;
      if (n < p.N) {
        const int part = n / p.part_rows;
        out[(size_t)part * p.part_size + (size_t)p.part_rows * p.K + (n - part * p.part_rows)] = red[tid] + red[tid + 128];
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256, 3) void gemm_tn_kernel(GemmTN<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  tn_body<T>(p, bid / p.tiles_nk, bid % p.tiles_nk, smem);
}

// grouped form: many weight-gradient products (all with plain row-major operands and the same reduction length M)
// in ONE launch; block -> problem by binary search over the problems' first block.
template <typename T>
__global__ __launch_bounds__(256, 3) void gemm_tn_grouped_kernel(const eg_tn_problem* __restrict__ probs, int nprob,
                                                                 int M, int splits, int rows_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int pi_s;
  // XCD-aware order: the tiles of one (problem, row range) read the same dY / X rows, so they must share an L2
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  if (threadIdx.x == 0) {
    int lo = 0, hi = nprob - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (probs[mid].blk0 <= bid) lo = mid; else hi = mid - 1;
    }
    pi_s = lo;
  }
  __syncthreads();
  const eg_tn_problem q = probs[pi_s];
  GemmTN<T> p;
  p.dY = (const T*)q.dY; p.X = (const T*)q.X; p.partial = (float*)q.partial;
  p.y.row_stride = q.ldy; p.y.group_stride = 0; p.y.rows_per_group = 0;
  p.x.row_stride = q.ldx; p.x.group_stride = 0; p.x.rows_per_group = 0;
  p.M = M; p.N = q.N; p.K = q.K; p.splits = splits; p.rows_per_split = rows_per_split;
  p.tiles_k = (q.K + 127) / 128;
  p.tiles_nk = p.tiles_k * ((q.N + 127) / 128);
  p.x_tile_stride = 128;
  p.part_rows = q.part_rows > 0 ? q.part_rows : q.N;
  p.has_bias = q.has_bias;
  p.part_size = (long long)p.part_rows * q.K + (q.has_bias ? p.part_rows : 0);
  p.slab = (long long)(q.N / p.part_rows) * p.part_size;
  const int local = bid - q.blk0;
  tn_body<T>(p, local / p.tiles_nk, local % p.tiles_nk, smem);
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 weight-gradient tile (16-bit operands, N % 256 == 0, K % 256 == 0): 512 threads = 8 waves as 4 (k groups of 64) x 2
// (n groups of 128), 4 x 8 accumulator tiles per wave.  A 128 x 128 tile streams 512 B per reduction row for 16 K outputs and
// depends on its neighbours hitting the same rows in L2 at the same time (measured: 2.77 GB fetched for 1.63 GB of operands);
// this tile streams 1024 B per row for 64 K outputs -- half the traffic per output before any sharing.  Same LDS image (rows of
// 256 columns, 16-B chunks XOR-swizzled by tn_swz), same transposed fragment reads, same k-ordered accumulation per output
// element as tn_body: the partial slabs are bit-identical to the 128 x 128 tile's for equal row splits.
// ------------------------------------------------------------------------------------------------
// 512 zero bytes: the LDS-DMA source of the rows beyond a split's end (a DMA cannot write a constant)
__device__ __attribute__((aligned(512))) char eg_zero_row[512];

__device__ __forceinline__ void tn_dma16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// 256 x 256 weight-gradient tile over one row split.  The loop is fed by LDS-DMA through a FOUR-stage ring of 32-row stages
// (2 x 16 KB each, 128 KB): three stages = 96 KB per CU are always in flight behind the one being multiplied.  (Before: 64-row
// stages staged through registers, ONE stage in flight -- requested after a barrier, needed one multiply phase later -- so every
// iteration cost one HBM latency: 5 600 cycles per 64 rows against 1 024 cycles of MFMA, 3.9 TB/s by the PMC counters.)
// One raw barrier per stage; counted s_waitcnt vmcnt(N) (each wave issues exactly four DMAs per stage).
template <typename T>
__device__ __forceinline__ void tn_body256(const GemmTN<T>& p, const int split, const int t, char* smem) {
  constexpr int ROWB = 512, RS = 32, TILEB = RS * ROWB, STAGEB = 2 * TILEB, NST = 4;    // 16 KiB per operand and stage
  typedef typename H16<T>::frag frag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 1, wn = wave & 1;
  const int n0 = (t / p.tiles_k) * 256, k0 = (t % p.tiles_k) * 256;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // DMA map: an instruction moves rows 2q, 2q+1 of a stage (lane -> row half lane / 32, LDS chunk position lane % 32 holding the
  // row's chunk pos ^ swz(row)); wave w issues q = w and w + 8 of dY and of X
  const int dhalf = lane >> 5, dpos = lane & 31;
  const char* const zsrc = eg_zero_row + dpos * 16;
  auto issue = [&](int it) {
    char* st = smem + (it & (NST - 1)) * STAGEB;
    const int mt = mbeg + it * RS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = wave + 8 * i;
      const int row = 2 * q + dhalf;
      const int m = mt + row;
      const int ch = (dpos ^ tn_swz(row)) << 4;
      const bool ok = m < mend;
      const int mc = ok ? m : mend - 1;
      const char* sy = (const char*)(p.dY + row_off(p.y, mc) + n0) + ch;
      const char* sx = (const char*)(p.X + row_off(p.x, mc) + k0) + ch;
      tn_dma16(ok ? sy : zsrc, st + q * 1024);
      tn_dma16(ok ? sx : zsrc, st + TILEB + q * 1024);
    }
  };
  const int nt = (mend - mbeg + RS - 1) / RS;
  const bool do_bias = p.has_bias && (t % p.tiles_k) == 0;
  float bsum = 0.f;
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < nt) issue(i);
  for (int it = 0; it < nt; ++it) {
    // this wave's part of stage `it` has landed once only the later stages' DMAs (four each) are outstanding
    const int later = min(NST - 2, nt - 1 - it);
    if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage it visible to all; nobody reads stage it - 1 any more
    if (it + NST - 1 < nt) issue(it + NST - 1);                        // ... so its buffer takes stage it + 3
    asm volatile("" ::: "memory");
    {
      const char* bufY = smem + (it & (NST - 1)) * STAGEB;
      const char* bufX = bufY + TILEB;
      // The transposed fragment reads are inline asm: as builtins (or plain loads) they carry LDS memory operands, and the compiler
      // then puts an s_waitcnt vmcnt(0) in front of them because an LDS-DMA "may alias" -- which drains the very ring this loop
      // keeps in flight.  The counted wait above is the real dependency; the lgkmcnt wait below is tied to the fragments.
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      const int r = 8 * g + q;                                 // row of half h: r + 4 h (same swizzle: tn_swz ignores bit 2)
      const uint32_t ly = (uint32_t)(size_t)(__attribute__((address_space(3))) const char*)bufY + r * ROWB;
      const uint32_t lx = ly + TILEB;
      const int fsw = tn_swz(r);
      s16x4 yp[8][2], xp[4][2];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int cy = wn * 128 + i * 16 + 4 * pp;
        const uint32_t a = ly + (((cy >> 3) ^ fsw) << 4) + ((cy >> 2) & 1) * 8;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(yp[i][0]) : "v"(a));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(yp[i][1]) : "v"(a));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cx = wk * 64 + i * 16 + 4 * pp;
        const uint32_t a = lx + (((cx >> 3) ^ fsw) << 4) + ((cx >> 2) & 1) * 8;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xp[i][0]) : "v"(a));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(xp[i][1]) : "v"(a));
      }
      frag yf[8], xf[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s16x8 ty = {yp[i][0][0], yp[i][0][1], yp[i][0][2], yp[i][0][3], yp[i][1][0], yp[i][1][1], yp[i][1][2], yp[i][1][3]};
        yf[i] = __builtin_bit_cast(frag, ty);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s16x8 tx = {xp[i][0][0], xp[i][0][1], xp[i][0][2], xp[i][0][3], xp[i][1][0], xp[i][1][1], xp[i][1][2], xp[i][1][3]};
        xf[i] = __builtin_bit_cast(frag, tx);
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(yf[0]), "+v"(yf[1]), "+v"(yf[2]), "+v"(yf[3]), "+v"(yf[4]), "+v"(yf[5]), "+v"(yf[6]), "+v"(yf[7]),
                     "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]));
#pragma unroll
      for (int ki = 0; ki < 4; ++ki)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) acc[ki][ni] = H16<T>::mfma(xf[ki], yf[ni], acc[ki][ni]);
      if (do_bias) {       // column sums of the dY stage: thread -> column tid & 255, rows 16 * (tid >> 8) .. +15
        const int col = tid & 255, half = tid >> 8;
        float s16 = 0.f;                                  // (summed per 16 rows first, as tn_tile_colsum does: same rounding)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = half * 16 + i;
          s16 += H16<T>::ld(*(const T*)(bufY + r * ROWB + (((col >> 3) ^ tn_swz(r)) << 4) + (col & 7) * 2));
        }
        bsum += s16;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every wave has left the ring (the bias sums reuse it)
  float* out = p.partial + (size_t)split * p.slab;
  const int l15 = lane & 15;
#pragma unroll
  for (int ki = 0; ki < 4; ++ki)
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
      const int n = n0 + wn * 128 + ni * 16 + l15;
      const int k = k0 + wk * 64 + ki * 16 + 4 * g;
      const int part = n / p.part_rows;
      *(f32x4*)(out + (size_t)part * p.part_size + (size_t)(n - part * p.part_rows) * p.K + k) = acc[ki][ni];
    }
  if (do_bias) {
    float* red = (float*)smem;
    red[tid] = bsum;
    __syncthreads();
    if (tid < 256) {
      const int n = n0 + tid;
      const int part = n / p.part_rows;
      out[(size_t)part * p.part_size + (size_t)p.part_rows * p.K + (n - part * p.part_rows)] = red[tid] + red[tid + 256];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(512, 2) void gemm_tn256_kernel(GemmTN<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  tn_body256<T>(p, bid / p.tiles_nk, bid % p.tiles_nk, smem);
}

template <typename T>
__global__ __launch_bounds__(512, 2) void gemm_tn_grouped256_kernel(const eg_tn_problem* __restrict__ probs, int nprob,
                                                                    int M, int splits, int rows_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int pi_s;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  if (threadIdx.x == 0) {
    int lo = 0, hi = nprob - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (probs[mid].blk0 <= bid) lo = mid; else hi = mid - 1;
    }
    pi_s = lo;
  }
  __syncthreads();
  const eg_tn_problem q = probs[pi_s];
  GemmTN<T> p;
  p.dY = (const T*)q.dY; p.X = (const T*)q.X; p.partial = (float*)q.partial;
  p.y.row_stride = q.ldy; p.y.group_stride = 0; p.y.rows_per_group = 0;
  p.x.row_stride = q.ldx; p.x.group_stride = 0; p.x.rows_per_group = 0;
  p.M = M; p.N = q.N; p.K = q.K; p.splits = splits; p.rows_per_split = rows_per_split;
  p.tiles_k = q.K / 256;
  p.tiles_nk = p.tiles_k * (q.N / 256);
  p.x_tile_stride = 256;
  p.part_rows = q.part_rows > 0 ? q.part_rows : q.N;
  p.has_bias = q.has_bias;
  p.part_size = (long long)p.part_rows * q.K + (q.has_bias ? p.part_rows : 0);
  p.slab = (long long)(q.N / p.part_rows) * p.part_size;
  const int local = bid - q.blk0;
  tn_body256<T>(p, local / p.tiles_nk, local % p.tiles_nk, smem);
}

// table-driven form of the reduce below: entry e sums `splits` slabs of n floats into out
__global__ __launch_bounds__(256) void reduce_table_kernel(const eg_reduce_entry* __restrict__ tab, int nent) {
  __shared__ f32x4 red[32][8];
  __shared__ int ei_s;
  if (threadIdx.x == 0) {
    int lo = 0, hi = nent - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tab[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    ei_s = lo;
  }
  __syncthreads();
  const eg_reduce_entry e = tab[ei_s];
  const float* partial = (const float*)e.partial;
  float* out = (float*)e.out;
  if (e.splits <= EG_REDUCE_WIDE_SPLITS) {
    // few splits (weight-gradient slabs): 256 float4 columns per block, each thread walks the splits in order --
    // every load is a full-wave 1 KiB row and nothing goes through LDS
    const long long i4 = ((long long)(blockIdx.x - e.blk0) * 256 + threadIdx.x) * 4;
    if (i4 + 4 <= e.n) {
      f32x4 t = *(const f32x4*)(partial + i4);
      for (int k = 1; k < e.splits; ++k) t += *(const f32x4*)(partial + (size_t)k * e.stride + i4);
      *(f32x4*)(out + i4) = t;
    }
    return;
  }
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const long long i4 = ((long long)(blockIdx.x - e.blk0) * 8 + tx) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 + 4 <= e.n)
    for (int k = ty; k < e.splits; k += 32) s += *(const f32x4*)(partial + (size_t)k * e.stride + i4);
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i4 + 4 <= e.n) {
    f32x4 t = red[0][tx];
#pragma unroll
    for (int r = 1; r < 32; ++r) t += red[r][tx];
    *(f32x4*)(out + i4) = t;
  }
}

// out[i] = sum_s partial[s*sstride + i].  256 threads = 8 float4 columns x 32 split lanes, so short outputs
// (bias / LayerNorm-gain gradients, n ~ 256) still spread their `splits` loads over many lanes.
// group > 0 (first stage of a long reduction, grid.y groups): block (x, g) sums splits [g * group, (g + 1) * group) and leaves the
// sum IN PLACE in the group's first row; a second launch then sums those rows.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* partial, float* out,
                                                              long long n, int splits, long long sstride, int accumulate, int group) {
  __shared__ f32x4 red[32][8];
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const long long i4 = ((long long)blockIdx.x * 8 + tx) * 4;
  if (group > 0) {
    const int k0 = blockIdx.y * group;
    partial += (size_t)k0 * sstride;
    out = const_cast<float*>(partial);
    splits = min(group, splits - k0);
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 + 4 <= n) {
    for (int k = ty; k < splits; k += 32) s += *(const f32x4*)(partial + (size_t)k * sstride + i4);
  } else if (i4 < n) {
    for (int k = ty; k < splits; k += 32)
      for (int e = 0; e < (int)(n - i4); ++e) s[e] += partial[(size_t)k * sstride + i4 + e];
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i4 < n) {
    f32x4 t = red[0][tx];
#pragma unroll
    for (int r = 1; r < 32; ++r) t += red[r][tx];
    if (i4 + 4 <= n) {
      if (accumulate) t += *(const f32x4*)(out + i4);
      *(f32x4*)(out + i4) = t;
    } else {
      for (int e = 0; e < (int)(n - i4); ++e) out[i4 + e] = (accumulate ? out[i4 + e] : 0.f) + t[e];
    }
  }
}

// conv weight gradient: partial [splits][N][Kp] in tap-major order (k = tap*Cp + c) -> dW [N][Cin][k] (parameter layout).
// One block per output channel n: the split sums are read as coalesced 16-B columns, the (tap, c) -> (c, tap) transposition
// happens in LDS, and the [Cin*k] parameter row leaves as one contiguous stream.
// 1024 threads and four splits requested per trip: the launch streams splits x N x Kp floats (65 MB for conv-0) and used to run at
// 1.9 TB/s with 256 threads issuing one dependent 16-B load at a time (35 us per launch); the split sums keep their order.
__global__ __launch_bounds__(1024) void unpack_conv_wgrad_kernel(const float* __restrict__ partial, float* __restrict__ dW,
                                                                 int splits, int N, int Cin, int k, int Cp, int Kp) {
  extern __shared__ __attribute__((aligned(16))) float row[];   // [Cin * k]
  const int n = blockIdx.x;
  const size_t slab = (size_t)N * Kp;
  const float* src = partial + (size_t)n * Kp;
  for (int i4 = threadIdx.x * 4; i4 < Kp; i4 += 4 * blockDim.x) {
    f32x4 s = *(const f32x4*)(src + i4);
    int sp = 1;
    for (; sp + 4 <= splits; sp += 4) {
      const f32x4 a0 = *(const f32x4*)(src + (size_t)sp * slab + i4), a1 = *(const f32x4*)(src + (size_t)(sp + 1) * slab + i4);
      const f32x4 a2 = *(const f32x4*)(src + (size_t)(sp + 2) * slab + i4), a3 = *(const f32x4*)(src + (size_t)(sp + 3) * slab + i4);
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; sp < splits; ++sp) s += *(const f32x4*)(src + (size_t)sp * slab + i4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kk = i4 + e, tap = kk / Cp, c = kk - tap * Cp;
      if (tap < k && c < Cin) row[c * k + tap] = s[e];
    }
  }
  __syncthreads();
  float* dst = dW + (size_t)n * Cin * k;
  for (int j = threadIdx.x; j < Cin * k; j += blockDim.x) dst[j] = row[j];
}

// column sums of a [M, N] matrix: block b sums rows [b*rpb, (b+1)*rpb) -> partial[b, N]
// A row is N / 8 16-B chunks; the 256 threads are 256 / (N / 8) row lanes x N / 8 chunk lanes, every row lane walks its rows two at a
// time.  (Until round 3 the map was 8 row lanes x 32 chunk lanes whatever N: at N = 64 -- the spectrogram convolutions -- 24 of 32
// lanes idled and a block had 8 rows in flight: 235 us per launch at the reference's default C = 32.)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ Y, RowMap y, int M, int N, int rpb,
                                                     float* __restrict__ partial) {
  __shared__ float red[256][9];
  const int tid = threadIdx.x;
  const int cpr = N >> 3;                                    // chunks per row (<= 128)
  const int nrl = 256 / cpr;                                 // row lanes (>= 2)
  const int c = tid % cpr, rl = tid / cpr;
  const int mbeg = blockIdx.x * rpb, mend = min(M, mbeg + rpb);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (rl < nrl) {
    int m = mbeg + rl;
    for (; m + nrl < mend; m += 2 * nrl) {
      float v0[8], v1[8];
      load8(Y + row_off(y, m) + c * 8, v0);
      load8(Y + row_off(y, m + nrl) + c * 8, v1);
#pragma unroll
      for (int e = 0; e < 8; ++e) { acc[e] += v0[e]; acc[e] += v1[e]; }
    }
    if (m < mend) {
      float v0[8];
      load8(Y + row_off(y, m) + c * 8, v0);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v0[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tid][e] = acc[e];
  __syncthreads();
  for (int n = tid; n < N; n += 256) {
    const int cc = n >> 3, e = n & 7;
    float s = 0.f;
    for (int r = 0; r < nrl; ++r) s += red[r * cpr + cc][e];
    partial[(size_t)blockIdx.x * N + n] = s;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
template <typename T>
static int launch_gemm_nt(const eg_gemm_desc* d, hipStream_t s) {
  GemmNT<T> p;
  p.A = (const T*)d->A; p.W = (const T*)d->W; p.C = (T*)d->C; p.bias = d->bias;
  p.residual = (const T*)d->residual; p.gate = (const T*)d->gate; p.out_pre = (T*)d->out_pre; p.st = d->state;
  p.a = to_rowmap(d->a); p.c = to_rowmap(d->c); p.r = to_rowmap(d->r); p.pm = to_rowmap(d->p);
  p.M = d->M; p.N = d->N; p.K = d->K; p.ldw = d->ldw; p.act = d->act;
  p.tiles_n = (d->N + BN - 1) / BN;
  p.nblocks = p.tiles_n * ((d->M + BM - 1) / BM);
  p.d1 = make_drop(d->drop1_p, d->drop1_site);
  p.d2 = make_drop(d->drop2_p, d->drop2_site);
  p.gate_scale = d->gate_scale == 0.f ? 1.0f : d->gate_scale;
  p.seg_tiles = d->a_seg_len > 0 ? d->a_seg_len / (128 / (int)sizeof(T)) : 0;
  p.seg_stride_bytes = (long long)d->a_seg_stride * (long long)sizeof(T);
  static const bool use_narrow = [] { const char* e = getenv("EYEGAZE_NT_NARROW"); return !e || atoi(e) != 0; }();
  if (use_narrow && d->N <= 64 && d->M >= 1024) {                     // one column tile of <= 64: 256 x 64 tiles (gemm_nt_kernel, NARROW)
    p.tiles_n = 1;
    p.nblocks = (d->M + 255) / 256;
    constexpr int lds = 32768 + 8192;
    if (d->act == EG_ACT_RELU) hipLaunchKernelGGL((gemm_nt_kernel<T, EG_ACT_RELU, true>), dim3(p.nblocks), dim3(256), lds, s, p);
    else if (d->act == EG_ACT_GELU) hipLaunchKernelGGL((gemm_nt_kernel<T, EG_ACT_GELU, true>), dim3(p.nblocks), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<T, EG_ACT_NONE, true>), dim3(p.nblocks), dim3(256), lds, s, p);
    EG_LAUNCH_CHECK("gemm_nt (narrow)");
    return 0;
  }
  if (d->act == EG_ACT_RELU) hipLaunchKernelGGL((gemm_nt_kernel<T, EG_ACT_RELU>), dim3(p.nblocks), dim3(256), NT_LDS_BYTES, s, p);
  else if (d->act == EG_ACT_GELU) hipLaunchKernelGGL((gemm_nt_kernel<T, EG_ACT_GELU>), dim3(p.nblocks), dim3(256), NT_LDS_BYTES, s, p);
  else hipLaunchKernelGGL((gemm_nt_kernel<T, EG_ACT_NONE>), dim3(p.nblocks), dim3(256), NT_LDS_BYTES, s, p);
  EG_LAUNCH_CHECK("gemm_nt");
  return 0;
}

int eg_rs_gemm_try(const eg_gemm_desc* d, hipStream_t s);   // rsgemm.hip: register-stationary row-stream kernel (K == 256)
int eg_wide_gemm_try(const eg_gemm_desc* d, hipStream_t s); // widegemm.hip: 160x256 tile, LDS-DMA ring (N == 256)

bool eg_rs_gemm_ok(const eg_gemm_desc* d);
bool eg_wide_gemm_ok(const eg_gemm_desc* d);
static int gemm_knob(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 1; }

// which kernel eg_gemm_nt launches for this descriptor (measurement aid: bench.py attributes its per-launch timings with it)
extern "C" int eg_gemm_nt_route(const eg_gemm_desc* d) {
  if (!d) return -1;
  static const int use_rs = gemm_knob("EYEGAZE_RS"), use_wide = gemm_knob("EYEGAZE_WIDE");
  if (use_wide && eg_wide_gemm_ok(d)) return EG_ROUTE_WIDE;
  if (use_rs && eg_rs_gemm_ok(d)) return EG_ROUTE_ROWSTREAM;
  return EG_ROUTE_TILED;
}

extern "C" int eg_gemm_nt(const eg_gemm_desc* d, void* stream) {
  EG_CHECK(d && d->A && d->W && d->C, "eg_gemm_nt: null operand");
  EG_CHECK(d->M > 0 && d->N > 0 && d->K > 0, "eg_gemm_nt: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  EG_CHECK(d->dtype == EG_F32 || d->dtype == EG_BF16 || d->dtype == EG_F16, "eg_gemm_nt: bad dtype %d", d->dtype);
  const int bk = d->dtype == EG_F32 ? 32 : 64;
  const int al = d->dtype == EG_F32 ? 4 : 8;  // elements per 16 B
  EG_CHECK(d->K % bk == 0, "eg_gemm_nt: K=%d must be a multiple of %d", d->K, bk);
  EG_CHECK(d->N % 8 == 0, "eg_gemm_nt: N=%d must be a multiple of 8", d->N);
  EG_CHECK(d->ldw >= d->K && d->ldw % al == 0, "eg_gemm_nt: ldw=%d", d->ldw);
  EG_CHECK(d->a.row_stride % al == 0 && d->a.group_stride % al == 0, "eg_gemm_nt: A rows must be 16-B aligned");
  EG_CHECK(d->c.row_stride % 8 == 0 && d->c.group_stride % 8 == 0, "eg_gemm_nt: C rows must be 8-element aligned");
  EG_CHECK(!d->residual || (d->r.row_stride % 8 == 0 && d->r.group_stride % 8 == 0), "eg_gemm_nt: residual rows");
  EG_CHECK(!d->out_pre || (d->p.row_stride % 8 == 0 && d->p.group_stride % 8 == 0), "eg_gemm_nt: out_pre rows");
  EG_CHECK(d->a_seg_len == 0 || (d->a_seg_len % bk == 0 && d->K % d->a_seg_len == 0 && d->a_seg_stride % al == 0),
           "eg_gemm_nt: segmented A rows need a_seg_len (=%d) to be a multiple of %d dividing K", d->a_seg_len, bk);
  EG_CHECK((d->drop1_p == 0.f && d->drop2_p == 0.f) || d->state, "eg_gemm_nt: dropout needs a step state");
  EG_CHECK(d->drop1_p >= 0.f && d->drop1_p < 1.f && d->drop2_p >= 0.f && d->drop2_p < 1.f, "eg_gemm_nt: dropout p");
  EG_CHECK((long long)d->M * d->N < (1ll << 32), "eg_gemm_nt: M*N exceeds the 32-bit dropout index");
  EG_CHECK(((uintptr_t)d->A | (uintptr_t)d->W | (uintptr_t)d->C) % 16 == 0, "eg_gemm_nt: operands must be 16-B aligned");
  hipStream_t s = (hipStream_t)stream;
  static const int use_rs = gemm_knob("EYEGAZE_RS"), use_wide = gemm_knob("EYEGAZE_WIDE");
  if (use_wide) {                                  // N == 256 (any K): one workgroup per 160 whole rows
    const int rc = eg_wide_gemm_try(d, s);
    if (rc == 0) return 0;
    if (rc != -1) return eg_fail("wide gemm launch failed");
  }
  if (use_rs) {
    const int rc = eg_rs_gemm_try(d, s);
    if (rc == 0) return 0;
    if (rc != -1) return eg_fail("rs_gemm launch failed");
  }
  return d->dtype == EG_BF16 ? launch_gemm_nt<bf16_t>(d, s) : d->dtype == EG_F16 ? launch_gemm_nt<f16_t>(d, s) : launch_gemm_nt<float>(d, s);
}

template <typename T>
static int launch_gemm_tn(const eg_gemm_tn_desc* d, hipStream_t s) {
  GemmTN<T> p;
  p.dY = (const T*)d->dY; p.X = (const T*)d->X; p.partial = d->partial;
  p.y = to_rowmap(d->y); p.x = to_rowmap(d->x);
  p.M = d->M; p.N = d->N; p.K = d->K; p.splits = d->splits;
  int rps = (d->M + d->splits - 1) / d->splits;
  p.rows_per_split = (rps + 63) / 64 * 64;
  p.tiles_k = (d->K + 127) / 128;
  p.tiles_nk = p.tiles_k * ((d->N + 127) / 128);
  p.x_tile_stride = d->x_tile_stride > 0 ? d->x_tile_stride : 128;
  p.part_rows = d->part_rows > 0 ? d->part_rows : d->N;
  p.has_bias = d->has_bias ? 1 : 0;
  p.part_size = (long long)p.part_rows * d->K + (p.has_bias ? p.part_rows : 0);
  p.slab = (long long)(d->N / p.part_rows) * p.part_size;
  if constexpr (sizeof(T) == 2) {
    if (d->tile == 256) {
      p.tiles_k = d->K / 256;
      p.tiles_nk = p.tiles_k * (d->N / 256);
      constexpr int lds256 = 4 * 2 * 32 * 512;          // tn_body256's four-stage ring
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_tn256_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
        attr = true;
      }
      hipLaunchKernelGGL(gemm_tn256_kernel<T>, dim3(p.tiles_nk * d->splits), dim3(512), lds256, s, p);
      EG_LAUNCH_CHECK("gemm_tn256");
      return 0;
    }
  }
  const int lds = 2 * TNCfg<T>::STAGE_ROWS * TNCfg<T>::ROWB;
  hipLaunchKernelGGL(gemm_tn_kernel<T>, dim3(p.tiles_nk * d->splits), dim3(256), lds, s, p);
  EG_LAUNCH_CHECK("gemm_tn");
  return 0;
}

extern "C" int eg_gemm_tn(const eg_gemm_tn_desc* d, void* stream) {
  EG_CHECK(d && d->dY && d->X && d->partial, "eg_gemm_tn: null operand");
  EG_CHECK(d->M > 0 && d->N > 0 && d->K > 0 && d->splits > 0, "eg_gemm_tn: bad shape");
  EG_CHECK(d->dtype == EG_F32 || d->dtype == EG_BF16 || d->dtype == EG_F16, "eg_gemm_tn: bad dtype %d", d->dtype);
  EG_CHECK(d->N % 8 == 0 && d->K % 8 == 0, "eg_gemm_tn: N=%d, K=%d must be multiples of 8", d->N, d->K);
  const int al = d->dtype == EG_F32 ? 4 : 8;
  EG_CHECK(d->y.row_stride % al == 0 && d->y.group_stride % al == 0 && d->x.row_stride % al == 0 &&
               d->x.group_stride % al == 0, "eg_gemm_tn: rows must be 16-B aligned");
  EG_CHECK(((uintptr_t)d->dY | (uintptr_t)d->X | (uintptr_t)d->partial) % 16 == 0, "eg_gemm_tn: alignment");
  EG_CHECK(d->part_rows == 0 || (d->N % d->part_rows == 0 && d->part_rows % 4 == 0), "eg_gemm_tn: part_rows=%d must divide N", d->part_rows);
  EG_CHECK(!d->has_bias || d->K % 4 == 0, "eg_gemm_tn: fused bias sums need K %% 4 == 0");
  EG_CHECK(d->x_tile_stride == 0 || (d->x_tile_stride % al == 0 && d->K % 128 == 0), "eg_gemm_tn: x_tile_stride needs K %% 128 == 0");
  EG_CHECK(d->tile == 0 || d->tile == 128 || (d->tile == 256 && d->dtype != EG_F32 && d->N % 256 == 0 && d->K % 256 == 0 &&
                                              (d->x_tile_stride == 0 || d->x_tile_stride == 128)),
           "eg_gemm_tn: tile=%d (256 needs a 16-bit dtype, N and K multiples of 256, contiguous X rows)", d->tile);
  hipStream_t s = (hipStream_t)stream;
  return d->dtype == EG_BF16 ? launch_gemm_tn<bf16_t>(d, s) : d->dtype == EG_F16 ? launch_gemm_tn<f16_t>(d, s) : launch_gemm_tn<float>(d, s);
}

extern "C" int eg_reduce_partials(float* partial, float* out, int64_t n, int splits, int64_t split_stride,
                                  int accumulate, void* stream) {
  EG_CHECK(partial && out && n > 0 && splits > 0 && split_stride >= n, "eg_reduce_partials: bad arguments");
  EG_CHECK(((uintptr_t)partial | (uintptr_t)out) % 16 == 0 && split_stride % 4 == 0,
           "eg_reduce_partials: 16-B alignment (split_stride %% 4 == 0)");
  const long long ncol = (n + 3) / 4;
  const unsigned gx = (unsigned)((ncol + 7) / 8);
  // A long reduction over a short vector (the spectrogram conv-1 gradient: 16 384 per-image partials of 320 floats at C = 32) left
  // one to nine workgroups walking thousands of rows each: 175 us of exposed load latency per call.  Two stages instead: 64 groups
  // of rows are summed in parallel, IN PLACE into each group's first row (the partial buffer is scratch), then those 64 rows.
  if (splits >= 2048 && gx <= 64) {
    const int groups = 64, group = (splits + groups - 1) / groups;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, (unsigned)((splits + group - 1) / group)), dim3(256), 0, (hipStream_t)stream,
                       partial, out, (long long)n, splits, (long long)split_stride, 0, group);
    EG_LAUNCH_CHECK("reduce_partials (stage 1)");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, partial, out, (long long)n,
                       (splits + group - 1) / group, (long long)split_stride * group, accumulate, 0);
    EG_LAUNCH_CHECK("reduce_partials");
    return 0;
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx), dim3(256), 0,
                     (hipStream_t)stream, partial, out, (long long)n, splits, (long long)split_stride, accumulate, 0);
  EG_LAUNCH_CHECK("reduce_partials");
  return 0;
}

// First stage of a long split reduction for other files (spec.hip): groups of `group` consecutive splits are summed IN PLACE into each
// group's first slab (the partial buffer is scratch); the caller then reads ceil(splits / group) slabs at stride `stride * group`.
int eg_reduce_groups_inplace(float* partial, long long n, int splits, long long stride, int groups, int* group_out, hipStream_t s) {
  const int group = (splits + groups - 1) / groups;
  const unsigned gx = (unsigned)((n + 31) / 32);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, (unsigned)((splits + group - 1) / group)), dim3(256), 0, s, partial, partial, n,
                     splits, stride, 0, group);
  *group_out = group;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int eg_unpack_conv_wgrad(const float* partial, float* dW, int splits, int N, int Cin, int k, int Cp, int Kp,
                                    void* stream) {
  EG_CHECK(partial && dW && splits > 0 && N > 0 && Cin > 0 && k > 0 && Cp >= Cin && Kp >= k * Cp,
           "eg_unpack_conv_wgrad: bad arguments");
  EG_CHECK(Kp % 4 == 0 && (size_t)Cin * k * sizeof(float) <= 64 * 1024, "eg_unpack_conv_wgrad: Kp=%d must be a multiple of 4 and Cin*k*4 <= 64 KiB", Kp);
  hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3(N), dim3(1024), (size_t)Cin * k * sizeof(float), (hipStream_t)stream,
                     partial, dW, splits, N, Cin, k, Cp, Kp);
  EG_LAUNCH_CHECK("unpack_conv_wgrad");
  return 0;
}

extern "C" int eg_colsum(const void* Y, eg_rowmap y, int M, int N, float* partial, int nblk, int dtype,
                         void* stream) {
  EG_CHECK(Y && partial && M > 0 && N > 0 && nblk > 0, "eg_colsum: bad arguments");
  EG_CHECK(N % 8 == 0 && N <= 1024, "eg_colsum: N=%d must be a multiple of 8 and <= 1024", N);
  const int rpb = (M + nblk - 1) / nblk;
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Y,
                       to_rowmap(y), M, N, rpb, partial);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(colsum_kernel<f16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const f16_t*)Y,
                       to_rowmap(y), M, N, rpb, partial);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float*)Y,
                       to_rowmap(y), M, N, rpb, partial);
  else
    return eg_fail("eg_colsum: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("colsum");
  return 0;
}

extern "C" int eg_gemm_tn_grouped(const eg_tn_problem* probs, int nprob, int total_blocks, int M, int splits, int dtype,
                                  void* stream) {
  EG_CHECK(probs && nprob > 0 && total_blocks > 0 && M > 0 && splits > 0, "eg_gemm_tn_grouped: bad arguments");
  EG_CHECK(dtype == EG_F32 || dtype == EG_BF16 || dtype == EG_F16, "eg_gemm_tn_grouped: bad dtype %d", dtype);
  int rps = (M + splits - 1) / splits;
  rps = (rps + 63) / 64 * 64;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(gemm_tn_grouped_kernel<bf16_t>, dim3(total_blocks), dim3(256), 2 * TNCfg<bf16_t>::STAGE_ROWS * TNCfg<bf16_t>::ROWB,
                       s, probs, nprob, M, splits, rps);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(gemm_tn_grouped_kernel<f16_t>, dim3(total_blocks), dim3(256), 2 * TNCfg<f16_t>::STAGE_ROWS * TNCfg<f16_t>::ROWB,
                       s, probs, nprob, M, splits, rps);
  else
    hipLaunchKernelGGL(gemm_tn_grouped_kernel<float>, dim3(total_blocks), dim3(256), 2 * TNCfg<float>::STAGE_ROWS * TNCfg<float>::ROWB,
                       s, probs, nprob, M, splits, rps);
  EG_LAUNCH_CHECK("gemm_tn_grouped");
  return 0;
}

// the same problem table served by 256 x 256 tiles: blk0 counts (N/256) * (K/256) * splits blocks per problem
extern "C" int eg_gemm_tn_grouped256(const eg_tn_problem* probs, int nprob, int total_blocks, int M, int splits, int dtype,
                                     void* stream) {
  EG_CHECK(probs && nprob > 0 && total_blocks > 0 && M > 0 && splits > 0, "eg_gemm_tn_grouped256: bad arguments");
  EG_CHECK(dtype == EG_BF16 || dtype == EG_F16, "eg_gemm_tn_grouped256: 16-bit dtypes only (got %d)", dtype);
  int rps = (M + splits - 1) / splits;
  rps = (rps + 63) / 64 * 64;
  hipStream_t s = (hipStream_t)stream;
  constexpr int lds = 4 * 2 * 32 * 512;                  // tn_body256's four-stage ring
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_grouped256_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)gemm_tn_grouped256_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(gemm_tn_grouped256_kernel<bf16_t>, dim3(total_blocks), dim3(512), lds, s, probs, nprob, M, splits, rps);
  else
    hipLaunchKernelGGL(gemm_tn_grouped256_kernel<f16_t>, dim3(total_blocks), dim3(512), lds, s, probs, nprob, M, splits, rps);
  EG_LAUNCH_CHECK("gemm_tn_grouped256");
  return 0;
}

extern "C" int eg_reduce_table(const eg_reduce_entry* table, int nentries, int total_blocks, void* stream) {
  EG_CHECK(table && nentries > 0 && total_blocks > 0, "eg_reduce_table: bad arguments");
  hipLaunchKernelGGL(reduce_table_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table, nentries);
  EG_LAUNCH_CHECK("reduce_table");
  return 0;
}
