// Multi-head attention core for short sequences (S <= 160, d_k = 32) on gfx950.
// One wave owns one (sample, head): the whole S x S score matrix of a head lives in that wave's registers.
//   scores^T tile = K_tile(16 keys x 32) * Q_tile^T            -> v_mfma_f32_16x16x32_bf16, key on rows, query on lanes
//   soft-max over keys = in-register over tiles + 2 cross-lane steps (lanes l, l^16, l^32, l^48 share a query)
//   O^T = V^T * P^T: the score accumulators ARE the P^T operand (no LDS round trip); V^T fragments come from a
//   row-major LDS image through ds_read_b64_tr_b16 with the key permutation the accumulator layout implies:
//   k-slot 8g+j  <->  key 32*kp + 16*(j>>2) + 4g + (j&3).
// Backward recomputes P from the saved log-sum-exp (flash style): pass A (key rows / query lanes) gives dQ,
// pass B (query rows / key lanes) gives dK and dV; delta = rowsum(dO * O).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) short s16x8;

// T = bf16_t or f16_t: the 16-bit storage type; FR<T> = its MFMA operand fragment (8 elements per lane)
template <typename T> using FR = typename H16<T>::frag;

template <typename T>
__device__ __forceinline__ FR<T> ld_frag_global(const T* p, bool valid) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if (valid) v = *(const u32x4*)p;
  return __builtin_bit_cast(FR<T>, v);
}
// LDS image: [rows][32 bf16] = 64-B rows; the two 32-B halves of a row are swapped when (row>>2)&1 so that the
// transposed reads of 8 consecutive rows hit 8 distinct 32-B slots of the 256-B bank row.
__device__ __forceinline__ int img_chunk_off(int row, int c4) {
  return row * 64 + ((((c4 >> 1) ^ ((row >> 2) & 1))) << 5) + ((c4 & 1) << 4);
}
template <typename T>
__device__ __forceinline__ FR<T> ld_frag_lds_row(const char* img, int row, int g) {
  return *(const FR<T>*)(img + img_chunk_off(row, g));
}
// transposed fragment: slot 8g+j <-> row rbase + 16*(j>>2) + 4g + (j&3), column d0 + (lane&15)
template <typename T>
__device__ __forceinline__ FR<T> ld_frag_lds_tr(const char* img, int rbase, int dt, int lane) {
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  s16x4 part[2];
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int row = rbase + 16 * h2 + 4 * g + qq;
    const int off = row * 64 + ((dt ^ (g & 1)) << 5) + pp * 8;
    part[h2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + off));
  }
  s16x8 t = {part[0][0], part[0][1], part[0][2], part[0][3], part[1][0], part[1][1], part[1][2], part[1][3]};
  return __builtin_bit_cast(FR<T>, t);
}
template <typename T>
__device__ __forceinline__ FR<T> pack_frag(const f32x4& a, const f32x4& b) {
  u32x4 v;
  v[0] = H16<T>::pack2(a[0], a[1]);
  v[1] = H16<T>::pack2(a[2], a[3]);
  v[2] = H16<T>::pack2(b[0], b[1]);
  v[3] = H16<T>::pack2(b[2], b[3]);
  return __builtin_bit_cast(FR<T>, v);
}
template <typename T>
__device__ __forceinline__ void stage_rows(char* img, const T* src, long long ld, int S, int SP, int first, int step) {
  for (int id = first; id < SP * 4; id += step) {
    const int row = id >> 2, c4 = id & 3;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < S) v = *(const u32x4*)(src + (long long)row * ld + c4 * 8);
    *(u32x4*)(img + img_chunk_off(row, c4)) = v;
  }
}

// stage_rows in two halves, so that a kernel can request SEVERAL images (and whatever else its prologue reads) before the first wait:
// every load -> wait -> LDS store pair left to itself cost one HBM latency (stamped: 9 200 cycles for the three images of the
// single-sweep backward, 28 % of a wave's lifetime)
template <typename T, int NCH>
__device__ __forceinline__ void rows_request(u32x4 (&r)[NCH], const T* src, long long ld, int S, int first) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int id = first + 128 * i;
    const int row = id >> 2, c4 = id & 3;
    r[i] = (u32x4){0u, 0u, 0u, 0u};
    if (row < S) r[i] = *(const u32x4*)(src + (long long)row * ld + c4 * 8);
  }
}
template <int NCH>
__device__ __forceinline__ void rows_store(char* img, const u32x4 (&r)[NCH], int SP, int first) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int id = first + 128 * i;
    if (id < SP * 4) *(u32x4*)(img + img_chunk_off(id >> 2, id & 3)) = r[i];
  }
}

constexpr float kScale = 0.17677669529663687f;  // 1/sqrt(32)

template <typename T, int SP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ ctx,
                                                       float* __restrict__ lse, int NB, int S, int H, int kv_shift,
                                                       DropCfg dc, const eg_step_state* st) {
  constexpr int NKT = SP / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // two waves share one (window, head): they split the staging of V and take alternate query tiles
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, role = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  int pid = blockIdx.x * 2 + (wave >> 1);
  const bool valid = pid < NB * H;
  if (!valid) pid = NB * H - 1;
  const int b = pid / H, h = pid % H;
  const int bk = (b + kv_shift) % NB;
  const int D = H * 32;
  const long long ld = 3ll * D;
  const T* qbase = qkv + (long long)b * S * ld + h * 32;
  const T* kbase = qkv + (long long)bk * S * ld + D + h * 32;
  const T* vbase = kbase + D;
  char* vimg = smem + (wave >> 1) * (SP * 64);
  constexpr int NCH = SP / 32;
  u32x4 rv[NCH];
  rows_request<T, NCH>(rv, vbase, ld, S, lane + 64 * role);    // (stored below, after every other request of the prologue)
  const int nkt = (S + 15) >> 4;
  FR<T> kf[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const int key = kt * 16 + l15;
    kf[kt] = ld_frag_global<T>(kbase + (long long)key * ld + g * 8, key < S);
  }
  uint32_t seed_lo = 0, seed_hi = 0;
  if (dc.thresh) { seed_lo = st->seed_lo; seed_hi = st->seed_hi; }
  rows_store<NCH>(vimg, rv, SP, lane + 64 * role);
  __syncthreads();
  // (Requesting every query tile's fragment before the loop, unrolling it, and a one-register tail key tile for S = 16 n + 1 were
  // measured on this kernel too: 52.9 -> 65.7 us at SP = 128, 69.3 -> 84.4 us at SP = 160, 27.7 -> 30.4 us at SP = 96 -- more registers,
  // one wave per SIMD fewer.  Only the batched prologue stayed.)
  for (int qt = role; qt < nkt; qt += 2) {
    const int q = qt * 16 + l15;
    const FR<T> qf = ld_frag_global<T>(qbase + (long long)q * ld + g * 8, q < S);
    f32x4 s[NKT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (kt < nkt) {   // wave-uniform: key tiles beyond the sequence cost nothing (S = 65 uses 5 of the 6 tiles)
        s[kt] = H16<T>::mfma(kf[kt], qf, s[kt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * g + r;
          const float v = key < S ? s[kt][r] * kScale : -INFINITY;
          s[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(s[kt][r] - mx);
          s[kt][r] = p;
          sum += p;
        }
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (g == 0 && q < S && valid) lse[((long long)b * H + h) * S + q] = mx + __logf(sum);
    const uint32_t rowidx = (uint32_t)((b * H + h) * S + q) * (uint32_t)((S + 1) & ~1);  // even row pitch: aligned pairs
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
        float pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = s[kt][r] * inv;
        eg_dropout_run<4>(pv, dc, seed_lo, seed_hi, rowidx + (uint32_t)(kt * 16 + 4 * g));
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kt][r] = pv[r];
      }
    }
    f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kp = 0; kp < NKT / 2; ++kp) {
      if (2 * kp < nkt) {
        const FR<T> pf = pack_frag<T>(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const FR<T> vf = ld_frag_lds_tr<T>(vimg, 32 * kp, dt, lane);
          o[dt] = H16<T>::mfma(vf, pf, o[dt]);
        }
      }
    }
    if (q < S && valid) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        float v[4] = {o[dt][0], o[dt][1], o[dt][2], o[dt][3]};
        store4(ctx + ((long long)b * S + q) * D + h * 32 + 16 * dt + 4 * g, v);
      }
    }
  }
}

template <typename T, int SP>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ ctx,
                                                       const T* __restrict__ dctx, const float* __restrict__ lse,
                                                       T* __restrict__ dqkv, int NB, int S, int H, int kv_shift,
                                                       DropCfg dc, const eg_step_state* st) {
  constexpr int NKT = SP / 16;
  constexpr int WB = 3 * SP * 64 + 2 * SP * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // two waves share one (window, head) and its LDS images: role 0 runs pass A (dQ) and the last k-tiles of pass B,
  // role 1 the first k-tiles of pass B (dK, dV) -- twice the resident waves for the same LDS footprint
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, role = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  int pid = blockIdx.x * 2 + (wave >> 1);
  const bool valid = pid < NB * H;
  if (!valid) pid = NB * H - 1;
  const int b = pid / H, h = pid % H;
  const int bk = (b + kv_shift) % NB;
  const int D = H * 32;
  const long long ld = 3ll * D;
  const T* qbase = qkv + (long long)b * S * ld + h * 32;
  const T* kbase = qkv + (long long)bk * S * ld + D + h * 32;
  const T* vbase = kbase + D;
  const T* dobase = dctx + (long long)b * S * D + h * 32;
  const T* obase = ctx + (long long)b * S * D + h * 32;
  char* base = smem + (wave >> 1) * WB;
  char* qimg = base;
  char* kimg = base + SP * 64;
  char* doimg = base + 2 * SP * 64;
  float* lsel = (float*)(base + 3 * SP * 64);
  float* dl = lsel + SP;
  // every global read of the prologue is requested before the first wait (see rows_request)
  constexpr int NCH = SP / 32, NQ = (SP + 127) / 128;
  const int first = lane + 64 * role;
  u32x4 rq[NCH], rk[NCH], rd[NCH];
  rows_request<T, NCH>(rq, qbase, ld, S, first);
  rows_request<T, NCH>(rk, kbase, ld, S, first);
  rows_request<T, NCH>(rd, dobase, D, S, first);
  u32x4 da[NQ][4], oa[NQ][4];
  float lq0[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int q = first + 128 * j;
    lq0[j] = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) { da[j][c4] = (u32x4){0u, 0u, 0u, 0u}; oa[j][c4] = (u32x4){0u, 0u, 0u, 0u}; }
    if (q < S) {
      lq0[j] = lse[((long long)b * H + h) * S + q];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        da[j][c4] = *(const u32x4*)(dobase + (long long)q * D + c4 * 8);
        oa[j][c4] = *(const u32x4*)(obase + (long long)q * D + c4 * 8);
      }
    }
  }
  rows_store<NCH>(qimg, rq, SP, first);
  rows_store<NCH>(kimg, rk, SP, first);
  rows_store<NCH>(doimg, rd, SP, first);
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int q = first + 128 * j;
    if (q < SP) {
      float dsum = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        float a[8], o[8];
        load8((const T*)&da[j][c4], a);
        load8((const T*)&oa[j][c4], o);
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += a[e] * o[e];
      }
      lsel[q] = lq0[j];
      dl[q] = dsum;
    }
  }
  const int nkt = (S + 15) >> 4;
  FR<T> kf[NKT], vf[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const int key = kt * 16 + l15;
    kf[kt] = ld_frag_global<T>(kbase + (long long)key * ld + g * 8, key < S);
    vf[kt] = ld_frag_global<T>(vbase + (long long)key * ld + g * 8, key < S);
  }
  uint32_t seed_lo = 0, seed_hi = 0;
  if (dc.thresh) { seed_lo = st->seed_lo; seed_hi = st->seed_hi; }
  const uint32_t headidx = (uint32_t)((b * H + h) * S);
  __syncthreads();

  // ---- pass A: key rows / query lanes -> dQ ----
  const int kt_split = (3 * nkt + 2) / 5;   // role 1 takes k-tiles [0, kt_split), role 0 the rest after pass A
  for (int qt = 0; qt < (role == 0 ? nkt : 0); ++qt) {
    const int q = qt * 16 + l15;
    const FR<T> qf = ld_frag_lds_row<T>(qimg, q, g);
    const FR<T> dof = ld_frag_lds_row<T>(doimg, q, g);
    const float lq = lsel[q], dq = dl[q];
    const uint32_t Sp2 = (uint32_t)((S + 1) & ~1);
    const uint32_t rowidx = (headidx + (uint32_t)q) * Sp2;
    f32x4 ds[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      ds[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (kt < nkt) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 sT = H16<T>::mfma(kf[kt], qf, z);
        const f32x4 dpT = H16<T>::mfma(vf[kt], dof, z);
        float dpv[4] = {dpT[0], dpT[1], dpT[2], dpT[3]};
        eg_dropout_run<4>(dpv, dc, seed_lo, seed_hi, rowidx + (uint32_t)(kt * 16 + 4 * g));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * g + r;
          const float p = key < S ? __expf(sT[r] * kScale - lq) : 0.f;
          ds[kt][r] = p * (dpv[r] - dq);
        }
      }
    }
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kp = 0; kp < NKT / 2; ++kp) {
      if (2 * kp < nkt) {
        const FR<T> dsf = pack_frag<T>(ds[2 * kp], ds[2 * kp + 1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const FR<T> ktr = ld_frag_lds_tr<T>(kimg, 32 * kp, dt, lane);
          acc[dt] = H16<T>::mfma(ktr, dsf, acc[dt]);
        }
      }
    }
    if (q < S && valid) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        float v[4] = {acc[dt][0] * kScale, acc[dt][1] * kScale, acc[dt][2] * kScale, acc[dt][3] * kScale};
        store4(dqkv + ((long long)b * S + q) * ld + h * 32 + 16 * dt + 4 * g, v);
      }
    }
  }

  // ---- pass B: query rows / key lanes -> dK, dV ----
  for (int kt = (role == 0 ? kt_split : 0); kt < (role == 0 ? nkt : kt_split); ++kt) {
    const int key = kt * 16 + l15;
    f32x4 dk[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    f32x4 dv[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    // kf/vf are indexed with a runtime kt here: pick the fragment with a uniform select chain (no scratch)
    FR<T> kfr = kf[0], vfr = vf[0];
#pragma unroll
    for (int i = 1; i < NKT; ++i)
      if (i == kt) { kfr = kf[i]; vfr = vf[i]; }
#pragma unroll
    for (int qp = 0; qp < NKT / 2; ++qp) {
      if (2 * qp < nkt) {
        f32x4 pd2[2], ds2[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int qt = 2 * qp + h2;
          pd2[h2] = (f32x4){0.f, 0.f, 0.f, 0.f};
          ds2[h2] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (qt < nkt) {
            const FR<T> qrow = ld_frag_lds_row<T>(qimg, qt * 16 + l15, g);
            const FR<T> dorow = ld_frag_lds_row<T>(doimg, qt * 16 + l15, g);
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 s = H16<T>::mfma(qrow, kfr, z);
            const f32x4 dp = H16<T>::mfma(dorow, vfr, z);
            const f32x4 l4 = *(const f32x4*)(lsel + qt * 16 + 4 * g);
            const f32x4 d4 = *(const f32x4*)(dl + qt * 16 + 4 * g);
            const uint32_t Sp2b = (uint32_t)((S + 1) & ~1);
            // One hash serves the elements (q, key) and (q, key ^ 1), which here sit in NEIGHBOURING LANES: a lane hashes two
            // of its four query rows (even keys rows 0-1, odd keys rows 2-3) and takes the other two from lane ^ 1.
            uint32_t hh[4] = {0u, 0u, 0u, 0u};
            if (dc.thresh) {
              const uint32_t odd = (uint32_t)key & 1u;
              const uint32_t ia = (headidx + (uint32_t)(qt * 16 + 4 * g) + 2u * odd) * Sp2b + (uint32_t)key;
              const uint32_t ha = eg_hash(seed_lo, seed_hi, dc.site, ia >> 1);
              const uint32_t hb = eg_hash(seed_lo, seed_hi, dc.site, (ia + Sp2b) >> 1);
              const uint32_t pa = (uint32_t)__shfl_xor((int)ha, 1, 64), pb = (uint32_t)__shfl_xor((int)hb, 1, 64);
              hh[0] = odd ? pa : ha; hh[1] = odd ? pb : hb;
              hh[2] = odd ? ha : pa; hh[3] = odd ? hb : pb;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int qq = qt * 16 + 4 * g + r;
              const float p = (key < S && qq < S) ? __expf(s[r] * kScale - l4[r]) : 0.f;
              float m = 1.0f;
              if (dc.thresh) {
                const uint32_t half = ((uint32_t)key & 1u) ? (hh[r] >> 16) : (hh[r] & 0xFFFFu);
                m = half >= dc.thresh ? dc.scale : 0.0f;
              }
              pd2[h2][r] = p * m;
              ds2[h2][r] = p * (dp[r] * m - d4[r]);
            }
          }
        }
        const FR<T> pdf = pack_frag<T>(pd2[0], pd2[1]);
        const FR<T> dsf = pack_frag<T>(ds2[0], ds2[1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const FR<T> dotr = ld_frag_lds_tr<T>(doimg, 32 * qp, dt, lane);
          dv[dt] = H16<T>::mfma(dotr, pdf, dv[dt]);
          const FR<T> qtr = ld_frag_lds_tr<T>(qimg, 32 * qp, dt, lane);
          dk[dt] = H16<T>::mfma(qtr, dsf, dk[dt]);
        }
      }
    }
    if (key < S && valid) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        float a[4] = {dk[dt][0] * kScale, dk[dt][1] * kScale, dk[dt][2] * kScale, dk[dt][3] * kScale};
        float c[4] = {dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]};
        T* row = dqkv + ((long long)bk * S + key) * ld + h * 32 + 16 * dt + 4 * g;
        store4(row + D, a);
        store4(row + 2 * D, c);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Single-sweep backward (S <= 128): every 16 x 16 block of P / dS is computed ONCE.
// The two-pass kernel above evaluates exp and the dropout hash of each block twice (pass B in the transposed layout) and is
// VALU-issue-bound on exactly that.  Here a wave owns alternate KEY tiles and walks all query tiles in pairs.  Per block it
// forms, in the key-rows / query-lanes layout, P*mask and dS; dS goes straight into the dQ MFMA of its query tile (B operand:
// query lanes, key k-slots -- the other key tile of the pair is a zero half), and both are TRANSPOSED through a 2-KB
// wave-private LDS image (written as 8-B row pieces, read back with ds_read_b64_tr_b16 as B operands with key lanes and the
// query pair in the k-slots) for the dV / dK MFMAs of the wave's key tile.  dK / dV of a key tile are complete in the wave
// (16 registers, stored once per key tile); dQ accumulates per query tile across the wave's key tiles (8 registers per tile) and
// the two waves of a head exchange halves through LDS in a fixed order (bit-reproducible).  Registers: 8 NKT + 16 accumulators
// instead of the 16 NKT of a query-major sweep, LDS 23 KB per head: three workgroups per CU, as the two-pass kernel.
// ------------------------------------------------------------------------------------------------
// NKTX > 0: the tile count is a compile-time constant and, with TAIL, the last key tile holds ONE valid key (S = 16 n + 1: the
// S = 65 windows of the benchmark); <SP, 0, false> is the general form.
template <int N> struct eg_int { static constexpr int value = N; };
// One key tile of the single-sweep backward (attn_bwd1_kernel): all of the wave's query pairs against key tile kt.
template <typename T, int SP, int NKTX, bool tail>
__device__ __forceinline__ void bwd1_key_tile(const int kt, const int S, const int nkt, const bool valid, const int lane, const char* kimg,
                                              const char* qimg, const char* doimg, const float* lsel, const float* dl, char* pimg,
                                              char* simg, const T* vbase, const long long ld, FR<T>& vnext, f32x4 (&accq)[SP / 16][2],
                                              const DropCfg& dc, const uint32_t seed_lo, const uint32_t seed_hi, const uint32_t headidx,
                                              const uint32_t Sp2, T* dqkv, const int bk, const int h, const int D) {
  constexpr int NKT = SP / 16;
  const int l15 = lane & 15, g = lane >> 4;
  const int qq = l15 >> 2, pp = l15 & 3;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  const int key = kt * 16 + l15;
  const FR<T> kfr = ld_frag_lds_row<T>(kimg, key, g);       // the staged image (zero rows beyond S) instead of a second global read
  const FR<T> vfr = vnext;                                   // requested one key tile ahead
  {
    const int key2 = key + 32;
    vnext = ld_frag_global<T>(vbase + (long long)key2 * ld + g * 8, key2 < S && kt + 2 < nkt);
  }
  FR<T> ktr[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) ktr[dt] = ld_frag_lds_tr<T>(kimg, 32 * (kt >> 1), dt, lane);
  f32x4 dk[2] = {zero4, zero4}, dv[2] = {zero4, zero4};
#pragma unroll
  for (int qp = 0; qp < NKT / 2; ++qp) {
    if (2 * qp < nkt) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int qt = 2 * qp + h2;
        const int off = (16 * h2 + l15) * 32 + g * 8;      // the block in the wave's images: query slot 16 h2 + l15, keys 4g .. 4g+3
        if (NKTX ? qt >= NKTX : qt >= nkt) {                // a query tile beyond the sequence: zero slots, no work
          *(u32x2*)(pimg + off) = (u32x2){0u, 0u};
          *(u32x2*)(simg + off) = (u32x2){0u, 0u};
          continue;
        }
        const int q = qt * 16 + l15;
        const FR<T> qf = ld_frag_lds_row<T>(qimg, q, g);
        const FR<T> dof = ld_frag_lds_row<T>(doimg, q, g);
        const float lq = lsel[q], dq = dl[q];
        const f32x4 sT = H16<T>::mfma(kfr, qf, zero4);
        const f32x4 dpT = H16<T>::mfma(vfr, dof, zero4);
        float pd[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 dsv = zero4;
        if (tail) {
          const float m0 = eg_dropout(1.f, dc, seed_lo, seed_hi, (headidx + (uint32_t)q) * Sp2 + (uint32_t)(kt * 16 + 4 * g));
          const float p = (g == 0 && q < S) ? __expf(sT[0] * kScale - lq) : 0.f;
          pd[0] = p * m0;
          dsv[0] = p * (dpT[0] * m0 - dq);
        } else {
          float m[4] = {1.f, 1.f, 1.f, 1.f};
          eg_dropout_run<4>(m, dc, seed_lo, seed_hi, (headidx + (uint32_t)q) * Sp2 + (uint32_t)(kt * 16 + 4 * g));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kk = kt * 16 + 4 * g + r;
            const float p = (kk < S && q < S) ? __expf(sT[r] * kScale - lq) : 0.f;
            pd[r] = p * m[r];
            dsv[r] = p * (dpT[r] * m[r] - dq);
          }
        }
        // dQ of this query tile: dS is the B operand as it stands (query lanes, key k-slots); the other key tile of the pair is zero
        const FR<T> dsf = (kt & 1) ? pack_frag<T>(zero4, dsv) : pack_frag<T>(dsv, zero4);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) accq[qt][dt] = H16<T>::mfma(ktr[dt], dsf, accq[qt][dt]);
        // the block into the wave's images (8 B of a 32-B row)
        u32x2 pw, sw;
        pw[0] = H16<T>::pack2(pd[0], pd[1]);   pw[1] = H16<T>::pack2(pd[2], pd[3]);
        sw[0] = H16<T>::pack2(dsv[0], dsv[1]); sw[1] = H16<T>::pack2(dsv[2], dsv[3]);
        *(u32x2*)(pimg + off) = pw;
        *(u32x2*)(simg + off) = sw;
      }
      // dV, dK of this key tile: the transposed blocks (key lanes, the query pair in the k-slots) against dO^T / Q^T
      s16x4 pa[2], sa[2];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int off = (16 * h2 + 4 * g + qq) * 32 + pp * 8;
        pa[h2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pimg + off));
        sa[h2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(simg + off));
      }
      const s16x8 pt = {pa[0][0], pa[0][1], pa[0][2], pa[0][3], pa[1][0], pa[1][1], pa[1][2], pa[1][3]};
      const s16x8 stt = {sa[0][0], sa[0][1], sa[0][2], sa[0][3], sa[1][0], sa[1][1], sa[1][2], sa[1][3]};
      const FR<T> pdf = __builtin_bit_cast(FR<T>, pt), dsT = __builtin_bit_cast(FR<T>, stt);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const FR<T> dotr = ld_frag_lds_tr<T>(doimg, 32 * qp, dt, lane);
        dv[dt] = H16<T>::mfma(dotr, pdf, dv[dt]);
        const FR<T> qtr = ld_frag_lds_tr<T>(qimg, 32 * qp, dt, lane);
        dk[dt] = H16<T>::mfma(qtr, dsT, dk[dt]);
      }
    }
  }
  if (key < S && valid) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      float a[4] = {dk[dt][0] * kScale, dk[dt][1] * kScale, dk[dt][2] * kScale, dk[dt][3] * kScale};
      float c[4] = {dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]};
      T* row = dqkv + ((long long)bk * S + key) * ld + h * 32 + 16 * dt + 4 * g;
      store4(row + D, a);
      store4(row + 2 * D, c);
    }
  }
}

template <typename T, int SP, int NKTX, bool TAIL>
__global__ __launch_bounds__(256, SP <= 96 ? 3 : 2) void attn_bwd1_kernel(const T* __restrict__ qkv, const T* __restrict__ ctx,
                                                        const T* __restrict__ dctx, const float* __restrict__ lse,
                                                        T* __restrict__ dqkv, int NB, int S, int H, int kv_shift,
                                                        DropCfg dc, const eg_step_state* st) {
  constexpr int NKT = SP / 16;
  constexpr int IMG = SP * 64;
  constexpr int SCR = 2 * 32 * 32;                            // one wave's P and dS images: [32 query slots][16 keys] each
  constexpr int WB = 3 * IMG + 2 * SP * 4 + 2 * SCR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, role = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  int pid = blockIdx.x * 2 + (wave >> 1);
  const bool valid = pid < NB * H;
  if (!valid) pid = NB * H - 1;
  const int b = pid / H, h = pid % H;
  const int bk = (b + kv_shift) % NB;
  const int D = H * 32;
  const long long ld = 3ll * D;
  const T* qbase = qkv + (long long)b * S * ld + h * 32;
  const T* kbase = qkv + (long long)bk * S * ld + D + h * 32;
  const T* vbase = kbase + D;
  const T* dobase = dctx + (long long)b * S * D + h * 32;
  const T* obase = ctx + (long long)b * S * D + h * 32;
  char* base = smem + (wave >> 1) * WB;
  char* qimg = base;
  char* kimg = base + IMG;
  char* doimg = base + 2 * IMG;
  float* lsel = (float*)(base + 3 * IMG);
  float* dl = lsel + SP;
  char* pimg = base + 3 * IMG + 2 * SP * 4 + role * SCR;
  char* simg = pimg + 32 * 32;
  // every global read of the prologue is requested before the first wait: the three images (this wave's half) and the lane's
  // dO / O rows and log-sum-exp for delta = rowsum(dO * O)
  constexpr int NCH = SP / 32;                               // 16-B chunks per lane and image: SP * 4 / 128
  const int first = lane + 64 * role;
  u32x4 rq[NCH], rk[NCH], rd[NCH];
  rows_request<T, NCH>(rq, qbase, ld, S, first);
  rows_request<T, NCH>(rk, kbase, ld, S, first);
  rows_request<T, NCH>(rd, dobase, D, S, first);
  constexpr int NQ = (SP + 127) / 128;                       // delta rows per lane (query first + 128 j)
  u32x4 da[NQ][4], oa[NQ][4];
  float lq0[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int qd = first + 128 * j;
    lq0[j] = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) { da[j][c4] = (u32x4){0u, 0u, 0u, 0u}; oa[j][c4] = (u32x4){0u, 0u, 0u, 0u}; }
    if (qd < S) {
      lq0[j] = lse[((long long)b * H + h) * S + qd];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        da[j][c4] = *(const u32x4*)(dobase + (long long)qd * D + c4 * 8);
        oa[j][c4] = *(const u32x4*)(obase + (long long)qd * D + c4 * 8);
      }
    }
  }
  rows_store<NCH>(qimg, rq, SP, first);
  rows_store<NCH>(kimg, rk, SP, first);
  rows_store<NCH>(doimg, rd, SP, first);
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int qd = first + 128 * j;
    if (qd < SP) {
      float dsum = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        float a[8], o[8];
        load8((const T*)&da[j][c4], a);
        load8((const T*)&oa[j][c4], o);
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += a[e] * o[e];
      }
      lsel[qd] = lq0[j];
      dl[qd] = dsum;
    }
  }
  const int nkt = (S + 15) >> 4;        // (run-time also when NKTX names it: the pair loop below keeps one basic block per pair --
                                        //  with a constant count the scheduler hoisted every pair's operand reads and spilled 31 registers)
  uint32_t seed_lo = 0, seed_hi = 0;
  if (dc.thresh) { seed_lo = st->seed_lo; seed_hi = st->seed_hi; }
  const uint32_t headidx = (uint32_t)((b * H + h) * S);
  const uint32_t Sp2 = (uint32_t)((S + 1) & ~1);
  __syncthreads();

  f32x4 accq[NKT][2];
#pragma unroll
  for (int qt = 0; qt < NKT; ++qt)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) accq[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int qq = l15 >> 2, pp = l15 & 3;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // A last key tile with a single valid key (TAIL) evaluates one accumulator register per lane instead of four, and the second
  // query tile of a pair is skipped when it lies beyond the sequence (odd tile counts): at S = 65 the wave with key tiles 0, 2, 4
  // did 18 full blocks against the other's 12; now 10 + 5 quarter blocks against 10.  The tail tile is its own copy of the body
  // (compile-time flag), so neither copy carries a branch inside its unrolled blocks.
  FR<T> vnext = ld_frag_global<T>(vbase + (long long)(role * 16 + l15) * ld + g * 8, role * 16 + l15 < S && role < nkt);
  const int nfull = TAIL ? NKTX - 1 : nkt;
#define EG_BWD1_ARGS S, nkt, valid, lane, kimg, qimg, doimg, lsel, dl, pimg, simg, vbase, ld, vnext, accq, dc, seed_lo, seed_hi, headidx, Sp2, dqkv, bk, h, D
  for (int kt = role; kt < nfull; kt += 2) bwd1_key_tile<T, SP, NKTX, false>(kt, EG_BWD1_ARGS);
  if (TAIL && role == ((NKTX - 1) & 1)) bwd1_key_tile<T, SP, NKTX, true>(NKTX - 1, EG_BWD1_ARGS);
#undef EG_BWD1_ARGS

  // ---- dQ: role 0 finishes query tiles [0, NKT/2), role 1 the rest; each hands the other its partial of the other's tiles ----
  __syncthreads();                                            // the images are dead: they carry the partials
  f32x4* xbuf = (f32x4*)base;                                 // [2 roles][NKT/2][2][64 lanes]
#pragma unroll
  for (int i = 0; i < NKT / 2; ++i) {
    const int qt = (1 - role) * (NKT / 2) + i;                // a tile the OTHER wave finishes
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      // Both candidates are pinned in registers before the choice: left to itself the compiler turns "role ? accq[i] : accq[j]" (as a
      // select or as a branch) into ONE load from a role-dependent address, which puts all of accq in scratch memory for the whole
      // kernel (784 us against 421 us per step when that happened to the S = 65 instantiation).
      f32x4 lo = accq[i][dt], hi = accq[NKT / 2 + i][dt];
      asm volatile("" : "+v"(lo), "+v"(hi));
      xbuf[((role * (NKT / 2) + i) * 2 + dt) * 64 + lane] = role ? lo : hi;
    }
    (void)qt;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NKT / 2; ++i) {
    const int qt = role * (NKT / 2) + i;
    const int q = qt * 16 + l15;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const f32x4 other = xbuf[(((1 - role) * (NKT / 2) + i) * 2 + dt) * 64 + lane];
      f32x4 lo = accq[i][dt], hi = accq[NKT / 2 + i][dt];
      asm volatile("" : "+v"(lo), "+v"(hi));
      const f32x4 mine = role ? hi : lo;
      const f32x4 tot = role ? other + mine : mine + other;    // fixed order: role 0's partial + role 1's partial
      if (q < S && valid) {
        float v[4] = {tot[0] * kScale, tot[1] * kScale, tot[2] * kScale, tot[3] * kScale};
        store4(dqkv + ((long long)b * S + q) * ld + h * 32 + 16 * dt + 4 * g, v);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Exact-fp32 attention (EG_F32): the parity path.  One workgroup per (sample, head), one thread per query
// (forward, dQ) or per key (dK, dV); Q/K/V/dO head slices live in LDS and are read as broadcasts.
// Plain fmaf chains, no MFMA: this path exists to pin the orchestration bit-tight against the fp32 oracle.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot32(const float* a, const float* b) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < 32; ++d) s = fmaf(a[d], b[d], s);
  return s;
}

__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                           float* __restrict__ lse, int NB, int S, int H, int kv_shift,
                                                           DropCfg dc, const eg_step_state* st) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Kl = (float*)smem;       // [S][32]
  float* Vl = Kl + S * 32;        // [S][32]
  const int pid = blockIdx.x, b = pid / H, h = pid % H, bk = (b + kv_shift) % NB, D = H * 32;
  const long long ld = 3ll * D;
  for (int i = threadIdx.x; i < S * 32; i += blockDim.x) {
    const int r = i >> 5, d = i & 31;
    Kl[i] = qkv[((long long)bk * S + r) * ld + D + h * 32 + d];
    Vl[i] = qkv[((long long)bk * S + r) * ld + 2 * D + h * 32 + d];
  }
  __syncthreads();
  const int q = threadIdx.x;
  if (q >= S) return;
  float qv[32], o[32];
#pragma unroll
  for (int d = 0; d < 32; ++d) { qv[d] = qkv[((long long)b * S + q) * ld + h * 32 + d]; o[d] = 0.f; }
  float mx = -INFINITY;
  for (int k = 0; k < S; ++k) mx = fmaxf(mx, dot32(qv, Kl + k * 32) * kScale);
  float sum = 0.f;
  for (int k = 0; k < S; ++k) sum += expf(dot32(qv, Kl + k * 32) * kScale - mx);
  const float inv = 1.0f / sum;
  uint32_t seed_lo = 0, seed_hi = 0;
  if (dc.thresh) { seed_lo = st->seed_lo; seed_hi = st->seed_hi; }
  const uint32_t rowidx = (uint32_t)((b * H + h) * S + q) * (uint32_t)((S + 1) & ~1);
  for (int k = 0; k < S; ++k) {
    float p = expf(dot32(qv, Kl + k * 32) * kScale - mx) * inv;
    if (dc.thresh) p = eg_dropout(p, dc, seed_lo, seed_hi, rowidx + (uint32_t)k);
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = fmaf(p, Vl[k * 32 + d], o[d]);
  }
  lse[((long long)b * H + h) * S + q] = mx + logf(sum);
#pragma unroll
  for (int d = 0; d < 32; ++d) ctx[((long long)b * S + q) * D + h * 32 + d] = o[d];
}

__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                           const float* __restrict__ dctx, const float* __restrict__ lse,
                                                           float* __restrict__ dqkv, int NB, int S, int H, int kv_shift,
                                                           DropCfg dc, const eg_step_state* st) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ql = (float*)smem;        // [S][32]
  float* Kl = Ql + S * 32;
  float* Vl = Kl + S * 32;
  float* Dl = Vl + S * 32;         // dO
  float* lsel = Dl + S * 32;       // [S]
  float* dl = lsel + S;            // [S]  delta = rowsum(dO * O)
  const int pid = blockIdx.x, b = pid / H, h = pid % H, bk = (b + kv_shift) % NB, D = H * 32;
  const long long ld = 3ll * D;
  for (int i = threadIdx.x; i < S * 32; i += blockDim.x) {
    const int r = i >> 5, d = i & 31;
    Ql[i] = qkv[((long long)b * S + r) * ld + h * 32 + d];
    Kl[i] = qkv[((long long)bk * S + r) * ld + D + h * 32 + d];
    Vl[i] = qkv[((long long)bk * S + r) * ld + 2 * D + h * 32 + d];
    Dl[i] = dctx[((long long)b * S + r) * D + h * 32 + d];
  }
  for (int r = threadIdx.x; r < S; r += blockDim.x) {
    lsel[r] = lse[((long long)b * H + h) * S + r];
    float s = 0.f;
    for (int d = 0; d < 32; ++d)
      s = fmaf(dctx[((long long)b * S + r) * D + h * 32 + d], ctx[((long long)b * S + r) * D + h * 32 + d], s);
    dl[r] = s;
  }
  __syncthreads();
  uint32_t seed_lo = 0, seed_hi = 0;
  if (dc.thresh) { seed_lo = st->seed_lo; seed_hi = st->seed_hi; }
  const uint32_t headidx = (uint32_t)((b * H + h) * S);
  const int t = threadIdx.x;
  if (t < S) {
    // thread = query t: dQ
    float acc[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) acc[d] = 0.f;
    for (int k = 0; k < S; ++k) {
      const float p = expf(dot32(Ql + t * 32, Kl + k * 32) * kScale - lsel[t]);
      float dp = dot32(Dl + t * 32, Vl + k * 32);
      if (dc.thresh) dp = eg_dropout(dp, dc, seed_lo, seed_hi, (headidx + (uint32_t)t) * (uint32_t)((S + 1) & ~1) + (uint32_t)k);
      const float ds = p * (dp - dl[t]);
#pragma unroll
      for (int d = 0; d < 32; ++d) acc[d] = fmaf(ds, Kl[k * 32 + d], acc[d]);
    }
#pragma unroll
    for (int d = 0; d < 32; ++d) dqkv[((long long)b * S + t) * ld + h * 32 + d] = acc[d] * kScale;
    // thread = key t: dK, dV
    float ak[32], av[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) { ak[d] = 0.f; av[d] = 0.f; }
    for (int q = 0; q < S; ++q) {
      const float p = expf(dot32(Ql + q * 32, Kl + t * 32) * kScale - lsel[q]);
      const float dpr = dot32(Dl + q * 32, Vl + t * 32);
      float m = 1.0f;
      if (dc.thresh) m = eg_dropout(1.0f, dc, seed_lo, seed_hi, (headidx + (uint32_t)q) * (uint32_t)((S + 1) & ~1) + (uint32_t)t);
      const float pd = p * m, ds = p * (dpr * m - dl[q]);
#pragma unroll
      for (int d = 0; d < 32; ++d) {
        av[d] = fmaf(pd, Dl[q * 32 + d], av[d]);
        ak[d] = fmaf(ds, Ql[q * 32 + d], ak[d]);
      }
    }
#pragma unroll
    for (int d = 0; d < 32; ++d) {
      dqkv[((long long)bk * S + t) * ld + D + h * 32 + d] = ak[d] * kScale;
      dqkv[((long long)bk * S + t) * ld + 2 * D + h * 32 + d] = av[d];
    }
  }
}

template <typename T, int SP>
int launch_fwd(const void* qkv, void* ctx, float* lse, int NB, int S, int H, int kv_shift, DropCfg dc,
               const eg_step_state* st, hipStream_t s) {
  const int nblk = (NB * H + 1) / 2;
  hipLaunchKernelGGL((attn_fwd_kernel<T, SP>), dim3(nblk), dim3(256), 2 * SP * 64, s, (const T*)qkv, (T*)ctx, lse,
                     NB, S, H, kv_shift, dc, st);
  return 0;
}
template <typename T, int SP>
int launch_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv, int NB, int S, int H,
               int kv_shift, DropCfg dc, const eg_step_state* st, hipStream_t s) {
  const int nblk = (NB * H + 1) / 2;
  static const int single = [] { const char* e = getenv("EYEGAZE_ATTN_BWD1"); return e ? atoi(e) : 1; }();
  // Single sweep (each P / dS block evaluated once; key-major, see attn_bwd1_kernel) for S <= 128: cfg3 step 3.87 -> 3.78 ms, cfg5
  // 5.63 -> 5.54 ms in round 2.  At SP = 128 it lost then (a5 9.04 vs 8.96 ms) -- with 272 B of scratch per lane nobody had noticed;
  // since accq stays in registers (see the kernel) it wins at every length: a5 (SP = 128) 7.38 -> 7.04 ms, a5c32 (SP = 160, 186 registers,
  // two waves per SIMD) 14.49 -> 13.84 ms.  EYEGAZE_ATTN_BWD1=0 forces the two-pass kernel.
  if (single) {
    constexpr int lds1 = 2 * (3 * SP * 64 + 2 * SP * 4 + 2 * 2 * 32 * 32);
    static bool attr1 = false;
    if (!attr1) {
      (void)hipFuncSetAttribute((const void*)attn_bwd1_kernel<T, SP, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
      (void)hipFuncSetAttribute((const void*)attn_bwd1_kernel<T, 96, 5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
      attr1 = true;
    }
    if (SP == 96 && S == 65)           // class token + 64 positions: exact tile count, single-key tail tile
      hipLaunchKernelGGL((attn_bwd1_kernel<T, 96, 5, true>), dim3(nblk), dim3(256), lds1, s, (const T*)qkv, (const T*)ctx,
                         (const T*)dctx, lse, (T*)dqkv, NB, S, H, kv_shift, dc, st);
    else
      hipLaunchKernelGGL((attn_bwd1_kernel<T, SP, 0, false>), dim3(nblk), dim3(256), lds1, s, (const T*)qkv, (const T*)ctx,
                         (const T*)dctx, lse, (T*)dqkv, NB, S, H, kv_shift, dc, st);
    return 0;
  }
  constexpr int lds = 2 * (3 * SP * 64 + 2 * SP * 4);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<T, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  hipLaunchKernelGGL((attn_bwd_kernel<T, SP>), dim3(nblk), dim3(256), lds, s, (const T*)qkv, (const T*)ctx,
                     (const T*)dctx, lse, (T*)dqkv, NB, S, H, kv_shift, dc, st);
  return 0;
}

}  // namespace

static int attn_check(const char* who, int NB, int S, int H, int kv_shift, int dtype, float p, const void* st) {
  EG_CHECK(NB > 0 && S > 0 && H > 0, "%s: bad shape NB=%d S=%d H=%d", who, NB, S, H);
  EG_CHECK(S <= 160, "%s: S=%d exceeds the register-resident limit of 160", who, S);
  EG_CHECK(kv_shift >= 0 && kv_shift < NB, "%s: kv_shift=%d out of range", who, kv_shift);
  EG_CHECK(dtype == EG_BF16 || dtype == EG_F32 || dtype == EG_F16, "%s: bad dtype %d", who, dtype);
  EG_CHECK(p >= 0.f && p < 1.f && (p == 0.f || st), "%s: dropout p=%f needs a step state", who, (double)p);
  EG_CHECK((long long)NB * H * S * (S + 1) < (1ll << 32), "%s: NB*H*S*S exceeds the 32-bit dropout index", who);
  return 0;
}

extern "C" int eg_attention_fwd(const void* qkv, void* ctx, float* lse, int NB, int S, int H, int kv_shift, int dtype,
                                float drop_p, uint32_t drop_site, const eg_step_state* state, void* stream) {
  EG_CHECK(qkv && ctx && lse, "eg_attention_fwd: null pointer");
  if (attn_check("eg_attention_fwd", NB, S, H, kv_shift, dtype, drop_p, state)) return 1;
  DropCfg dc = make_drop(drop_p, drop_site);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EG_F32) {
    hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(NB * H), dim3(256), 2 * S * 32 * 4, s, (const float*)qkv, (float*)ctx, lse,
                       NB, S, H, kv_shift, dc, state);
    EG_LAUNCH_CHECK("attention_fwd_f32");
    return 0;
  }
  if (dtype == EG_F16) {
    if (S <= 96) launch_fwd<f16_t, 96>(qkv, ctx, lse, NB, S, H, kv_shift, dc, state, s);
    else if (S <= 128) launch_fwd<f16_t, 128>(qkv, ctx, lse, NB, S, H, kv_shift, dc, state, s);
    else launch_fwd<f16_t, 160>(qkv, ctx, lse, NB, S, H, kv_shift, dc, state, s);
  } else {
    if (S <= 96) launch_fwd<bf16_t, 96>(qkv, ctx, lse, NB, S, H, kv_shift, dc, state, s);
    else if (S <= 128) launch_fwd<bf16_t, 128>(qkv, ctx, lse, NB, S, H, kv_shift, dc, state, s);
    else launch_fwd<bf16_t, 160>(qkv, ctx, lse, NB, S, H, kv_shift, dc, state, s);
  }
  EG_LAUNCH_CHECK("attention_fwd");
  return 0;
}

extern "C" int eg_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, void* dqkv, int NB,
                                int S, int H, int kv_shift, int dtype, float drop_p, uint32_t drop_site,
                                const eg_step_state* state, void* stream) {
  EG_CHECK(qkv && ctx && dctx && lse && dqkv, "eg_attention_bwd: null pointer");
  if (attn_check("eg_attention_bwd", NB, S, H, kv_shift, dtype, drop_p, state)) return 1;
  DropCfg dc = make_drop(drop_p, drop_site);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EG_F32) {
    const int lds = (4 * S * 32 + 2 * S) * 4;
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute((const void*)attn_bwd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (4 * 160 * 32 + 320) * 4);
      attr = true;
    }
    hipLaunchKernelGGL(attn_bwd_f32_kernel, dim3(NB * H), dim3(256), lds, s, (const float*)qkv, (const float*)ctx,
                       (const float*)dctx, lse, (float*)dqkv, NB, S, H, kv_shift, dc, state);
    EG_LAUNCH_CHECK("attention_bwd_f32");
    return 0;
  }
  if (dtype == EG_F16) {
    if (S <= 96) launch_bwd<f16_t, 96>(qkv, ctx, dctx, lse, dqkv, NB, S, H, kv_shift, dc, state, s);
    else if (S <= 128) launch_bwd<f16_t, 128>(qkv, ctx, dctx, lse, dqkv, NB, S, H, kv_shift, dc, state, s);
    else launch_bwd<f16_t, 160>(qkv, ctx, dctx, lse, dqkv, NB, S, H, kv_shift, dc, state, s);
  } else {
    if (S <= 96) launch_bwd<bf16_t, 96>(qkv, ctx, dctx, lse, dqkv, NB, S, H, kv_shift, dc, state, s);
    else if (S <= 128) launch_bwd<bf16_t, 128>(qkv, ctx, dctx, lse, dqkv, NB, S, H, kv_shift, dc, state, s);
    else launch_bwd<bf16_t, 160>(qkv, ctx, dctx, lse, dqkv, NB, S, H, kv_shift, dc, state, s);
  }
  EG_LAUNCH_CHECK("attention_bwd");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Attention probabilities for the analysis hooks (eeg_metrics.py:433-452 reads them through a forward hook on the
// attention-dropout module): probs[b, h, q, k] = exp(q.k / sqrt(32) - lse[b, h, q]) in fp32.  Not on the training path.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_probs_kernel(const T* __restrict__ qkv, const float* __restrict__ lse,
                                                         float* __restrict__ probs, int NB, int S, int H, int kv_shift) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ql = (float*)smem;   // [S][33]
  float* Kl = Ql + S * 33;    // [S][33]
  const int pid = blockIdx.x, b = pid / H, h = pid % H, bk = (b + kv_shift) % NB, D = H * 32;
  const long long ld = 3ll * D;
  for (int i = threadIdx.x; i < S * 32; i += blockDim.x) {
    const int r = i >> 5, d = i & 31;
    Ql[r * 33 + d] = Elem<T>::ld(qkv + ((long long)b * S + r) * ld + h * 32 + d);
    Kl[r * 33 + d] = Elem<T>::ld(qkv + ((long long)bk * S + r) * ld + D + h * 32 + d);
  }
  __syncthreads();
  const float* lrow = lse + ((long long)b * H + h) * S;
  float* out = probs + ((long long)b * H + h) * S * S;
  for (int i = threadIdx.x; i < S * S; i += blockDim.x) {
    const int q = i / S, k = i - q * S;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) acc = fmaf(Ql[q * 33 + d], Kl[k * 33 + d], acc);
    out[i] = expf(acc * kScale - lrow[q]);
  }
}

extern "C" int eg_attention_probs(const void* qkv, const float* lse, float* probs, int NB, int S, int H, int kv_shift,
                                  int dtype, void* stream) {
  EG_CHECK(qkv && lse && probs, "eg_attention_probs: null pointer");
  if (attn_check("eg_attention_probs", NB, S, H, kv_shift, dtype, 0.f, nullptr)) return 1;
  const int lds = 2 * S * 33 * 4;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EG_F32)
    hipLaunchKernelGGL(attn_probs_kernel<float>, dim3(NB * H), dim3(256), lds, s, (const float*)qkv, lse, probs, NB, S, H, kv_shift);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(attn_probs_kernel<f16_t>, dim3(NB * H), dim3(256), lds, s, (const f16_t*)qkv, lse, probs, NB, S, H, kv_shift);
  else
    hipLaunchKernelGGL(attn_probs_kernel<bf16_t>, dim3(NB * H), dim3(256), lds, s, (const bf16_t*)qkv, lse, probs, NB, S, H, kv_shift);
  EG_LAUNCH_CHECK("attention_probs");
  return 0;
}
