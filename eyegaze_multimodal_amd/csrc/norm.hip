// LayerNorm(eps = 1e-5) forward / backward, one wave per row, fp32 statistics.  HBM-bound.
#include "common.h"

namespace {

constexpr int LN_MAXCH = 4;  // lane owns chunks lane + 64*j of 4 elements => D <= 1024

template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ stats, int M, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = D >> 2;
  float v[LN_MAXCH][4];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXCH; ++j) {
    const int c = lane + 64 * j;
    if (c < nch) {
      load4(x + (size_t)row * D + c * 4, v[j]);
      s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    }
  }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXCH; ++j) {
    const int c = lane + 64 * j;
    if (c < nch)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[j][e] - mean;
        q += d * d;
      }
  }
  const float rstd = rsqrtf(wave_sum(q) / D + 1e-5f);
#pragma unroll
  for (int j = 0; j < LN_MAXCH; ++j) {
    const int c = lane + 64 * j;
    if (c < nch) {
      float g[4], b[4], o[4];
      load4(gamma + c * 4, g);
      load4(beta + c * 4, b);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * g[e] + b[e];
      store4(y + (size_t)row * D + c * 4, o);
    }
  }
  if (stats && lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

// each wave walks rows wave_id, wave_id + nwaves, ...; dgamma/dbeta accumulate in registers and are
// reduced per block into partial[blk][0][D] (dgamma) and partial[blk][1][D] (dbeta)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, T* __restrict__ dx,
                                                            T* __restrict__ dx_drop, float* __restrict__ partial, int M,
                                                            int D, DropCfg d1, DropCfg d2, const eg_step_state* st) {
  __shared__ float red[4][2][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D >> 2;
  float g[LN_MAXCH][4], dg[LN_MAXCH][4], db[LN_MAXCH][4];
#pragma unroll
  for (int j = 0; j < LN_MAXCH; ++j) {
    const int c = lane + 64 * j;
#pragma unroll
    for (int e = 0; e < 4; ++e) { g[j][e] = 0.f; dg[j][e] = 0.f; db[j][e] = 0.f; }
    if (c < nch) load4(gamma + c * 4, g[j]);
  }
  uint32_t seed_lo = 0, seed_hi = 0;
  const bool drop = (d1.thresh | d2.thresh) != 0 && dx_drop != nullptr;
  if (drop) { seed_lo = st->seed_lo; seed_hi = st->seed_hi; }
  const int nwaves = gridDim.x * 4;
  for (int row = blockIdx.x * 4 + wave; row < M; row += nwaves) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float xh[LN_MAXCH][4], dv[LN_MAXCH][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        float xv[4];
        load4(x + (size_t)row * D + c * 4, xv);
        load4(dy + (size_t)row * D + c * 4, dv[j]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[j][e] = (xv[e] - mean) * rstd;
          dg[j][e] += dv[j][e] * xh[j][e];
          db[j][e] += dv[j][e];
          const float dxh = dv[j][e] * g[j][e];
          s1 += dxh;
          s2 += dxh * xh[j][e];
        }
      }
    }
    const float c1 = wave_sum(s1) / D, c2 = wave_sum(s2) / D;
#pragma unroll
    for (int j = 0; j < LN_MAXCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (dv[j][e] * g[j][e] - c1 - xh[j][e] * c2);
        store4(dx + (size_t)row * D + c * 4, o);
        if (dx_drop) {
          if (drop) {
            const uint32_t idx = (uint32_t)row * (uint32_t)D + (uint32_t)(c * 4);
            eg_dropout_run<4>(o, d1, seed_lo, seed_hi, idx);
            eg_dropout_run<4>(o, d2, seed_lo, seed_hi, idx);
          }
          store4(dx_drop + (size_t)row * D + c * 4, o);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < LN_MAXCH; ++j) {
    const int c = lane + 64 * j;
    if (c < nch)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[wave][0][c * 4 + e] = dg[j][e];
        red[wave][1][c * 4 + e] = db[j][e];
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    const int which = i / D, n = i % D;
    partial[(size_t)blockIdx.x * 2 * D + i] = red[0][which][n] + red[1][which][n] + red[2][which][n] + red[3][which][n];
  }
}

// ---- D == 256 fast path: half a wave (32 lanes x 8 elements = 16-B accesses) per row, two rows per wave ----
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd256_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, T* __restrict__ y,
                                                               float* __restrict__ stats, int M) {
  const int lane = threadIdx.x & 63, l = lane & 31;
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  const bool ok = row < M;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (ok) load8(x + (size_t)row * 256 + l * 8, v);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += v[e];
  const float mean = half_sum(s) * (1.0f / 256.f);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; q += d * d; }
  const float rstd = rsqrtf(half_sum(q) * (1.0f / 256.f) + 1e-5f);
  if (!ok) return;
  float g[8], b[8], o[8];
  load8(gamma + l * 8, g);
  load8(beta + l * 8, b);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (v[e] - mean) * rstd * g[e] + b[e];
  store8(y + (size_t)row * 256 + l * 8, o);
  if (stats && l == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd256_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                               const float* __restrict__ stats,
                                                               const float* __restrict__ gamma, T* __restrict__ dx,
                                                               T* __restrict__ dx_drop, float* __restrict__ partial, int M,
                                                               DropCfg d1, DropCfg d2, const eg_step_state* st) {
  __shared__ float red[8][2][256];
  const int lane = threadIdx.x & 63, l = lane & 31, hw = threadIdx.x >> 5;  // 8 half-waves per block
  float g[8], dg[8], db[8];
  load8(gamma + l * 8, g);
#pragma unroll
  for (int e = 0; e < 8; ++e) { dg[e] = 0.f; db[e] = 0.f; }
  uint32_t seed_lo = 0, seed_hi = 0;
  const bool drop = (d1.thresh | d2.thresh) != 0 && dx_drop != nullptr;
  if (drop) { seed_lo = st->seed_lo; seed_hi = st->seed_hi; }
  const int nrows = gridDim.x * 8;
  // both halves of a wave must run the same trip count (the shuffles below are wave-wide)
  const int trips = (M + nrows - 1) / nrows;
  // software pipeline: the next trip's rows are requested before this trip's arithmetic (4 x 16 B in flight per lane)
  constexpr int NV = (int)sizeof(T) / 2;                     // 16-B vectors per 8 elements
  struct Row { u32x4 x[NV], d[NV]; float mean, rstd; };      // rows wait as loaded (packed): four of them are in flight per lane
  auto fetch = [&](int it, Row& r) {
    const int row = it * nrows + blockIdx.x * 8 + hw;
#pragma unroll
    for (int v = 0; v < NV; ++v) { r.x[v] = (u32x4){0u, 0u, 0u, 0u}; r.d[v] = (u32x4){0u, 0u, 0u, 0u}; }
    r.mean = 0.f; r.rstd = 0.f;
    if (it < trips && row < M) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        r.x[v] = *(const u32x4*)((const char*)(x + (size_t)row * 256 + l * 8) + 16 * v);
        r.d[v] = *(const u32x4*)((const char*)(dy + (size_t)row * 256 + l * 8) + 16 * v);
      }
      r.mean = stats[2 * row];
      r.rstd = stats[2 * row + 1];
    }
  };
  // TWO trips' rows are in flight (8 x 16 B per lane): a block walks four trips at the benchmark size, and with one trip ahead the
  // launch was a chain of five HBM latencies (17 us for 68 MB); the rows are still consumed in trip order (same sums)
  auto consume = [&](int it, const Row& cur) {
    const int row = it * nrows + blockIdx.x * 8 + hw;
    const bool ok = row < M;
    const float mean = cur.mean, rstd = cur.rstd;
    float xv[8], dv[8];
    load8((const T*)&cur.x[0], xv);
    load8((const T*)&cur.d[0], dv);
    float o[8];
    eg_ln_bwd_row8(xv, dv, g, mean, rstd, dg, db, o);
    if (ok) {
      store8(dx + (size_t)row * 256 + l * 8, o);
      if (dx_drop) {
        if (drop) {
          const uint32_t idx = (uint32_t)row * 256u + (uint32_t)(l * 8);
          eg_dropout_run<8>(o, d1, seed_lo, seed_hi, idx);
          eg_dropout_run<8>(o, d2, seed_lo, seed_hi, idx);
        }
        store8(dx_drop + (size_t)row * 256 + l * 8, o);
      }
    }
  };
  Row ra, rb, rc, rd;
  fetch(0, ra);
  fetch(1, rb);
  for (int it = 0; it < trips; it += 2) {
    fetch(it + 2, rc);
    fetch(it + 3, rd);
    consume(it, ra);
    if (it + 1 < trips) consume(it + 1, rb);
    ra = rc;
    rb = rd;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[hw][0][l * 8 + e] = dg[e];
    red[hw][1][l * 8 + e] = db[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int which = i >> 8, n = i & 255;
    float s = 0.f;
#pragma unroll
    for (int h = 0; h < 8; ++h) s += red[h][which][n];
    partial[(size_t)blockIdx.x * 512 + i] = s;
  }
}

}  // namespace

extern "C" int eg_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int M, int D,
                                int dtype, void* stream) {
  EG_CHECK(x && gamma && beta && y, "eg_layernorm_fwd: null pointer");
  EG_CHECK(M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "eg_layernorm_fwd: D=%d must be a multiple of 4, <= 1024", D);
  dim3 grid((M + 3) / 4);
  if (D == 256 && (dtype == EG_BF16 || dtype == EG_F32 || dtype == EG_F16)) {
    dim3 g8((M + 7) / 8);
    if (dtype == EG_BF16)
      hipLaunchKernelGGL(layernorm_fwd256_kernel<bf16_t>, g8, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, gamma, beta,
                         (bf16_t*)y, stats, M);
    else if (dtype == EG_F16)
      hipLaunchKernelGGL(layernorm_fwd256_kernel<f16_t>, g8, dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, gamma, beta,
                         (f16_t*)y, stats, M);
    else
      hipLaunchKernelGGL(layernorm_fwd256_kernel<float>, g8, dim3(256), 0, (hipStream_t)stream, (const float*)x, gamma, beta,
                         (float*)y, stats, M);
    EG_LAUNCH_CHECK("layernorm_fwd256");
    return 0;
  }
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(layernorm_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, gamma,
                       beta, (bf16_t*)y, stats, M, D);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(layernorm_fwd_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, gamma,
                       beta, (f16_t*)y, stats, M, D);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(layernorm_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, gamma, beta,
                       (float*)y, stats, M, D);
  else
    return eg_fail("eg_layernorm_fwd: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("layernorm_fwd");
  return 0;
}

extern "C" int eg_layernorm_bwd(const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                                void* dx_drop, float* partial, int nblk, int partial_capacity_blocks, int M, int D, int dtype,
                                float drop1_p, uint32_t drop1_site, float drop2_p, uint32_t drop2_site,
                                const eg_step_state* state, void* stream) {
  EG_CHECK(dy && x && stats && gamma && dx && partial, "eg_layernorm_bwd: null pointer");
  EG_CHECK(M > 0 && D > 0 && D % 4 == 0 && D <= 1024 && nblk > 0, "eg_layernorm_bwd: bad shape");
  // every workgroup writes its own [2, D] row of `partial`: a grid larger than the buffer is an out-of-bounds store (round 2: a
  // 2080-block sweep into a 2048-block buffer ended in a GPU memory access fault), so the caller states the capacity
  EG_CHECK(nblk <= partial_capacity_blocks, "eg_layernorm_bwd: nblk %d exceeds the partial buffer's capacity of %d blocks",
           nblk, partial_capacity_blocks);
  EG_CHECK((drop1_p == 0.f && drop2_p == 0.f) || state, "eg_layernorm_bwd: dropout needs a step state");
  EG_CHECK((long long)M * D < (1ll << 32), "eg_layernorm_bwd: M*D exceeds the 32-bit dropout index");
  DropCfg d1 = make_drop(drop1_p, drop1_site), d2 = make_drop(drop2_p, drop2_site);
  if (D == 256 && (dtype == EG_BF16 || dtype == EG_F32 || dtype == EG_F16)) {
    if (dtype == EG_BF16)
      hipLaunchKernelGGL(layernorm_bwd256_kernel<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                         (const bf16_t*)x, stats, gamma, (bf16_t*)dx, (bf16_t*)dx_drop, partial, M, d1, d2, state);
    else if (dtype == EG_F16)
      hipLaunchKernelGGL(layernorm_bwd256_kernel<f16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const f16_t*)dy,
                         (const f16_t*)x, stats, gamma, (f16_t*)dx, (f16_t*)dx_drop, partial, M, d1, d2, state);
    else
      hipLaunchKernelGGL(layernorm_bwd256_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                         (const float*)x, stats, gamma, (float*)dx, (float*)dx_drop, partial, M, d1, d2, state);
    EG_LAUNCH_CHECK("layernorm_bwd256");
    return 0;
  }
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(layernorm_bwd_kernel<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)x, stats, gamma, (bf16_t*)dx, (bf16_t*)dx_drop, partial, M, D, d1, d2, state);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(layernorm_bwd_kernel<f16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const f16_t*)dy,
                       (const f16_t*)x, stats, gamma, (f16_t*)dx, (f16_t*)dx_drop, partial, M, D, d1, d2, state);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                       (const float*)x, stats, gamma, (float*)dx, (float*)dx_drop, partial, M, D, d1, d2, state);
  else
    return eg_fail("eg_layernorm_bwd: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("layernorm_bwd");
  return 0;
}
