// Optimiser step over the FLAT parameter / gradient buffers (T:221-222, T:401-405):
//   global L2 norm -> clip coefficient (device resident, no host sync) -> AdamW (decoupled weight decay).
// HBM-bound: one pass reads g, p, m, v and writes p, m, v (28 B/param).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long long n,
                                                             float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const long long stride = (long long)gridDim.x * 256 * 4;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const f32x4 v = *(const f32x4*)(g + i);
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    } else {
      for (long long j = i; j < n; ++j) s += g[j] * g[j];
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void clip_coef_kernel(const float* __restrict__ partial, int n, float max_norm, eg_step_state* st) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]) * st->grad_scale;
    st->grad_norm = norm;
    float coef = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.0f;
    st->clip_coef = fminf(coef, 1.0f);
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n,
                                                    float beta1, float beta2, float eps, float wd,
                                                    const eg_step_state* __restrict__ st) {
  const float lr = st->lr, bc1 = st->bias_corr1, bc2s = sqrtf(st->bias_corr2);
  const float gs = st->grad_scale * st->clip_coef;
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const int cnt = (int)min(4ll, n - i);
  float pv[4], gv[4], mv[4], vv[4];
  if (cnt == 4) {
    load4(p + i, pv); load4(g + i, gv); load4(m + i, mv); load4(v + i, vv);
  } else {
    for (int e = 0; e < cnt; ++e) { pv[e] = p[i + e]; gv[e] = g[i + e]; mv[e] = m[i + e]; vv[e] = v[i + e]; }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e < cnt) {
      const float gg = gv[e] * gs;
      pv[e] *= 1.0f - lr * wd;
      mv[e] = beta1 * mv[e] + (1.0f - beta1) * gg;
      vv[e] = beta2 * vv[e] + (1.0f - beta2) * gg * gg;
      const float denom = sqrtf(vv[e]) / bc2s + eps;
      pv[e] -= (lr / bc1) * (mv[e] / denom);
    }
  }
  if (cnt == 4) {
    store4(p + i, pv); store4(m + i, mv); store4(v + i, vv);
  } else {
    for (int e = 0; e < cnt; ++e) { p[i + e] = pv[e]; m[i + e] = mv[e]; v[i + e] = vv[e]; }
  }
}

__global__ void fill_kernel(float* __restrict__ p, long long n, float val) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = val;
}

}  // namespace

extern "C" int eg_grad_sqnorm(const float* g, int64_t n, float* partial, int nblk, void* stream) {
  EG_CHECK(g && partial && n > 0 && nblk > 0 && nblk <= 4096, "eg_grad_sqnorm: bad arguments");
  EG_CHECK((uintptr_t)g % 16 == 0, "eg_grad_sqnorm: alignment");
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, (long long)n, partial);
  EG_LAUNCH_CHECK("grad_sqnorm");
  return 0;
}

extern "C" int eg_clip_coef(const float* partial, int nblk, float max_norm, eg_step_state* state, void* stream) {
  EG_CHECK(partial && state && nblk > 0, "eg_clip_coef: bad arguments");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, max_norm, state);
  EG_LAUNCH_CHECK("clip_coef");
  return 0;
}

extern "C" int eg_adamw(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                        float weight_decay, const eg_step_state* state, void* stream) {
  EG_CHECK(p && g && m && v && state && n > 0, "eg_adamw: bad arguments");
  EG_CHECK(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "eg_adamw: alignment");
  const long long nt = (n + 3) / 4;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (long long)n, beta1, beta2, eps, weight_decay, state);
  EG_LAUNCH_CHECK("adamw");
  return 0;
}

extern "C" int eg_fill_f32(float* p, int64_t n, float value, void* stream) {
  EG_CHECK(p && n > 0, "eg_fill_f32: bad arguments");
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, (long long)n,
                     value);
  EG_LAUNCH_CHECK("fill");
  return 0;
}
