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
    const float ls = st->scaler_on ? st->loss_scale : 1.0f;
    const float norm = (float)sqrt(red[0]) * (st->grad_scale / ls);
    st->grad_norm = norm;
    float coef = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.0f;
    st->clip_coef = fminf(coef, 1.0f);
    st->found_inf = (st->scaler_on && !(fabsf(norm) <= 3.0e38f)) ? 1u : 0u;   // inf or NaN
  }
}

__global__ void set_step_state_kernel(eg_step_state* st, uint32_t seed_lo, uint32_t seed_hi, float lr, float bc1, float bc2,
                                      float grad_scale, int reset_scaler, float init_scale, int use_dev_t) {
  st->seed_lo = seed_lo; st->seed_hi = seed_hi; st->lr = lr; st->bias_corr1 = bc1; st->bias_corr2 = bc2;
  st->grad_scale = grad_scale; st->use_dev_t = (uint32_t)use_dev_t;
  if (reset_scaler) {
    st->scaler_on = reset_scaler == 1 ? 1u : 0u;
    st->loss_scale = reset_scaler == 1 ? init_scale : 1.0f;
    st->found_inf = 0u; st->good_steps = 0u; st->opt_steps = 0u; st->skipped = 0u;
    st->clip_coef = 1.0f; st->grad_norm = 0.0f;
  }
}

__global__ void scaler_update_kernel(eg_step_state* st, float growth, float backoff, int growth_interval) {
  if (st->scaler_on && st->found_inf) {
    st->loss_scale *= backoff;
    st->good_steps = 0u;
    st->skipped += 1u;
    return;
  }
  st->opt_steps += 1u;
  if (!st->scaler_on) return;
  st->good_steps += 1u;
  if ((int)st->good_steps >= growth_interval) {
    st->loss_scale *= growth;
    st->good_steps = 0u;
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n,
                                                    float beta1, float beta2, float eps, float wd, float lr_mult,
                                                    const eg_step_state* __restrict__ st) {
  if (st->scaler_on && st->found_inf) return;        // GradScaler.step: non-finite gradients -> no update at all
  float bc1 = st->bias_corr1, bc2 = st->bias_corr2;
  if (st->use_dev_t) {                               // skipped steps do not advance t: the device keeps the count
    const float t = (float)(st->opt_steps + 1u);
    bc1 = 1.0f - powf(beta1, t);
    bc2 = 1.0f - powf(beta2, t);
  }
  const float lr = st->lr * lr_mult, bc2s = sqrtf(bc2);
  const float gs = st->grad_scale * st->clip_coef / (st->scaler_on ? st->loss_scale : 1.0f);
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

extern "C" int eg_adamw_group(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                              float weight_decay, float lr_mult, const eg_step_state* state, void* stream) {
  EG_CHECK(p && g && m && v && state && n > 0, "eg_adamw: bad arguments");
  EG_CHECK(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "eg_adamw: alignment");
  const long long nt = (n + 3) / 4;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (long long)n, beta1, beta2, eps, weight_decay, lr_mult, state);
  EG_LAUNCH_CHECK("adamw");
  return 0;
}

extern "C" int eg_adamw(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                        float weight_decay, const eg_step_state* state, void* stream) {
  return eg_adamw_group(p, g, m, v, n, beta1, beta2, eps, weight_decay, 1.0f, state, stream);
}

extern "C" int eg_set_step_state(eg_step_state* state, uint32_t seed_lo, uint32_t seed_hi, float lr, float bias_corr1,
                                 float bias_corr2, float grad_scale, int reset_scaler, float init_scale, int use_dev_t,
                                 void* stream) {
  EG_CHECK(state && (uintptr_t)state % 16 == 0, "eg_set_step_state: state must be a 16-B aligned device pointer");
  EG_CHECK(reset_scaler >= 0 && reset_scaler <= 2, "eg_set_step_state: reset_scaler %d", reset_scaler);
  EG_CHECK(reset_scaler != 1 || init_scale > 0.f, "eg_set_step_state: init_scale must be positive");
  hipLaunchKernelGGL(set_step_state_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, seed_lo, seed_hi, lr, bias_corr1,
                     bias_corr2, grad_scale, reset_scaler, init_scale, use_dev_t);
  EG_LAUNCH_CHECK("set_step_state");
  return 0;
}

extern "C" int eg_scaler_update(eg_step_state* state, float growth, float backoff, int growth_interval, void* stream) {
  EG_CHECK(state && growth >= 1.f && backoff > 0.f && backoff <= 1.f && growth_interval > 0, "eg_scaler_update: bad arguments");
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, growth, backoff, growth_interval);
  EG_LAUNCH_CHECK("scaler_update");
  return 0;
}

extern "C" int eg_fill_f32(float* p, int64_t n, float value, void* stream) {
  EG_CHECK(p && n > 0, "eg_fill_f32: bad arguments");
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, (long long)n,
                     value);
  EG_LAUNCH_CHECK("fill");
  return 0;
}
