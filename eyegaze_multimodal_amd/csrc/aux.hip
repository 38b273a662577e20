// Batch-level auxiliary losses of the dual-stream classifier (3_Models/backbones/dual_eeg_transformer.py:1255-1371), each
// with its gradient w.r.t. the [B, d] tokens it reads (for an upstream gradient of 1; the caller scales).  All fp32; the
// tensors are [B, d]-sized (B <= 1024), so these are small latency-bound launches, not MFMA work:
//   eg_aux_symmetry  F.mse_loss(cls1, cls2)                                              :1255-1260
//   eg_aux_infonce   cross_entropy(normalize(ibs) . normalize(cat[cls1, cls2])^T / tau, arange(B))   :1262-1304
//   eg_aux_supcon    supervised contrastive loss over normalize(ibs), exp(sim) without max shift, 1e-8 guards, mean over
//                    the rows that have a positive; 0 when no row has one                 :1306-1371
// F.normalize(x) = x / max(||x||, 1e-12).
#include "common.h"

namespace {

constexpr int kMaxB = 1024;

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// n[r,:] = x[r,:] / max(||x[r,:]||, 1e-12), inv[r] = that factor.  One block per row.
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, float* __restrict__ n, float* __restrict__ inv,
                                                      int D) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) { const float v = x[(size_t)r * D + d]; s += v * v; }
  const float f = 1.0f / fmaxf(sqrtf(block_sum(s, red)), 1e-12f);
  for (int d = threadIdx.x; d < D; d += 256) n[(size_t)r * D + d] = x[(size_t)r * D + d] * f;
  if (threadIdx.x == 0) inv[r] = f;
}

// dx[r,:] = (dn - n (n . dn)) * inv[r]   (backward of the row normalisation).  Called inside the row kernels below.
__device__ __forceinline__ void normalize_bwd_row(const float* __restrict__ n, const float* dn_lds, float inv, float* __restrict__ dx,
                                                  int D, float* red) {
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) s += n[d] * dn_lds[d];
  const float dot = block_sum(s, red);
  for (int d = threadIdx.x; d < D; d += 256) dx[d] = (dn_lds[d] - n[d] * dot) * inv;
}

__global__ __launch_bounds__(256) void symmetry_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ loss, float* __restrict__ da, float* __restrict__ db,
                                                       long long n) {
  __shared__ float red[4];
  const float scale = 2.0f / (float)n;
  float s = 0.f;
  for (long long i = threadIdx.x; i < n; i += 256) {
    const float d = a[i] - b[i];
    s += d * d;
    da[i] = d * scale;
    db[i] = -d * scale;
  }
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = tot / (float)n;
}

// InfoNCE row i: logits over the 2B normalised targets, loss_i = lse - logit_i, dsim = (softmax - onehot_i) / B
__global__ __launch_bounds__(256) void infonce_rows_kernel(const float* __restrict__ nI, const float* __restrict__ nT,
                                                           float* __restrict__ dsim, float* __restrict__ lossrow, int B, int D,
                                                           float inv_tau) {
  extern __shared__ float sm[];   // [D] query row, [2B] logits
  __shared__ float red[4];
  float* q = sm;
  float* lg = sm + D;
  const int i = blockIdx.x, T2 = 2 * B;
  for (int d = threadIdx.x; d < D; d += 256) q[d] = nI[(size_t)i * D + d];
  __syncthreads();
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < T2; j += 256) {
    const float* t = nT + (size_t)j * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(q[d], t[d], s);
    s *= inv_tau;
    lg[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = block_max(mx, red);
  float se = 0.f;
  for (int j = threadIdx.x; j < T2; j += 256) se += expf(lg[j] - mx);
  const float lse = mx + logf(block_sum(se, red));
  for (int j = threadIdx.x; j < T2; j += 256)
    dsim[(size_t)i * T2 + j] = (expf(lg[j] - lse) - (j == i ? 1.f : 0.f)) / (float)B;
  if (threadIdx.x == 0) lossrow[i] = lse - lg[i];
}

// rows [0,B): d nI_r = sum_j dsim[r][j] nT_j / tau ; rows [B,3B): d nT_t = sum_i dsim[i][t] nI_i / tau ; then the
// normalisation backward of that row -> d_ibs / d_cls1 / d_cls2
__global__ __launch_bounds__(256) void infonce_grad_kernel(const float* __restrict__ nI, const float* __restrict__ nT,
                                                           const float* __restrict__ invI, const float* __restrict__ invT,
                                                           const float* __restrict__ dsim, float* __restrict__ d_ibs,
                                                           float* __restrict__ d_cls1, float* __restrict__ d_cls2, int B, int D,
                                                           float inv_tau) {
  extern __shared__ float sm[];   // [D] dn
  __shared__ float red[4];
  const int r = blockIdx.x, T2 = 2 * B;
  if (r < B) {
    for (int d = threadIdx.x; d < D; d += 256) {
      float s = 0.f;
      for (int j = 0; j < T2; ++j) s = fmaf(dsim[(size_t)r * T2 + j], nT[(size_t)j * D + d], s);
      sm[d] = s * inv_tau;
    }
    __syncthreads();
    normalize_bwd_row(nI + (size_t)r * D, sm, invI[r], d_ibs + (size_t)r * D, D, red);
  } else {
    const int t = r - B;
    for (int d = threadIdx.x; d < D; d += 256) {
      float s = 0.f;
      for (int i = 0; i < B; ++i) s = fmaf(dsim[(size_t)i * T2 + t], nI[(size_t)i * D + d], s);
      sm[d] = s * inv_tau;
    }
    __syncthreads();
    float* dst = t < B ? d_cls1 + (size_t)t * D : d_cls2 + (size_t)(t - B) * D;
    normalize_bwd_row(nT + (size_t)t * D, sm, invT[t], dst, D, red);
  }
}

__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = tot / (float)n;
}

// supervised contrastive, row i: e_ij = exp(z_i . z_j / tau); A = sum_j e_ij pos_ij; Dn = sum_{j != i} e_ij
// stat[i] = (A, Dn, has, loss_i)
__global__ __launch_bounds__(256) void supcon_rows_kernel(const float* __restrict__ z, const long long* __restrict__ labels,
                                                          float* __restrict__ e, float* __restrict__ stat, int B, int D,
                                                          float inv_tau) {
  extern __shared__ float sm[];   // [D]
  __shared__ float red[4];
  const int i = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += 256) sm[d] = z[(size_t)i * D + d];
  __syncthreads();
  const long long li = labels[i];
  float A = 0.f, Dn = 0.f, np = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    const float* t = z + (size_t)j * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(sm[d], t[d], s);
    const float ev = expf(s * inv_tau);
    e[(size_t)i * B + j] = ev;
    if (j != i) {
      Dn += ev;
      if (labels[j] == li) { A += ev; np += 1.f; }
    }
  }
  A = block_sum(A, red);
  Dn = block_sum(Dn, red);
  np = block_sum(np, red);
  if (threadIdx.x == 0) {
    const float has = np > 0.f ? 1.f : 0.f;
    stat[4 * i + 0] = A;
    stat[4 * i + 1] = Dn;
    stat[4 * i + 2] = has;
    stat[4 * i + 3] = -logf(A / (Dn + 1e-8f) + 1e-8f);
  }
}

// loss = mean of loss_i over rows with a positive (0 if none); hdr[0] = 1/n_has (0 if none)
__global__ __launch_bounds__(256) void supcon_finish_kernel(const float* __restrict__ stat, float* __restrict__ loss,
                                                            float* __restrict__ hdr, int B) {
  __shared__ float red[4];
  float n = 0.f, s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    if (stat[4 * i + 2] > 0.f) { n += 1.f; s += stat[4 * i + 3]; }
  }
  n = block_sum(n, red);
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    loss[0] = n > 0.f ? s / n : 0.f;
    hdr[0] = n > 0.f ? 1.0f / n : 0.f;
  }
}

// dL/ds_ij = -(1/n_has) has_i (1/r_i) e_ij (pos_ij / (Dn_i + eps) - A_i / (Dn_i + eps)^2)   for j != i, 0 on the diagonal;
// dz_i = sum_j (ds_ij + ds_ji) z_j / tau ; then the normalisation backward
__global__ __launch_bounds__(256) void supcon_grad_kernel(const float* __restrict__ z, const float* __restrict__ inv,
                                                          const long long* __restrict__ labels, const float* __restrict__ e,
                                                          const float* __restrict__ stat, const float* __restrict__ hdr,
                                                          float* __restrict__ d_ibs, int B, int D, float inv_tau) {
  extern __shared__ float sm[];   // [D] dz, [B] G
  __shared__ float red[4];
  float* dz = sm;
  float* G = sm + D;
  const int i = blockIdx.x;
  const float c = hdr[0];
  const long long li = labels[i];
  const float Ai = stat[4 * i], Di = stat[4 * i + 1] + 1e-8f, hi = stat[4 * i + 2];
  const float ri = Ai / Di + 1e-8f;
  for (int j = threadIdx.x; j < B; j += 256) {
    float g = 0.f;
    if (j != i) {
      const float pos = labels[j] == li ? 1.f : 0.f;
      const float ev = e[(size_t)i * B + j];
      const float Aj = stat[4 * j], Dj = stat[4 * j + 1] + 1e-8f, hj = stat[4 * j + 2];
      const float rj = Aj / Dj + 1e-8f;
      g = -c * ev * (hi / ri * (pos / Di - Ai / (Di * Di)) + hj / rj * (pos / Dj - Aj / (Dj * Dj)));
    }
    G[j] = g;
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = 0.f;
    for (int j = 0; j < B; ++j) s = fmaf(G[j], z[(size_t)j * D + d], s);
    dz[d] = s * inv_tau;
  }
  __syncthreads();
  normalize_bwd_row(z + (size_t)i * D, dz, inv[i], d_ibs + (size_t)i * D, D, red);
}

}  // namespace

extern "C" int eg_aux_symmetry(const float* cls1, const float* cls2, float* loss, float* d_cls1, float* d_cls2, int B, int D,
                               void* stream) {
  EG_CHECK(cls1 && cls2 && loss && d_cls1 && d_cls2 && B > 0 && D > 0, "eg_aux_symmetry: bad arguments");
  hipLaunchKernelGGL(symmetry_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, cls1, cls2, loss, d_cls1, d_cls2,
                     (long long)B * D);
  EG_LAUNCH_CHECK("aux_symmetry");
  return 0;
}

extern "C" int eg_aux_infonce(const float* ibs, const float* cls1, const float* cls2, float temperature, float* loss,
                              float* d_ibs, float* d_cls1, float* d_cls2, float* work, int B, int D, void* stream) {
  EG_CHECK(ibs && cls1 && cls2 && loss && d_ibs && d_cls1 && d_cls2 && work, "eg_aux_infonce: null pointer");
  EG_CHECK(B > 0 && B <= kMaxB && D > 0 && D <= 4096 && temperature > 0.f, "eg_aux_infonce: B=%d (<= %d), D=%d, tau=%f", B, kMaxB, D,
           (double)temperature);
  hipStream_t s = (hipStream_t)stream;
  float* nI = work;
  float* nT = nI + (size_t)B * D;
  float* invI = nT + (size_t)2 * B * D;
  float* invT = invI + B;
  float* dsim = invT + 2 * B;
  float* lossrow = dsim + (size_t)2 * B * B;
  hipLaunchKernelGGL(rownorm_kernel, dim3(B), dim3(256), 0, s, ibs, nI, invI, D);
  hipLaunchKernelGGL(rownorm_kernel, dim3(B), dim3(256), 0, s, cls1, nT, invT, D);
  hipLaunchKernelGGL(rownorm_kernel, dim3(B), dim3(256), 0, s, cls2, nT + (size_t)B * D, invT + B, D);
  const float inv_tau = 1.0f / temperature;
  hipLaunchKernelGGL(infonce_rows_kernel, dim3(B), dim3(256), (size_t)(D + 2 * B) * 4, s, nI, nT, dsim, lossrow, B, D, inv_tau);
  hipLaunchKernelGGL(infonce_grad_kernel, dim3(3 * B), dim3(256), (size_t)D * 4, s, nI, nT, invI, invT, dsim, d_ibs, d_cls1, d_cls2,
                     B, D, inv_tau);
  hipLaunchKernelGGL(mean_rows_kernel, dim3(1), dim3(256), 0, s, lossrow, loss, B);
  EG_LAUNCH_CHECK("aux_infonce");
  return 0;
}

extern "C" int eg_aux_supcon(const float* ibs, const int64_t* labels, float temperature, float* loss, float* d_ibs, float* work,
                             int B, int D, void* stream) {
  EG_CHECK(ibs && labels && loss && d_ibs && work, "eg_aux_supcon: null pointer");
  EG_CHECK(B > 0 && B <= kMaxB && D > 0 && D <= 4096 && temperature > 0.f, "eg_aux_supcon: B=%d (<= %d), D=%d, tau=%f", B, kMaxB, D,
           (double)temperature);
  hipStream_t s = (hipStream_t)stream;
  float* z = work;
  float* inv = z + (size_t)B * D;
  float* e = inv + B;
  float* stat = e + (size_t)B * B;
  float* hdr = stat + 4 * B;
  const float inv_tau = 1.0f / temperature;
  hipLaunchKernelGGL(rownorm_kernel, dim3(B), dim3(256), 0, s, ibs, z, inv, D);
  hipLaunchKernelGGL(supcon_rows_kernel, dim3(B), dim3(256), (size_t)D * 4, s, z, (const long long*)labels, e, stat, B, D, inv_tau);
  hipLaunchKernelGGL(supcon_finish_kernel, dim3(1), dim3(256), 0, s, stat, loss, hdr, B);
  hipLaunchKernelGGL(supcon_grad_kernel, dim3(B), dim3(256), (size_t)(D + B) * 4, s, z, inv, (const long long*)labels, e, stat, hdr,
                     d_ibs, B, D, inv_tau);
  EG_LAUNCH_CHECK("aux_supcon");
  return 0;
}
