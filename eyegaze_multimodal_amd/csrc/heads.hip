// Small per-sample kernels around the encoder: sequence assembly (CLS / shared tokens), pooling +
// symmetric fusion operands, the final class projection fused with cross-entropy, and their backward forms.
// These touch [B, d]-sized data; they are latency-bound, so each is a single short launch.
#include "common.h"

namespace {

// dst[b, off + r, :] = src[(b % src_nb), r, :] (+ pos[off + r, :])      src fp32 (parameters such as cls_token)
template <typename T>
__global__ void rows_bcast_f32_kernel(const float* __restrict__ src, const float* __restrict__ pos, T* __restrict__ dst,
                                      int NB, int S, int D, int R, int off, int src_nb) {
  const int b = blockIdx.y, r = blockIdx.x;
  const float* s = src + ((size_t)(b % src_nb) * R + r) * D;
  T* d = dst + ((size_t)b * S + off + r) * D;
  for (int n = threadIdx.x; n < D; n += blockDim.x) {
    float v = s[n];
    if (pos) v += pos[(size_t)(off + r) * D + n];
    Elem<T>::st(d + n, v);
  }
}

// dst[b_dst0 + b, off + r, :] = src[b_src0 + b, off + r, :]   (copy the shared synchrony tokens to stream 2)
template <typename T>
__global__ void rows_copy_kernel(T* __restrict__ seq, int S, int D, int R, int off, int b_src0, int b_dst0) {
  const int b = blockIdx.y, r = blockIdx.x;
  const T* s = seq + ((size_t)(b_src0 + b) * S + off + r) * D;
  T* d = seq + ((size_t)(b_dst0 + b) * S + off + r) * D;
  for (int n = threadIdx.x; n < D; n += blockDim.x) d[n] = s[n];
}

// pooling + fusion operands (D:1193-1212, D:933-938, D:1222-1223)
//   z [2B, S, D]; stream 1 = samples [0,B), stream 2 = [B,2B)
//   cls1/cls2 fp32 [B,D]; comb [B,3D] = [a+b, a*b, |a-b|]; zf[:, D:2D] = mean_{s>=off} z1, zf[:, 2D:3D] = same for z2
//   ibs_pool fp32+T [B,D] = mean_{1<=s<1+n_ibs} z1   (n_ibs may be 0)
template <typename T>
__global__ void pool_fuse_fwd_kernel(const T* __restrict__ z, float* __restrict__ cls1, float* __restrict__ cls2,
                                     T* __restrict__ comb, T* __restrict__ zf, float* __restrict__ ibs_pool_f,
                                     T* __restrict__ ibs_pool, int B, int S, int D, int off, int n_ibs, int ibs_first) {
  const int b = blockIdx.x;
  const T* z1 = z + (size_t)b * S * D;
  const T* z2 = z + (size_t)(b + B) * S * D;
  for (int n = threadIdx.x; n < D; n += blockDim.x) {
    const float a = Elem<T>::ld(z1 + n), c = Elem<T>::ld(z2 + n);
    cls1[(size_t)b * D + n] = a;
    cls2[(size_t)b * D + n] = c;
    Elem<T>::st(comb + (size_t)b * 3 * D + n, a + c);
    Elem<T>::st(comb + (size_t)b * 3 * D + D + n, a * c);
    Elem<T>::st(comb + (size_t)b * 3 * D + 2 * D + n, fabsf(a - c));
    float m1 = 0.f, m2 = 0.f;
#pragma unroll 8
    for (int s = off; s < S; ++s) {          // unrolled: 16 independent 2-byte loads in flight instead of 2
      m1 += Elem<T>::ld(z1 + (size_t)s * D + n);
      m2 += Elem<T>::ld(z2 + (size_t)s * D + n);
    }
    const float inv = 1.0f / (float)(S - off);
    Elem<T>::st(zf + (size_t)b * 3 * D + D + n, m1 * inv);
    Elem<T>::st(zf + (size_t)b * 3 * D + 2 * D + n, m2 * inv);
    if (n_ibs > 0) {
      float mi = 0.f;
      for (int s = ibs_first; s < ibs_first + n_ibs; ++s) mi += Elem<T>::ld(z1 + (size_t)s * D + n);
      mi /= (float)n_ibs;
      ibs_pool_f[(size_t)b * D + n] = mi;
      Elem<T>::st(ibs_pool + (size_t)b * D + n, mi);
    }
  }
}

// backward of the above: writes the full dz [2B, S, D] (zeros where nothing flows)
template <typename T>
__global__ void pool_fuse_bwd_kernel(const T* __restrict__ z, const T* __restrict__ dcomb, const T* __restrict__ dzf,
                                     const float* __restrict__ gcls1, const float* __restrict__ gcls2,
                                     const T* __restrict__ dibs_pool, const float* __restrict__ gibs_pool,
                                     T* __restrict__ dz, int B, int S, int D, int off, int n_ibs, int ibs_first) {
  const int b = blockIdx.x;
  const T* z1 = z + (size_t)b * S * D;
  const T* z2 = z + (size_t)(b + B) * S * D;
  T* d1 = dz + (size_t)b * S * D;
  T* d2 = dz + (size_t)(b + B) * S * D;
  const float invp = 1.0f / (float)(S - off);
  for (int n = threadIdx.x; n < D; n += blockDim.x) {
    const float a = Elem<T>::ld(z1 + n), c = Elem<T>::ld(z2 + n);
    const float g0 = Elem<T>::ld(dcomb + (size_t)b * 3 * D + n);
    const float g1 = Elem<T>::ld(dcomb + (size_t)b * 3 * D + D + n);
    const float g2 = Elem<T>::ld(dcomb + (size_t)b * 3 * D + 2 * D + n);
    const float sg = (a > c) ? 1.f : ((a < c) ? -1.f : 0.f);
    float da = g0 + g1 * c + g2 * sg;
    float dc = g0 + g1 * a - g2 * sg;
    if (gcls1) da += gcls1[(size_t)b * D + n];
    if (gcls2) dc += gcls2[(size_t)b * D + n];
    const float dm1 = Elem<T>::ld(dzf + (size_t)b * 3 * D + D + n) * invp;
    const float dm2 = Elem<T>::ld(dzf + (size_t)b * 3 * D + 2 * D + n) * invp;
    float di = 0.f;
    if (n_ibs > 0) {
      if (dibs_pool) di += Elem<T>::ld(dibs_pool + (size_t)b * D + n);
      if (gibs_pool) di += gibs_pool[(size_t)b * D + n];
      di /= (float)n_ibs;
    }
    for (int s = 0; s < S; ++s) {
      float v1 = 0.f, v2 = 0.f;
      if (s == 0) { v1 = da; v2 = dc; }
      if (s >= off) { v1 += dm1; v2 += dm2; }
      if (n_ibs > 0 && s >= ibs_first && s < ibs_first + n_ibs) v1 += di;
      Elem<T>::st(d1 + (size_t)s * D + n, v1);
      Elem<T>::st(d2 + (size_t)s * D + n, v2);
    }
  }
}

// logits = h W^T + b  (ncls <= 16), per-sample CE; one wave per sample (D:1104-1105 / 1078 + D:1244 / 1250)
template <typename T>
__global__ __launch_bounds__(256) void classifier_ce_fwd_kernel(const T* __restrict__ h, const float* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                const long long* __restrict__ labels,
                                                                float* __restrict__ logits, float* __restrict__ sample_loss,
                                                                int B, int K, int ncls) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float lg[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) lg[c] = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float x = Elem<T>::ld(h + (size_t)b * K + k);
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < ncls) lg[c] += x * W[(size_t)c * K + k];
  }
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 16; ++c)
    if (c < ncls) {
      lg[c] = wave_sum(lg[c]) + bias[c];
      mx = fmaxf(mx, lg[c]);
    }
  if (lane == 0) {
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < ncls) {
        logits[(size_t)b * ncls + c] = lg[c];
        se += expf(lg[c] - mx);
      }
    if (labels && sample_loss) {
      const int y = (int)labels[b];
      float ly = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c == y) ly = lg[c];
      sample_loss[b] = (mx + logf(se)) - ly;
    }
  }
}

__global__ void mean_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0] / (float)n;
}

// dlogits[b,c] = gloss * (softmax - onehot)/B + glogits[b,c];  dh[b,k] = (sum_c dlogits[b,c] W[c,k]) * gate(h>0)*gate_scale
template <typename T>
__global__ __launch_bounds__(256) void classifier_ce_bwd_kernel(const T* __restrict__ h, const float* __restrict__ W,
                                                                const float* __restrict__ logits,
                                                                const long long* __restrict__ labels,
                                                                const float* __restrict__ gloss,
                                                                const float* __restrict__ glogits,
                                                                float* __restrict__ dlogits, T* __restrict__ dh, int B,
                                                                int K, int ncls, int use_gate, float gate_scale) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float dl[16];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    dl[c] = 0.f;
    if (c < ncls) mx = fmaxf(mx, logits[(size_t)b * ncls + c]);
  }
  float se = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c)
    if (c < ncls) {
      dl[c] = expf(logits[(size_t)b * ncls + c] - mx);
      se += dl[c];
    }
  const float gl = (gloss && labels) ? *gloss / (float)B : 0.f;
  const int y = labels ? (int)labels[b] : -1;
#pragma unroll
  for (int c = 0; c < 16; ++c)
    if (c < ncls) {
      float d = gl * (dl[c] / se - (c == y ? 1.f : 0.f));
      if (glogits) d += glogits[(size_t)b * ncls + c];
      dl[c] = d;
      if (lane == 0) dlogits[(size_t)b * ncls + c] = d;
    }
  for (int k = lane; k < K; k += 64) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < ncls) s += dl[c] * W[(size_t)c * K + k];
    if (use_gate) s = Elem<T>::ld(h + (size_t)b * K + k) > 0.f ? s * gate_scale : 0.f;
    Elem<T>::st(dh + (size_t)b * K + k, s);
  }
}

// dW[c,k] = sum_b dlogits[b,c] h[b,k];  db[c] = sum_b dlogits[b,c]      grid = (ncls, ceil(K/64)); 64 columns x 4 batch lanes
template <typename T>
__global__ __launch_bounds__(256) void classifier_wgrad_kernel(const T* __restrict__ h, const float* __restrict__ dlogits,
                                                               float* __restrict__ dW, float* __restrict__ db, int B, int K,
                                                               int ncls) {
  __shared__ float red[4][64], redb[4];
  const int c = blockIdx.x, k = blockIdx.y * 64 + (threadIdx.x & 63), bl = threadIdx.x >> 6;
  float s = 0.f, sb = 0.f;
#pragma unroll 8
  for (int b = bl; b < B; b += 4) {
    const float dl = dlogits[(size_t)b * ncls + c];
    if (k < K) s = fmaf(dl, Elem<T>::ld(h + (size_t)b * K + k), s);
    sb += dl;
  }
  red[bl][threadIdx.x & 63] = s;
  if ((threadIdx.x & 63) == 0) redb[bl] = sb;
  __syncthreads();
  if (bl == 0) {
    const int kk = threadIdx.x & 63;
    if (k < K) dW[(size_t)c * K + k] = red[0][kk] + red[1][kk] + red[2][kk] + red[3][kk];
    if (blockIdx.y == 0 && kk == 0) db[c] = redb[0] + redb[1] + redb[2] + redb[3];
  }
}

// out[s, :] = sum_b dseq[b, s, :]  (position-embedding gradient; row 0 is also the cls_token gradient)
// grid = (rows, ceil(D/64)); 64 columns x 4 batch lanes per block
template <typename T>
__global__ __launch_bounds__(256) void batch_rowsum_kernel(const T* __restrict__ dseq, float* __restrict__ out, int NB, int S,
                                                           int D) {
  __shared__ float red[4][64];
  const int s = blockIdx.x, n = blockIdx.y * 64 + (threadIdx.x & 63), bl = threadIdx.x >> 6;
  float a = 0.f;
  if (n < D) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // four independent loads in flight per thread
    int b = bl;
    for (; b + 12 < NB; b += 16) {
      a0 += Elem<T>::ld(dseq + ((size_t)b * S + s) * D + n);
      a1 += Elem<T>::ld(dseq + ((size_t)(b + 4) * S + s) * D + n);
      a2 += Elem<T>::ld(dseq + ((size_t)(b + 8) * S + s) * D + n);
      a3 += Elem<T>::ld(dseq + ((size_t)(b + 12) * S + s) * D + n);
    }
    for (; b < NB; b += 4) a0 += Elem<T>::ld(dseq + ((size_t)b * S + s) * D + n);
    a = (a0 + a1) + (a2 + a3);
  }
  red[bl][threadIdx.x & 63] = a;
  __syncthreads();
  if (bl == 0 && n < D) {
    const int c = threadIdx.x & 63;
    out[(size_t)s * D + n] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
  }
}

// dst[b, r, :] = (src[b, off+r, :] (+ src[b+B2, off+r, :])) * (gate[b, r, :] > 0 ? gate_scale : 0)
//   dst rows are addressed by a rowmap (e.g. the zero-padded dY buffer of the strided-conv backward)
template <typename T>
__global__ void rows_gather_gate_kernel(const T* __restrict__ src, const T* __restrict__ gate, T* __restrict__ dst,
                                        RowMap dmap, int S, int D, int R, int off, int pair_shift, float gate_scale) {
  const int b = blockIdx.y, r = blockIdx.x;
  const T* s = src + ((size_t)b * S + off + r) * D;
  const T* s2 = pair_shift ? src + ((size_t)(b + pair_shift) * S + off + r) * D : nullptr;
  const T* gt = gate ? gate + ((size_t)b * R + r) * D : nullptr;
  T* d = dst + row_off(dmap, b * R + r);
  for (int n = threadIdx.x * 4; n < D; n += blockDim.x * 4) {
    float v[4];
    load4(s + n, v);
    if (s2) {
      float w[4];
      load4(s2 + n, w);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += w[e];
    }
    if (gt) {
      float gv[4];
      load4(gt + n, gv);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gv[e] > 0.f ? v[e] * gate_scale : 0.f;
    }
    store4(d + n, v);
  }
}

// FuzzyGatingFusion.forward (3_Models/fusion/fuzzy_gating_fusion.py:297-390), one thread per sample, K <= 16 classes.
// prm = [tau_img, tau_eeg, c_unrel_img, c_unrel_eeg, ls_rel_img, ls_rel_eeg, ls_unrel_img, ls_unrel_eeg, beta0..3]
// mode: 0 full, 1 no_temperature, 2 no_fuzzification, 3 fixed_weights
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float entropy_k(const float* z, int K, float eps_log) {
  float mx = -INFINITY, se = 0.f, h = 0.f;
  for (int c = 0; c < K; ++c) mx = fmaxf(mx, z[c]);
  for (int c = 0; c < K; ++c) se += expf(z[c] - mx);
  for (int c = 0; c < K; ++c) {
    const float p = expf(z[c] - mx) / se;
    h -= p * logf(p + eps_log);
  }
  return h;
}
__global__ void fuzzy_gate_fwd_kernel(const float* __restrict__ zi, const float* __restrict__ ze, const float* __restrict__ prm,
                                      float* __restrict__ fused, float* __restrict__ alpha_out, int B, int K, int mode,
                                      float eps_temp, float eps_log, float eps_div) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float a[16], e[16];
  float ti = 1.f, te = 1.f;
  if (mode == 0 || mode == 2) { ti = softplus_f(prm[0]) + eps_temp; te = softplus_f(prm[1]) + eps_temp; }
  for (int c = 0; c < K; ++c) { a[c] = zi[(size_t)b * K + c] / ti; e[c] = ze[(size_t)b * K + c] / te; }
  const float Hi = entropy_k(a, K, eps_log), He = entropy_k(e, K, eps_log);
  float al;
  if (mode == 3) {
    al = 0.5f;
  } else if (mode == 2) {
    const float hmax = logf((float)K);
    const float ci = fmaxf(1.0f - Hi / (hmax + eps_div), 0.f), ce = fmaxf(1.0f - He / (hmax + eps_div), 0.f);
    al = fminf(fmaxf(ci / (ci + ce + eps_div), 0.f), 1.f);
  } else {
    auto mu = [&](float x, float c, float ls) { const float s = expf(ls); return expf(-((x - c) * (x - c)) / (2.f * s * s + eps_div)); };
    const float ir = mu(Hi, 0.f, prm[4]), iu = mu(Hi, prm[2], prm[6]);
    const float er = mu(He, 0.f, prm[5]), eu = mu(He, prm[3], prm[7]);
    const float w[4] = {ir * eu, iu * er, ir * er, iu * eu};
    float num = 0.f, den = 0.f;
    for (int k = 0; k < 4; ++k) { num += w[k] / (1.f + expf(-prm[8 + k])); den += w[k]; }
    al = fminf(fmaxf(num / (den + eps_div), 0.f), 1.f);
  }
  alpha_out[b] = al;
  for (int c = 0; c < K; ++c) fused[(size_t)b * K + c] = al * a[c] + (1.f - al) * e[c];
}

// FuzzyGatingFusion backward: given d fused [B,K] and (optionally) d alpha [B], the gradients of both logit sets and of the
// 12 scalar parameters.  One thread per sample recomputes the forward (12 scalars, K <= 16 logits) and applies the chain rule
// by hand; parameter gradients are reduced over the block into partial[blockIdx][12] (summed in block order by the caller).
//   clamp(): torch passes the gradient where min <= x <= max (inclusive), as does this kernel.
__device__ __forceinline__ void entropy_bwd_k(const float* z, int K, float eps_log, float dH, float* dz) {
  // H = -sum p log(p + eps); dH/dp_c = -(log(p_c + eps) + p_c / (p_c + eps)); soft-max: dz_j = p_j (g_j - sum_c p_c g_c)
  float mx = -INFINITY, se = 0.f;
  for (int c = 0; c < K; ++c) mx = fmaxf(mx, z[c]);
  for (int c = 0; c < K; ++c) se += expf(z[c] - mx);
  float pg = 0.f;
  for (int c = 0; c < K; ++c) {
    const float p = expf(z[c] - mx) / se;
    pg += p * (-(logf(p + eps_log) + p / (p + eps_log)));
  }
  for (int c = 0; c < K; ++c) {
    const float p = expf(z[c] - mx) / se;
    const float g = -(logf(p + eps_log) + p / (p + eps_log));
    dz[c] += dH * p * (g - pg);
  }
}

__global__ __launch_bounds__(128) void fuzzy_gate_bwd_kernel(const float* __restrict__ zi, const float* __restrict__ ze,
                                                             const float* __restrict__ prm, const float* __restrict__ dfused,
                                                             const float* __restrict__ dalpha, float* __restrict__ dzi,
                                                             float* __restrict__ dze, float* __restrict__ partial, int B, int K,
                                                             int mode, float eps_temp, float eps_log, float eps_div) {
  __shared__ float red[2][12];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  float dp[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) dp[i] = 0.f;
  if (b < B) {
    float a[16], e[16], da[16], de[16];
    float ti = 1.f, te = 1.f;
    const bool temp = (mode == 0 || mode == 2);
    if (temp) { ti = softplus_f(prm[0]) + eps_temp; te = softplus_f(prm[1]) + eps_temp; }
    for (int c = 0; c < K; ++c) { a[c] = zi[(size_t)b * K + c] / ti; e[c] = ze[(size_t)b * K + c] / te; }
    const float Hi = entropy_k(a, K, eps_log), He = entropy_k(e, K, eps_log);
    // ---- forward pieces needed again ----
    float al_pre = 0.5f;
    float ir = 0.f, iu = 0.f, er = 0.f, eu = 0.f, num = 0.f, den = 0.f, th[4] = {0.f, 0.f, 0.f, 0.f};
    float ci = 0.f, ce = 0.f, ci_raw = 0.f, ce_raw = 0.f;
    auto D_of = [&](float ls) { const float s = expf(ls); return 2.f * s * s + eps_div; };
    if (mode == 2) {
      const float hmax = logf((float)K);
      ci_raw = 1.0f - Hi / (hmax + eps_div); ce_raw = 1.0f - He / (hmax + eps_div);
      ci = fmaxf(ci_raw, 0.f); ce = fmaxf(ce_raw, 0.f);
      al_pre = ci / (ci + ce + eps_div);
    } else if (mode != 3) {
      ir = expf(-(Hi * Hi) / D_of(prm[4]));
      iu = expf(-((Hi - prm[2]) * (Hi - prm[2])) / D_of(prm[6]));
      er = expf(-(He * He) / D_of(prm[5]));
      eu = expf(-((He - prm[3]) * (He - prm[3])) / D_of(prm[7]));
      const float w[4] = {ir * eu, iu * er, ir * er, iu * eu};
      for (int k = 0; k < 4; ++k) { th[k] = 1.f / (1.f + expf(-prm[8 + k])); num += w[k] * th[k]; den += w[k]; }
      al_pre = num / (den + eps_div);
    }
    const float al = fminf(fmaxf(al_pre, 0.f), 1.f);
    // ---- fusion: fused = al * a + (1 - al) * e ----
    float dal = dalpha ? dalpha[b] : 0.f;
    for (int c = 0; c < K; ++c) {
      const float g = dfused[(size_t)b * K + c];
      da[c] = al * g;
      de[c] = (1.f - al) * g;
      dal += g * (a[c] - e[c]);
    }
    if (!(al_pre >= 0.f && al_pre <= 1.f)) dal = 0.f;
    float dHi = 0.f, dHe = 0.f;
    if (mode == 2) {
      const float s = ci + ce + eps_div, hmax = logf((float)K);
      const float dci = dal * (ce + eps_div) / (s * s), dce = -dal * ci / (s * s);
      if (ci_raw >= 0.f) dHi = -dci / (hmax + eps_div);
      if (ce_raw >= 0.f) dHe = -dce / (hmax + eps_div);
    } else if (mode != 3) {
      const float dd = den + eps_div;
      const float dnum = dal / dd, dden = -dal * num / (dd * dd);
      const float w[4] = {ir * eu, iu * er, ir * er, iu * eu};
      float dw[4];
      for (int k = 0; k < 4; ++k) {
        dw[k] = dnum * th[k] + dden;
        dp[8 + k] = dnum * w[k] * th[k] * (1.f - th[k]);
      }
      const float dir = dw[0] * eu + dw[2] * er, deu = dw[0] * ir + dw[3] * iu;
      const float diu = dw[1] * er + dw[3] * eu, der = dw[1] * iu + dw[2] * ir;
      // mu(x; c, ls) = exp(-(x-c)^2 / D), D = 2 exp(2 ls) + eps:  d/dx = mu * (-2 (x-c) / D), d/dc = -d/dx,
      //                                                          d/dls = mu * (x-c)^2 / D^2 * 4 exp(2 ls)
      auto mu_bwd = [&](float x, float c, float ls, float m, float dm, float& dx, float& dc, float& dls) {
        const float D = D_of(ls), s2 = expf(2.f * ls), t = x - c;
        const float gx = dm * m * (-2.f * t / D);
        dx += gx;
        dc -= gx;
        dls += dm * m * (t * t) / (D * D) * 4.f * s2;
      };
      float dummy = 0.f;
      mu_bwd(Hi, 0.f, prm[4], ir, dir, dHi, dummy, dp[4]);
      mu_bwd(Hi, prm[2], prm[6], iu, diu, dHi, dp[2], dp[6]);
      mu_bwd(He, 0.f, prm[5], er, der, dHe, dummy, dp[5]);
      mu_bwd(He, prm[3], prm[7], eu, deu, dHe, dp[3], dp[7]);
    }
    if (mode != 3) {
      entropy_bwd_k(a, K, eps_log, dHi, da);
      entropy_bwd_k(e, K, eps_log, dHe, de);
    }
    // ---- temperature: a = z / T, T = softplus(tau) + eps ----
    float dTi = 0.f, dTe = 0.f;
    for (int c = 0; c < K; ++c) {
      dzi[(size_t)b * K + c] = da[c] / ti;
      dze[(size_t)b * K + c] = de[c] / te;
      dTi -= da[c] * a[c] / ti;
      dTe -= de[c] * e[c] / te;
    }
    if (temp) {
      dp[0] = dTi / (1.f + expf(-prm[0]));   // softplus' = sigmoid
      dp[1] = dTe / (1.f + expf(-prm[1]));
    }
  }
  // block reduction of the 12 parameter gradients (2 waves of 64)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float v = wave_sum(dp[i]);
    if (lane == 0) red[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 12) partial[(size_t)blockIdx.x * 12 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x];
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F16, CALL_F32, who) \
  if ((dtype) == EG_BF16) { CALL_BF16; }                      \
  else if ((dtype) == EG_F16) { CALL_F16; }                   \
  else if ((dtype) == EG_F32) { CALL_F32; }                   \
  else return eg_fail("%s: bad dtype %d", who, (int)(dtype));

extern "C" int eg_rows_bcast_f32(const float* src, const float* pos, void* seq, int NB, int S, int D, int R, int off,
                                 int src_nb, int dtype, void* stream) {
  EG_CHECK(src && seq && NB > 0 && R > 0 && off >= 0 && off + R <= S && src_nb > 0, "eg_rows_bcast_f32: bad arguments");
  dim3 grid(R, NB);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(rows_bcast_f32_kernel<bf16_t>, grid, dim3(256), 0, s, src, pos, (bf16_t*)seq, NB, S, D, R, off, src_nb),
             hipLaunchKernelGGL(rows_bcast_f32_kernel<f16_t>, grid, dim3(256), 0, s, src, pos, (f16_t*)seq, NB, S, D, R, off, src_nb),
             hipLaunchKernelGGL(rows_bcast_f32_kernel<float>, grid, dim3(256), 0, s, src, pos, (float*)seq, NB, S, D, R, off, src_nb),
             "eg_rows_bcast_f32");
  EG_LAUNCH_CHECK("rows_bcast_f32");
  return 0;
}

extern "C" int eg_rows_copy(void* seq, int S, int D, int R, int off, int b_src0, int b_dst0, int nb, int dtype,
                            void* stream) {
  EG_CHECK(seq && nb > 0 && R > 0 && off >= 0 && off + R <= S, "eg_rows_copy: bad arguments");
  dim3 grid(R, nb);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(rows_copy_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)seq, S, D, R, off, b_src0, b_dst0),
             hipLaunchKernelGGL(rows_copy_kernel<f16_t>, grid, dim3(256), 0, s, (f16_t*)seq, S, D, R, off, b_src0, b_dst0),
             hipLaunchKernelGGL(rows_copy_kernel<float>, grid, dim3(256), 0, s, (float*)seq, S, D, R, off, b_src0, b_dst0),
             "eg_rows_copy");
  EG_LAUNCH_CHECK("rows_copy");
  return 0;
}

extern "C" int eg_pool_fuse_fwd(const void* z, float* cls1, float* cls2, void* comb, void* zf, float* ibs_pool_f,
                                void* ibs_pool, int B, int S, int D, int off, int n_ibs, int ibs_first, int dtype,
                                void* stream) {
  EG_CHECK(z && cls1 && cls2 && comb && zf, "eg_pool_fuse_fwd: null pointer");
  EG_CHECK(B > 0 && off > 0 && off < S, "eg_pool_fuse_fwd: bad shape");
  EG_CHECK(n_ibs == 0 || (ibs_pool_f && ibs_pool && ibs_first >= 1 && ibs_first + n_ibs <= S), "eg_pool_fuse_fwd: ibs range");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pool_fuse_fwd_kernel<bf16_t>, dim3(B), dim3(256), 0, s, (const bf16_t*)z, cls1, cls2, (bf16_t*)comb, (bf16_t*)zf, ibs_pool_f, (bf16_t*)ibs_pool, B, S, D, off, n_ibs, ibs_first),
             hipLaunchKernelGGL(pool_fuse_fwd_kernel<f16_t>, dim3(B), dim3(256), 0, s, (const f16_t*)z, cls1, cls2, (f16_t*)comb, (f16_t*)zf, ibs_pool_f, (f16_t*)ibs_pool, B, S, D, off, n_ibs, ibs_first),
             hipLaunchKernelGGL(pool_fuse_fwd_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)z, cls1, cls2, (float*)comb, (float*)zf, ibs_pool_f, (float*)ibs_pool, B, S, D, off, n_ibs, ibs_first),
             "eg_pool_fuse_fwd");
  EG_LAUNCH_CHECK("pool_fuse_fwd");
  return 0;
}

extern "C" int eg_pool_fuse_bwd(const void* z, const void* dcomb, const void* dzf, const float* gcls1, const float* gcls2,
                                const void* dibs_pool, const float* gibs_pool, void* dz, int B, int S, int D, int off,
                                int n_ibs, int ibs_first, int dtype, void* stream) {
  EG_CHECK(z && dcomb && dzf && dz, "eg_pool_fuse_bwd: null pointer");
  EG_CHECK(B > 0 && off > 0 && off < S, "eg_pool_fuse_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pool_fuse_bwd_kernel<bf16_t>, dim3(B), dim3(256), 0, s, (const bf16_t*)z, (const bf16_t*)dcomb, (const bf16_t*)dzf, gcls1, gcls2, (const bf16_t*)dibs_pool, gibs_pool, (bf16_t*)dz, B, S, D, off, n_ibs, ibs_first),
             hipLaunchKernelGGL(pool_fuse_bwd_kernel<f16_t>, dim3(B), dim3(256), 0, s, (const f16_t*)z, (const f16_t*)dcomb, (const f16_t*)dzf, gcls1, gcls2, (const f16_t*)dibs_pool, gibs_pool, (f16_t*)dz, B, S, D, off, n_ibs, ibs_first),
             hipLaunchKernelGGL(pool_fuse_bwd_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)z, (const float*)dcomb, (const float*)dzf, gcls1, gcls2, (const float*)dibs_pool, gibs_pool, (float*)dz, B, S, D, off, n_ibs, ibs_first),
             "eg_pool_fuse_bwd");
  EG_LAUNCH_CHECK("pool_fuse_bwd");
  return 0;
}

extern "C" int eg_classifier_ce_fwd(const void* h, const float* W, const float* bias, const int64_t* labels,
                                    float* logits, float* sample_loss, float* loss, int B, int K, int ncls, int dtype,
                                    void* stream) {
  EG_CHECK(h && W && bias && logits, "eg_classifier_ce_fwd: null pointer");
  EG_CHECK(B > 0 && K > 0 && ncls > 0 && ncls <= 16, "eg_classifier_ce_fwd: ncls=%d must be in [1,16]", ncls);
  EG_CHECK(!labels || (sample_loss && loss), "eg_classifier_ce_fwd: labels need sample_loss and loss outputs");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((B + 3) / 4);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(classifier_ce_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)h, W, bias, (const long long*)labels, logits, sample_loss, B, K, ncls),
             hipLaunchKernelGGL(classifier_ce_fwd_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)h, W, bias, (const long long*)labels, logits, sample_loss, B, K, ncls),
             hipLaunchKernelGGL(classifier_ce_fwd_kernel<float>, grid, dim3(256), 0, s, (const float*)h, W, bias, (const long long*)labels, logits, sample_loss, B, K, ncls),
             "eg_classifier_ce_fwd");
  if (labels) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, sample_loss, loss, B);
  EG_LAUNCH_CHECK("classifier_ce_fwd");
  return 0;
}

extern "C" int eg_classifier_ce_bwd(const void* h, const float* W, const float* logits, const int64_t* labels,
                                    const float* gloss, const float* glogits, float* dlogits, void* dh, float* dW,
                                    float* db, int B, int K, int ncls, int use_gate, float gate_scale, int dtype,
                                    void* stream) {
  EG_CHECK(h && W && logits && dlogits && dh && dW && db, "eg_classifier_ce_bwd: null pointer");
  EG_CHECK(B > 0 && K > 0 && ncls > 0 && ncls <= 16, "eg_classifier_ce_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((B + 3) / 4);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(classifier_ce_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)h, W, logits, (const long long*)labels, gloss, glogits, dlogits, (bf16_t*)dh, B, K, ncls, use_gate, gate_scale);
             hipLaunchKernelGGL(classifier_wgrad_kernel<bf16_t>, dim3(ncls, (K + 63) / 64), dim3(256), 0, s, (const bf16_t*)h, dlogits, dW, db, B, K, ncls),
             hipLaunchKernelGGL(classifier_ce_bwd_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)h, W, logits, (const long long*)labels, gloss, glogits, dlogits, (f16_t*)dh, B, K, ncls, use_gate, gate_scale);
             hipLaunchKernelGGL(classifier_wgrad_kernel<f16_t>, dim3(ncls, (K + 63) / 64), dim3(256), 0, s, (const f16_t*)h, dlogits, dW, db, B, K, ncls),
             hipLaunchKernelGGL(classifier_ce_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)h, W, logits, (const long long*)labels, gloss, glogits, dlogits, (float*)dh, B, K, ncls, use_gate, gate_scale);
             hipLaunchKernelGGL(classifier_wgrad_kernel<float>, dim3(ncls, (K + 63) / 64), dim3(256), 0, s, (const float*)h, dlogits, dW, db, B, K, ncls),
             "eg_classifier_ce_bwd");
  EG_LAUNCH_CHECK("classifier_ce_bwd");
  return 0;
}

extern "C" int eg_batch_rowsum(const void* dseq, float* out, int NB, int S, int D, int rows, int dtype, void* stream) {
  EG_CHECK(dseq && out && NB > 0 && rows > 0 && rows <= S, "eg_batch_rowsum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(batch_rowsum_kernel<bf16_t>, dim3(rows, (D + 63) / 64), dim3(256), 0, s, (const bf16_t*)dseq, out, NB, S, D),
             hipLaunchKernelGGL(batch_rowsum_kernel<f16_t>, dim3(rows, (D + 63) / 64), dim3(256), 0, s, (const f16_t*)dseq, out, NB, S, D),
             hipLaunchKernelGGL(batch_rowsum_kernel<float>, dim3(rows, (D + 63) / 64), dim3(256), 0, s, (const float*)dseq, out, NB, S, D),
             "eg_batch_rowsum");
  EG_LAUNCH_CHECK("batch_rowsum");
  return 0;
}

extern "C" int eg_rows_gather_gate(const void* src, const void* gate, void* dst, eg_rowmap dmap, int nb, int S, int D,
                                   int R, int off, int pair_shift, float gate_scale, int dtype, void* stream) {
  EG_CHECK(src && dst && nb > 0 && R > 0 && off >= 0 && off + R <= S && D % 4 == 0, "eg_rows_gather_gate: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(R, nb);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(rows_gather_gate_kernel<bf16_t>, grid, dim3(64), 0, s, (const bf16_t*)src, (const bf16_t*)gate, (bf16_t*)dst, to_rowmap(dmap), S, D, R, off, pair_shift, gate_scale),
             hipLaunchKernelGGL(rows_gather_gate_kernel<f16_t>, grid, dim3(64), 0, s, (const f16_t*)src, (const f16_t*)gate, (f16_t*)dst, to_rowmap(dmap), S, D, R, off, pair_shift, gate_scale),
             hipLaunchKernelGGL(rows_gather_gate_kernel<float>, grid, dim3(64), 0, s, (const float*)src, (const float*)gate, (float*)dst, to_rowmap(dmap), S, D, R, off, pair_shift, gate_scale),
             "eg_rows_gather_gate");
  EG_LAUNCH_CHECK("rows_gather_gate");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// The loss of the multimodal logit-fusion step and its gradients on the [B, K] logits, in ONE single-workgroup launch
// (train_multimodal_fuzzy_fusion.py:436-460): L = CE(fused) + l_img CE(z_img / T_img) + l_eeg CE(z_eeg / T_eeg) + l_reg R(T),
// temperatures DETACHED in the auxiliary terms (fuzzy_gating_fusion.py:334), R = relu(T - t_max) + relu(t_min - T) over both
// temperatures (:392-419), T = softplus(tau) + eps_temp.  As torch autograd on 2 x [B, 3] tensors this was ~70 tiny launches with
// 10-40 us of host time between them: 0.8 ms of idle GPU per step of BASELINE configs[4].
//   losses[5] = total, ce, aux_img, aux_eeg, reg
//   dfused    = dL/dfused  (feed eg_fuzzy_gate_bwd);  daux_img / daux_eeg = the auxiliary terms' direct gradients on the logits
//   dtau[2]   = l_reg dR/dtau_img, l_reg dR/dtau_eeg
// every gradient is multiplied by the loss scale of `state` when its scaler is on (GradScaler.scale(loss).backward(), :462).
// Summation over the batch in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fusion_loop_loss_kernel(const float* __restrict__ fused, const float* __restrict__ zi,
                                                               const float* __restrict__ ze, const long long* __restrict__ labels,
                                                               const float* __restrict__ prm, float* __restrict__ losses,
                                                               float* __restrict__ dfused, float* __restrict__ dai,
                                                               float* __restrict__ dae, float* __restrict__ dtau, int B, int K,
                                                               int mode, float eps_temp, float l_img, float l_eeg, float l_reg,
                                                               float t_min, float t_max, const eg_step_state* st) {
  __shared__ float red[3][256];
  const float scale = (st && st->scaler_on) ? st->loss_scale : 1.0f;
  float ti = 1.f, te = 1.f;
  if (mode == 0 || mode == 2) { ti = softplus_f(prm[0]) + eps_temp; te = softplus_f(prm[1]) + eps_temp; }
  const float invB = 1.0f / (float)B;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < B; b += 256) {
    const int y = (int)labels[b];
    const float* rows[3] = {fused + (size_t)b * K, zi + (size_t)b * K, ze + (size_t)b * K};
    float* outs[3] = {dfused + (size_t)b * K, dai + (size_t)b * K, dae + (size_t)b * K};
    const float tdiv[3] = {1.f, ti, te};
    const float wgt[3] = {1.f, l_img, l_eeg};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      float z[16], mx = -INFINITY, se = 0.f;
      for (int c = 0; c < K; ++c) { z[c] = rows[q][c] / tdiv[q]; mx = fmaxf(mx, z[c]); }
      for (int c = 0; c < K; ++c) se += expf(z[c] - mx);
      const float lse = mx + logf(se);
      acc[q] += lse - z[y];                                    // -log softmax(z)[y]
      const float g = wgt[q] * invB * scale / tdiv[q];
      for (int c = 0; c < K; ++c) outs[q][c] = (expf(z[c] - lse) - (c == y ? 1.f : 0.f)) * g;
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) red[q][threadIdx.x] = acc[q];
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot[3] = {0.f, 0.f, 0.f};
    for (int q = 0; q < 3; ++q) {
      for (int i = 0; i < 256; ++i) tot[q] += red[q][i];
      tot[q] *= invB;
    }
    // the regulariser always reads the learnable temperatures (fuzzy_gating_fusion.py:392-419), whatever the mode
    const float Ti = softplus_f(prm[0]) + eps_temp, Te = softplus_f(prm[1]) + eps_temp;
    const float reg = fmaxf(Ti - t_max, 0.f) + fmaxf(t_min - Ti, 0.f) + fmaxf(Te - t_max, 0.f) + fmaxf(t_min - Te, 0.f);
    auto sig = [](float x) { return 1.0f / (1.0f + expf(-x)); };
    dtau[0] = l_reg * scale * ((Ti > t_max ? 1.f : 0.f) - (Ti < t_min ? 1.f : 0.f)) * sig(prm[0]);
    dtau[1] = l_reg * scale * ((Te > t_max ? 1.f : 0.f) - (Te < t_min ? 1.f : 0.f)) * sig(prm[1]);
    losses[1] = tot[0]; losses[2] = tot[1]; losses[3] = tot[2]; losses[4] = reg;
    losses[0] = tot[0] + l_img * tot[1] + l_eeg * tot[2] + l_reg * reg;
  }
}

extern "C" int eg_fusion_loop_loss(const float* fused, const float* z_img, const float* z_eeg, const int64_t* labels,
                                   const float* params, float* losses, float* dfused, float* daux_img, float* daux_eeg,
                                   float* dtau, int B, int K, int mode, float eps_temp, float lambda_aux_img,
                                   float lambda_aux_eeg, float lambda_reg, float t_min, float t_max, const eg_step_state* state,
                                   void* stream) {
  EG_CHECK(fused && z_img && z_eeg && labels && params && losses && dfused && daux_img && daux_eeg && dtau,
           "eg_fusion_loop_loss: null pointer");
  EG_CHECK(B > 0 && K > 1 && K <= 16 && mode >= 0 && mode <= 3, "eg_fusion_loop_loss: bad shape / mode");
  hipLaunchKernelGGL(fusion_loop_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, fused, z_img, z_eeg,
                     (const long long*)labels, params, losses, dfused, daux_img, daux_eeg, dtau, B, K, mode, eps_temp,
                     lambda_aux_img, lambda_aux_eeg, lambda_reg, t_min, t_max, state);
  EG_LAUNCH_CHECK("fusion_loop_loss");
  return 0;
}

extern "C" int eg_fuzzy_gate_bwd(const float* z_img, const float* z_eeg, const float* params, const float* dfused,
                                 const float* dalpha, float* dz_img, float* dz_eeg, float* partial, int B, int K, int mode,
                                 float eps_temp, float eps_log, float eps_div, void* stream) {
  EG_CHECK(z_img && z_eeg && params && dfused && dz_img && dz_eeg && partial, "eg_fuzzy_gate_bwd: null pointer");
  EG_CHECK(B > 0 && K > 1 && K <= 16 && mode >= 0 && mode <= 3, "eg_fuzzy_gate_bwd: bad shape / mode");
  hipLaunchKernelGGL(fuzzy_gate_bwd_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, z_img, z_eeg, params,
                     dfused, dalpha, dz_img, dz_eeg, partial, B, K, mode, eps_temp, eps_log, eps_div);
  EG_LAUNCH_CHECK("fuzzy_gate_bwd");
  return 0;
}

extern "C" int eg_fuzzy_gate_fwd(const float* z_img, const float* z_eeg, const float* params, float* fused, float* alpha,
                                 int B, int K, int mode, float eps_temp, float eps_log, float eps_div, void* stream) {
  EG_CHECK(z_img && z_eeg && params && fused && alpha, "eg_fuzzy_gate_fwd: null pointer");
  EG_CHECK(B > 0 && K > 1 && K <= 16 && mode >= 0 && mode <= 3, "eg_fuzzy_gate_fwd: bad shape / mode");
  hipLaunchKernelGGL(fuzzy_gate_fwd_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, z_img, z_eeg, params,
                     fused, alpha, B, K, mode, eps_temp, eps_log, eps_div);
  EG_LAUNCH_CHECK("fuzzy_gate_fwd");
  return 0;
}
