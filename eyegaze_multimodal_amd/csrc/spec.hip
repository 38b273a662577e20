// 2-D CNN over the STFT image (dual_eeg_transformer.py:70-77, 124-127):
//   Conv2d(1,32,3,p1)+ReLU+MaxPool2  -> direct kernel (one workgroup per image; K = 9 is too shallow for MFMA)
//   Conv2d(32,64,3,p1)+ReLU          -> eg_gemm_nt over segmented channel-last rows (3 segments of 4 x 32 channels)
//   AdaptiveAvgPool2d(4,4)+flatten   -> pooling kernel that emits PyTorch's (c, py, px) flatten order
// Layouts (per image, "padded" = one zero row on top/bottom, one zero column left, three right):
//   p1   [Hp+2][Wp+4][32]  pooled conv-1 output, channel-last      out2 [Hp+2][Wp][64] conv-2 output (rows >= Hp unused)
//   d2   [Hp+2][Wp+4][64]  gradient of out2 (padded)               dp1  [Hp+2][Wp][32] gradient of p1 (rows >= Hp unused)
#include "common.h"

namespace {

constexpr int C1 = 32, C2 = 64;

template <typename T>
__global__ __launch_bounds__(256) void spec_conv1_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                             const float* __restrict__ bias, T* __restrict__ p1, int F,
                                                             int nfr) {
  extern __shared__ float sm[];
  const int Hp = F / 2, Wp = nfr / 2, PW = nfr + 2;
  float* I = sm;                       // [(F+2)][PW] zero-padded image
  float* wl = I + (F + 2) * PW;        // [32][9]
  float* bl = wl + C1 * 9;             // [32]
  const int im = blockIdx.x;
  for (int i = threadIdx.x; i < (F + 2) * PW; i += blockDim.x) {
    const int y = i / PW - 1, x = i % PW - 1;
    I[i] = (y >= 0 && y < F && x >= 0 && x < nfr) ? img[((size_t)im * F + y) * nfr + x] : 0.f;
  }
  for (int i = threadIdx.x; i < C1 * 9; i += blockDim.x) wl[i] = w[i];
  if (threadIdx.x < C1) bl[threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const int ch = threadIdx.x & 31;
  float wk[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) wk[i] = wl[ch * 9 + i];
  const float bb = bl[ch];
  T* ob = p1 + (size_t)im * (Hp + 2) * (Wp + 4) * C1;
  for (int cell = threadIdx.x >> 5; cell < Hp * Wp; cell += 8) {
    const int y = cell / Wp, x = cell % Wp;
    float best = 0.f;  // ReLU floor: max(relu(a_i)) = max(0, max a_i)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float a = bb;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) a = fmaf(wk[ky * 3 + kx], I[(2 * y + dy + ky) * PW + (2 * x + dx + kx)], a);
        best = fmaxf(best, a);
      }
    Elem<T>::st(ob + ((size_t)(y + 1) * (Wp + 4) + (x + 1)) * C1 + ch, best);
  }
}

// recompute conv-1, route the pooled gradient to the arg-max (ReLU-gated) and accumulate dW [32][9] + db [32]
template <typename T>
__global__ __launch_bounds__(256) void spec_conv1_bwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const T* __restrict__ dp1,
                                                             float* __restrict__ partial, int F, int nfr) {
  extern __shared__ float sm[];
  const int Hp = F / 2, Wp = nfr / 2, PW = nfr + 2;
  float* I = sm;
  float* wl = I + (F + 2) * PW;
  float* bl = wl + C1 * 9;
  float* red = bl + C1;  // [8][32][10]
  const int im = blockIdx.x;
  for (int i = threadIdx.x; i < (F + 2) * PW; i += blockDim.x) {
    const int y = i / PW - 1, x = i % PW - 1;
    I[i] = (y >= 0 && y < F && x >= 0 && x < nfr) ? img[((size_t)im * F + y) * nfr + x] : 0.f;
  }
  for (int i = threadIdx.x; i < C1 * 9; i += blockDim.x) wl[i] = w[i];
  if (threadIdx.x < C1) bl[threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const int ch = threadIdx.x & 31, grp = threadIdx.x >> 5;
  float wk[9], acc[10];
#pragma unroll
  for (int i = 0; i < 9; ++i) wk[i] = wl[ch * 9 + i];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = 0.f;
  const float bb = bl[ch];
  const T* gb = dp1 + (size_t)im * (Hp + 2) * Wp * C1;
  for (int cell = grp; cell < Hp * Wp; cell += 8) {
    const int y = cell / Wp, x = cell % Wp;
    float best = -INFINITY;
    int by = 0, bx = 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float a = bb;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) a = fmaf(wk[ky * 3 + kx], I[(2 * y + dy + ky) * PW + (2 * x + dx + kx)], a);
        if (a > best) { best = a; by = dy; bx = dx; }
      }
    if (best > 0.f) {
      const float g = Elem<T>::ld(gb + ((size_t)y * Wp + x) * C1 + ch);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = fmaf(g, I[(2 * y + by + ky) * PW + (2 * x + bx + kx)], acc[ky * 3 + kx]);
      acc[9] += g;
    }
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) red[(grp * C1 + ch) * 10 + i] = acc[i];
  __syncthreads();
  // partial[im][0:288] = dW[ch][9], partial[im][288:320] = db[ch]
  for (int i = threadIdx.x; i < C1 * 10; i += blockDim.x) {
    const int c = i / 10, j = i % 10;
    float s = 0.f;
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) s += red[(g8 * C1 + c) * 10 + j];
    partial[(size_t)im * 320 + (j < 9 ? c * 9 + j : 288 + c)] = s;
  }
}

// out2 [Hp+2][Wp][64] (post-ReLU) -> pooled [1024] = (c, py, px), mean over (Hp/4) x (Wp/4) windows
template <typename T>
__global__ __launch_bounds__(128) void spec_avgpool_fwd_kernel(const T* __restrict__ out2, T* __restrict__ pooled, int Hp, int Wp) {
  // thread <-> (window, 8 channels): 16-B reads, the window's positions summed in the same (y, x) order as the element-per-thread
  // form it replaces (which moved 2 B per lane: 284 us at C = 32)
  const int im = blockIdx.x, wy = Hp / 4, wx = Wp / 4;
  const T* ib = out2 + (size_t)im * (Hp + 2) * Wp * C2;
  const int o = threadIdx.x;                                  // 16 windows x 8 channel groups
  const int c8 = (o & 7) * 8, cell = o >> 3, py = cell >> 2, px = cell & 3;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  for (int y = py * wy; y < (py + 1) * wy; ++y) {
    const T* row = ib + ((size_t)y * Wp + px * wx) * C2 + c8;
    int x = 0;
    for (; x + 2 <= wx; x += 2) {
      float a[8], b[8];
      load8(row + (size_t)x * C2, a);
      load8(row + (size_t)(x + 1) * C2, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += a[e]; s[e] += b[e]; }
    }
    if (x < wx) {
      float a[8];
      load8(row + (size_t)x * C2, a);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += a[e];
    }
  }
  const float d = (float)(wy * wx);
#pragma unroll
  for (int e = 0; e < 8; ++e) Elem<T>::st(pooled + (size_t)im * (C2 * 16) + (c8 + e) * 16 + py * 4 + px, s[e] / d);
}

// d2 [Hp+2][Wp+4][64] interior = dpooled / (wy*wx) where out2 > 0
template <typename T>
__global__ __launch_bounds__(256) void spec_avgpool_bwd_kernel(const T* __restrict__ out2, const T* __restrict__ dpooled, T* __restrict__ d2,
                                                               int Hp, int Wp) {
  // 16-B accesses (8 channels per thread) with the image's 64 x 16 pooled gradients staged once in LDS as [window][channel]; the
  // element-per-thread form moved 2 B per lane and re-read the pooled gradient from global memory per element (611 us at C = 32)
  __shared__ float gs[16][C2 + 4];
  const int im = blockIdx.x, wy = Hp / 4, wx = Wp / 4;
  const T* ob = out2 + (size_t)im * (Hp + 2) * Wp * C2;
  T* db = d2 + (size_t)im * (Hp + 2) * (Wp + 4) * C2;
  const float inv = 1.0f / (float)(wy * wx);
  for (int j = threadIdx.x; j < C2 * 16; j += blockDim.x)
    gs[j & 15][j >> 4] = Elem<T>::ld(dpooled + (size_t)im * (C2 * 16) + j) * inv;
  __syncthreads();
  for (int i8 = threadIdx.x; i8 < Hp * Wp * (C2 / 8); i8 += blockDim.x) {
    const int c8 = (i8 & (C2 / 8 - 1)) * 8, xy = i8 / (C2 / 8), x = xy % Wp, y = xy / Wp;
    const float* g = gs[(y / wy) * 4 + (x / wx)] + c8;
    float a[8], o[8];
    load8(ob + ((size_t)y * Wp + x) * C2 + c8, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = a[e] > 0.f ? g[e] : 0.f;
    store8(db + ((size_t)(y + 1) * (Wp + 4) + (x + 1)) * C2 + c8, o);
  }
}

// forward layout: dst[n][(ky*4 + kx)*Cin + c] = w[n][c][ky][kx] (kx = 3 -> 0)
// transposed (backward-data) layout: dst[c][(ky*4 + kx)*N + n] = w[n][c][2-ky][2-kx] (kx = 3 -> 0)
template <typename T>
__global__ void pack_conv2d_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int N, int Cin, int transposed) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)N * Cin * 12;
  if (i >= total) return;
  float v = 0.f;
  if (!transposed) {
    const int n = (int)(i / (12 * Cin)), r = (int)(i % (12 * Cin)), kk = r / Cin, c = r % Cin, ky = kk / 4, kx = kk % 4;
    if (kx < 3) v = w[(((size_t)n * Cin + c) * 3 + ky) * 3 + kx];
  } else {
    const int c = (int)(i / (12 * N)), r = (int)(i % (12 * N)), kk = r / N, n = r % N, ky = kk / 4, kx = kk % 4;
    if (kx < 3) v = w[(((size_t)n * Cin + c) * 3 + (2 - ky)) * 3 + (2 - kx)];
  }
  Elem<T>::st(dst + i, v);
}

// dW[n][c][ky][kx] = sum_s partial[s][n][(ky*4 + kx)*Cin + c].  Thread <-> one element of a [N][12 * Cin] slab, c fastest, so a wave
// reads 256 contiguous bytes of every slab (the former map walked (kx, ky, c): 4-B reads Cin floats apart, 72 blocks streaming up to
// 768 slabs: 81 us per launch); four slabs are requested per trip, the sums keep their order.
__global__ __launch_bounds__(256) void unpack_conv2d_wgrad_kernel(const float* __restrict__ partial, float* __restrict__ dW, int splits,
                                                                  long long stride, int N, int Cin) {
  const int K = 12 * Cin;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * K) return;
  const int n = (int)(i / K), r = (int)(i % K), kk = r / Cin, c = r % Cin, ky = kk >> 2, kx = kk & 3;
  if (kx == 3) return;                                       // pad tap of the 3 x 4 window
  const float* src = partial + i;
  float s = 0.f;
  int sp = 0;
  for (; sp + 4 <= splits; sp += 4) {
    const float a0 = src[(size_t)sp * stride], a1 = src[(size_t)(sp + 1) * stride], a2 = src[(size_t)(sp + 2) * stride],
                a3 = src[(size_t)(sp + 3) * stride];
    s += a0; s += a1; s += a2; s += a3;
  }
  for (; sp < splits; ++sp) s += src[(size_t)sp * stride];
  dW[(((size_t)n * Cin + c) * 3 + ky) * 3 + kx] = s;
}

}  // namespace

#define SPEC_DISPATCH(dtype, BF, F16, F32, who)   \
  if ((dtype) == EG_BF16) { BF; }                  \
  else if ((dtype) == EG_F16) { F16; }             \
  else if ((dtype) == EG_F32) { F32; }             \
  else return eg_fail("%s: bad dtype %d", who, (int)(dtype));

static int spec_shape_ok(const char* who, int nimg, int F, int nfr) {
  EG_CHECK(nimg > 0 && F >= 8 && F % 8 == 0 && nfr >= 8 && (nfr / 2) % 4 == 0,
           "%s: image %dx%d unsupported (F %% 8 == 0 and floor(frames/2) %% 4 == 0 required)", who, F, nfr);
  return 0;
}

extern "C" int eg_spec_conv1_fwd(const float* img, const float* w, const float* bias, void* p1, int nimg, int F, int nfr,
                                 int dtype, void* stream) {
  EG_CHECK(img && w && bias && p1, "eg_spec_conv1_fwd: null pointer");
  if (spec_shape_ok("eg_spec_conv1_fwd", nimg, F, nfr)) return 1;
  const int lds = ((F + 2) * (nfr + 2) + C1 * 10) * 4;
  hipStream_t s = (hipStream_t)stream;
  SPEC_DISPATCH(dtype,
                hipLaunchKernelGGL(spec_conv1_fwd_kernel<bf16_t>, dim3(nimg), dim3(256), lds, s, img, w, bias, (bf16_t*)p1, F, nfr),
                hipLaunchKernelGGL(spec_conv1_fwd_kernel<f16_t>, dim3(nimg), dim3(256), lds, s, img, w, bias, (f16_t*)p1, F, nfr),
                hipLaunchKernelGGL(spec_conv1_fwd_kernel<float>, dim3(nimg), dim3(256), lds, s, img, w, bias, (float*)p1, F, nfr),
                "eg_spec_conv1_fwd");
  EG_LAUNCH_CHECK("spec_conv1_fwd");
  return 0;
}

extern "C" int eg_spec_conv1_bwd(const float* img, const float* w, const float* bias, const void* dp1, float* partial,
                                 int nimg, int F, int nfr, int dtype, void* stream) {
  EG_CHECK(img && w && bias && dp1 && partial, "eg_spec_conv1_bwd: null pointer");
  if (spec_shape_ok("eg_spec_conv1_bwd", nimg, F, nfr)) return 1;
  const int lds = ((F + 2) * (nfr + 2) + C1 * 10 + 8 * C1 * 10) * 4;
  hipStream_t s = (hipStream_t)stream;
  SPEC_DISPATCH(dtype,
                hipLaunchKernelGGL(spec_conv1_bwd_kernel<bf16_t>, dim3(nimg), dim3(256), lds, s, img, w, bias, (const bf16_t*)dp1, partial, F, nfr),
                hipLaunchKernelGGL(spec_conv1_bwd_kernel<f16_t>, dim3(nimg), dim3(256), lds, s, img, w, bias, (const f16_t*)dp1, partial, F, nfr),
                hipLaunchKernelGGL(spec_conv1_bwd_kernel<float>, dim3(nimg), dim3(256), lds, s, img, w, bias, (const float*)dp1, partial, F, nfr),
                "eg_spec_conv1_bwd");
  EG_LAUNCH_CHECK("spec_conv1_bwd");
  return 0;
}

extern "C" int eg_spec_avgpool_fwd(const void* out2, void* pooled, int nimg, int Hp, int Wp, int dtype, void* stream) {
  EG_CHECK(out2 && pooled && nimg > 0 && Hp % 4 == 0 && Wp % 4 == 0, "eg_spec_avgpool_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  SPEC_DISPATCH(dtype,
                hipLaunchKernelGGL(spec_avgpool_fwd_kernel<bf16_t>, dim3(nimg), dim3(128), 0, s, (const bf16_t*)out2, (bf16_t*)pooled, Hp, Wp),
                hipLaunchKernelGGL(spec_avgpool_fwd_kernel<f16_t>, dim3(nimg), dim3(128), 0, s, (const f16_t*)out2, (f16_t*)pooled, Hp, Wp),
                hipLaunchKernelGGL(spec_avgpool_fwd_kernel<float>, dim3(nimg), dim3(128), 0, s, (const float*)out2, (float*)pooled, Hp, Wp),
                "eg_spec_avgpool_fwd");
  EG_LAUNCH_CHECK("spec_avgpool_fwd");
  return 0;
}

extern "C" int eg_spec_avgpool_bwd(const void* out2, const void* dpooled, void* d2, int nimg, int Hp, int Wp, int dtype,
                                   void* stream) {
  EG_CHECK(out2 && dpooled && d2 && nimg > 0 && Hp % 4 == 0 && Wp % 4 == 0, "eg_spec_avgpool_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  SPEC_DISPATCH(dtype,
                hipLaunchKernelGGL(spec_avgpool_bwd_kernel<bf16_t>, dim3(nimg), dim3(256), 0, s, (const bf16_t*)out2, (const bf16_t*)dpooled, (bf16_t*)d2, Hp, Wp),
                hipLaunchKernelGGL(spec_avgpool_bwd_kernel<f16_t>, dim3(nimg), dim3(256), 0, s, (const f16_t*)out2, (const f16_t*)dpooled, (f16_t*)d2, Hp, Wp),
                hipLaunchKernelGGL(spec_avgpool_bwd_kernel<float>, dim3(nimg), dim3(256), 0, s, (const float*)out2, (const float*)dpooled, (float*)d2, Hp, Wp),
                "eg_spec_avgpool_bwd");
  EG_LAUNCH_CHECK("spec_avgpool_bwd");
  return 0;
}

extern "C" int eg_pack_conv2d_weight(const float* w, void* dst, int N, int Cin, int transposed, int dtype, void* stream) {
  EG_CHECK(w && dst && N > 0 && Cin > 0, "eg_pack_conv2d_weight: bad arguments");
  const long long n = (long long)N * Cin * 12;
  dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  SPEC_DISPATCH(dtype,
                hipLaunchKernelGGL(pack_conv2d_weight_kernel<bf16_t>, grid, dim3(256), 0, s, w, (bf16_t*)dst, N, Cin, transposed),
                hipLaunchKernelGGL(pack_conv2d_weight_kernel<f16_t>, grid, dim3(256), 0, s, w, (f16_t*)dst, N, Cin, transposed),
                hipLaunchKernelGGL(pack_conv2d_weight_kernel<float>, grid, dim3(256), 0, s, w, (float*)dst, N, Cin, transposed),
                "eg_pack_conv2d_weight");
  EG_LAUNCH_CHECK("pack_conv2d_weight");
  return 0;
}

extern "C" int eg_unpack_conv2d_wgrad(float* partial, float* dW, int splits, int N, int Cin, void* stream) {
  EG_CHECK(partial && dW && splits > 0 && N > 0 && Cin > 0, "eg_unpack_conv2d_wgrad: bad arguments");
  const long long slab = (long long)N * 12 * Cin;
  EG_CHECK(slab % 4 == 0, "eg_unpack_conv2d_wgrad: N * 12 * Cin must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  long long stride = slab;
  if (splits > 64) {                  // long reductions in two stages: 64 groups of slabs are summed in place first (partial is scratch)
    int group = 1;
    if (eg_reduce_groups_inplace(partial, slab, splits, slab, 64, &group, s)) return eg_fail("eg_unpack_conv2d_wgrad: stage-1 launch failed");
    splits = (splits + group - 1) / group;
    stride = slab * group;
  }
  hipLaunchKernelGGL(unpack_conv2d_wgrad_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, s, partial, dW, splits, stride, N, Cin);
  EG_LAUNCH_CHECK("unpack_conv2d_wgrad");
  return 0;
}
