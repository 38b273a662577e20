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
  // a 32-lane group walks pooled rows grp, grp + 8, ... left to right: the 4 x 4 input patch of a cell slides by two columns, so a
  // cell costs 8 LDS reads and no index arithmetic beyond two pointer bumps (the cell = y * Wp + x form spent two thirds of its
  // instructions on division and addressing: 267 us at C = 32).  Per output the multiply-add chain is unchanged.
  for (int y = threadIdx.x >> 5; y < Hp; y += 8) {
    const float* r0 = I + (2 * y) * PW;
    T* orow = ob + ((size_t)(y + 1) * (Wp + 4) + 1) * C1 + ch;
    float pt[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { pt[r][2] = r0[r * PW]; pt[r][3] = r0[r * PW + 1]; }
    for (int x = 0; x < Wp; ++x) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pt[r][0] = pt[r][2]; pt[r][1] = pt[r][3];
        pt[r][2] = r0[r * PW + 2 * x + 2]; pt[r][3] = r0[r * PW + 2 * x + 3];
      }
      float best = 0.f;  // ReLU floor: max(relu(a_i)) = max(0, max a_i)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          float a = bb;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a = fmaf(wk[ky * 3 + kx], pt[dy + ky][dx + kx], a);
          best = fmaxf(best, a);
        }
      Elem<T>::st(orow + (size_t)x * C1, best);
    }
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
  // rows grp, grp + 8, ... left to right with a sliding 4 x 4 patch, as the forward kernel.  (The cell = y * Wp + x form gave a group
  // the cells grp, grp + 8, ...; the eight fp32 partial sums per channel now group the cells by pooled row -- equal to rounding.)
  for (int y = grp; y < Hp; y += 8) {
    const float* r0 = I + (2 * y) * PW;
    const T* grow = gb + (size_t)y * Wp * C1 + ch;
    float pt[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { pt[r][2] = r0[r * PW]; pt[r][3] = r0[r * PW + 1]; }
    for (int x = 0; x < Wp; ++x) {
      const float g = Elem<T>::ld(grow + (size_t)x * C1);        // requested before the recomputation, used after it
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pt[r][0] = pt[r][2]; pt[r][1] = pt[r][3];
        pt[r][2] = r0[r * PW + 2 * x + 2]; pt[r][3] = r0[r * PW + 2 * x + 3];
      }
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
            for (int kx = 0; kx < 3; ++kx) a = fmaf(wk[ky * 3 + kx], pt[dy + ky][dx + kx], a);
          if (a > best) { best = a; by = dy; bx = dx; }
        }
      if (best > 0.f) {
        const float* q = r0 + by * PW + 2 * x + bx;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = fmaf(g, q[ky * PW + kx], acc[ky * 3 + kx]);
        acc[9] += g;
      }
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


// ------------------------------------------------------------------------------------------------
// conv-2 weight gradient as a FLAT correlation (16-bit operands, N = 64 outputs, Cin = 32).
// Activations p1 and output gradients d2 live in the same padded pixel rows [image row][Wp + 4 pixels][channels]; with q the flat
// padded pixel index, the 3 x 3 window of output pixel q is p1[q + ky * rowpx + kx] and its gradient is d2[q + rowpx + 1].  The pads
// of d2 are zero (eg_spec_avgpool_bwd writes interiors only), so the sum may run over EVERY q:
//     dW[n][ky][kx][c] = sum_q d2[q + rowpx + 1][n] * p1[q + ky * rowpx + kx][c]
// -- nine GEMMs over q whose X operand is the same LDS image read at nine row offsets.  The im2col form (eg_gemm_tn with a sliding
// row map) fetched every p1 pixel twelve times and every d2 row three times through L2 and ran half of its 128-column MFMA tile on
// padding: 0.93 ms at C = 32; here each operand byte reaches LDS once.
// Workgroup = one split of q; 256-row stages (Y 32 KB + X 20 KB incl. a 64-row halo) in ONE buffer, the next stage waits in
// registers.  Wave w multiplies all four 16-wide n-tiles by channel tile (w & 1) of taps 0..4 (w < 2) or 5..8: 20 or 16 MFMA per
// 32-row step against 9 KB of fragment reads.  Fragments by ds_read_b64_tr_b16 (the recipe of tn_mma); 32-B slots swizzled so that
// the eight rows a half-wave touches never share a slot, at any tap offset.
// Output: partial[split][n][(ky * 4 + kx) * 32 + c], the slab layout eg_unpack_conv2d_wgrad reduces (pad taps kx = 3 unwritten).
// ------------------------------------------------------------------------------------------------
constexpr int CW_QC = 256, CW_HALO = 64;    // halo >= 2 rowpx + 2: rowpx <= 31
constexpr int CW_YB = CW_QC * 128, CW_XB = (CW_QC + CW_HALO) * 64;

__device__ __forceinline__ int cw_yoff(int row, int slot) { return row * 128 + ((slot ^ (((row >> 1) & 1) | (((row >> 3) & 1) << 1))) << 5); }
__device__ __forceinline__ int cw_xoff(int row, int slot) { return row * 64 + ((slot ^ ((row >> 3) & 1)) << 5); }

template <typename T>
__global__ __launch_bounds__(256, 2) void conv2d_wgrad_flat_kernel(const T* __restrict__ d2, const T* __restrict__ p1,
                                                                   float* __restrict__ partial, float* __restrict__ bias_partial,
                                                                   long long Q, long long p1_rows, int rowpx,
                                                                   long long rows_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename H16<T>::frag frag;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  char* const bufY = smem;
  char* const bufX = smem + CW_YB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cj = wave & 1, tap0 = (wave >> 1) * 5, ntap = 5 - (wave >> 1);
  // wave 3 has a free fifth accumulator slot: with an all-ones X fragment it holds the column sums of the gradient rows = the bias
  // gradient (the pads of d2 are zero), at four MFMA per step and no extra LDS or HBM traffic (a separate column-sum pass re-read d2)
  const bool bias_wave = bias_partial != nullptr && wave == 3;
  const long long qbeg = (long long)blockIdx.x * rows_per_split;
  const long long qend = qbeg + rows_per_split < Q ? qbeg + rows_per_split : Q;
  const T* const yb = d2 + (size_t)(rowpx + 1) * 64;             // gradient of output pixel q

  f32x4 acc[5][4];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int toff[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int tt = min(tap0 + t, 8);
    toff[t] = (tt / 3) * rowpx + (tt % 3);
  }

  // staging map: Y chunk c = tid + 256 i (i < 8): row c >> 3, 16-B piece c & 7;  X chunk c = tid + 256 i (i < 5): row c >> 2, piece c & 3
  u32x4 ry[8], rx[5];
  auto load_stage = [&](long long q0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i, row = c >> 3, pc = c & 7;
      u32x4 z = {0u, 0u, 0u, 0u};
      ry[i] = z;
      if (q0 + row < qend) ry[i] = *(const u32x4*)(yb + (size_t)(q0 + row) * 64 + pc * 8);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = tid + 256 * i, row = c >> 2, pc = c & 3;
      u32x4 z = {0u, 0u, 0u, 0u};
      rx[i] = z;
      if (row < CW_QC + CW_HALO && q0 + row < p1_rows) rx[i] = *(const u32x4*)(p1 + (size_t)(q0 + row) * 32 + pc * 8);
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i, row = c >> 3, pc = c & 7;
      *(u32x4*)(bufY + cw_yoff(row, pc >> 1) + (pc & 1) * 16) = ry[i];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = tid + 256 * i, row = c >> 2, pc = c & 3;
      if (row < CW_QC + CW_HALO) *(u32x4*)(bufX + cw_xoff(row, pc >> 1) + (pc & 1) * 16) = rx[i];
    }
  };

  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  if (qbeg < qend) load_stage(qbeg);
  for (long long q0 = qbeg; q0 < qend; q0 += CW_QC) {
    store_stage();
    __syncthreads();
    if (q0 + CW_QC < qend) load_stage(q0 + CW_QC);
#pragma unroll 2
    for (int ks = 0; ks < CW_QC / 32; ++ks) {
      const int r0 = 32 * ks + 8 * g + qq, r1 = r0 + 4;
      frag yf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bufY + cw_yoff(r0, i) + 8 * pp));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bufY + cw_yoff(r1, i) + 8 * pp));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        yf[i] = __builtin_bit_cast(frag, v);
      }
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        if (t < ntap) {
          const int a0 = r0 + toff[t], a1 = r1 + toff[t];
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bufX + cw_xoff(a0, cj) + 8 * pp));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bufX + cw_xoff(a1, cj) + 8 * pp));
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          const frag xf = __builtin_bit_cast(frag, v);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t][i] = H16<T>::mfma(xf, yf[i], acc[t][i]);
        }
      }
      if (bias_wave) {
        const uint32_t one2 = H16<T>::pack2(1.f, 1.f);
        const u32x4 ones = {one2, one2, one2, one2};
        const frag xf = __builtin_bit_cast(frag, ones);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4][i] = H16<T>::mfma(xf, yf[i], acc[4][i]);
      }
    }
    __syncthreads();
  }
  if (bias_wave && lane < 16) {                                   // every row of the ones-product holds the sums: lane g = 0, r = 0
#pragma unroll
    for (int i = 0; i < 4; ++i) bias_partial[(size_t)blockIdx.x * 64 + i * 16 + lane] = acc[4][i][0];
  }
  // D[i = channel][j = n]: lane holds 4 consecutive channels (4 g + r) of column n = lane & 15
  float* out = partial + (size_t)blockIdx.x * (64 * 384);
  const int l15 = lane & 15;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    if (t < ntap) {
      const int tt = tap0 + t, kcol = ((tt / 3) * 4 + (tt % 3)) * 32 + cj * 16 + 4 * g;
#pragma unroll
      for (int i = 0; i < 4; ++i) *(f32x4*)(out + (size_t)(i * 16 + l15) * 384 + kcol) = acc[t][i];
    }
  }
}


// ------------------------------------------------------------------------------------------------
// 3 x 3 convolutions on the same flat pixel index (16-bit operands), forward and backward-data alike:
//     out[g * Wp + x][n] = act(b[n] + sum over the nine taps and CIN channels of in[q + ky * rowpx + kx][c] * W[n][(ky * 4 + kx) * CIN + c]),
//     q = g * rowpx + x, x < Wp
// forward: in = p1 (CIN 32), NOUT 64, W = eg_pack_conv2d_weight(transposed = 0), ReLU;  backward-data: in = d2 (CIN 64), NOUT 32,
// W = the transposed packing (taps flipped), out = dp1.  The segmented-row eg_gemm_nt re-read every input pixel twelve times through
// L2 (0.37 + 0.58 ms at C = 32); here a stage of QC pixels (+ 64-pixel halo) sits in LDS once and the nine taps are nine row offsets of
// that image.  A wave owns a quarter of a stage's m-tiles (16 pixels each) and ALL outputs: the 36 weight fragments (NOUT/16 n-tiles
// x 9 taps x CIN/32 k-steps) stay in registers for the whole launch; an m-tile costs 9 CIN/32 fragment reads (ds_read_b128, chunk
// index XOR-swizzled by the row) and 36 MFMA.  One third of the MFMA work is the pad slots x >= Wp (never stored); the launches are
// bound by HBM, not by MFMA.
// ------------------------------------------------------------------------------------------------
// LDS image of the input pixels, conflict-free for ds_read_b128 at EVERY row offset (the nine taps shift the rows a fragment reads;
// the instruction serves its lanes in the four 16-lane groups {0-3, 12-15, 20-27}, ...: checked by enumeration over all offsets):
// 32 channels: 64-B rows at a 96-B pitch, no swizzle;  64 channels: 128-B rows, 16-B chunk index XOR (row & 7).
// (The first form -- chunk ^ ((row / rows-per-bank-row) & mask) -- was two ways conflicted on every read: SQ_LDS_BANK_CONFLICT 41 %.)
template <int CIN> struct CfImg;
template <> struct CfImg<32> { static constexpr int PITCH = 96; };
template <> struct CfImg<64> { static constexpr int PITCH = 128; };
template <int CIN>
__device__ __forceinline__ int cf_xoff(int row, int chunk) {
  return CIN == 32 ? row * 96 + (chunk << 4) : row * 128 + ((chunk ^ (row & 7)) << 4);
}

template <typename T, int ACT, int CIN, int NOUT>
__global__ __launch_bounds__(256, 2) void conv2d_flat_kernel(const T* __restrict__ in, const T* __restrict__ W,
                                                             const float* __restrict__ bias, T* __restrict__ out, int Q, int in_rows,
                                                             int rowpx, int Wp, int rows_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename H16<T>::frag frag;
  constexpr int QC = CIN == 32 ? 256 : 128;                      // pixels per stage
  constexpr int CPR = CIN / 8, KS = CIN / 32, NT = NOUT / 16;
  constexpr int NCH = (QC + CW_HALO) * CPR / 256;                // 16-B chunks per thread and stage (5 / 6)
  constexpr int MTW = QC / 64;                                   // m-tiles per wave and stage
  static_assert((QC + CW_HALO) * CPR % 256 == 0 && NT * KS == 4, "conv2d_flat_kernel: shape");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int qbeg = blockIdx.x * rows_per_split;
  const int qend = min(Q, qbeg + rows_per_split);

  frag wf[9][KS][NT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < NT; ++i)
        wf[t][ks][i] = *(const frag*)(W + (size_t)(i * 16 + l15) * (12 * CIN) + ((t / 3) * 4 + (t % 3)) * CIN + ks * 32 + 8 * g);
  float bv[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[i][r] = bias ? bias[i * 16 + 4 * g + r] : 0.f;
  int toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = (t / 3) * rowpx + (t % 3);

  // staging map: chunk c = tid + 256 i (i < NCH): row c / CPR, 16-B piece c % CPR of the (QC + 64)-row image
  u32x4 rx[NCH];
  auto load_stage = [&](int q0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + 256 * i, row = c / CPR, pc = c % CPR;
      u32x4 z = {0u, 0u, 0u, 0u};
      rx[i] = z;
      if (q0 + row < in_rows) rx[i] = *(const u32x4*)(in + (size_t)(q0 + row) * CIN + pc * 8);
    }
  };
  if (qbeg < qend) load_stage(qbeg);
  for (int q0 = qbeg; q0 < qend; q0 += QC) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + 256 * i;
      *(u32x4*)(smem + cf_xoff<CIN>(c / CPR, c % CPR)) = rx[i];
    }
    __syncthreads();
    if (q0 + QC < qend) load_stage(q0 + QC);
#pragma unroll 1
    for (int mt = 0; mt < MTW; ++mt) {
      const int m0 = (wave * MTW + mt) * 16;                      // first pixel of the m-tile within the stage
      if (q0 + m0 >= qend) break;
      f32x4 acc[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const frag xf = *(const frag*)(smem + cf_xoff<CIN>(m0 + l15 + toff[t], ks * 4 + g));
#pragma unroll
          for (int i = 0; i < NT; ++i) acc[i] = H16<T>::mfma(wf[t][ks][i], xf, acc[i]);
        }
      // D[i = n][j = pixel]: lane holds outputs n = 16 i + 4 g + r of pixel m0 + l15
      const int q = q0 + m0 + l15;
      const int grp = q / rowpx, x = q - grp * rowpx;
      if (q < qend && x < Wp) {
        T* o = out + ((size_t)grp * Wp + x) * NOUT + 4 * g;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[i][r] + bv[i][r];
            if (ACT == EG_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
          }
          *(uint2*)(o + i * 16) = make_uint2(H16<T>::pack2(v[0], v[1]), H16<T>::pack2(v[2], v[3]));
        }
      }
    }
    __syncthreads();
  }
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

extern "C" int eg_conv2d_wgrad_flat(const void* d2, const void* p1, float* partial, float* bias_partial, long long Q, long long p1_rows,
                                    int rowpx, int splits, int dtype, void* stream) {
  EG_CHECK(d2 && p1 && partial && Q > 0 && rowpx >= 4 && 2 * rowpx + 2 <= CW_HALO && splits > 0, "eg_conv2d_wgrad_flat: bad arguments");
  EG_CHECK(p1_rows >= Q + 2 * rowpx + 2, "eg_conv2d_wgrad_flat: p1 holds %lld pixel rows, the windows of %lld pixels reach %lld", p1_rows, Q,
           Q + 2 * rowpx + 2);
  EG_CHECK(dtype == EG_BF16 || dtype == EG_F16, "eg_conv2d_wgrad_flat: 16-bit operands only (dtype %d); fp32 goes through eg_gemm_tn", dtype);
  // rows per split: a multiple of the 256-row stage; every split must own at least one row
  long long rps = ((Q + splits - 1) / splits + CW_QC - 1) / CW_QC * CW_QC;
  EG_CHECK((long long)(splits - 1) * rps < Q, "eg_conv2d_wgrad_flat: %d splits of %lld rows overrun Q = %lld (use eg_conv2d_wgrad_flat_splits)",
           splits, rps, Q);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv2d_wgrad_flat_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, CW_YB + CW_XB);
    (void)hipFuncSetAttribute((const void*)conv2d_wgrad_flat_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, CW_YB + CW_XB);
    attr = true;
  }
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(conv2d_wgrad_flat_kernel<bf16_t>, dim3(splits), dim3(256), CW_YB + CW_XB, s, (const bf16_t*)d2, (const bf16_t*)p1,
                       partial, bias_partial, Q, p1_rows, rowpx, rps);
  else
    hipLaunchKernelGGL(conv2d_wgrad_flat_kernel<f16_t>, dim3(splits), dim3(256), CW_YB + CW_XB, s, (const f16_t*)d2, (const f16_t*)p1,
                       partial, bias_partial, Q, p1_rows, rowpx, rps);
  EG_LAUNCH_CHECK("conv2d_wgrad_flat");
  return 0;
}

// the largest split count <= want whose 256-row-aligned splits all own rows
extern "C" int eg_conv2d_wgrad_flat_splits(long long Q, int want) {
  if (Q <= 0 || want <= 0) return 0;
  int splits = want;
  while (splits > 1) {
    const long long rps = ((Q + splits - 1) / splits + CW_QC - 1) / CW_QC * CW_QC;
    if ((long long)(splits - 1) * rps < Q) break;
    --splits;
  }
  return splits;
}

extern "C" int eg_conv2d_flat(const void* in, const void* W, const float* bias, void* out, long long Q, long long in_rows, int rowpx,
                              int Wp, int cin, int nout, int act, int splits, int dtype, void* stream) {
  EG_CHECK(in && W && out && Q > 0 && Q < (1ll << 31) - 4096 && in_rows < (1ll << 31) && rowpx >= 4 && 2 * rowpx + 2 <= CW_HALO &&
           Wp > 0 && Wp <= rowpx && splits > 0, "eg_conv2d_flat: bad arguments");
  EG_CHECK(dtype == EG_BF16 || dtype == EG_F16, "eg_conv2d_flat: 16-bit operands only (dtype %d); fp32 goes through eg_gemm_nt", dtype);
  EG_CHECK((cin == 32 && nout == 64) || (cin == 64 && nout == 32), "eg_conv2d_flat: %d -> %d channels (32 -> 64 or 64 -> 32)", cin, nout);
  EG_CHECK(act == EG_ACT_NONE || act == EG_ACT_RELU, "eg_conv2d_flat: act %d (none or ReLU)", act);
  EG_CHECK(in_rows >= Q + 2 * rowpx + 2, "eg_conv2d_flat: the input holds %lld pixel rows, the windows of %lld pixels reach %lld", in_rows, Q,
           Q + 2 * rowpx + 2);
  const int qc = cin == 32 ? 256 : 128;
  const long long rps = ((Q + splits - 1) / splits + qc - 1) / qc * qc;
  const int nblk = (int)((Q + rps - 1) / rps);
  const int lds = (qc + CW_HALO) * (cin == 32 ? CfImg<32>::PITCH : CfImg<64>::PITCH);
  hipStream_t s = (hipStream_t)stream;
#define EG_CF_LAUNCH(T, A, CI, NO) \
  hipLaunchKernelGGL((conv2d_flat_kernel<T, A, CI, NO>), dim3(nblk), dim3(256), lds, s, (const T*)in, (const T*)W, bias, (T*)out, (int)Q, \
                     (int)in_rows, rowpx, Wp, (int)rps)
#define EG_CF_SHAPE(T) \
  if (cin == 32) { if (act == EG_ACT_RELU) EG_CF_LAUNCH(T, EG_ACT_RELU, 32, 64); else EG_CF_LAUNCH(T, EG_ACT_NONE, 32, 64); } \
  else           { if (act == EG_ACT_RELU) EG_CF_LAUNCH(T, EG_ACT_RELU, 64, 32); else EG_CF_LAUNCH(T, EG_ACT_NONE, 64, 32); }
  if (dtype == EG_BF16) { EG_CF_SHAPE(bf16_t) } else { EG_CF_SHAPE(f16_t) }
#undef EG_CF_SHAPE
#undef EG_CF_LAUNCH
  EG_LAUNCH_CHECK("conv2d_flat");
  return 0;
}
