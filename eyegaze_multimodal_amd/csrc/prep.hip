// Staging kernels: input windows [B,C,T] -> channel-last padded rows, fp32 master parameters -> compute-dtype
// copies in the layouts the GEMMs read.  All HBM-bound, coalesced on both sides via an LDS transpose.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

thread_local char eg_err_buf[512] = {0};
int eg_fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(eg_err_buf, sizeof(eg_err_buf), fmt, ap);
  va_end(ap);
  return 1;
}
extern "C" const char* eg_last_error(void) { return eg_err_buf; }
extern "C" int eg_abi_version(void) { return EG_ABI_VERSION; }
extern "C" int eg_device_info(int* cu_count, char* arch, int arch_len) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
    return eg_fail("eg_device_info: no HIP device");
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
  return 0;
}

namespace {

// x [NB, C, T] f32 -> xt [NB, Tp, Cp]; one block = 256 consecutive padded time steps of one window
template <typename T>
__global__ __launch_bounds__(256) void window_pack_kernel(const float* __restrict__ x, T* __restrict__ xt, int C, int Tn,
                                                          int Cp, int pad_front, int Tp) {
  extern __shared__ float tile[];  // [256][Cp + 1]
  const int nb = blockIdx.y, tp0 = blockIdx.x * 256, tid = threadIdx.x;
  const int pitch = Cp + 1;
  const int t = tp0 + tid - pad_front;
  const bool tv = t >= 0 && t < Tn && tp0 + tid < Tp;
  const float* xb = x + (size_t)nb * C * Tn;
  for (int c = 0; c < Cp; ++c) tile[tid * pitch + c] = (tv && c < C) ? xb[(size_t)c * Tn + t] : 0.f;
  __syncthreads();
  const int rows = min(256, Tp - tp0);
  const int nchunk = rows * Cp / 8;
  T* ob = xt + ((size_t)nb * Tp + tp0) * Cp;
  for (int ch = tid; ch < nchunk; ch += 256) {
    const int e = ch * 8, r = e / Cp, c = e % Cp;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[r * pitch + c + j];
    store8(ob + e, v);
  }
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long long n) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 4 <= n) {
    float v[4];
    load4(src + i, v);
    store4(dst + i, v);
  } else {
    for (long long j = i; j < n; ++j) Elem<T>::st(dst + j, src[j]);
  }
}

// src [R, Cc] -> dst[c * ldd + r]
template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int R,
                                                             int Cc, int ldd) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? src[(size_t)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < R) Elem<T>::st(dst + (size_t)c * ldd + r, tile[tx][i]);
  }
}

// w [N, Cin, k] -> dst [N, Kp]: dst[n][tap*Cp + c] = w[n][c][tap]
template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int N, int Cin, int k, int Cp,
                                        int Kp) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * Kp) return;
  const int n = (int)(i / Kp), kk = (int)(i % Kp);
  const int tap = kk / Cp, c = kk % Cp;
  const float v = (tap < k && c < Cin) ? w[((size_t)n * Cin + c) * k + tap] : 0.f;
  Elem<T>::st(dst + i, v);
}

// backward-data weights: dst[p][c][j*N + n] = w[n][c][s*(J-1-j) + p]   (0 when the tap index exceeds k-1)
template <typename T>
__global__ void pack_convT_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int N, int Cin, int k, int s,
                                         int J) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long per_phase = (long long)Cin * J * N;
  if (i >= per_phase * s) return;
  const int p = (int)(i / per_phase);
  const long long rem = i % per_phase;
  const int c = (int)(rem / (J * N));
  const int jn = (int)(rem % (J * N));
  const int j = jn / N, n = jn % N;
  const int tap = s * (J - 1 - j) + p;
  const float v = tap < k ? w[((size_t)n * Cin + c) * k + tap] : 0.f;
  Elem<T>::st(dst + i, v);
}

// table-driven parameter staging: ONE launch re-casts / transposes every weight of the model.
// mode 0: dst[i] = cast(src[i]) over rows*cols contiguous elements; mode 1: dst[c*ldd + r] = cast(src[r*cols + c]);
// mode 2: like 0 but the destination is fp32 (bias vectors gathered into fused buffers); modes 3-6: eg_ffn_chain's fragment
// order (16-bit dtypes).  Each block handles one 32x32 tile (mode 1), 1024 elements (modes 0/2) or 2048 elements (modes 3-6);
// blk0 is the entry's first block.
template <typename T>
__global__ __launch_bounds__(256) void pack_table_kernel(const eg_pack_entry* __restrict__ tab, int nent) {
  __shared__ float tile[32][33];
  __shared__ int ent_s;
  if (threadIdx.x == 0) {
    int lo = 0, hi = nent - 1;
    while (lo < hi) {  // last entry with blk0 <= blockIdx.x
      const int mid = (lo + hi + 1) >> 1;
      if (tab[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    ent_s = lo;
  }
  __syncthreads();
  const eg_pack_entry e = tab[ent_s];
  const int lb = blockIdx.x - e.blk0;
  const float* src = (const float*)e.src;
  if (e.mode == 1) {
    const int tiles_c = (e.cols + 31) / 32;
    const int r0 = (lb / tiles_c) * 32, c0 = (lb % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
      const int r = r0 + i, c = c0 + tx;
      tile[i][tx] = (r < e.rows && c < e.cols) ? src[(size_t)r * e.cols + c] : 0.f;
    }
    __syncthreads();
    T* dst = (T*)e.dst;
    for (int i = ty; i < 32; i += 8) {
      const int c = c0 + i, r = r0 + tx;
      if (c < e.cols && r < e.rows) Elem<T>::st(dst + (size_t)c * e.ldd + r, tile[tx][i]);
    }
  } else if (e.mode == 7 || e.mode == 8) {
    // eg_attn_block_fwd's fragment order (16-bit dtypes), source fp32 [256, 256] = one of q_proj / k_proj / v_proj (mode 7, part
    // e.ldd = 0 / 1 / 2 into the shared [chunk c: 4][wave: 4][k-step s: 8][tile j: 3][lane] image, tile t = 3 wave + j of the chunk
    // = head t / 6, part (t % 6) / 2, half t % 2) or out_proj (mode 8: [chunk c: 4][wave: 4][k-step s: 2][tile j: 4][lane]).
    const int u = lb * 256 + threadIdx.x;                  // 8192 chunks of 8 elements per [256, 256] source
    const int lane = u & 63, l15 = lane & 15, g4 = lane >> 4;
    int n, k0;
    size_t q;
    if (e.mode == 7) {
      const int s8 = (u >> 6) & 7, half = (u >> 9) & 1, th = (u >> 10) & 1, c = u >> 11;
      const int t = th * 6 + e.ldd * 2 + half, wn = t / 3, j = t % 3;
      n = (2 * c + th) * 32 + 16 * half + l15; k0 = 32 * s8 + 8 * g4;
      q = ((((size_t)c * 4 + wn) * 8 + s8) * 3 + j) * 64 + lane;
    } else {
      const int j = (u >> 6) & 3, s2 = (u >> 8) & 1, wn = (u >> 9) & 3, c = u >> 11;
      n = 64 * wn + 16 * j + l15; k0 = 64 * c + 32 * s2 + 8 * g4;
      q = (size_t)u;
    }
    if (u < 8192) {
      float v[8];
      load8(src + (size_t)n * 256 + k0, v);
      if constexpr (sizeof(T) == 2) store8((T*)e.dst + q * 8, v);
    }
  } else if (e.mode >= 3 && e.mode <= 6) {
    // MFMA-fragment order of eg_ffn_chain's weights: destination chunk q (8 elements) is what lane q%64 of a wave loads as
    // its operand of one v_mfma_f32_16x16x32, so a fragment load is ONE contiguous 1-KB read.  The logical matrix is the source
    // (modes 3, 5) or its transpose (modes 4, 6); role 1 = [F, 256] (modes 3, 4), role 2 = [256, F] (modes 5, 6).
    const int q = lb * 256 + threadIdx.x;
    const bool tr = e.mode == 4 || e.mode == 6;
    const int lane = q & 63, l15 = lane & 15, g4 = lane >> 4;
    int n, k0, N, K;
    if (e.mode <= 4) {                       // role 1: [chunk c][wn][s: 8][j: 2][lane]
      const int j = (q >> 6) & 1, s8 = (q >> 7) & 7, wn = (q >> 10) & 3, c = q >> 12;
      n = 128 * c + 32 * wn + 16 * j + l15; k0 = 32 * s8 + 8 * g4;
      K = 256; N = tr ? e.cols : e.rows;
    } else {                                 // role 2: [chunk c][wn][s: 4][j: 4][lane]
      const int j = (q >> 6) & 3, s4 = (q >> 8) & 3, wn = (q >> 10) & 3, c = q >> 12;
      n = 64 * wn + 16 * j + l15; k0 = 128 * c + 32 * s4 + 8 * g4;
      N = 256; K = tr ? e.rows : e.cols;
    }
    if ((long long)q * 8 < (long long)e.rows * e.cols) {
      float v[8];
      if (!tr) load8(src + (size_t)n * K + k0, v);
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = src[(size_t)(k0 + i) * N + n];
      }
      if constexpr (sizeof(T) == 2) store8((T*)e.dst + (size_t)q * 8, v);
    }
  } else {
    const long long n = (long long)e.rows * e.cols;
    const long long i0 = (long long)lb * 1024 + threadIdx.x * 4;
    if (e.mode == 2) {
      float* dst = (float*)e.dst;
      for (long long i = i0; i < min(n, i0 + 4); ++i) dst[i] = src[i];
    } else {
      T* dst = (T*)e.dst;
      if (i0 + 4 <= n) {
        float v[4];
        load4(src + i0, v);
        store4(dst + i0, v);
      } else {
        for (long long i = i0; i < n; ++i) Elem<T>::st(dst + i, src[i]);
      }
    }
  }
}

}  // namespace

extern "C" int eg_pack_table(const eg_pack_entry* table, int nentries, int total_blocks, int dtype, void* stream) {
  EG_CHECK(table && nentries > 0 && total_blocks > 0, "eg_pack_table: bad arguments");
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(pack_table_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table, nentries);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(pack_table_kernel<f16_t>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table, nentries);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(pack_table_kernel<float>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table, nentries);
  else
    return eg_fail("eg_pack_table: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("pack_table");
  return 0;
}

extern "C" int eg_window_pack(const float* x, void* xt, int NB, int C, int T, int Cp, int pad_front, int Tp, int dtype,
                              void* stream) {
  EG_CHECK(x && xt, "eg_window_pack: null pointer");
  EG_CHECK(NB > 0 && C > 0 && T > 0, "eg_window_pack: bad shape NB=%d C=%d T=%d", NB, C, T);
  EG_CHECK(Cp >= C && Cp % 8 == 0 && Cp <= 256, "eg_window_pack: Cp=%d must be a multiple of 8 in [C, 256]", Cp);
  EG_CHECK(pad_front >= 0 && Tp >= T + pad_front, "eg_window_pack: Tp=%d < T+pad_front", Tp);
  dim3 grid((Tp + 255) / 256, NB);
  const size_t lds = 256 * (Cp + 1) * sizeof(float);
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(window_pack_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, x, (bf16_t*)xt, C, T, Cp,
                       pad_front, Tp);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(window_pack_kernel<f16_t>, grid, dim3(256), lds, (hipStream_t)stream, x, (f16_t*)xt, C, T, Cp,
                       pad_front, Tp);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(window_pack_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, x, (float*)xt, C, T, Cp,
                       pad_front, Tp);
  else
    return eg_fail("eg_window_pack: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("window_pack");
  return 0;
}

extern "C" int eg_cast(const float* src, void* dst, int64_t n, int dtype, void* stream) {
  EG_CHECK(src && dst && n > 0, "eg_cast: bad arguments");
  EG_CHECK(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 8 == 0), "eg_cast: alignment");
  const long long nt = (n + 3) / 4;
  dim3 grid((unsigned)((nt + 255) / 256));
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(cast_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long long)n);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(cast_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (f16_t*)dst, (long long)n);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(cast_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, (long long)n);
  else
    return eg_fail("eg_cast: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("cast");
  return 0;
}

extern "C" int eg_transpose_cast(const float* src, void* dst, int R, int Cc, int ldd, int dtype, void* stream) {
  EG_CHECK(src && dst && R > 0 && Cc > 0 && ldd >= R, "eg_transpose_cast: bad arguments");
  dim3 grid((Cc + 31) / 32, (R + 31) / 32);
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(transpose_cast_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, R, Cc,
                       ldd);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(transpose_cast_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (f16_t*)dst, R, Cc,
                       ldd);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(transpose_cast_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, R, Cc,
                       ldd);
  else
    return eg_fail("eg_transpose_cast: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("transpose_cast");
  return 0;
}

extern "C" int eg_pack_conv_weight(const float* w, void* dst, int N, int Cin, int k, int Cp, int Kp, int dtype,
                                   void* stream) {
  EG_CHECK(w && dst && N > 0 && Cin > 0 && k > 0, "eg_pack_conv_weight: bad arguments");
  EG_CHECK(Cp >= Cin && Kp >= k * Cp, "eg_pack_conv_weight: Cp=%d Kp=%d too small", Cp, Kp);
  const long long n = (long long)N * Kp;
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(pack_conv_weight_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)dst, N, Cin,
                       k, Cp, Kp);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(pack_conv_weight_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, w, (f16_t*)dst, N, Cin,
                       k, Cp, Kp);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(pack_conv_weight_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, w, (float*)dst, N, Cin, k,
                       Cp, Kp);
  else
    return eg_fail("eg_pack_conv_weight: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("pack_conv_weight");
  return 0;
}

extern "C" int eg_pack_convT_weight(const float* w, void* dst, int N, int Cin, int k, int stride, int dtype,
                                    void* stream) {
  EG_CHECK(w && dst && N > 0 && Cin > 0 && k > 0 && stride > 0, "eg_pack_convT_weight: bad arguments");
  const int J = (k + stride - 1) / stride;
  const long long n = (long long)stride * Cin * J * N;
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(pack_convT_weight_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)dst, N,
                       Cin, k, stride, J);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(pack_convT_weight_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)stream, w, (f16_t*)dst, N,
                       Cin, k, stride, J);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(pack_convT_weight_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, w, (float*)dst, N, Cin,
                       k, stride, J);
  else
    return eg_fail("eg_pack_convT_weight: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("pack_convT_weight");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Window normalisation of the data path (1_Data/processed/dual_eeg_dataset.py:142-168, 194-202).
//   raw [N, 2, C, T] f32 (window n: player-1 channels then player-2 channels, as the shards store them)
//   -> eeg1 [N, C, T], eeg2 [N, C, T] f32
//   mode 0 (enable_preprocessing = False, the yaml default): (x - mean(x)) / (std_pop(x) + 1e-8) over the whole window
//   mode 1 (enable_preprocessing = True): common-average reference (subtract the channel mean at every time step),
//          then per-channel (v - mean) / (std_pop + 1e-8)
// One block per (window, player).  Sums run in double: the reference reduces in float32 with pairwise summation, a
// double accumulator is at least as accurate, and the pass is bandwidth-bound either way (three reads from L2, one write).
// ---------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

__global__ __launch_bounds__(256) void window_normalize_kernel(const float* __restrict__ raw, float* __restrict__ eeg1,
                                                               float* __restrict__ eeg2, int C, int T, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[4];
  float* car = (float*)smem;  // [T] (mode 1)
  const int n = blockIdx.x >> 1, who = blockIdx.x & 1;
  const long long CT = (long long)C * T;
  const float* x = raw + ((long long)n * 2 + who) * CT;
  float* y = (who ? eeg2 : eeg1) + (long long)n * CT;
  if (mode == 0) {
    double s = 0.0;
    for (long long i = threadIdx.x; i < CT; i += blockDim.x) s += (double)x[i];
    const double mean = block_sum_d(s, red) / (double)CT;
    const float meanf = (float)mean;
    double q = 0.0;
    for (long long i = threadIdx.x; i < CT; i += blockDim.x) {
      const double d = (double)x[i] - mean;
      q += d * d;
    }
    const float stdf = (float)sqrt(block_sum_d(q, red) / (double)CT);
    const float den = stdf + 1e-8f;
    for (long long i = threadIdx.x; i < CT; i += blockDim.x) y[i] = (x[i] - meanf) / den;
    return;
  }
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += x[(long long)c * T + t];
    car[t] = s / (float)C;
  }
  __syncthreads();
  for (int c = 0; c < C; ++c) {
    const float* xc = x + (long long)c * T;
    double s = 0.0;
    for (int t = threadIdx.x; t < T; t += blockDim.x) s += (double)(xc[t] - car[t]);
    const double mean = block_sum_d(s, red) / (double)T;
    const float meanf = (float)mean;
    double q = 0.0;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
      const double d = (double)(xc[t] - car[t]) - mean;
      q += d * d;
    }
    const float den = (float)sqrt(block_sum_d(q, red) / (double)T) + 1e-8f;
    for (int t = threadIdx.x; t < T; t += blockDim.x) y[(long long)c * T + t] = ((xc[t] - car[t]) - meanf) / den;
  }
}

}  // namespace

extern "C" int eg_window_normalize(const float* raw, float* eeg1, float* eeg2, int N, int C, int T, int mode, void* stream) {
  EG_CHECK(raw && eeg1 && eeg2, "eg_window_normalize: null pointer");
  EG_CHECK(N > 0 && C > 0 && T > 0, "eg_window_normalize: bad shape N=%d C=%d T=%d", N, C, T);
  EG_CHECK(mode == 0 || mode == 1, "eg_window_normalize: mode %d (0 = window z-score, 1 = CAR + channel z-score)", mode);
  EG_CHECK(T <= 16384, "eg_window_normalize: T=%d exceeds the 16384-sample staging row", T);
  hipLaunchKernelGGL(window_normalize_kernel, dim3(2 * N), dim3(256), mode ? (size_t)T * 4 : 0, (hipStream_t)stream, raw, eeg1,
                     eeg2, C, T, mode);
  EG_LAUNCH_CHECK("window_normalize");
  return 0;
}
