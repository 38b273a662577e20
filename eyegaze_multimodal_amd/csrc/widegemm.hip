// Wide-tile NT GEMM for the products whose output is one d_model-wide row block (N == 256, any K % 64 == 0, 16-bit operands):
//   C[M, 256] = epilogue(A[M, K] * W[256, K]^T)     out-proj, FFN-2, every backward-data product of the encoder, both 1-D convs
//
// Why a second tile: with N == 256 the 128x128 tile launches M/128 x 2 = 520 workgroups at the benchmark size -- 0.68 of one
// resident round -- and each pair of workgroups pulls the same A rows through two CUs' load paths.  Here ONE workgroup owns 160
// rows x all 256 columns: A crosses the load path once, the grid (208 workgroups for 33 280 rows) is a single round, and the
// K loop is fed by LDS-DMA (global_load_lds, 16 B per lane) through a 3-stage ring with ONE raw barrier per 64-deep K step and
// counted s_waitcnt vmcnt(N) waits, so two K steps of loads are always in flight behind the MFMAs.
//   512 threads = 8 waves as 2 (row halves of 80) x 4 (column groups of 64); a wave holds 5 x 4 accumulator tiles (80 VGPRs).
//   LDS rows are 128 B (one K step) with the 16-B-chunk XOR swizzle on the DMA's per-lane SOURCE address and on the fragment
//   reads (conflict-free ds_read_b128).  Epilogue: accumulators -> wave-private fp32 LDS image (in the drained ring) -> whole
//   128-B row segments with bias / activation / gate / dropout / second output / residual, exactly gemm_nt_kernel's order.
// Arithmetic is the same k-ordered MFMA chain as gemm_nt_kernel (bit-identical results).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int WBM = 160, WBN = 256;
constexpr int WST_A = WBM * 128, WST_W = WBN * 128, WSTAGE = WST_A + WST_W;   // 20 KB + 32 KB per stage
constexpr int WNST = 3;
constexpr int WLDS = WNST * WSTAGE;                                         // 159,744 B
constexpr int WTP = 68;                                                      // fp32 image pitch (floats): 64 + 4

template <typename T>
struct WideNT {
  const T* A; const T* W; T* C; const float* bias; const T* residual; const T* gate; T* out_pre;
  const eg_step_state* st;
  RowMap a, c, r, pm;
  int M, N, K, ldw;
  DropCfg d1, d2;
  float gate_scale;
};

__device__ __forceinline__ void wdma16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int ACT>
__device__ __forceinline__ float wide_act(float v) {
  if (ACT == EG_ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == EG_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  return v;
}

template <typename T, int ACT>
__global__ __launch_bounds__(512, 2) void gemm_nt_wide_kernel(WideNT<T> p) {
  typedef typename H16<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int m0 = blockIdx.x * WBM;
  const int nk = p.K >> 6;

  // ---- DMA addressing: an instruction moves 8 rows x 128 B; lane -> row lane/8, LDS chunk position lane%8 holding global
  //      chunk pos ^ (row & 7).  A: 20 instructions per stage (waves 0-3 issue 3, waves 4-7 issue 2); W: 32 (4 per wave). ----
  const int drow = lane >> 3, dpos = lane & 7;
  const int dsw = (dpos ^ drow) << 4;                        // (8q + drow) & 7 == drow
  const char* asrc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int r = 8 * (wave + 8 * i) + drow;
    asrc[i] = (const char*)(p.A + row_off(p.a, min(m0 + min(r, WBM - 1), p.M - 1))) + dsw;
  }
  const char* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wsrc[i] = (const char*)(p.W + (size_t)(8 * (wave + 8 * i) + drow) * (size_t)p.ldw) + dsw;
  const int na = wave < 4 ? 3 : 2;                           // A instructions of this wave per stage
  auto issue = [&](int kt, int slot) {
    char* sa = smem + slot * WSTAGE;
    const size_t ko = (size_t)kt * 128;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < na) wdma16(asrc[i] + ko, sa + (wave + 8 * i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) wdma16(wsrc[i] + ko, sa + WST_A + (wave + 8 * i) * 1024);
  };

  f32x4 acc[5][4];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue(0, 0);
  if (nk > 1) issue(1, 1);
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's part of stage kt has landed once at most the next stage's DMAs are outstanding
    if (kt + 1 < nk) {
      if (wave < 4) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // stage kt visible; stage kt-1 no longer read by anyone
    if (kt + 2 < nk) issue(kt + 2, slot == 0 ? 2 : slot - 1);        // (kt+2) % 3 == (slot + 2) % 3
    asm volatile("" ::: "memory");
    const char* sa = smem + slot * WSTAGE + (80 * wm + l15) * 128;
    const char* sw = smem + slot * WSTAGE + WST_A + (64 * wn + l15) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = ((kk * 4 + g4) ^ (l15 & 7)) << 4;
      frag xf[5], wf[4];
#pragma unroll
      for (int i = 0; i < 5; ++i) xf[i] = *(const frag*)(sa + i * 16 * 128 + ch);
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *(const frag*)(sw + j * 16 * 128 + ch);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = H16<T>::mfma(wf[j], xf[i], acc[i][j]);
    }
    slot = slot == 2 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every wave has left the ring: it becomes epilogue scratch

  // ---- epilogue: per 16-row tile through a wave-private fp32 image [16][68]; a lane then owns 16 consecutive columns of a row ----
  float* timg = (float*)(smem + wave * (16 * WTP * 4));
  uint32_t seed_lo = 0, seed_hi = 0;
  if (p.d1.thresh | p.d2.thresh) { seed_lo = p.st->seed_lo; seed_hi = p.st->seed_hi; }
  const int er = lane >> 2, ec = lane & 3;                   // row of the tile, 16-column group
  const int n = 64 * wn + 16 * ec;
  float bv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bv[j] = 0.f;
  if (p.bias) { load8(p.bias + n, bv); load8(p.bias + n + 8, bv + 8); }
  // the epilogue operand (residual, else gate) of all of this lane's rows is requested up front: one exposed latency, not five
  const T* const eop = p.residual ? p.residual : p.gate;
  const RowMap& emap = p.residual ? p.r : p.c;
  u32x4 eraw[5][2];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    eraw[i][0] = (u32x4){0u, 0u, 0u, 0u};
    eraw[i][1] = (u32x4){0u, 0u, 0u, 0u};
    const int m = m0 + 80 * wm + 16 * i + er;
    if (eop && m < p.M) {
      const T* pe = eop + row_off(emap, m) + n;
      eraw[i][0] = *(const u32x4*)pe;
      eraw[i][1] = *(const u32x4*)(pe + 8);
    }
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int m = m0 + 80 * wm + 16 * i + er;
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(timg + l15 * WTP + 16 * j + 4 * g4) = acc[i][j];
    if (m0 + 80 * wm + 16 * i >= p.M) break;                 // wave-uniform: tiles wholly beyond M
    float v[16];
    load8(timg + er * WTP + 16 * ec, v);
    load8(timg + er * WTP + 16 * ec + 8, v + 8);
    if (m < p.M) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = wide_act<ACT>(v[j] + bv[j]);
      const long long coff = row_off(p.c, m) + n;
      if (p.gate) {
        float gv[16];
        if (!p.residual) {
          load8((const T*)&eraw[i][0], gv);
          load8((const T*)&eraw[i][1], gv + 8);
        } else {
          load8(p.gate + coff, gv);
          load8(p.gate + coff + 8, gv + 8);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = gv[j] > 0.f ? v[j] * p.gate_scale : 0.f;
      }
      if (p.d1.thresh | p.d2.thresh) {
        const uint32_t idx = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
        float (&v0)[8] = *(float (*)[8])v;
        float (&v1)[8] = *(float (*)[8])(v + 8);
        eg_dropout_run<8>(v0, p.d1, seed_lo, seed_hi, idx);
        eg_dropout_run<8>(v0, p.d2, seed_lo, seed_hi, idx);
        eg_dropout_run<8>(v1, p.d1, seed_lo, seed_hi, idx + 8);
        eg_dropout_run<8>(v1, p.d2, seed_lo, seed_hi, idx + 8);
      }
      if (p.out_pre) {
        T* po = p.out_pre + row_off(p.pm, m) + n;
        store8(po, v);
        store8(po + 8, v + 8);
      }
      if (p.residual) {
        float rv[16];
        load8((const T*)&eraw[i][0], rv);
        load8((const T*)&eraw[i][1], rv + 8);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] += rv[j];
      }
      store8(p.C + coff, v);
      store8(p.C + coff + 8, v + 8);
    }
  }
}

template <typename T>
static int wide_launch(const eg_gemm_desc* d, hipStream_t s) {
  WideNT<T> p;
  p.A = (const T*)d->A; p.W = (const T*)d->W; p.C = (T*)d->C; p.bias = d->bias;
  p.residual = (const T*)d->residual; p.gate = (const T*)d->gate; p.out_pre = (T*)d->out_pre; p.st = d->state;
  p.a = to_rowmap(d->a); p.c = to_rowmap(d->c); p.r = to_rowmap(d->r); p.pm = to_rowmap(d->p);
  p.M = d->M; p.N = d->N; p.K = d->K; p.ldw = d->ldw;
  p.d1 = make_drop(d->drop1_p, d->drop1_site);
  p.d2 = make_drop(d->drop2_p, d->drop2_site);
  p.gate_scale = d->gate_scale == 0.f ? 1.0f : d->gate_scale;
  const dim3 grid((d->M + WBM - 1) / WBM), blk(512);
#define WIDE_LAUNCH(A_)                                                                                                \
  do {                                                                                                                 \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      (void)hipFuncSetAttribute((const void*)gemm_nt_wide_kernel<T, A_>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                WLDS);                                                                                 \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((gemm_nt_wide_kernel<T, A_>), grid, blk, WLDS, s, p);                                           \
  } while (0)
  if (d->act == EG_ACT_RELU) WIDE_LAUNCH(EG_ACT_RELU);
  else if (d->act == EG_ACT_GELU) WIDE_LAUNCH(EG_ACT_GELU);
  else WIDE_LAUNCH(EG_ACT_NONE);
#undef WIDE_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// eligibility + launch; returns -1 when the product does not fit this kernel (caller falls back)
bool eg_wide_gemm_ok(const eg_gemm_desc* d) {
  if ((d->dtype != EG_BF16 && d->dtype != EG_F16) || d->N != WBN || d->K % 64 != 0 || d->K < 128) return false;
  if (d->a_seg_len || !d->C) return false;
  if (d->M < 1024) return false;               // the head products (M = batch) keep the 128x128 tile
  // the epilogue reads residual / gate rows and writes out_pre rows as 16-B vectors: misaligned bases keep the 128x128 tile
  if (((uintptr_t)d->residual | (uintptr_t)d->gate | (uintptr_t)d->out_pre | (uintptr_t)d->C | (uintptr_t)d->A | (uintptr_t)d->W) % 16)
    return false;
  return d->ldw % 8 == 0;
}

int eg_wide_gemm_try(const eg_gemm_desc* d, hipStream_t s) {
  if (!eg_wide_gemm_ok(d)) return -1;
  return d->dtype == EG_F16 ? wide_launch<f16_t>(d, s) : wide_launch<bf16_t>(d, s);
}
