// Deep-K NT GEMM for the products that sit ABOVE the MFMA ridge (N == 256, K >= 1536, 16-bit operands): the two strided 1-D
// convolutions as overlapping-row GEMMs -- conv-1 forward (K = 6400) and its four backward-data phases (K = 1792) at the
// benchmark size -- which the 160 x 256 wide tile runs at 0.78-0.87 PFLOP/s = a third of the dense bf16 peak.
//
// Why the wide tile stalls there: both operands go through LDS (0.45 KB of fragment reads per MFMA plus the weight tile's
// DMA writes: the LDS pipe is busier than the matrix pipe), and its eight waves meet at one barrier per 64-deep K step with
// only two further steps of loads in flight.
// Here (gfx950), one 256-thread workgroup per CU, FOUR waves = one per SIMD, each owning all rows x 64 columns:
//   * WEIGHTS NEVER TOUCH LDS: they are pre-packed in MFMA-fragment order (eg_frag_order_rows), so a wave's four fragments of
//     a 32-deep k-step are one contiguous 4-KB read straight into registers, through a ring TWO K-steps deep; no wave loads a
//     fragment another wave also loads;
//   * LDS carries only the activation rows: a SIX-stage LDS-DMA ring of [16 NT rows x 128 B] (NT = 8, 9 or 10 row tiles per
//     workgroup, chosen so that one round of workgroups covers M), 0.25-0.31 KB of fragment reads per MFMA;
//   * one raw barrier per K-step and NO hand-counted vmcnt: a wave's wait for the weight fragments of step k (requested two
//     steps ago, AFTER the activation DMAs of stage k+3 in its in-order vector-memory queue) also proves its share of stage
//     k+1 has landed; the barrier then publishes it;
//   * accumulators NT x 4 tiles (160 VGPRs at NT = 10); the fragments of the next 32-deep half-step -- across the K-step
//     boundary too: the next stage is already visible -- are read while the current one is multiplied.
// Arithmetic is the same k-ordered MFMA chain and epilogue as gemm_nt_kernel / gemm_nt_wide_kernel: bit-identical results.
//
// MEASURED (MI355X, conv-1 forward 32 768 x 256 x 6400 / a backward-data phase 35 840 x 256 x 1792): 128 / 52 us against the wide
// tile's 124 / 42 us -- no gain, so the engine does not hand it the fragment-ordered weights by default (EYEGAZE_TALL_CONV=1
// opts in).  With every memory operation of the K loop switched off the compiler-scheduled loop of 64 MFMAs + 16 ds_read_b128
// per wave and K-step still runs at about half the MFMA issue rate: one wave per SIMD has nobody to cover its own
// s_waitcnt / barrier gaps, which two lock-stepped waves of the wide tile at least halve.  Kept as the parity-tested starting
// point for a hand-scheduled K loop.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int T_RD = 6;                       // activation ring depth (stages of 64 k)
constexpr int T_WD = 2;                       // weight fragment ring depth (K-steps)
constexpr int T_TP = 68;                      // fp32 image pitch of the epilogue (floats)

template <typename T>
struct TallNT {
  const T* A; const T* Wf; T* C; const float* bias; const T* residual; const T* gate; T* out_pre;
  const eg_step_state* st;
  RowMap a, c, r, pm;
  int M, N, K;
  DropCfg d1, d2;
  float gate_scale;
};

__device__ __forceinline__ void tdma16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int ACT>
__device__ __forceinline__ float tall_act(float v) {
  if (ACT == EG_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

template <typename T, int NT, int ACT>
__global__ __launch_bounds__(256, 1) void gemm_nt_tall_kernel(TallNT<T> p) {
  typedef typename H16<T>::frag frag;
  constexpr int ROWS = 16 * NT;
  constexpr int STAGE = ROWS * 128;           // bytes per ring stage
  constexpr int NDMA = 2 * NT;                // 1-KB DMA instructions per stage (8 rows x 128 B each)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  const int m0 = blockIdx.x * ROWS;
  const int nk = p.K >> 6;

  // ---- activation DMA: instruction q moves rows 8q .. 8q+7 (lane -> row lane/8, LDS chunk position lane%8 holding global chunk
  //      pos ^ (row & 7)); wave w issues q = w, w+4, .. ----
  const int drow = lane >> 3, dpos = lane & 7;
  const int dsw = (dpos ^ drow) << 4;
  constexpr int QPW = (NDMA + 3) / 4;         // instructions per wave (the last one may not exist for every wave)
  const char* asrc[QPW];
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int r = 8 * min(wn + 4 * i, NDMA - 1) + drow;
    asrc[i] = (const char*)(p.A + row_off(p.a, min(m0 + r, p.M - 1))) + dsw;
  }
  // branch-free: past the end of K the last stage is fetched again (into a slot nobody reads any more), and where NDMA is not a
  // multiple of 4 the surplus instruction of waves 2-3 repeats the last row group (same bytes to the same place)
  auto issue_a = [&](int kt) {
    char* sa = smem + (kt % T_RD) * STAGE;
    const size_t ko = (size_t)min(kt, nk - 1) * 128;
#pragma unroll
    for (int i = 0; i < QPW; ++i) tdma16(asrc[i] + ko, sa + min(wn + 4 * i, NDMA - 1) * 1024);
  };
  // ---- weight fragments: [k-step s of 32][wave][j][lane][8] ----
  const T* const wp = p.Wf + (size_t)wn * (4 * 512) + lane * 8;
  frag wr[T_WD][2][4];
  auto issue_w = [&](int kt, int slot) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j) wr[slot][kk][j] = *(const frag*)(wp + (size_t)(2 * min(kt, nk - 1) + kk) * (4 * 4 * 512) + j * 512);
  };

  f32x4 acc[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: stages 0 .. RD-2 and weight steps 0 .. WD-1
#pragma unroll
  for (int s = 0; s < T_RD - 1; ++s) issue_a(s);
  asm volatile("" ::: "memory");
#pragma unroll
  for (int s = 0; s < T_WD; ++s) issue_w(s, s);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int sw7 = l15 & 7;
  // fragments of the first 32-deep half of the CURRENT stage: read during the previous K-step (stage kt + 1 is already visible
  // in step kt: by the barrier that ended step kt - 1 every wave had waited for its W(kt - 1) fragments, which it requested
  // after its DMAs of stage kt + 1)
  frag xa[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) xa[i] = *(const frag*)(smem + (16 * i + l15) * 128 + ((g4 ^ sw7) << 4));
  auto kstep = [&](int kt, int slot) {
    // (a) the activation stage RD-1 steps ahead, into the slot whose stage (kt - 1) everybody left at the last barrier
    issue_a(kt + T_RD - 1);
    asm volatile("" ::: "memory");
    const char* sa = smem + (kt % T_RD) * STAGE + l15 * 128;
    const char* sn = smem + ((kt + 1) % T_RD) * STAGE + l15 * 128;     // (past the end: a stale but valid stage, never multiplied)
    frag xb[NT], xn[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) xb[i] = *(const frag*)(sa + i * 16 * 128 + (((4 + g4) ^ sw7) << 4));
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = H16<T>::mfma(wr[slot][0][j], xa[i], acc[i][j]);
#pragma unroll
    for (int i = 0; i < NT; ++i) xn[i] = *(const frag*)(sn + i * 16 * 128 + ((g4 ^ sw7) << 4));
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = H16<T>::mfma(wr[slot][1][j], xb[i], acc[i][j]);
    // (b) this slot's fragments are consumed: refill it WD steps ahead (after (a) in the vector-memory queue)
    issue_w(kt + T_WD, slot);
#pragma unroll
    for (int i = 0; i < NT; ++i) xa[i] = xn[i];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  for (int kt = 0; kt < nk; kt += T_WD) {
#pragma unroll
    for (int u = 0; u < T_WD; ++u) kstep(kt + u, u);
  }

  // ---- epilogue: per 16-row tile through a wave-private fp32 image [16][68] in the drained ring; a lane then owns 16
  //      consecutive columns of a row (same order of operations as gemm_nt_kernel) ----
  float* timg = (float*)(smem + wn * (16 * T_TP * 4));
  uint32_t seed_lo = 0, seed_hi = 0;
  if (p.d1.thresh | p.d2.thresh) { seed_lo = p.st->seed_lo; seed_hi = p.st->seed_hi; }
  const int er = lane >> 2, ec = lane & 3;
  const int n = 64 * wn + 16 * ec;
  float bv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bv[j] = 0.f;
  if (p.bias) { load8(p.bias + n, bv); load8(p.bias + n + 8, bv + 8); }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int m = m0 + 16 * i + er;
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(timg + l15 * T_TP + 16 * j + 4 * g4) = acc[i][j];
    if (m0 + 16 * i >= p.M) break;                           // wave-uniform: tiles wholly beyond M
    float v[16];
    load8(timg + er * T_TP + 16 * ec, v);
    load8(timg + er * T_TP + 16 * ec + 8, v + 8);
    if (m < p.M) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = tall_act<ACT>(v[j] + bv[j]);
      const long long coff = row_off(p.c, m) + n;
      if (p.gate) {
        float gv[16];
        load8(p.gate + coff, gv);
        load8(p.gate + coff + 8, gv + 8);
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
        const T* pr = p.residual + row_off(p.r, m) + n;
        load8(pr, rv);
        load8(pr + 8, rv + 8);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] += rv[j];
      }
      store8(p.C + coff, v);
      store8(p.C + coff + 8, v + 8);
    }
  }
}

template <typename T, int NT>
static int tall_launch_nt(const TallNT<T>& p, int act, hipStream_t s) {
  constexpr int lds = T_RD * 16 * NT * 128;
  const dim3 grid((p.M + 16 * NT - 1) / (16 * NT)), blk(256);
#define TALL_LAUNCH(A_)                                                                                                \
  do {                                                                                                                 \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      (void)hipFuncSetAttribute((const void*)gemm_nt_tall_kernel<T, NT, A_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((gemm_nt_tall_kernel<T, NT, A_>), grid, blk, lds, s, p);                                        \
  } while (0)
  if (act == EG_ACT_RELU) TALL_LAUNCH(EG_ACT_RELU); else TALL_LAUNCH(EG_ACT_NONE);
#undef TALL_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename T>
static int tall_launch(const eg_gemm_desc* d, hipStream_t s, int cus) {
  TallNT<T> p;
  p.A = (const T*)d->A; p.Wf = (const T*)d->W_frag; p.C = (T*)d->C; p.bias = d->bias;
  p.residual = (const T*)d->residual; p.gate = (const T*)d->gate; p.out_pre = (T*)d->out_pre; p.st = d->state;
  p.a = to_rowmap(d->a); p.c = to_rowmap(d->c); p.r = to_rowmap(d->r); p.pm = to_rowmap(d->p);
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.d1 = make_drop(d->drop1_p, d->drop1_site);
  p.d2 = make_drop(d->drop2_p, d->drop2_site);
  p.gate_scale = d->gate_scale == 0.f ? 1.0f : d->gate_scale;
  // row tiles per workgroup: the fewest (8..10) for which one round of workgroups covers M; larger M runs several rounds of 10
  const int tiles = (d->M + 15) / 16;
  int nt = (tiles + cus - 1) / cus;
  nt = nt < 8 ? 8 : (nt > 10 ? 10 : nt);
  if (nt == 8) return tall_launch_nt<T, 8>(p, d->act, s);
  if (nt == 9) return tall_launch_nt<T, 9>(p, d->act, s);
  return tall_launch_nt<T, 10>(p, d->act, s);
}

// [N = 256, ldw] row-major 16-bit weight -> fragment order [k-step of 32][wave 0..3][j 0..3][lane][8]
__global__ __launch_bounds__(256) void frag_order_rows_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int K,
                                                              int ldw) {
  const int q = blockIdx.x * 256 + threadIdx.x;               // destination 16-B chunk
  if (q >= 256 * K / 8) return;
  src += (size_t)blockIdx.y * 256 * ldw;                      // matrix blockIdx.y of a stack
  dst += (size_t)blockIdx.y * 256 * K;
  const int lane = q & 63, j = (q >> 6) & 3, wn = (q >> 8) & 3, s = q >> 10;
  const int nrow = 64 * wn + 16 * j + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
  *(u32x4*)(dst + (size_t)q * 8) = *(const u32x4*)(src + (size_t)nrow * ldw + k0);
}

}  // namespace

bool eg_tall_gemm_ok(const eg_gemm_desc* d) {
  if ((d->dtype != EG_BF16 && d->dtype != EG_F16) || d->N != 256 || !d->W_frag) return false;
  if (d->K < 1536 || d->K % (64 * T_WD) != 0 || d->M < 2048) return false;
  if (d->a_seg_len || d->ln_mode || d->row_tile || !d->C || d->act == EG_ACT_GELU) return false;
  return true;
}

int eg_tall_gemm_try(const eg_gemm_desc* d, hipStream_t s) {
  if (!eg_tall_gemm_ok(d)) return -1;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  return d->dtype == EG_F16 ? tall_launch<f16_t>(d, s, cus) : tall_launch<bf16_t>(d, s, cus);
}

extern "C" int eg_frag_order_rows(const void* src, void* dst, int K, int ldw, int count, void* stream) {
  EG_CHECK(src && dst && K > 0 && K % 32 == 0 && ldw >= K && ldw % 8 == 0 && count > 0, "eg_frag_order_rows: K=%d ldw=%d count=%d", K,
           ldw, count);
  EG_CHECK(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "eg_frag_order_rows: alignment");
  const int chunks = 256 * K / 8;
  hipLaunchKernelGGL(frag_order_rows_kernel, dim3((chunks + 255) / 256, count), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)src, (uint16_t*)dst, K, ldw);
  EG_LAUNCH_CHECK("frag_order_rows");
  return 0;
}
