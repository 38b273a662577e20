// Row-stream GEMM for the d_model-deep products of the encoder (K == 256, N a multiple of 256, 16-bit operands):
//   C[M, N] = epilogue(A[M, 256] * W[N, 256]^T)      q|k|v, out-proj, FFN-1 and their backward-data twins
//
// These products sit far below the MFMA ridge (FLOP/B ~ 100-200): the roofline that binds them is HBM, and with only
// 130 rows per CU at the benchmark size a tiled kernel spends its life in load -> barrier -> MFMA -> barrier latency chains
// while every 128x128 tile pulls 4 bytes through the CU's load path per byte it writes.
// Design (gfx950):
//   * weights are REGISTER-stationary: a 512-thread workgroup (8 waves, 2 per SIMD) owns a 256-column slice of W; wave w keeps
//     its 32 columns x 256 k as MFMA operands in 64 VGPRs for the whole launch -- W costs no LDS bandwidth and leaves L2 once
//     per workgroup;
//   * one workgroup per CU walks a contiguous range of 16-row blocks of A, two blocks (32 rows) per iteration; ranges are
//     balanced to +-1 block so all CUs finish together whatever M is.  Rows arrive by LDS-DMA (global_load_lds, 16 B per lane)
//     into a 4-pair ring three pairs ahead of their use, the residual / gate rows into a second ring two pairs ahead;
//   * the XOR swizzle of the rings lives on the DMA's per-lane SOURCE address (the LDS side of a DMA is lane-linear); the
//     matching XOR on the reads makes the ds_read_b128 (fragments) and ds_read_b64 (epilogue operand) conflict-free;
//   * ONE raw s_barrier per 32 rows; waits are counted s_waitcnt vmcnt(N) -- loads, DMAs and stores share the counter in issue
//     order, so the kernel keeps its own tally of issued vector-memory instructions, and row index clamping (never
//     predication) keeps that tally exact;
//   * the epilogue is wave-local: accumulators -> bias / ReLU / gate / dropout / residual in the MFMA layout -> a 2.5 KB
//     wave-private LDS image -> 16-B stores (16 rows x 64 B per instruction);
//   * software pipeline: between two barriers a wave runs the MFMAs of pair t and the epilogue of pair t-1 (two accumulator
//     sets); waves 4-7 (the SIMD partners of waves 0-3) take these two independent halves in the opposite order, so one
//     wave's matrix burst sits beside its partner's vector / LDS / store work.
// Arithmetic is the same k-ordered MFMA chain as gemm_nt_kernel, so results are bit-identical to it.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int RS_RA = 4;                    // A ring depth (pairs of row blocks)
constexpr int RS_RE = 4;                    // epilogue-operand ring depth (pairs)
constexpr int RS_PAIR = 32 * 512;           // one pair: 32 rows x 256 16-bit elements
constexpr int RS_TP = 80;                   // pitch (bytes) of the wave-private output image: 32 x 16-bit + 16
constexpr int RS_TW = 32 * RS_TP;           // one wave's image: 32 rows
constexpr int RS_LDS = (RS_RA + RS_RE) * RS_PAIR + 8 * RS_TW;   // 149,504 B

template <typename T>
struct RsGemm {
  const T* A; const T* W; T* C; const float* bias; const T* E; T* out_pre;
  const eg_step_state* st;
  int lda, ldc, lde, ldp;                   // row strides (elements); M * stride < 2^32 is checked by the host
  int M, N, ldw;
  int nblk, groups, ns;                     // 16-row blocks, row groups, 256-column slices
  DropCfg d1, d2;
  float gate_scale;
};

__device__ __forceinline__ void dma16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the field is an immediate); n is rounded DOWN (= waits for more)
__device__ __forceinline__ void wait_vmcnt(int n) {
  n = __builtin_amdgcn_readfirstlane(n) >> 1;      // even counts only: at most one more instruction is waited for
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
  }
}

// 4 floats -> 4 elements of the 16-bit storage type (two packed converts)
template <typename T> __device__ __forceinline__ u32x2 rs_pack4(const float v[4]);
template <> __device__ __forceinline__ u32x2 rs_pack4<bf16_t>(const float v[4]) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const bf16x2 lo = __builtin_convertvector((f32x2){v[0], v[1]}, bf16x2);
  const bf16x2 hi = __builtin_convertvector((f32x2){v[2], v[3]}, bf16x2);
  u32x2 o;
  o[0] = __builtin_bit_cast(uint32_t, lo);
  o[1] = __builtin_bit_cast(uint32_t, hi);
  return o;
}
template <> __device__ __forceinline__ u32x2 rs_pack4<f16_t>(const float v[4]) {
  u32x2 o;
  o[0] = pack2h(v[0], v[1]);
  o[1] = pack2h(v[2], v[3]);
  return o;
}

// EMODE: 0 no epilogue operand, 1 residual (added last), 2 gate (ReLU backward: zero where gate <= 0)
// RELU:  ReLU on acc + bias;  DROP: the dropout sites of the descriptor are live (each still checks its own threshold)
template <typename T, int EMODE, int RELU, int DROP>
__global__ __launch_bounds__(512, 2) void rs_gemm_kernel(RsGemm<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ringA = smem;
  char* const ringE = smem + RS_RA * RS_PAIR;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // owns columns 32*wave .. +31 of the slice, all 32 rows of a pair
  const int l15 = lane & 15, g4 = lane >> 4;
  char* const timg = smem + (RS_RA + RS_RE) * RS_PAIR + wave * RS_TW;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);          // neighbours (same rows, other column slices) share an XCD / L2
  const int rg = bid / p.ns, slice = bid - rg * p.ns;
  const int b0 = (int)((long long)rg * p.nblk / p.groups);
  const int nb = (int)((long long)(rg + 1) * p.nblk / p.groups) - b0;
  const int npairs = (nb + 1) >> 1;
  const int n0 = slice * 256;
  const int row0 = b0 * 16;

  // ---- DMA addressing: per pair a wave moves rows 4w .. 4w+3 (two instructions of 2 rows x 512 B); LDS position `pos` of
  //      row r holds global 16-B chunk pos ^ (r & 15) ----
  const int dpos = lane & 31;
  int vm_issued = 0;                                         // vector-memory instructions this wave has issued so far
  auto issueA = [&](int pair, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 4 * wave + 2 * i + (lane >> 5);          // row within the pair
      const int row = min(row0 + pair * 32 + r, p.M - 1);
      dma16((const char*)(p.A + (size_t)((uint32_t)row * (uint32_t)p.lda)) + ((dpos ^ (r & 15)) << 4),
            ringA + slot * RS_PAIR + (4 * wave + 2 * i) * 512);
    }
    vm_issued += 2;
  };
  auto issueE = [&](int pair, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 4 * wave + 2 * i + (lane >> 5);
      const int row = min(row0 + pair * 32 + r, p.M - 1);
      dma16((const char*)(p.E + (size_t)((uint32_t)row * (uint32_t)p.lde) + n0) + ((dpos ^ (r & 15)) << 4),
            ringE + slot * RS_PAIR + (4 * wave + 2 * i) * 512);
    }
    vm_issued += 2;
  };

  // ---- prologue: the first pair goes out before the weights (the first MFMA needs both; a wait for an ordinary load also
  //      drains every DMA issued before it, so the rest of the ring is issued after that wait) ----
  if (npairs > 0) {
    issueA(0, 0);
    if (EMODE) issueE(0, 0);
  }
  asm volatile("" ::: "memory");

  // register-stationary weights: this wave's 32 columns as MFMA A-operands (rows of W)
  typedef typename H16<T>::frag frag;
  frag wf[2][8];
  {
    const T* wrow = p.W + (size_t)(n0 + 32 * wave + l15) * (size_t)p.ldw + 8 * g4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 8; ++s) wf[j][s] = *(const frag*)(wrow + (size_t)(16 * j) * (size_t)p.ldw + 32 * s);
  }
  // per-lane epilogue constants: accumulator register q of tile j is column n0 + 32w + 16j + 4*g4 + q of row l15
  const int ncol = n0 + 32 * wave + 4 * g4;
  float bv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[j][q] = 0.f;
    if (p.bias) load4(p.bias + ncol + 16 * j, bv[j]);
  }
  uint32_t seed_lo = 0, seed_hi = 0;
  if (DROP) { seed_lo = p.st->seed_lo; seed_hi = p.st->seed_hi; }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // weights in registers, pair 0 in the ring (this wave's part)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(wf[j][s]));
  // the rest of the rings goes in flight (A: pairs 1, 2; epilogue operand: pair 1); the tally of issued instructions starts here
  vm_issued = 0;
  if (1 < npairs) {
    issueA(1, 1);
    if (EMODE) issueE(1, 1);
  }
  const int mark_a = vm_issued;
  if (2 < npairs) issueA(2, 2);
  int mkA2 = mark_a, mkA1 = vm_issued, mkE1 = mark_a;        // tallies right after the request of pair t+1 (A) / t+2 (A) / t+1 (E)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // pair 0 is visible to every wave

  // epilogue of one pair from the accumulators `a` (wave-local: bias / ReLU / gate / dropout / residual in the MFMA layout,
  // then through the wave's 32 x 32 LDS image so that a lane stores 16 B and an instruction covers 16 rows x 64 B)
  auto epilogue = [&](int pair, const f32x4 (&a)[2][2]) {
    const int slotE = pair % RS_RE;
    const int nvalid = min(2, nb - 2 * pair);
    // rows this lane stores after the transposition: lane/4 of each block, 16 B at column 8*(lane%4)
    const int sr = lane >> 2, sc = lane & 3;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (b >= nvalid) break;                                // the odd tail of the range has no second block
      const int r = 16 * b + l15;                            // row within the pair (MFMA layout)
      const uint32_t mrow = (uint32_t)min(row0 + pair * 32 + r, p.M - 1);
      const uint32_t drow = mrow * (uint32_t)p.N;
      const uint32_t ms = (uint32_t)min(row0 + pair * 32 + 16 * b + sr, p.M - 1);
      float v[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[j][q] = a[b][j][q] + bv[j][q];
          if (RELU) v[j][q] = fmaxf(v[j][q], 0.f);
        }
        if (EMODE == 2) {
          float ev[4];
          const int ch = (4 * wave + 2 * j + (g4 >> 1)) ^ (r & 15);
          load4((const T*)(ringE + slotE * RS_PAIR + r * 512 + (ch << 4) + ((g4 & 1) << 3)), ev);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[j][q] = ev[q] > 0.f ? v[j][q] * p.gate_scale : 0.f;
        }
        if (DROP) {
          const uint32_t idx = drow + (uint32_t)(ncol + 16 * j);
          eg_dropout_run<4>(v[j], p.d1, seed_lo, seed_hi, idx);
          eg_dropout_run<4>(v[j], p.d2, seed_lo, seed_hi, idx);
        }
      }
      char* const tb = timg + b * (16 * RS_TP);
      if (p.out_pre) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *(u32x2*)(tb + l15 * RS_TP + (16 * j + 4 * g4) * 2) = rs_pack4<T>(v[j]);
        const u32x4 o = *(const u32x4*)(tb + sr * RS_TP + sc * 16);
        *(u32x4*)(p.out_pre + (size_t)(ms * (uint32_t)p.ldp) + n0 + 32 * wave + 8 * sc) = o;
        vm_issued += 1;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (EMODE == 1) {
          float ev[4];
          const int ch = (4 * wave + 2 * j + (g4 >> 1)) ^ (r & 15);
          load4((const T*)(ringE + slotE * RS_PAIR + r * 512 + (ch << 4) + ((g4 & 1) << 3)), ev);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[j][q] += ev[q];
        }
        *(u32x2*)(tb + l15 * RS_TP + (16 * j + 4 * g4) * 2) = rs_pack4<T>(v[j]);
      }
      const u32x4 o = *(const u32x4*)(tb + sr * RS_TP + sc * 16);
      *(u32x4*)(p.C + (size_t)(ms * (uint32_t)p.ldc) + n0 + 32 * wave + 8 * sc) = o;
      vm_issued += 1;
    }
  };
  auto frags_mma = [&](int slot, f32x4 (&a)[2][2]) {
    // fragments of the pair (swizzled rows), 32 MFMAs: D[n][m] = sum_k W[n][k] X[m][k]
    frag xf[2][8];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const char* ab = ringA + slot * RS_PAIR + (16 * b + l15) * 512;
#pragma unroll
      for (int s = 0; s < 8; ++s) xf[b][s] = *(const frag*)(ab + (((4 * s + g4) ^ l15) << 4));
    }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 2; ++j) a[b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) a[b][j] = H16<T>::mfma(wf[j][s], xf[b][s], a[b][j]);
  };

  // Software pipeline: between two barriers a wave runs the MFMAs of pair t AND the epilogue of pair t-1 (two accumulator
  // sets, roles swapped every step), so the matrix pipe works while the vector / LDS / store side of the previous pair drains.
  f32x4 accP[2][2], accQ[2][2];
  int slotA = 0;
  auto step = [&](int t, f32x4 (&cur)[2][2], f32x4 (&prev)[2][2]) {
    if (t + 3 < npairs) issueA(t + 3, (t + 3) % RS_RA);
    if (EMODE && t + 2 < npairs) issueE(t + 2, (t + 2) % RS_RE);
    const int mk0 = vm_issued;
    asm volatile("" ::: "memory");
    // waves 4-7 are the SIMD partners of waves 0-3: they take the two independent halves of a step in the opposite order, so
    // one wave's MFMA burst runs beside its partner's vector / LDS / store work instead of beside another MFMA burst
    if (wave < 4) {
      frags_mma(slotA, cur);
      if (t > 0) epilogue(t - 1, prev);
    } else {
      if (t > 0) epilogue(t - 1, prev);
      frags_mma(slotA, cur);
    }
    asm volatile("" ::: "memory");
    // pair t+1 (and its epilogue operand) must have landed before the barrier that publishes it
    if (t + 1 < npairs) wait_vmcnt(vm_issued - (EMODE ? mkE1 : mkA2));
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    mkA2 = mkA1;
    mkA1 = mk0;
    mkE1 = mk0;
    slotA = slotA + 1 == RS_RA ? 0 : slotA + 1;
  };
  for (int t = 0; t < npairs; t += 2) {
    step(t, accP, accQ);
    if (t + 1 < npairs) step(t + 1, accQ, accP);
  }
  if (npairs > 0) {
    if (npairs & 1) epilogue(npairs - 1, accP); else epilogue(npairs - 1, accQ);
  }
}

}  // namespace

// eligibility + launch; returns -1 when the product does not fit this kernel (caller falls back to gemm_nt)
template <typename T>
static int rs_gemm_launch(const eg_gemm_desc* d, hipStream_t s, int cus) {
  RsGemm<T> p;
  p.A = (const T*)d->A; p.W = (const T*)d->W; p.C = (T*)d->C; p.bias = d->bias;
  p.E = (const T*)(d->residual ? d->residual : d->gate);
  p.out_pre = (T*)d->out_pre; p.st = d->state;
  p.lda = (int)d->a.row_stride; p.ldc = (int)d->c.row_stride; p.lde = (int)(d->residual ? d->r.row_stride : d->c.row_stride);
  p.ldp = (int)d->p.row_stride;
  p.M = d->M; p.N = d->N; p.ldw = d->ldw;
  const int e_mode = d->residual ? 1 : (d->gate ? 2 : 0);
  p.nblk = (d->M + 15) / 16;
  p.ns = d->N / 256;
  p.groups = cus / p.ns > 0 ? cus / p.ns : 1;
  if (p.groups > p.nblk) p.groups = p.nblk;
  p.d1 = make_drop(d->drop1_p, d->drop1_site);
  p.d2 = make_drop(d->drop2_p, d->drop2_site);
  p.gate_scale = d->gate_scale == 0.f ? 1.0f : d->gate_scale;
  const dim3 grid(p.groups * p.ns), blk(512);
#define RS_LAUNCH(E_, R_, D_)                                                                                          \
  do {                                                                                                                 \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      (void)hipFuncSetAttribute((const void*)rs_gemm_kernel<T, E_, R_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                RS_LDS);                                                                               \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((rs_gemm_kernel<T, E_, R_, D_>), grid, blk, RS_LDS, s, p);                                      \
  } while (0)
#define RS_PICK(E_)                                                                                                    \
  do {                                                                                                                 \
    if (relu) { if (drop) RS_LAUNCH(E_, 1, 1); else RS_LAUNCH(E_, 1, 0); }                                             \
    else      { if (drop) RS_LAUNCH(E_, 0, 1); else RS_LAUNCH(E_, 0, 0); }                                             \
  } while (0)
  const bool relu = d->act == EG_ACT_RELU, drop = (p.d1.thresh | p.d2.thresh) != 0;
  if (e_mode == 1) RS_PICK(1); else if (e_mode == 2) RS_PICK(2); else RS_PICK(0);
#undef RS_PICK
#undef RS_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

bool eg_rs_gemm_ok(const eg_gemm_desc* d) {
  if ((d->dtype != EG_BF16 && d->dtype != EG_F16) || d->K != 256 || d->N % 256 != 0 || d->ldw != 256) return false;
  if (d->a.rows_per_group || d->c.rows_per_group || d->r.rows_per_group || d->p.rows_per_group) return false;
  if (d->a_seg_len || d->act == EG_ACT_GELU) return false;
  if (d->residual && d->gate) return false;
  if (!d->C) return false;
  if (d->a.row_stride % 8 || d->c.row_stride % 8 || (d->residual && d->r.row_stride % 8) || (d->out_pre && d->p.row_stride % 8)) return false;
  {
    const long long lim = 1ll << 32, M = d->M;
    if (M * d->a.row_stride >= lim || M * d->c.row_stride >= lim || (d->residual && M * d->r.row_stride >= lim) ||
        (d->out_pre && M * d->p.row_stride >= lim)) return false;
  }
  if (((uintptr_t)d->A | (uintptr_t)d->W | (uintptr_t)d->C | (uintptr_t)d->residual | (uintptr_t)d->gate | (uintptr_t)d->out_pre) % 16) return false;
  return true;
}

int eg_rs_gemm_try(const eg_gemm_desc* d, hipStream_t s) {
  if (!eg_rs_gemm_ok(d)) return -1;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return d->dtype == EG_F16 ? rs_gemm_launch<f16_t>(d, s, cus) : rs_gemm_launch<bf16_t>(d, s, cus);
}
