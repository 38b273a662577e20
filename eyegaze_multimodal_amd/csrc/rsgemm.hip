// Row-stream GEMM for the d_model-deep products of the encoder (K == 256, N a multiple of 256, 16-bit operands):
//   C[M, N] = epilogue(A[M, 256] * W[N, 256]^T)      q|k|v, out-proj, FFN-1 and their backward-data twins
//
// These products sit far below the MFMA ridge (FLOP/B ~ 100-200): the roofline that binds them is HBM, and with only
// 130 rows per CU at the benchmark size a tiled kernel spends its life in load -> barrier -> MFMA -> barrier latency chains.
// Design (gfx950):
//   * weights are REGISTER-stationary: a 512-thread workgroup (8 waves, 2 per SIMD) owns a 256-column slice of W; wave w keeps
//     its 32 columns x 256 k as MFMA operands in 64 VGPRs for the whole launch -- W costs no LDS bandwidth at all;
//   * one workgroup per CU walks a contiguous range of 16-row blocks of A (ranges balanced to +-1 block, so all 256 CUs finish
//     together whatever M is); row blocks arrive by LDS-DMA (global_load_lds, 16 B per lane) into a 7-deep ring, so ~56 KB of
//     activation rows (and as many residual / gate rows) are in flight per CU at all times and HBM never idles behind a barrier;
//   * the XOR swizzle of the ring lives on the DMA's per-lane SOURCE address (the LDS side of a DMA is lane-linear), the matching
//     XOR on the fragment reads makes the ds_read_b128 conflict-free;
//   * one raw s_barrier per row block; waits are counted s_waitcnt vmcnt(N) (loads, DMAs and stores share the counter in issue
//     order, so the kernel keeps its own tally of issued vector-memory instructions);
//   * the epilogue goes through a double-buffered fp32 LDS image so every global access is a whole 512-B row segment; with
//     N == 256 a half-wave owns a complete row and LayerNorm (A:293) runs right there (ln_mode 1).
// Arithmetic is the same k-ordered MFMA chain as gemm_nt_kernel, so results are bit-identical to it.
#include "common.h"

namespace {

constexpr int RS_R = 7;                     // ring depth (row blocks)
constexpr int RS_BLK = 16 * 512;            // one row block: 16 rows x 256 16-bit elements
constexpr int RS_SP = 260;                  // fp32 staging pitch (floats)
constexpr int RS_STAGE = 16 * RS_SP * 4;    // one staging image
constexpr int RS_LDS = 2 * RS_R * RS_BLK + 2 * RS_STAGE;   // 147,968 B

struct RsGemm {
  const bf16_t* A; const bf16_t* W; bf16_t* C; const float* bias; const bf16_t* E; bf16_t* out_pre;
  const eg_step_state* st;
  long long lda, ldc, lde, ldp;             // row strides (elements)
  int M, N, ldw, act, e_mode;               // e_mode: 0 none, 1 residual (added last), 2 gate (ReLU backward)
  int nblk, groups, ns;                     // 16-row blocks, row groups, 256-column slices
  DropCfg d1, d2;
  float gate_scale;
  int ln_mode; const float* gamma; const float* beta; float* stats; bf16_t* ln_out;
};

__device__ __forceinline__ void dma16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the field is an immediate); n is clamped DOWN to 31 = waits for more
#define RS_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vmcnt(int n) {
  n = __builtin_amdgcn_readfirstlane(n);
  switch (n) {
    RS_W(0) RS_W(1) RS_W(2) RS_W(3) RS_W(4) RS_W(5) RS_W(6) RS_W(7) RS_W(8) RS_W(9) RS_W(10) RS_W(11) RS_W(12) RS_W(13)
    RS_W(14) RS_W(15) RS_W(16) RS_W(17) RS_W(18) RS_W(19) RS_W(20) RS_W(21) RS_W(22) RS_W(23) RS_W(24) RS_W(25) RS_W(26)
    RS_W(27) RS_W(28) RS_W(29) RS_W(30)
    default: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
  }
}
#undef RS_W

__device__ __forceinline__ float rs_act(float v, int act) {
  if (act == EG_ACT_RELU) return fmaxf(v, 0.f);
  if (act == EG_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  return v;
}
__device__ __forceinline__ float rs_half_sum32(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int EMODE, int LN>
__global__ __launch_bounds__(512, 2) void rs_gemm_kernel(RsGemm p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ringA = smem;
  char* const ringE = smem + RS_R * RS_BLK;
  float* const stage = (float*)(smem + 2 * RS_R * RS_BLK);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);          // neighbours (same rows, other column slices) share an XCD / L2
  const int rg = bid / p.ns, slice = bid - rg * p.ns;
  const int b0 = (int)((long long)rg * p.nblk / p.groups);
  const int nb = (int)((long long)(rg + 1) * p.nblk / p.groups) - b0;
  const int n0 = slice * 256;

  // ---- DMA source addressing: this lane moves 16 B of row (2*wave + lane/32) of every block; chunk position pos = lane%32 of
  //      the LDS row holds global chunk pos ^ (row & 15) (A ring) or pos (E ring) ----
  const int drow = 2 * wave + (lane >> 5), dpos = lane & 31;
  const int dqA = dpos ^ (drow & 15);
  int vm_issued = 0;                                         // vector-memory instructions this wave has issued so far
  int mark[RS_R];                                            // vm_issued right after the last DMA of the block in ring slot u
#pragma unroll
  for (int u = 0; u < RS_R; ++u) mark[u] = 0;

  auto issueA = [&](int item, int slot) {
    const int row = min((b0 + item) * 16 + drow, p.M - 1);
    dma16((const char*)(p.A + (long long)row * p.lda) + dqA * 16, ringA + slot * RS_BLK + wave * 1024);
    vm_issued += 1;
  };
  auto issueE = [&](int item, int slot) {
    const int row = min((b0 + item) * 16 + drow, p.M - 1);
    dma16((const char*)(p.E + (long long)row * p.lde + n0) + dpos * 16, ringE + slot * RS_BLK + wave * 1024);
    vm_issued += 1;
  };

  // ---- prologue: two row blocks (and their epilogue operands) first, then the weights (L2 hits), then the rest of the ring.
  //      The first MFMA needs the weights, and a wait for an ordinary load also drains every DMA issued before it, so only the
  //      blocks the first two iterations consume are issued ahead of the weight loads. ----
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (j < nb) {
      issueA(j, j);
      if (EMODE) issueE(j, j);
    }
  asm volatile("" ::: "memory");

  // ---- register-stationary weights: wave w owns columns n0 + 32w .. +31 ----
  bf16x8 wf[2][8];
  {
    const bf16_t* wrow = p.W + (long long)(n0 + 32 * wave + l15) * p.ldw + 8 * g4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 8; ++s) wf[j][s] = *(const bf16x8*)(wrow + (long long)(16 * j) * p.ldw + 32 * s);
  }

  // ---- per-thread epilogue constants: thread owns row (tid/32) of a block and 8 consecutive columns ----
  const int er = tid >> 5, ec = tid & 31;
  const int n = n0 + ec * 8;
  float bv[8], gam[8], bet[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bv[j] = 0.f; gam[j] = 1.f; bet[j] = 0.f; }
  if (p.bias) load8(p.bias + n, bv);
  if (LN) { load8(p.gamma + n, gam); load8(p.beta + n, bet); }
  uint32_t seed_lo = 0, seed_hi = 0;
  if (p.d1.thresh | p.d2.thresh) { seed_lo = p.st->seed_lo; seed_hi = p.st->seed_hi; }

  // everything issued so far must have landed (weights in registers, blocks 0 and 1 in the ring) ...
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(wf[j][s]));
  // ... then the rest of the ring goes in flight: blocks 2..R-2, epilogue operands 2..R-3; the tally starts here
  vm_issued = 0;
#pragma unroll
  for (int j = 2; j < RS_R - 1; ++j)
    if (j < nb) {
      issueA(j, j);
      mark[j] = vm_issued;
    }
  if (EMODE) {
#pragma unroll
    for (int j = 2; j < RS_R - 2; ++j)
      if (j < nb) {
        issueE(j, j);
        mark[j] = vm_issued;
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // blocks 0 and 1 are visible to every wave

  for (int t0 = 0; t0 < nb; t0 += RS_R) {
#pragma unroll
    for (int u = 0; u < RS_R; ++u) {
      const int t = t0 + u;
      if (t >= nb) break;
      // -- keep the rings full: block t+R-1 goes into the slot block t-1 has just left; the epilogue operand trails by one --
      const int slotAn = (u + RS_R - 1) % RS_R, slotEn = (u + RS_R - 2) % RS_R;
      if (t + RS_R - 1 < nb) {
        issueA(t + RS_R - 1, slotAn);
        mark[slotAn] = vm_issued;
      }
      if (EMODE && t + RS_R - 2 < nb) {
        issueE(t + RS_R - 2, slotEn);
        mark[slotEn] = vm_issued;
      }
      asm volatile("" ::: "memory");
      // -- fragments of block t (swizzled rows), 16 MFMAs: D[n][m] = sum_k W[n][k] X[m][k] --
      const char* ab = ringA + u * RS_BLK + l15 * 512;
      bf16x8 xf[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) xf[s] = *(const bf16x8*)(ab + (((4 * s + g4) ^ l15) << 4));
      f32x4 acc[2];
      acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][s], xf[s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][s], xf[s], acc[1], 0, 0, 0);
      }
      float* const sb = stage + ((t & 1) ? 16 * RS_SP : 0);
      *(f32x4*)(sb + l15 * RS_SP + 32 * wave + 4 * g4) = acc[0];
      *(f32x4*)(sb + l15 * RS_SP + 32 * wave + 16 + 4 * g4) = acc[1];
      // -- block t+1 (and its epilogue operand) must have landed before the barrier that publishes it --
      if (t + 1 < nb) wait_vmcnt(vm_issued - mark[(u + 1) % RS_R]);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // -- epilogue of block t: whole rows, 16 B per lane --
      const int m = min((b0 + t) * 16 + er, p.M - 1);
      float v[8];
      load8(sb + er * RS_SP + ec * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = rs_act(v[j] + bv[j], p.act);
      u32x4 eraw = {0u, 0u, 0u, 0u};
      if (EMODE) eraw = *(const u32x4*)(ringE + u * RS_BLK + er * 512 + ec * 16);
      if (EMODE == 2) {
        float gv[8];
        load8((const bf16_t*)&eraw, gv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gv[j] > 0.f ? v[j] * p.gate_scale : 0.f;
      }
      if (p.d1.thresh | p.d2.thresh) {
        const uint32_t idx = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
        eg_dropout_run<8>(v, p.d1, seed_lo, seed_hi, idx);
        eg_dropout_run<8>(v, p.d2, seed_lo, seed_hi, idx);
      }
      asm volatile("" ::: "memory");
      if (p.out_pre) {
        store8(p.out_pre + (long long)m * p.ldp + n, v);
        vm_issued += 1;
      }
      if (EMODE == 1) {
        float rv[8];
        load8((const bf16_t*)&eraw, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rv[j];
      }
      store8(p.C + (long long)m * p.ldc + n, v);
      vm_issued += 1;
      if (LN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bf2f(f2bf(v[j]));   // normalise exactly what was stored
        float s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s1 += v[j];
        const float mean = rs_half_sum32(s1) * (1.0f / 256.f);
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float dlt = v[j] - mean; s2 += dlt * dlt; }
        const float rstd = rsqrtf(rs_half_sum32(s2) * (1.0f / 256.f) + 1e-5f);
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = (v[j] - mean) * rstd * gam[j] + bet[j];
        store8(p.ln_out + (long long)m * 256 + n, y);
        vm_issued += 1;
        if (ec == 0) {                       // not counted: a predicated 8-B store (under-counting only waits longer)
          p.stats[2 * (long long)m] = mean;
          p.stats[2 * (long long)m + 1] = rstd;
        }
      }
      asm volatile("" ::: "memory");
    }
  }
}

}  // namespace

// eligibility + launch; returns -1 when the product does not fit this kernel (caller falls back to gemm_nt)
int eg_rs_gemm_try(const eg_gemm_desc* d, hipStream_t s) {
  if (d->dtype != EG_BF16 || d->K != 256 || d->N % 256 != 0 || d->ldw != 256) return -1;
  if (d->a.rows_per_group || d->c.rows_per_group || d->r.rows_per_group || d->p.rows_per_group) return -1;
  if (d->a_seg_len || d->ln_mode == 2 || d->row_tile) return -1;
  if (d->residual && d->gate) return -1;
  if (d->ln_mode == 1 && d->N != 256) return -1;
  if (!d->C) return -1;
  if (d->a.row_stride % 8 || d->c.row_stride % 8 || (d->residual && d->r.row_stride % 8) || (d->out_pre && d->p.row_stride % 8)) return -1;
  if (((uintptr_t)d->A | (uintptr_t)d->W | (uintptr_t)d->C | (uintptr_t)d->residual | (uintptr_t)d->gate | (uintptr_t)d->out_pre) % 16) return -1;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  RsGemm p;
  p.A = (const bf16_t*)d->A; p.W = (const bf16_t*)d->W; p.C = (bf16_t*)d->C; p.bias = d->bias;
  p.E = (const bf16_t*)(d->residual ? d->residual : d->gate);
  p.out_pre = (bf16_t*)d->out_pre; p.st = d->state;
  p.lda = d->a.row_stride; p.ldc = d->c.row_stride; p.lde = d->residual ? d->r.row_stride : d->c.row_stride;
  p.ldp = d->p.row_stride;
  p.M = d->M; p.N = d->N; p.ldw = d->ldw; p.act = d->act;
  p.e_mode = d->residual ? 1 : (d->gate ? 2 : 0);
  p.nblk = (d->M + 15) / 16;
  p.ns = d->N / 256;
  p.groups = cus / p.ns > 0 ? cus / p.ns : 1;
  if (p.groups > p.nblk) p.groups = p.nblk;
  p.d1 = make_drop(d->drop1_p, d->drop1_site);
  p.d2 = make_drop(d->drop2_p, d->drop2_site);
  p.gate_scale = d->gate_scale == 0.f ? 1.0f : d->gate_scale;
  p.ln_mode = d->ln_mode; p.gamma = d->ln_gamma; p.beta = d->ln_beta; p.stats = d->ln_stats; p.ln_out = (bf16_t*)d->ln_out;
  const dim3 grid(p.groups * p.ns), blk(512);
#define RS_LAUNCH(E_, L_)                                                                                              \
  do {                                                                                                                 \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      hipFuncSetAttribute((const void*)rs_gemm_kernel<E_, L_>, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);    \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((rs_gemm_kernel<E_, L_>), grid, blk, RS_LDS, s, p);                                             \
  } while (0)
  if (p.ln_mode == 1) {
    if (p.e_mode == 1) RS_LAUNCH(1, 1); else if (p.e_mode == 2) RS_LAUNCH(2, 1); else RS_LAUNCH(0, 1);
  } else {
    if (p.e_mode == 1) RS_LAUNCH(1, 0); else if (p.e_mode == 2) RS_LAUNCH(2, 0); else RS_LAUNCH(0, 0);
  }
#undef RS_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
