// LayerNorm backward + the backward-data product that consumes it, in ONE launch over 80-row tiles (d_model = 256):
//   dr   = LayerNormBackward(dy; x, mean, rstd, gamma)          (A:293 backward: gradient of the pre-norm sum, the residual path's)
//   dyo  = dropout(dr)                                         (the branch dropout of A:292: what flows into out_proj's backward)
//   dC   = dyo * W[256, 256]^T                                 (A:213 backward: grad_input of out_proj, W in fragment order)
//   partial[workgroup][0:256 | 256:512] = this tile's sums of dy * xhat | dy  (gain / bias gradients, reduced by eg_reduce_*)
// As two launches (eg_layernorm_bwd, eg_gemm_nt) the masked rows are written, read back by the product's A-operand DMA, and each launch
// pays its own ramp: 16 + 14 us per layer at the benchmark size.  Here the LayerNorm arithmetic FILLS the resident A tile of an
// eg_ffn_chain-style product (weights as MFMA fragments straight from L2, no streamed operand -- so no HBM-latency DMA sits in front of
// the fragment loads in the waves' in-order vmcnt queues, the reason the streamed form of this tile was dropped).
//
// One 256-thread workgroup per 80 rows, two per CU (80 KB of LDS):
//   1. LDS-DMA: dy rows -> tile Y, x rows -> tile X ([80][512 B], 16-B pieces XOR-swizzled by 2 (row & 7));
//   2. a half-wave per row (32 lanes x 8 elements), rows hw, hw + 8, ..: EXACTLY layernorm_bwd256_kernel's arithmetic and reduction
//      order (dr and dyo are bit-identical to eg_layernorm_bwd's); dr and dyo leave as 512-B rows, dyo also replaces the dy piece in
//      tile Y in place -- tile Y becomes the product's A operand;
//   3. dC: a wave owns 80 x 64 (5 x 4 accumulator tiles), K = 256 in 8 k-steps, weight fragments two k-steps ahead (eg_pack_table
//      modes 5 / 6: [chunk][wave][k-step: 4][tile: 4][lane]); same k-ordered MFMA chains as eg_gemm_nt: dC is bit-identical to it;
//   4. epilogue through a wave-private fp32 image (in tile X, free by then): 16-bit stores of dC.
#include "common.h"

namespace {

constexpr int PR = 80;
constexpr int PD = 256;
constexpr int P_T = PR * PD * 2;              // 40,960 B per tile
constexpr int P_LDS = 2 * P_T;                // 81,920 B
constexpr int P_TP = 68;

template <typename T>
struct LnProjArgs {
  const T* dy; const T* x; const float* stats; const float* gamma; const T* Wf;
  T* dr; T* dyo; T* dC; float* partial;
  const eg_step_state* st;
  int M;
  DropCfg d1, d2;
};

__device__ __forceinline__ void pdma16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ float p_half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256, 2) void ln_bwd_proj_kernel(LnProjArgs<T> p) {
  typedef typename H16<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ty = smem;                       // dy rows, then dyo rows: the product's A operand
  char* const tx = smem + P_T;                 // x rows; later the gain / bias sums and the epilogue image
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  const int m0 = blockIdx.x * PR;

  // ---- 1. both tiles by LDS-DMA: instruction q moves rows 2q, 2q + 1 (lane -> row half lane / 32, piece lane % 32 holding the row's
  //         piece pos ^ 2 (row & 7)); wave w issues q = w, w + 4, .. ----
  {
    const int half = lane >> 5, pos = lane & 31;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int q = wn + 4 * i;
      const int r = 2 * q + half;
      const size_t row = (size_t)min(m0 + r, p.M - 1);
      const int sw = (pos ^ ((r & 7) << 1)) << 4;
      pdma16((const char*)(p.dy + row * PD) + sw, ty + q * 1024);
      pdma16((const char*)(p.x + row * PD) + sw, tx + q * 1024);
    }
  }
  // weight fragments of the first two k-steps and this half-wave's statistics travel meanwhile
  const char* const wu = (const char*)(p.Wf + (size_t)wn * (4 * 4 * 512));      // + c * (4*4*4*512) + (s4 * 4 + j) * 512   [elements]
  const uint32_t wl = (uint32_t)lane * 16u;
  frag wr[2][4];
  auto req_w = [&](int s, int slot) {            // k-step s = 0 .. 7: chunk s / 4, step s % 4
#pragma unroll
    for (int j = 0; j < 4; ++j)
      wr[slot][j] = *(const frag*)(wu + ((size_t)(s >> 2) * (4 * 4 * 4 * 512) + ((s & 3) * 4 + j) * 512) * 2 + wl);
  };
  const int l = lane & 31, hw = tid >> 5;        // half-wave hw = 0 .. 7 owns rows hw, hw + 8, .. (10 rows)
  float mean[10], rstd[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const int m = m0 + hw + 8 * j;
    mean[j] = 0.f; rstd[j] = 0.f;
    if (m < p.M) { mean[j] = p.stats[2 * (size_t)m]; rstd[j] = p.stats[2 * (size_t)m + 1]; }
  }
  float g[8], dg[8], db[8];
  load8(p.gamma + l * 8, g);
#pragma unroll
  for (int e = 0; e < 8; ++e) { dg[e] = 0.f; db[e] = 0.f; }
  uint32_t seed_lo = 0, seed_hi = 0;
  const bool drop = (p.d1.thresh | p.d2.thresh) != 0;
  if (drop) { seed_lo = p.st->seed_lo; seed_hi = p.st->seed_hi; }
  req_w(0, 0);
  req_w(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- 2. LayerNorm backward, a half-wave per row (layernorm_bwd256_kernel's arithmetic) ----
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const int r = hw + 8 * j;
    const int m = m0 + r;
    const bool ok = m < p.M;
    char* const py = ty + r * 512 + ((l ^ ((r & 7) << 1)) << 4);
    const char* const px = tx + r * 512 + ((l ^ ((r & 7) << 1)) << 4);
    float xv[8], dv[8];
    load8((const T*)px, xv);
    load8((const T*)py, dv);
    if (!ok) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { xv[e] = 0.f; dv[e] = 0.f; }
    }
    float o[8];
    eg_ln_bwd_row8(xv, dv, g, mean[j], rstd[j], dg, db, o);
    if (!ok) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
    }
    if (ok) store8(p.dr + (size_t)m * PD + l * 8, o);
    if (drop && ok) {
      const uint32_t idx = (uint32_t)m * 256u + (uint32_t)(l * 8);
      eg_dropout_run<8>(o, p.d1, seed_lo, seed_hi, idx);
      eg_dropout_run<8>(o, p.d2, seed_lo, seed_hi, idx);
    }
    if (ok) store8(p.dyo + (size_t)m * PD + l * 8, o);
    store8((T*)py, o);                           // (zeros for rows beyond M) the piece becomes part of the A operand
  }
  __syncthreads();                               // tile Y = dyo complete; tile X is free

  // gain / bias sums of the tile: [half-wave][dg | db][256] through LDS (tile X), then one sum per column
  {
    float* red = (float*)tx;                     // [8][2][256]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(hw * 2 + 0) * 256 + l * 8 + e] = dg[e];
      red[(hw * 2 + 1) * 256 + l * 8 + e] = db[e];
    }
    __syncthreads();
    for (int i = tid; i < 512; i += 256) {
      const int which = i >> 8, n = i & 255;
      float s = 0.f;
#pragma unroll
      for (int h = 0; h < 8; ++h) s += red[(h * 2 + which) * 256 + n];
      p.partial[(size_t)blockIdx.x * 512 + i] = s;
    }
  }

  // ---- 3. dC = dyo * W^T over K = 256 ----
  f32x4 acc[5][4];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int sw7 = (l15 & 7) << 1;      // chunk ^ 2 (row & 7): conflict-free under ds_read_b128's lane groups on 512-B rows (ffn.hip)
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    frag xf[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) xf[i] = *(const frag*)(ty + (l15 + 16 * i) * 512 + (((4 * s + g4) ^ sw7) << 4));
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = H16<T>::mfma(wr[s & 1][j], xf[i], acc[i][j]);
    if (s + 2 < 8) req_w(s + 2, s & 1);
  }

  // ---- 4. epilogue: per 16-row tile through a wave-private fp32 image [16][68] (tile X, behind the 16 KB of sums) ----
  float* timg = (float*)(tx + 16384 + wn * (16 * P_TP * 4));
  const int er = lane >> 2, ec = lane & 3;
  const int n = 64 * wn + 16 * ec;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int m = m0 + 16 * i + er;
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(timg + l15 * P_TP + 16 * j + 4 * g4) = acc[i][j];
    if (m0 + 16 * i >= p.M) break;                 // workgroup-uniform: tiles wholly beyond M
    float v[16];
    load8(timg + er * P_TP + 16 * ec, v);
    load8(timg + er * P_TP + 16 * ec + 8, v + 8);
    if (m < p.M) {
      T* pc = p.dC + (size_t)m * PD + n;
      store8(pc, v);
      store8(pc + 8, v + 8);
    }
  }
}

template <typename T>
static int lnproj_launch(const eg_ln_bwd_proj_desc* d, hipStream_t s) {
  LnProjArgs<T> p;
  p.dy = (const T*)d->dy; p.x = (const T*)d->x; p.stats = d->stats; p.gamma = d->gamma; p.Wf = (const T*)d->W_frag;
  p.dr = (T*)d->dx; p.dyo = (T*)d->dx_drop; p.dC = (T*)d->dC; p.partial = d->partial; p.st = d->state; p.M = d->M;
  p.d1 = make_drop(d->drop1_p, d->drop1_site);
  p.d2 = make_drop(d->drop2_p, d->drop2_site);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)ln_bwd_proj_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
    attr = true;
  }
  hipLaunchKernelGGL((ln_bwd_proj_kernel<T>), dim3((d->M + PR - 1) / PR), dim3(256), P_LDS, s, p);
  EG_LAUNCH_CHECK("ln_bwd_proj");
  return 0;
}

}  // namespace

extern "C" int eg_ln_bwd_proj_blocks(int M) { return M > 0 ? (M + PR - 1) / PR : 0; }

extern "C" int eg_ln_bwd_proj(const eg_ln_bwd_proj_desc* d, void* stream) {
  EG_CHECK(d && d->dy && d->x && d->stats && d->gamma && d->W_frag && d->dx && d->dx_drop && d->dC && d->partial,
           "eg_ln_bwd_proj: null operand");
  EG_CHECK(d->dtype == EG_BF16 || d->dtype == EG_F16, "eg_ln_bwd_proj: 16-bit compute dtypes only (got %d)", d->dtype);
  EG_CHECK(d->M > 0 && d->d_model == PD, "eg_ln_bwd_proj: M=%d, d_model=%d (256 only)", d->M, d->d_model);
  EG_CHECK(d->partial_capacity_blocks >= eg_ln_bwd_proj_blocks(d->M),
           "eg_ln_bwd_proj: the partial buffer holds %d rows, the launch writes %d", d->partial_capacity_blocks, eg_ln_bwd_proj_blocks(d->M));
  EG_CHECK((long long)d->M * PD < (1ll << 32), "eg_ln_bwd_proj: M*D exceeds the 32-bit dropout index");
  EG_CHECK(d->drop1_p >= 0.f && d->drop1_p < 1.f && d->drop2_p >= 0.f && d->drop2_p < 1.f, "eg_ln_bwd_proj: dropout p");
  EG_CHECK((d->drop1_p == 0.f && d->drop2_p == 0.f) || d->state, "eg_ln_bwd_proj: dropout needs a step state");
  EG_CHECK(((uintptr_t)d->dy | (uintptr_t)d->x | (uintptr_t)d->W_frag | (uintptr_t)d->dx | (uintptr_t)d->dx_drop | (uintptr_t)d->dC) % 16 == 0,
           "eg_ln_bwd_proj: operands must be 16-B aligned");
  hipStream_t s = (hipStream_t)stream;
  return d->dtype == EG_F16 ? lnproj_launch<f16_t>(d, s) : lnproj_launch<bf16_t>(d, s);
}
