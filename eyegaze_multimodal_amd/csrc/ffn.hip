// Chained product pair of the position-wise feed-forward block (A:272) in ONE launch, forward and backward-data alike:
//   H[M, F]   = epi1(A[M, 256] * W1[F, 256]^T)      epi1: + bias1, ReLU, gate (ReLU backward), dropout      -> stored (backward and
//   C[M, 256] = epi2(H[M, F]   * W2[256, F]^T)                                                                 weight gradients need it)
//   forward : A = y1, W1 = linear1, W2 = linear2, epi2 = + bias2, dropout x2, + residual (the A rows themselves) -> r2
//   backward: A = dY, W1 = linear2^T, gate = saved hidden rows, W2 = linear1^T, epi2 = + residual (dr)           -> dy1
// As two launches the pair moves H through HBM twice (write, read back: 2 x 68 MB per layer and direction at the benchmark
// size) and A / the residual once more; here the hidden rows only leave the chip once, as the stored result.
//
// Design (gfx950), one 256-thread workgroup (4 waves = 4 column groups, one per SIMD) per 80 rows, TWO workgroups per CU (80 KB
// of LDS each): the two row halves of a CU share nothing, so they run as independent workgroups whose phases drift apart -- one's
// MFMA bursts sit beside the other's epilogue / barrier / store phases (as one 512-thread workgroup every wave hit the same
// barrier and the matrix pipe idled through each epilogue):
//   * the A tile [80 x 256] stays in LDS for the whole launch (40 KB, LDS-DMA, 16-B chunks XOR-swizzled by 2 (row & 7)); it is the
//     B operand of every product-1 MFMA and, in the forward pass, the residual of the final epilogue (no second HBM read);
//   * the hidden dimension is walked in chunks of 128 columns: product 1 gives a wave 80 x 32 of the chunk (5 x 2 accumulator
//     tiles over K = 256), its epilogue writes the 16-bit chunk into one of TWO 20 KB LDS buffers (one barrier per chunk), from
//     where (a) all threads stream it to HBM as whole 256-B row segments and (b) product 2 reads it back as the B operand of
//     the wave's 80 x 64 part of C (5 x 4 accumulator tiles that live in registers across all chunks);
//   * WEIGHTS NEVER TOUCH LDS: a wave's W1 / W2 fragments (pre-packed in fragment order: one contiguous 1-KB read each,
//     L2-resident, 1 MB per layer) go straight into registers through rolling rings four / two k-steps ahead, so the LDS pipe only carries
//     the activation fragments: 0.5 KB per MFMA in product 1, 0.25 KB in product 2 -- below the 0.5 KB / MFMA at which LDS and
//     matrix pipes balance, which the 128 x 128 and 160 x 256 tiles (weights through LDS) sit on;
//   * the backward gate ("the saved hidden value is > 0": ReLU and dropout cut together) travels as ONE BIT per element: the
//     forward launch writes it in MFMA-lane order (40 bits of one 8-byte word per lane and chunk), the backward launch of the
//     same geometry reads one word per lane and chunk, two chunks ahead -- 4.3 MB per layer instead of re-reading the 68 MB of
//     hidden rows, and no HBM-latency loads in front of the weight fragments in the waves' in-order vmcnt queues (with the gate
//     rows themselves loaded into registers the backward form took 75 us against 68 us for the two launches).  The hidden
//     rows themselves remain accepted as the gate (eg_ffn_desc.gate) for callers without a forward launch of this kernel.
// Arithmetic: the same k-ordered chains of v_mfma_f32_16x16x32 as eg_gemm_nt's kernels and the same epilogue order and dropout
// indices, so H and C are bit-identical to the two-launch path.
#include "common.h"

namespace {

constexpr int FR = 80;                        // rows per workgroup
constexpr int FD = 256;                       // d_model: K of product 1, N of product 2
constexpr int FC = 128;                       // hidden columns per chunk
constexpr int F_XT = FR * FD * 2;             // 40,960 B
constexpr int F_HT = FR * FC * 2;             // 20,480 B
constexpr int F_LDS = F_XT + 2 * F_HT;        // 81,920 B = half the LDS of a CU
constexpr int F_TP = 68;                      // fp32 image pitch of the final epilogue (floats)

template <typename T>
struct FfnArgs {
  const T* A; const T* W1; const T* W2; T* H; T* C; const float* bias1; const float* bias2; const T* gate; const T* residual;
  const unsigned long long* bits_in; unsigned long long* bits_out;
  const float* ln_gamma; const float* ln_beta; T* LN_OUT; float* ln_stats;   // LNF: y = LayerNorm(C) in the final epilogue
  const eg_step_state* st;
  long long lda, ldh, ldc, ldg, ldr;
  int M, F, relu, res_in_lds;
  DropCfg dh, dc1, dc2;
  float gate_scale;
};

__device__ __forceinline__ void fdma16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <typename T> __device__ __forceinline__ u32x2 f_pack4(const float v[4]) {
  u32x2 o;
  o[0] = H16<T>::pack2(v[0], v[1]);
  o[1] = H16<T>::pack2(v[2], v[3]);
  return o;
}

// EPI selects epilogue 1's bias / ReLU at compile time: 3 = bias + ReLU (the forward pass), 0 = neither (the backward pass),
// 4 = as the descriptor says at run time (a per-value select on the flag -- 80 extra vector instructions per chunk).
// LNF: the layer's second LayerNorm runs on the completed C rows in the final epilogue (eg_epilogue_layernorm256).
template <typename T, int GATE, int BOUT, int EPI, bool LNF = false>
__global__ __launch_bounds__(256, 2) void ffn_chain_kernel(FfnArgs<T> p) {
  typedef typename H16<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const xt = smem;
  char* const hb = smem + F_XT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave = column group
  const int l15 = lane & 15, g4 = lane >> 4;
  const int m0 = blockIdx.x * FR;
  const int nch = p.F / FC;

  // ---- A tile: instruction q moves rows 2q, 2q+1 (lane -> row half lane/32, LDS chunk position lane%32 holding global chunk
  //      pos ^ 2 (row & 7)); wave w issues q = w, w+4, .. (10 each) ----
  {
    const int half = lane >> 5, pos = lane & 31;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int q = wn + 4 * i;
      const int r = 2 * q + half;
      const int row = min(m0 + r, p.M - 1);
      fdma16((const char*)(p.A + (size_t)row * (size_t)p.lda) + ((pos ^ ((r & 7) << 1)) << 4), xt + q * 1024);
    }
  }

  // ---- weight fragment streams (global -> registers): the weights arrive in FRAGMENT ORDER (eg_pack_table modes 3-6), so a
  //      fragment load is one contiguous 1-KB read per wave.  (Row-major weights cost 64 L1 tag look-ups per load -- 16 rows x
  //      64 B per quarter wave -- and made the launch tag-rate-bound: 98 us against 72 us for the two launches it replaces.) ----
  const char* const w1u = (const char*)(p.W1 + (size_t)wn * (8 * 2 * 512));      // + c * (4*8*2*512) + (s * 2 + j) * 512   [elements]
  const char* const w2u = (const char*)(p.W2 + (size_t)wn * (4 * 4 * 512));      // + c * (4*4*4*512) + (s * 4 + j) * 512
  const uint32_t wl = (uint32_t)lane * 16u;                                      // wave-uniform base + 32-bit lane offset: saddr loads
  // Rolling rings, four (W1) / two (W2) k-steps ahead of their MFMAs and running on across chunk boundaries.  (Requesting a
  // whole chunk's fragments one phase ahead -- 16 + 16 live fragments -- was built: 67-88 spilled registers; the 256-register
  // budget of two waves per SIMD is spent on the 5 x 4 + 5 x 2 accumulator tiles.)
  frag w1r[4][2], w2r[2][4];
  auto req_w1 = [&](int c, int s, int slot) {
#pragma unroll
    for (int j = 0; j < 2; ++j) w1r[slot][j] = *(const frag*)(w1u + ((size_t)c * (4 * 8 * 2 * 512) + (s * 2 + j) * 512) * 2 + wl);
  };
  auto req_w2 = [&](int c, int s, int slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j) w2r[slot][j] = *(const frag*)(w2u + ((size_t)c * (4 * 4 * 4 * 512) + (s * 4 + j) * 512) * 2 + wl);
  };
#pragma unroll
  for (int s = 0; s < 4; ++s) req_w1(0, s, s);
#pragma unroll
  for (int s = 0; s < 2; ++s) req_w2(0, s, s);

  f32x4 acc2[5][4];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint32_t seed_lo = 0, seed_hi = 0;
  if (p.dh.thresh | p.dc1.thresh | p.dc2.thresh) { seed_lo = p.st->seed_lo; seed_hi = p.st->seed_hi; }

  // ---- weights into the Infinity Cache: in a training step a launch finds its 1 MB of weights in HBM (written a whole step
  //      earlier), and because all workgroups walk the chunks in lock-step each chunk's first touch stalled every one of them
  //      (74 us against 62 us when a replayed launch finds them cached).  So every workgroup touches its 1/grid share of ALL
  //      chunks' 128-B lines here, where the wave waits for its HBM-resident A rows anyway: by the time chunk 1 is needed the
  //      lines sit in the memory-side cache, one XCD's L2 miss away. ----
  uint32_t touched = 0;
  if (wn < 2) {
    const char* wb = (const char*)(wn == 0 ? p.W1 : p.W2);
    const int lines = p.F * FD * 2 / 128;                    // per matrix
    const int per = (lines + gridDim.x - 1) / gridDim.x;
    for (int t = lane; t < per; t += 64) {
      const int g = blockIdx.x * per + t;
      if (g < lines) touched += *(const uint32_t*)(wb + (size_t)g * 128);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's part of the A tile has landed (and the first fragments)
  asm volatile("" :: "v"(touched));
  __syncthreads();                                           // ... and everybody else's

  // gate bit words: [workgroup][chunk][wave][lane]
  const unsigned long long* const bits_p = p.bits_in ? p.bits_in + ((size_t)blockIdx.x * nch * 4 + wn) * 64 + lane : nullptr;
  unsigned long long gnext0 = 0, gnext1 = 0;
  if (GATE == 2) {
    gnext0 = bits_p[0];
    if (nch > 1) gnext1 = bits_p[256];
  }
  const int rbase = l15;                                     // row of tile i within the workgroup: rbase + 16 i
  const int sw7 = (l15 & 7) << 1;      // chunk ^ 2 (row & 7): free of bank conflicts under ds_read_b128's lane groups on 256-B and 512-B rows
                                        // (chunk ^ (row & 7) is two ways conflicted there: SQ_LDS_BANK_CONFLICT 44 % of the LDS cycles)
  // epilogue-1 lane constants: byte offset of this lane's 8-B slot (row l15, hidden column 32 wn + 16 j + 4 g4 of the chunk) in the
  // swizzled chunk image -- tile row i adds the immediate 4096 i -- and the dropout PAIR index of (row m0 + l15, column 32 wn + 4 g4)
  int ha[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int hcol = 32 * wn + 16 * j + 4 * g4;
    ha[j] = l15 * 256 + (((hcol >> 3) ^ sw7) << 4) + ((hcol & 4) << 1);
  }
  const uint32_t pair0 = ((uint32_t)(m0 + l15) * (uint32_t)p.F + (uint32_t)(32 * wn + 4 * g4)) >> 1;
  const uint32_t pairF = 8u * (uint32_t)p.F;                 // 16 rows further

  for (int c = 0; c < nch; ++c) {
    char* const hc = hb + (c & 1) * F_HT;
    const int f0 = FC * c + 32 * wn + 4 * g4;                // hidden column of accumulator register q of tile j: f0 + 16 j + q

    // epilogue-1 operands of this chunk, requested before the MFMAs that hide them
    float b1[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) b1[j][q] = 0.f;
      if (EPI == 3 || (EPI == 4 && p.bias1)) load4(p.bias1 + f0 + 16 * j, b1[j]);
    }

    u32x2 gt[5][2];
    if (GATE == 1) {
      int rb = rbase;
      asm volatile("" : "+v"(rb));                             // addresses are recomputed per chunk, not kept (and spilled) across the loop
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int m = min(m0 + rb + 16 * i, p.M - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) gt[i][j] = *(const u32x2*)(p.gate + (size_t)m * (size_t)p.ldg + f0 + 16 * j);
      }
    }
    // gate bits of a lane and chunk: word g (= tile rows 0-2 | 3-4 ... see below) collects, in the order the epilogue visits them, the
    // flags of its 32-bit stores (two values each): flag of the even value in the low half, of the odd value in the high half,
    // first store highest.  20 stores -> two words of 10 + 10 bits per half.
    uint32_t gw[2] = {0u, 0u}, ow[2] = {0u, 0u};
    if (GATE == 2) {                                           // this chunk's word was requested two chunks ago
      gw[0] = (uint32_t)gnext0;
      gw[1] = (uint32_t)(gnext0 >> 32);
      gnext0 = gnext1;
      if (c + 2 < nch) gnext1 = bits_p[(size_t)(c + 2) * 256];
    }

    // ---- product 1: acc1[i][j] = sum_k W1[hidden][k] * A[row][k] ----
    f32x4 acc1[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      frag xf[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) xf[i] = *(const frag*)(xt + (rbase + 16 * i) * 512 + (((4 * s + g4) ^ sw7) << 4));
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = H16<T>::mfma(w1r[s & 3][j], xf[i], acc1[i][j]);
      if (s < 4) req_w1(c, s + 4, s & 3);
      else if (c + 1 < nch) req_w1(c + 1, s - 4, s & 3);
    }

    // ---- epilogue 1 (MFMA layout: lane holds 4 consecutive hidden columns of row l15) -> 16-bit chunk image in LDS ----
    // Vector-instruction count matters here: the launch is bound by VALU issue (4 cycles per instruction and wave, two waves per
    // SIMD), not by the matrix pipe -- 817 vector instructions per wave and chunk against 160 MFMAs before this form.
    const uint32_t pairc = pair0 + (uint32_t)(FC / 2) * (uint32_t)c;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = 2 * i + j;                                // store pair 2t, 2t+1 of this chunk
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = acc1[i][j][q];
          if (EPI == 3 || EPI == 4) v[q] += b1[j][q];
          if (EPI == 3) v[q] = fmaxf(v[q], 0.f);
          if (EPI == 4 && p.relu) v[q] = fmaxf(v[q], 0.f);
        }
        if (GATE == 1) {
          float gv[4];
          load4((const T*)&gt[i][j], gv);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = gv[q] > 0.f ? v[q] * p.gate_scale : 0.f;
        }
        if (GATE == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = (2 * t + (q >> 1)) % 10;              // position among the word's 10 stores
            int m;                                               // 0 or -1 (as asm: the compiler turns the builtin back into and + compare + select)
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(gw[t / 5]), "n"((9 - n) + 16 * (q & 1)));
            v[q] = __uint_as_float(__float_as_uint(v[q] * p.gate_scale) & (uint32_t)m);
          }
        }
        if (p.dh.thresh) {                                      // = eg_dropout_run<4> at element (m0 + l15 + 16 i) * F + f0 + 16 j
          const uint32_t e = pairc + (uint32_t)(8 * j) + (uint32_t)i * pairF;
          const uint32_t h0 = eg_hash(seed_lo, seed_hi, p.dh.site, e), h1 = eg_hash(seed_lo, seed_hi, p.dh.site, e + 1u);
          v[0] = (h0 & 0xFFFFu) >= p.dh.thresh ? v[0] * p.dh.scale : 0.0f;
          v[1] = (h0 >> 16) >= p.dh.thresh ? v[1] * p.dh.scale : 0.0f;
          v[2] = (h1 & 0xFFFFu) >= p.dh.thresh ? v[2] * p.dh.scale : 0.0f;
          v[3] = (h1 >> 16) >= p.dh.thresh ? v[3] * p.dh.scale : 0.0f;
        }
        const u32x2 pk = f_pack4<T>(v);
        *(u32x2*)(hc + ha[j] + 4096 * i) = pk;
        if (BOUT) {                                            // "stored value > 0", taken from the stored 16-bit patterns:
#pragma unroll
          for (int h = 0; h < 2; ++h) {                        // min(max(as int16, 0), 1) per half = 1 exactly for +x, x != 0
            uint32_t z, f;                                       // (as asm: three instructions per store; the elementwise builtins
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(z) : "v"(pk[h]));                      // came back as compare / select / permute chains)
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(f) : "v"(z), "v"(0x00010001u));
            ow[t / 5] = (ow[t / 5] << 1) | f;
          }
        }
      }
    }
    if (BOUT) p.bits_out[((size_t)blockIdx.x * nch + c) * 256 + wn * 64 + lane] = (unsigned long long)ow[0] | ((unsigned long long)ow[1] << 32);
    __syncthreads();        // chunk c is complete in LDS; nobody reads buffer (c+1)&1 (chunk c-1) any more

    // ---- the stored result: whole 256-B row segments of the chunk, 16 B per thread ----
    int tq = tid >> 4;
    asm volatile("" : "+v"(tq));                               // (same: no loop-invariant address registers)
#pragma unroll
    for (int ps = 0; ps < 5; ++ps) {
      const int r = 16 * ps + tq, ch = tid & 15;
      const u32x4 o = *(const u32x4*)(hc + r * 256 + ((ch ^ ((r & 7) << 1)) << 4));
      if (m0 + r < p.M) *(u32x4*)(p.H + (size_t)(m0 + r) * (size_t)p.ldh + FC * c + 8 * ch) = o;
    }

    // ---- product 2: acc2[i][j] += sum_h W2[col][h] * H[row][h] over the chunk's 128 hidden columns ----
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      frag hf[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) hf[i] = *(const frag*)(hc + (rbase + 16 * i) * 256 + (((4 * s + g4) ^ sw7) << 4));
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[i][j] = H16<T>::mfma(w2r[s & 1][j], hf[i], acc2[i][j]);
      if (s < 2) req_w2(c, s + 2, s & 1);
      else if (c + 1 < nch) req_w2(c + 1, s - 2, s & 1);
    }
  }
  __syncthreads();          // every wave has left the chunk buffers: they become the fp32 image of the final epilogue

  // ---- epilogue 2: per 16-row tile through a wave-private fp32 image [16][68]; a lane then owns 16 consecutive columns of a row ----
  float* timg = (float*)(hb + wn * (16 * F_TP * 4));
  const int er = lane >> 2, ec = lane & 3;
  const int n = 64 * wn + 16 * ec;
  float bv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bv[j] = 0.f;
  if (p.bias2) { load8(p.bias2 + n, bv); load8(p.bias2 + n + 8, bv + 8); }
  u32x4 eraw[5][2];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    eraw[i][0] = (u32x4){0u, 0u, 0u, 0u};
    eraw[i][1] = (u32x4){0u, 0u, 0u, 0u};
    const int r = 16 * i + er;
    if (p.res_in_lds) {
      eraw[i][0] = *(const u32x4*)(xt + r * 512 + ((((n >> 3)) ^ ((r & 7) << 1)) << 4));
      eraw[i][1] = *(const u32x4*)(xt + r * 512 + ((((n >> 3) + 1) ^ ((r & 7) << 1)) << 4));
    } else if (p.residual && m0 + r < p.M) {
      const T* pe = p.residual + (size_t)(m0 + r) * (size_t)p.ldr + n;
      eraw[i][0] = *(const u32x4*)pe;
      eraw[i][1] = *(const u32x4*)(pe + 8);
    }
  }
  float vv[LNF ? 5 : 1][16];                       // LNF: the stored C values of this lane's rows, for the LayerNorm below
  if (LNF) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) vv[i][j] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int m = m0 + 16 * i + er;
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(timg + l15 * F_TP + 16 * j + 4 * g4) = acc2[i][j];
    if (m0 + 16 * i >= p.M) break;                 // workgroup-uniform: tiles wholly beyond M
    float v[16];
    load8(timg + er * F_TP + 16 * ec, v);
    load8(timg + er * F_TP + 16 * ec + 8, v + 8);
    if (m < p.M) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] += bv[j];
      if (p.dc1.thresh | p.dc2.thresh) {
        const uint32_t idx = (uint32_t)m * (uint32_t)FD + (uint32_t)n;
        float (&v0)[8] = *(float (*)[8])v;
        float (&v1)[8] = *(float (*)[8])(v + 8);
        eg_dropout_run<8>(v0, p.dc1, seed_lo, seed_hi, idx);
        eg_dropout_run<8>(v0, p.dc2, seed_lo, seed_hi, idx);
        eg_dropout_run<8>(v1, p.dc1, seed_lo, seed_hi, idx + 8);
        eg_dropout_run<8>(v1, p.dc2, seed_lo, seed_hi, idx + 8);
      }
      if (p.residual) {
        float rv[16];
        load8((const T*)&eraw[i][0], rv);
        load8((const T*)&eraw[i][1], rv + 8);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] += rv[j];
      }
      T* pc = p.C + (size_t)m * (size_t)p.ldc + n;
      store8(pc, v);
      store8(pc + 8, v + 8);
      if (LNF) {
#pragma unroll
        for (int j = 0; j < 16; ++j) vv[i][j] = round_store<T>(v[j]);
      }
    }
  }
  if constexpr (LNF)
    eg_epilogue_layernorm256<T>(vv, (float*)(hb + 4 * (16 * F_TP * 4)), wn, lane, min(FR, p.M - m0), (size_t)m0, p.ln_gamma, p.ln_beta,
                                p.LN_OUT, p.ln_stats);
}

template <typename T>
static int ffn_launch(const eg_ffn_desc* d, hipStream_t s) {
  FfnArgs<T> p;
  p.A = (const T*)d->A; p.W1 = (const T*)d->W1; p.W2 = (const T*)d->W2; p.H = (T*)d->H; p.C = (T*)d->C;
  p.bits_in = (const unsigned long long*)d->gate_bits_in; p.bits_out = (unsigned long long*)d->gate_bits_out;
  p.bias1 = d->bias1; p.bias2 = d->bias2; p.gate = (const T*)d->gate; p.residual = (const T*)d->residual; p.st = d->state;
  p.lda = d->lda; p.ldh = d->ldh; p.ldc = d->ldc; p.ldg = d->ldg; p.ldr = d->ldr;
  p.M = d->M; p.F = d->F;
  p.relu = d->act1 == EG_ACT_RELU;
  p.res_in_lds = d->residual && d->residual == d->A && d->ldr == d->lda;
  p.dh = make_drop(d->drop_h_p, d->drop_h_site);
  p.dc1 = make_drop(d->drop_c1_p, d->drop_c1_site);
  p.dc2 = make_drop(d->drop_c2_p, d->drop_c2_site);
  p.gate_scale = d->gate_scale == 0.f ? 1.0f : d->gate_scale;
  p.ln_gamma = d->ln_gamma; p.ln_beta = d->ln_beta; p.LN_OUT = (T*)d->ln_out; p.ln_stats = d->ln_stats;
  const dim3 grid((d->M + FR - 1) / FR);
#define FFN_LAUNCH(G_, B_, E_, L_)                                                                                     \
  do {                                                                                                                 \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      (void)hipFuncSetAttribute((const void*)ffn_chain_kernel<T, G_, B_, E_, L_>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS); \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((ffn_chain_kernel<T, G_, B_, E_, L_>), grid, dim3(256), F_LDS, s, p);                           \
  } while (0)
  const int gsel = d->gate_bits_in ? 2 : d->gate ? 1 : 0;
  const bool bout = d->gate_bits_out != nullptr;
  const bool lnf = d->ln_out != nullptr;                                                                      // (forward form only, checked below)
  if (gsel == 0 && d->bias1 && p.relu) {                                                                      // the forward pass
    if (lnf) { if (bout) FFN_LAUNCH(0, 1, 3, true); else FFN_LAUNCH(0, 0, 3, true); }
    else { if (bout) FFN_LAUNCH(0, 1, 3, false); else FFN_LAUNCH(0, 0, 3, false); }
  }
  else if (gsel == 2 && !d->bias1 && !p.relu) FFN_LAUNCH(2, 0, 0, false);                                    // the backward pass
  else if (gsel == 1 && !d->bias1 && !p.relu) FFN_LAUNCH(1, 0, 0, false);
  else if (gsel == 2) FFN_LAUNCH(2, 0, 4, false);                                                             // anything else
  else if (gsel == 1) FFN_LAUNCH(1, 0, 4, false);
  else if (bout) FFN_LAUNCH(0, 1, 4, false);
  else FFN_LAUNCH(0, 0, 4, false);
#undef FFN_LAUNCH
  EG_LAUNCH_CHECK("ffn_chain");
  return 0;
}

}  // namespace

extern "C" int eg_ffn_chain(const eg_ffn_desc* d, void* stream) {
  EG_CHECK(d && d->A && d->W1 && d->W2 && d->H && d->C, "eg_ffn_chain: null operand");
  EG_CHECK(d->dtype == EG_BF16 || d->dtype == EG_F16, "eg_ffn_chain: 16-bit compute dtypes only (got %d)", d->dtype);
  EG_CHECK(d->M > 0 && d->F > 0 && d->F % FC == 0, "eg_ffn_chain: M=%d, F=%d (F must be a multiple of %d)", d->M, d->F, FC);
  EG_CHECK(d->act1 == EG_ACT_NONE || d->act1 == EG_ACT_RELU, "eg_ffn_chain: act1 %d", d->act1);
  EG_CHECK(d->lda >= FD && d->ldc >= FD && d->ldh >= d->F && d->lda % 8 == 0 && d->ldc % 8 == 0 && d->ldh % 8 == 0,
           "eg_ffn_chain: row strides must be 16-B multiples covering the rows");
  EG_CHECK(!d->gate || (d->ldg >= d->F && d->ldg % 4 == 0), "eg_ffn_chain: gate stride");
  EG_CHECK(!(d->gate_bits_out && (d->gate_bits_in || d->gate)), "eg_ffn_chain: a launch either writes gate bits or applies a gate");
  EG_CHECK(((uintptr_t)d->gate_bits_in | (uintptr_t)d->gate_bits_out) % 8 == 0, "eg_ffn_chain: gate bit words must be 8-B aligned");
  EG_CHECK(!d->residual || (d->ldr >= FD && d->ldr % 8 == 0), "eg_ffn_chain: residual stride");
  EG_CHECK((long long)d->M * d->F < (1ll << 32), "eg_ffn_chain: M*F exceeds the 32-bit dropout index");
  EG_CHECK(!d->ln_out || (d->ln_gamma && d->ln_beta && d->ldc == FD && !d->gate && !d->gate_bits_in && d->bias1 && d->act1 == EG_ACT_RELU &&
                          (uintptr_t)d->ln_out % 16 == 0),
           "eg_ffn_chain: the fused LayerNorm serves the forward form (bias + ReLU, no gate) with contiguous C rows");
  const float ps[3] = {d->drop_h_p, d->drop_c1_p, d->drop_c2_p};
  for (float q : ps) EG_CHECK(q >= 0.f && q < 1.f, "eg_ffn_chain: dropout p");
  EG_CHECK((ps[0] == 0.f && ps[1] == 0.f && ps[2] == 0.f) || d->state, "eg_ffn_chain: dropout needs a step state");
  EG_CHECK(((uintptr_t)d->A | (uintptr_t)d->W1 | (uintptr_t)d->W2 | (uintptr_t)d->H | (uintptr_t)d->C | (uintptr_t)d->gate |
            (uintptr_t)d->residual) % 16 == 0, "eg_ffn_chain: operands must be 16-B aligned");
  hipStream_t s = (hipStream_t)stream;
  return d->dtype == EG_F16 ? ffn_launch<f16_t>(d, s) : ffn_launch<bf16_t>(d, s);
}

extern "C" int64_t eg_ffn_gate_bits_bytes(int M, int F) {
  if (M <= 0 || F <= 0) return 0;
  return (int64_t)((M + FR - 1) / FR) * (F / FC) * 256 * 8;
}
