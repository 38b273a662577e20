// Window-resident attention block (forward) of one post-LN encoder layer, A:202-213 + A:292-293, in ONE launch:
//   q|k|v = x Wqkv^T + b            (stored: the backward and the analysis hooks read it)
//   P     = dropout(softmax(q k^T / sqrt(32)))  per head, lse stored
//   ctx   = P v                      (stored: attention backward's delta term and the out-proj weight gradient read it)
//   r1    = x + dropout(ctx Wo^T + bo)
// As three launches (row-stream q|k|v GEMM, attention core, wide out-proj GEMM; the LayerNorm that follows is a fourth unless ln_out
// asks for it here) the layer moved
// x twice, q|k|v twice and ctx twice through HBM: 204 MB per layer at the benchmark size (33 280 rows).  Here a workgroup owns
// one WINDOW (S <= 80 token rows x 256 = 40 KB in LDS for the whole launch): x is read once and is also the residual, q|k|v and
// ctx leave the chip once as results and are consumed on chip from LDS images: 102 MB.
//
// Design (gfx950): one 256-thread workgroup (4 waves, one per SIMD) per window, TWO workgroups per CU (72 KB of LDS each) whose
// phases drift apart, as eg_ffn_chain's do.  The 8 heads are walked in 4 chunks of 2 heads:
//   P1  q|k|v projection of the chunk's 2 x 96 columns: wave w owns 3 of the 12 column tiles x all 5 row tiles over K = 256
//       (120 MFMAs); the x fragments come from the LDS tile, the weights straight from L2 into registers in FRAGMENT ORDER
//       (eg_pack_table modes 7 / 8: one contiguous 1-KB read per fragment, rolling ring 3 k-steps ahead);
//       epilogue: + bias, round to 16 bit, 8-B writes into six [rows][32] LDS images (q, k, v of the two heads) in the attention
//       core's swizzled row layout; the images are streamed to HBM as 128-B row segments (two adjacent heads);
//   P2  attention: two waves per head (alternate query tiles), EXACTLY eg_attention_fwd's arithmetic (scores^T = K Q^T with the
//       key on accumulator rows, soft-max in registers, the probabilities are the next MFMA's operand, V^T through
//       ds_read_b64_tr_b16) with K / Q fragments read from the images; ctx overwrites the head's q image row by row;
//   P3  out-proj partial sums: acc2[80 x 64 per wave] += ctx_chunk[80 x 64] Wo[:, chunk]^T (40 MFMAs, accumulators live in
//       registers across the four chunks, k runs in ascending order exactly as in the stand-alone product);
// final epilogue as eg_gemm_nt's: + bias, dropout, + residual (the LDS-resident x rows), 16-bit store.
// Measured (MI355X, 512 windows of S = 65, p = 0.1): 54.5 us per launch against 25 + 27 + 20 us for the three launches (47.7 us as
// the exact S = 65 instantiation, 53 us with norm1 in its tail); HBM traffic
// by PMC 21.6 MB fetched + 86.3 MB written = 1.06 x the algorithmic 102 MB.  The launch is bound by the attention core's vector
// work (exp + the dropout hash: ~1900 issue cycles per 16-query tile, 40 tiles per window over 4 waves, 3 : 2 between the two waves
// of a head at S = 65), not by bandwidth; static s_setprio for the matrix phases measured no change (57.9 vs 58.2 us by HIP events).
// Same MFMA chains, rounding points and dropout indices as the launches it replaces: q|k|v, lse, ctx and r1 are BIT-IDENTICAL
// to eg_gemm_nt -> eg_attention_fwd -> eg_gemm_nt (tests/test_gpu_attnblock.py).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short ab_s16x8;

constexpr int AR = 80;                         // rows of the window tile (5 MFMA row tiles): S <= 80
constexpr int AD = 256;                        // d_model
constexpr int AH = 8;                          // heads of 32
constexpr int A_XT = AR * AD * 2;              // 40,960 B: the window's x rows
constexpr int A_QK = AR * 64;                  // 5,120 B: one [80][32] image
constexpr int A_IMGH = 2 * A_QK + 96 * 64;     // 16,384 B per head of a chunk: q | k | v (v: 96 rows, rows 80..95 stay zero)
constexpr int A_LDS = A_XT + 2 * A_IMGH;       // 73,728 B
constexpr int A_TP = 68;                       // fp32 image pitch of the final epilogue (floats)
constexpr float kAScale = 0.17677669529663687f;   // 1/sqrt(32)

template <typename T>
struct ABArgs {
  const T* X; const T* Wqkv; const T* Wo; const float* bqkv; const float* bo;
  T* QKV; T* CTX; float* LSE; T* R1;
  const float* ln_gamma; const float* ln_beta; T* LN_OUT; float* ln_stats;     // LNF: y = LayerNorm(r1) in the final epilogue
  const eg_step_state* st;
  int NB, S;
  DropCfg da, d1;
};

__device__ __forceinline__ void ab_dma16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// [rows][32] 16-bit image, 64-B rows; the two 32-B halves of a row are swapped when (row >> 2) & 1 (csrc/attention.hip)
__device__ __forceinline__ int ab_img_off(int row, int c4) {
  return row * 64 + ((((c4 >> 1) ^ ((row >> 2) & 1))) << 5) + ((c4 & 1) << 4);
}
template <typename T>
__device__ __forceinline__ typename H16<T>::frag ab_frag_row(const char* img, int row, int g) {
  return *(const typename H16<T>::frag*)(img + ab_img_off(row, g));
}
// transposed fragment: slot 8g+j <-> row rbase + 16*(j>>2) + 4g + (j&3), column 16*dt + (lane&15)
template <typename T>
__device__ __forceinline__ typename H16<T>::frag ab_frag_tr(const char* img, int rbase, int dt, int lane) {
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  s16x4 part[2];
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int row = rbase + 16 * h2 + 4 * g + qq;
    const int off = row * 64 + ((dt ^ (g & 1)) << 5) + pp * 8;
    part[h2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + off));
  }
  ab_s16x8 t = {part[0][0], part[0][1], part[0][2], part[0][3], part[1][0], part[1][1], part[1][2], part[1][3]};
  return __builtin_bit_cast(typename H16<T>::frag, t);
}
template <typename T>
__device__ __forceinline__ typename H16<T>::frag ab_pack_frag(const f32x4& a, const f32x4& b) {
  u32x4 v;
  v[0] = H16<T>::pack2(a[0], a[1]);
  v[1] = H16<T>::pack2(a[2], a[3]);
  v[2] = H16<T>::pack2(b[0], b[1]);
  v[3] = H16<T>::pack2(b[2], b[3]);
  return __builtin_bit_cast(typename H16<T>::frag, v);
}

// NKTX > 0: the tile count is a compile-time constant (no wave-uniform branches inside the unrolled tile loops: the scheduler
// sees one straight-line body per query tile); TAIL: the last key tile holds ONE valid key (S = 16 n + 1) and evaluates one
// accumulator register per lane.  <0, false> is the general form.  LNF: the layer's first LayerNorm (A:293) runs in the final
// epilogue on the rows the workgroup has just completed (eg_epilogue_layernorm256) instead of as a launch that re-reads r1.
template <typename T, int NKTX, bool TAIL, bool LNF>
__global__ __launch_bounds__(256, 2) void attn_block_fwd_kernel(ABArgs<T> p) {
  typedef typename H16<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const xt = smem;
  char* const imgs = smem + A_XT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  const int b = blockIdx.x;                      // the window this workgroup owns
  const int S = p.S;
  const size_t row0 = (size_t)b * (size_t)S;     // its first token row
  const int nkt = NKTX ? NKTX : (S + 15) >> 4;   // key / query tiles in use (<= 5)

  // ---- x tile: instruction q moves rows 2q, 2q+1 (lane -> row half lane/32, LDS chunk position lane%32 holding global chunk
  //      pos ^ 2 (row & 7)); rows beyond S repeat row S-1 (finite values that never reach a stored result) ----
  {
    const int half = lane >> 5, pos = lane & 31;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int q = wn + 4 * i;
      const int r = 2 * q + half;
      const int row = min(r, S - 1);
      ab_dma16((const char*)(p.X + (row0 + row) * AD) + ((pos ^ ((r & 7) << 1)) << 4), xt + q * 1024);
    }
  }
  // rows 80..95 of both v images are read by the last key pair's transposed fragments and never written: zero them once
  if (tid < 128) *(u32x4*)(imgs + (tid >> 6) * A_IMGH + 2 * A_QK + 80 * 64 + (tid & 63) * 16) = (u32x4){0u, 0u, 0u, 0u};

  // ---- weight fragment streams (L2 -> registers), fragment order: Wqkv [chunk c][wave][k-step s: 8][tile j: 3][lane],
  //      Wo [chunk c][wave][k-step s: 2][tile j: 4][lane] ----
  const T* const wqp = p.Wqkv + (size_t)wn * (8 * 3 * 512) + lane * 8;
  const T* const wop = p.Wo + (size_t)wn * (2 * 4 * 512) + lane * 8;
  frag w1r[3][3], wor[2][4];
  auto req_w1 = [&](int c, int s, int slot) {
#pragma unroll
    for (int j = 0; j < 3; ++j) w1r[slot][j] = *(const frag*)(wqp + (size_t)c * (4 * 8 * 3 * 512) + (s * 3 + j) * 512);
  };
  auto req_wo = [&](int c) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) wor[s][j] = *(const frag*)(wop + (size_t)c * (4 * 2 * 4 * 512) + (s * 4 + j) * 512);
  };
#pragma unroll
  for (int s = 0; s < 3; ++s) req_w1(0, s, s);

  f32x4 acc2[5][4];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint32_t seed_lo = 0, seed_hi = 0;
  if (p.da.thresh | p.d1.thresh) { seed_lo = p.st->seed_lo; seed_hi = p.st->seed_hi; }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's part of the x tile has landed
  __syncthreads();                                           // ... and everybody else's

  const int hh = wn >> 1, role = wn & 1;                     // attention phase: two waves per head of the chunk
  char* const qimg = imgs + hh * A_IMGH;
  char* const kimg = qimg + A_QK;
  char* const vimg = qimg + 2 * A_QK;

  for (int c = 0; c < 4; ++c) {
    // addresses below derive from an opaque copy of the lane id, so they are recomputed per chunk instead of being hoisted out of
    // the loop (loop-invariant address registers pushed the kernel past its 256-register budget and into scratch)
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int l15 = lv & 15, g4 = lv >> 4, sw7 = (l15 & 7) << 1;   // chunk ^ 2 (row & 7): conflict-free under ds_read_b128's lane groups on 512-B rows (ffn.hip)
    // ================= P1: q|k|v columns of heads 2c, 2c+1 =================
    // column tile t = 3 wn + j of the chunk: head t / 6, part (t % 6) / 2 (q, k, v), 16-column half t % 2
    f32x4 acc1[5][3];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      frag xf[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) xf[i] = *(const frag*)(xt + (l15 + 16 * i) * 512 + (((4 * s + g4) ^ sw7) << 4));
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc1[i][j] = H16<T>::mfma(w1r[s % 3][j], xf[i], acc1[i][j]);
      if (s < 5) req_w1(c, s + 3, s % 3);       // ring three k-steps ahead (the next chunk's first fragments are requested after
    }                                             // the attention phase: held across it they pushed the kernel into scratch)
    float b1[3][4];
    int ioff[3];                                             // byte offset of tile j's images
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int t = 3 * wn + j;
      const int th = t / 6, part = (t % 6) >> 1, half = t & 1;
      load4(p.bqkv + part * AD + (2 * c + th) * 32 + 16 * half + 4 * g4, b1[j]);     // (lands while the barrier gathers the waves)
      ioff[j] = th * A_IMGH + part * A_QK;
    }
    __syncthreads();        // (A) every wave has left chunk c-1's images (its out-proj fragments are in registers)

    // epilogue 1: lane holds 4 consecutive columns of row l15 + 16 i  ->  8-B pieces of the six images
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int half = (3 * wn + j) & 1;
      const int c4 = 2 * half + (g4 >> 1);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int r = l15 + 16 * i;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc1[i][j][q] + b1[j][q];
        u32x2 pk;
        pk[0] = H16<T>::pack2(v[0], v[1]);
        pk[1] = H16<T>::pack2(v[2], v[3]);
        *(u32x2*)(imgs + ioff[j] + ab_img_off(r, c4) + ((g4 & 1) << 3)) = pk;
      }
    }
    __syncthreads();        // (B) the chunk's q, k, v images are complete

    // the stored k and v rows: per row and part 128 contiguous bytes (heads 2c, 2c+1), 16 B per thread.  (The q rows are
    // stored below by the wave that is about to overwrite them with ctx: no barrier needed between this loop and that write.)
    for (int idx = tid; idx < S * 16; idx += 256) {
      const int row = idx >> 4, pc = idx & 15;
      const int part = 1 + (pc >> 3), th = (pc >> 2) & 1, c4 = pc & 3;
      const u32x4 o = *(const u32x4*)(imgs + th * A_IMGH + part * A_QK + ab_img_off(row, c4));
      *(u32x4*)(p.QKV + (row0 + row) * (3 * AD) + part * AD + (2 * c + th) * 32 + c4 * 8) = o;
    }

    // ================= P2: attention of head 2c + hh, query tiles role, role + 2, .. =================
    {
      const int h = 2 * c + hh;
      constexpr int ktail = TAIL ? NKTX - 1 : -1;
      frag kf[5];
#pragma unroll
      for (int kt = 0; kt < 5; ++kt) kf[kt] = ab_frag_row<T>(kimg, kt * 16 + l15, g4);
      for (int qt = role; qt < nkt; qt += 2) {
        const int q = qt * 16 + l15;
        const frag qf = ab_frag_row<T>(qimg, q, g4);
        {                                                    // this tile's q rows leave the chip before ctx replaces them
          const int row = qt * 16 + (lane >> 2), c4 = lane & 3;
          if (row < S) *(u32x4*)(p.QKV + (row0 + row) * (3 * AD) + h * 32 + c4 * 8) = *(const u32x4*)(qimg + ab_img_off(row, c4));
        }
        f32x4 s[6];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 6; ++kt) {
          s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (kt < 5 && kt < nkt) {
            s[kt] = H16<T>::mfma(kf[kt < 5 ? kt : 0], qf, s[kt]);
            if (kt == ktail) {                               // single valid key: one accumulator register (csrc/attention.hip)
              const float v = g4 == 0 ? s[kt][0] * kAScale : -INFINITY;
              s[kt][0] = v;
              mx = fmaxf(mx, v);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * g4 + r;
                const float v = key < S ? s[kt][r] * kAScale : -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
              }
            }
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 5; ++kt) {
          if (kt < nkt) {
            if (kt == ktail) {
              const float pe = __expf(s[kt][0] - mx);
              s[kt] = (f32x4){pe, 0.f, 0.f, 0.f};
              sum += pe;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float pe = __expf(s[kt][r] - mx);
                s[kt][r] = pe;
                sum += pe;
              }
            }
          }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if (g4 == 0 && q < S) p.LSE[((size_t)b * AH + h) * S + q] = mx + __logf(sum);
        const uint32_t rowidx = (uint32_t)((b * AH + h) * S + q) * (uint32_t)((S + 1) & ~1);
#pragma unroll
        for (int kt = 0; kt < 5; ++kt) {
          if (kt < nkt) {
            if (kt == ktail) {
              s[kt][0] = eg_dropout(s[kt][0] * inv, p.da, seed_lo, seed_hi, rowidx + (uint32_t)(kt * 16 + 4 * g4));
            } else {
              float pv[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) pv[r] = s[kt][r] * inv;
              eg_dropout_run<4>(pv, p.da, seed_lo, seed_hi, rowidx + (uint32_t)(kt * 16 + 4 * g4));
#pragma unroll
              for (int r = 0; r < 4; ++r) s[kt][r] = pv[r];
            }
          }
        }
        f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kp = 0; kp < 3; ++kp) {
          if (2 * kp < nkt) {
            const frag pf = ab_pack_frag<T>(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
              const frag vf = ab_frag_tr<T>(vimg, 32 * kp, dt, lane);
              o[dt] = H16<T>::mfma(vf, pf, o[dt]);
            }
          }
        }
        // ctx rows of this query tile replace its q rows (read above, by this wave only): lane holds dims 16 dt + 4 g4 .. + 3
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          u32x2 pk;
          pk[0] = H16<T>::pack2(o[dt][0], o[dt][1]);
          pk[1] = H16<T>::pack2(o[dt][2], o[dt][3]);
          *(u32x2*)(qimg + ab_img_off(q, 2 * dt + (g4 >> 1)) + ((g4 & 1) << 3)) = pk;
        }
      }
    }
    req_wo(c);              // the chunk's out-proj fragments travel while the barrier gathers the waves
    if (c + 1 < 4) {        // ... and so do the next chunk's first q|k|v fragments (they land under P3)
#pragma unroll
      for (int s = 0; s < 3; ++s) req_w1(c + 1, s, s);
    }
    __syncthreads();        // (C) ctx of both heads is complete (in the q images)

    // the stored ctx rows: 128 contiguous bytes per row (heads 2c, 2c+1)
    for (int idx = tid; idx < S * 8; idx += 256) {
      const int row = idx >> 3, th = (idx >> 2) & 1, c4 = idx & 3;
      const u32x4 o = *(const u32x4*)(imgs + th * A_IMGH + ab_img_off(row, c4));
      *(u32x4*)(p.CTX + (row0 + row) * AD + (2 * c + th) * 32 + c4 * 8) = o;
    }

    // ================= P3: out-proj partial sums over the chunk's 64 ctx columns =================
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag cf[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) cf[i] = ab_frag_row<T>(imgs + s * A_IMGH, l15 + 16 * i, g4);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[i][j] = H16<T>::mfma(wor[s][j], cf[i], acc2[i][j]);
    }
  }
  __syncthreads();          // every wave has left the images: they become the fp32 image of the final epilogue

  // ---- final epilogue (eg_gemm_nt's order): + bias, dropout, + residual (the x rows, from LDS), store ----
  float* timg = (float*)(imgs + wn * (16 * A_TP * 4));
  const int er = lane >> 2, ec = lane & 3;
  const int n = 64 * wn + 16 * ec;
  float bv[16];
  load8(p.bo + n, bv);
  load8(p.bo + n + 8, bv + 8);
  float vv[LNF ? 5 : 1][16];                       // LNF: the stored r1 values of this lane's rows, for the LayerNorm below
  if (LNF) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) vv[i][j] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int r = 16 * i + er;
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(timg + l15 * A_TP + 16 * j + 4 * g4) = acc2[i][j];
    if (16 * i >= S) break;                        // workgroup-uniform: tiles wholly beyond the window
    float v[16];
    load8(timg + er * A_TP + 16 * ec, v);
    load8(timg + er * A_TP + 16 * ec + 8, v + 8);
    if (r < S) {
      const size_t m = row0 + r;
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] += bv[j];
      if (p.d1.thresh) {
        const uint32_t idx = (uint32_t)m * (uint32_t)AD + (uint32_t)n;
        float (&v0)[8] = *(float (*)[8])v;
        float (&v1)[8] = *(float (*)[8])(v + 8);
        eg_dropout_run<8>(v0, p.d1, seed_lo, seed_hi, idx);
        eg_dropout_run<8>(v1, p.d1, seed_lo, seed_hi, idx + 8);
      }
      const u32x4 e0 = *(const u32x4*)(xt + r * 512 + ((((n >> 3)) ^ ((r & 7) << 1)) << 4));
      const u32x4 e1 = *(const u32x4*)(xt + r * 512 + ((((n >> 3) + 1) ^ ((r & 7) << 1)) << 4));
      float rv[16];
      load8((const T*)&e0, rv);
      load8((const T*)&e1, rv + 8);
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] += rv[j];
      T* pc = p.R1 + m * AD + n;
      store8(pc, v);
      store8(pc + 8, v + 8);
      if (LNF) {
#pragma unroll
        for (int j = 0; j < 16; ++j) vv[i][j] = round_store<T>(v[j]);
      }
    }
  }
  if constexpr (LNF)
    eg_epilogue_layernorm256<T>(vv, (float*)(imgs + 4 * (16 * A_TP * 4)), wn, lane, S, row0, p.ln_gamma, p.ln_beta, p.LN_OUT, p.ln_stats);
}

template <typename T>
static int ab_launch(const eg_attn_block_desc* d, hipStream_t s) {
  ABArgs<T> p;
  p.X = (const T*)d->x; p.Wqkv = (const T*)d->wqkv_frag; p.Wo = (const T*)d->wo_frag; p.bqkv = d->bqkv; p.bo = d->bo;
  p.QKV = (T*)d->qkv; p.CTX = (T*)d->ctx; p.LSE = d->lse; p.R1 = (T*)d->r1; p.st = d->state;
  p.NB = d->NB; p.S = d->S;
  p.da = make_drop(d->attn_drop_p, d->attn_drop_site);
  p.d1 = make_drop(d->out_drop_p, d->out_drop_site);
  p.ln_gamma = d->ln_gamma; p.ln_beta = d->ln_beta; p.LN_OUT = (T*)d->ln_out; p.ln_stats = d->ln_stats;
  const bool lnf = d->ln_out != nullptr;
#define AB_LAUNCH(N_, T_, L_)                                                                                                  \
  do {                                                                                                                         \
    static bool attr = false;                                                                                                  \
    if (!attr) {                                                                                                               \
      (void)hipFuncSetAttribute((const void*)attn_block_fwd_kernel<T, N_, T_, L_>, hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS); \
      attr = true;                                                                                                             \
    }                                                                                                                          \
    hipLaunchKernelGGL((attn_block_fwd_kernel<T, N_, T_, L_>), dim3(d->NB), dim3(256), A_LDS, s, p);                           \
  } while (0)
  if (d->S == 65) { if (lnf) AB_LAUNCH(5, true, true); else AB_LAUNCH(5, true, false); }     // class token + 64 positions
  else { if (lnf) AB_LAUNCH(0, false, true); else AB_LAUNCH(0, false, false); }
#undef AB_LAUNCH
  EG_LAUNCH_CHECK("attn_block_fwd");
  return 0;
}

}  // namespace

extern "C" int eg_attn_block_ok(int S, int d_model, int num_heads, int dtype) {
  return (dtype == EG_BF16 || dtype == EG_F16) && d_model == AD && num_heads == AH && S >= 1 && S <= AR;
}

extern "C" int eg_attn_block_fwd(const eg_attn_block_desc* d, void* stream) {
  EG_CHECK(d && d->x && d->wqkv_frag && d->wo_frag && d->bqkv && d->bo && d->qkv && d->ctx && d->lse && d->r1,
           "eg_attn_block_fwd: null operand");
  EG_CHECK(eg_attn_block_ok(d->S, d->d_model, d->num_heads, d->dtype),
           "eg_attn_block_fwd: needs a 16-bit dtype, d_model == 256, 8 heads and S <= 80 (got dtype %d, d %d, H %d, S %d)", d->dtype,
           d->d_model, d->num_heads, d->S);
  EG_CHECK(d->NB > 0 && (long long)d->NB * d->S * 768 < (1ll << 31), "eg_attn_block_fwd: NB=%d", d->NB);
  EG_CHECK(!d->ln_out || (d->ln_gamma && d->ln_beta && (uintptr_t)d->ln_out % 16 == 0),
           "eg_attn_block_fwd: the fused LayerNorm needs gamma, beta and a 16-B aligned output");
  EG_CHECK((long long)d->NB * AH * d->S * ((d->S + 1) & ~1) < (1ll << 32), "eg_attn_block_fwd: NB*H*S*S exceeds the 32-bit dropout index");
  const float ps[2] = {d->attn_drop_p, d->out_drop_p};
  for (float q : ps) EG_CHECK(q >= 0.f && q < 1.f, "eg_attn_block_fwd: dropout p");
  EG_CHECK((ps[0] == 0.f && ps[1] == 0.f) || d->state, "eg_attn_block_fwd: dropout needs a step state");
  EG_CHECK(((uintptr_t)d->x | (uintptr_t)d->wqkv_frag | (uintptr_t)d->wo_frag | (uintptr_t)d->qkv | (uintptr_t)d->ctx |
            (uintptr_t)d->r1 | (uintptr_t)d->bqkv | (uintptr_t)d->bo) % 16 == 0, "eg_attn_block_fwd: operands must be 16-B aligned");
  hipStream_t s = (hipStream_t)stream;
  return d->dtype == EG_F16 ? ab_launch<f16_t>(d, s) : ab_launch<bf16_t>(d, s);
}
