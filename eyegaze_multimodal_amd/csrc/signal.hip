// Signal-processing front ends that feed extra tokens to the encoder (forward only: their inputs carry no gradient).
//   * inter-stream synchrony ("IBS") connectivity: per band an FFT-mask band-pass + FFT Hilbert transform, then seven
//     channel-by-channel reductions over time  (dual_eeg_transformer.py:473-819; scalar variant :178-470)
//   * STFT log-magnitude image of every channel  (dual_eeg_transformer.py:98-121)
// FFTs are radix-2 Stockham autosort transforms held entirely in LDS (one workgroup per signal); the pair
// reductions keep both players' band signals and phases of an 8x8 channel tile in LDS (128 KiB).
#include "common.h"

namespace {

constexpr int MAX_BANDS = 8;
struct BandTable {
  int nbands;
  float lo[MAX_BANDS], hi[MAX_BANDS];
};

typedef float2 cf;
__device__ __forceinline__ cf cmul(cf a, cf b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// in-LDS radix-4 Stockham autosort FFT of length N (power of two; one closing radix-2 stage when log2 N is odd), all threads of the
// block cooperate.  x: input/output buffer, y: scratch, tw: table exp(-2 pi i k / N), k < N/2.  inverse => conjugated twiddles, no
// scaling.  Returns the buffer that holds the result (natural order).
// A radix-4 stage is two radix-2 stages at one pass through LDS and one barrier: per output point 2.1 instead of 7 index / address
// instructions and half the LDS traffic (the radix-2 form spent 19 vector instructions per butterfly, 10 of them arithmetic, and
// was half of eg_ibs_analytic).  s, the stride, is a power of two: p = j >> log2 s.
__device__ cf* fft_stockham(cf* x, cf* y, const cf* tw, int N, bool inverse) {
  int n = N, ls = 0;
  const float isg = inverse ? 1.f : -1.f;                        // forward: multiply by -i, inverse: by +i
  while (n >= 4) {
    const int m = n >> 2;
    const int tstep = N / n;
    const int s = 1 << ls;
    for (int j = threadIdx.x; j < (N >> 2); j += blockDim.x) {
      const int p = j >> ls, q = j & (s - 1);
      cf w1 = tw[p * tstep], w2 = tw[2 * p * tstep];
      if (inverse) { w1.y = -w1.y; w2.y = -w2.y; }
      const cf w3 = cmul(w1, w2);
      const cf* in = x + q + s * p;
      const cf a0 = in[0], a1 = in[s * m], a2 = in[2 * s * m], a3 = in[3 * s * m];
      const cf t0 = make_float2(a0.x + a2.x, a0.y + a2.y), t1 = make_float2(a0.x - a2.x, a0.y - a2.y);
      const cf t2 = make_float2(a1.x + a3.x, a1.y + a3.y);
      const cf d3 = make_float2(a1.x - a3.x, a1.y - a3.y);
      const cf t3 = make_float2(-isg * d3.y, isg * d3.x);        // (a1 - a3) * (+-i)
      cf* out = y + q + s * 4 * p;
      out[0] = make_float2(t0.x + t2.x, t0.y + t2.y);
      out[s] = cmul(make_float2(t1.x + t3.x, t1.y + t3.y), w1);
      out[2 * s] = cmul(make_float2(t0.x - t2.x, t0.y - t2.y), w2);
      out[3 * s] = cmul(make_float2(t1.x - t3.x, t1.y - t3.y), w3);
    }
    __syncthreads();
    cf* t = x; x = y; y = t;
    n = m;
    ls += 2;
  }
  if (n == 2) {                                                  // closing radix-2 stage: s = N/2, p = 0, unit twiddle
    const int s = N >> 1;
    for (int q = threadIdx.x; q < s; q += blockDim.x) {
      const cf a = x[q], b = x[q + s];
      y[q] = make_float2(a.x + b.x, a.y + b.y);
      y[q + s] = make_float2(a.x - b.x, a.y - b.y);
    }
    __syncthreads();
    cf* t = x; x = y; y = t;
  }
  return x;
}

// three block sums behind one pair of barriers; per value the order of additions is block_sum's
__device__ __forceinline__ void block_sum3(float& a, float& b, float& c, float* red) {
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = a; red[8 + (threadIdx.x >> 6)] = b; red[16 + (threadIdx.x >> 6)] = c;
  }
  __syncthreads();
  a = b = c = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { a += red[i]; b += red[8 + i]; c += red[16 + i]; }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

// one block per signal (window, channel):  x [T] -> for every band: band-limited signal xb, instantaneous phase,
// (mean, 1/(std_unbiased+1e-8)) of xb and of xb^2;  plus the complex spectrum bins [0, nbin) of the raw signal.
__global__ __launch_bounds__(256) void ibs_analytic_kernel(const float* __restrict__ x, float* __restrict__ xb,
                                                           float* __restrict__ phase, float* __restrict__ stats,
                                                           cf* __restrict__ spec, int nsig, int T, float fs, int nbin,
                                                           BandTable bt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cf* bufX = (cf*)smem;        // [T] spectrum (kept)
  cf* bufA = bufX + T;         // [T] work
  cf* bufB = bufA + T;         // [T] work
  cf* tw = bufB + T;           // [T/2]
  __shared__ float red[24];
  const int sig = blockIdx.x;
  for (int k = threadIdx.x; k < (T >> 1); k += blockDim.x) {
    float sn, cs;
    sincospif(-2.0f * (float)k / (float)T, &sn, &cs);
    tw[k] = make_float2(cs, sn);
  }
  for (int t = threadIdx.x; t < T; t += blockDim.x) bufA[t] = make_float2(x[(size_t)sig * T + t], 0.f);
  __syncthreads();
  cf* X = fft_stockham(bufA, bufB, tw, T, false);
  for (int k = threadIdx.x; k < T; k += blockDim.x) bufX[k] = X[k];
  __syncthreads();
  for (int k = threadIdx.x; k < nbin; k += blockDim.x) spec[(size_t)sig * nbin + k] = bufX[k];
  const float df = fs / (float)T;
  const float invT = 1.0f / (float)T;
  for (int b = 0; b < bt.nbands; ++b) {
    for (int k = threadIdx.x; k < T; k += blockDim.x) {
      const float f = (float)k * df;
      float h = 0.f;
      if (k <= (T >> 1) && f >= bt.lo[b] && f <= bt.hi[b]) h = (k == 0 || k == (T >> 1)) ? 1.f : 2.f;
      bufA[k] = make_float2(bufX[k].x * h, bufX[k].y * h);
    }
    __syncthreads();
    cf* a = fft_stockham(bufA, bufB, tw, T, true);
    float s1 = 0.f, s2 = 0.f, s4 = 0.f;
    const size_t base = ((size_t)b * nsig + sig) * T;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
      const float re = a[t].x * invT, im = a[t].y * invT;
      xb[base + t] = re;
      phase[base + t] = atan2f(im, re);
      const float p = re * re;
      s1 += re; s2 += p; s4 += p * p;
    }
    block_sum3(s1, s2, s4, red);
    const float S1 = s1, S2 = s2, S4 = s4;
    if (threadIdx.x == 0) {
      const float n = (float)T;
      const float mx = S1 / n, mp = S2 / n;
      const float vx = fmaxf((S2 - n * mx * mx) / (n - 1.f), 0.f);
      const float vp = fmaxf((S4 - n * mp * mp) / (n - 1.f), 0.f);
      float* st = stats + ((size_t)b * nsig + sig) * 4;
      st[0] = mx; st[1] = 1.0f / (sqrtf(vx) + 1e-8f);
      st[2] = mp; st[3] = 1.0f / (sqrtf(vp) + 1e-8f);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

// sin and cos of an atan2f result (|x| <= pi): quadrant by Cody-Waite (pi/2 in two pieces), then the two degree-7 / degree-8
// minimax polynomials of the Cephes single-precision sinf / cosf on [-pi/4, pi/4].  Absolute error <= 9.3e-8 over [-pi, pi]
// (checked on 4 M points against float64; a correctly rounded result is off by up to 6.2e-8), at 25 vector instructions for the pair
// instead of the ~90 of sincosf, whose argument reduction has to cover every float.
__device__ __forceinline__ void sincos_atan2(float x, float& s, float& c) {
  const float kf = rintf(x * 0.636619772f);
  float r = fmaf(kf, -1.5707963705f, x);
  r = fmaf(kf, 4.3711388e-8f, r);
  const float z = r * r;
  float sp = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  sp = fmaf(sp * z, r, r);
  float cp = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  cp = fmaf(cp * z, z, fmaf(-0.5f, z, 1.0f));
  const int n = (int)kf;
  const bool sw = n & 1;
  const float ss = sw ? cp : sp, cc = sw ? sp : cp;
  s = (n & 2) ? -ss : ss;
  c = ((n + 1) & 2) ? -cc : cc;
}

// 1-D grid of B * nbands * tiles_i * tiles_j workgroups: an 8x8 channel tile of player 1 x player 2 each.  Consecutive workgroup ids
// go to consecutive XCDs, so the id is turned round first (XCD x owns one contiguous eighth of the logical ids): the tiles_i * tiles_j
// tiles of one (window pair, band) -- which read each channel row tiles_j (tiles_i) times -- then run next to each other on ONE
// XCD and find the rows in its L2 (round 3's first form, grid (B, nbands, tiles), had them 1536 workgroups apart: 3.44 GB fetched
// for 0.81 GB of rows at C = 32).
// out conn [B, nbands, 7, C, C] in the reference's feature order [PLV, PLI, wPLI, Coherence, Power_Corr, Phase_Diff, Time_Corr]
// One time step of a channel is staged into LDS ONCE per tile as the quad (band signal a, phase p, cos p, sin p), so the 64 pairs
// of the tile cost a handful of multiply-adds per time step: cos(p1 - p2) = c1 c2 + s1 s2, sin(p1 - p2) = s1 c2 - c1 s2 instead of
// one sincosf per PAIR and time step.  A lane owns one time step and walks its wave's 16 pairs (2 channels of player 1 x 8 of
// player 2); the two player-1 channels ride in the two halves of packed fp32 operations (v_pk_fma/mul/add_f32: 2 flops per lane and
// issue slot), which with sign(d) as clamp(d * 2^96 * 2^96, -1, 1) and |d| = sign(d) * d brings a pair and time step from 21.8
// vector instructions to 10 (the sum of w over time is closed-form from the channel statistics and not accumulated).  Per lane the time steps and so the summation order are those of the earlier kernels.
constexpr int IBS_RED_FLOATS = 32 * 68;   // per wave: 32 values x (64 lanes + 4 pad)
constexpr int IBS_TC = 256;          // time steps per LDS tile: 16 channels x 256 x 16 B = 64 KB -> two workgroups per CU;
                                     // one's staging (memory latency) runs beside the other's pair loop
__global__ __launch_bounds__(256, 2) void ibs_pairs_kernel(const float* __restrict__ xb, const float* __restrict__ phase,
                                                        const float* __restrict__ stats, const cf* __restrict__ spec,
                                                        float* __restrict__ conn, int B, int C, int T, float fs, int nbin,
                                                        BandTable bt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Tc = min(T, IBS_TC);
  f32x4* const lds = (f32x4*)smem;                               // [player * 8 + channel][Tc] quads (a, p, cos p, sin p)
  const int tj = (C + 7) / 8;
  const int tiles = tj * tj;
  int wg = blockIdx.x;
  if ((gridDim.x & 7) == 0) wg = (wg & 7) * (gridDim.x >> 3) + (wg >> 3);
  const int tile = wg % tiles, bb = wg / tiles;
  const int band = bb % bt.nbands, b = bb / bt.nbands;
  const int i0 = (tile / tj) * 8, j0 = (tile % tj) * 8;
  const int nsig = 2 * B * C;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x2 acc[8][8];                                               // [player-2 channel v][feature]; halves = player-1 channels 2 wave, 2 wave + 1
#pragma unroll
  for (int v = 0; v < 8; ++v)
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[v][f] = (f32x2){0.f, 0.f};
  // z-score constants of this wave's channels: player 1 channels 2 wave, 2 wave + 1; player 2 channels 0..7 of the tile
  f32x2 m1, r1, mp1, rp1;
  float m2[8], r2[8], mp2[8], rp2[8];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float* s1 = stats + ((size_t)band * nsig + (size_t)b * C + min(i0 + 2 * wave + u, C - 1)) * 4;
    m1[u] = s1[0]; r1[u] = s1[1]; mp1[u] = s1[2]; rp1[u] = s1[3];
  }
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const float* s2 = stats + ((size_t)band * nsig + (size_t)(b + B) * C + min(j0 + v, C - 1)) * 4;
    m2[v] = s2[0]; r2[v] = s2[1]; mp2[v] = s2[2]; rp2[v] = s2[3];
  }
  // staging: a thread takes one time step of a channel per element -- 4-B loads, coalesced across the wave, and ONE 16-B LDS write at
  // a 16-B lane stride (free of bank conflicts; four time steps per thread would write at a 64-B stride, four ways conflicted).
  // The (signal, phase) values of chunk c + 1 are requested BEFORE the pair loop of chunk c and converted after it, so their memory
  // latency runs under the loop (T is a power of two >= 64, so every chunk has Tc steps: 16 Tc / 256 <= 16 elements per thread).
  const int lt = __builtin_ctz(Tc);
  const int nel = 16 * Tc;
  const float* const xbB = xb + (size_t)band * nsig * T;         // this band's rows; element offsets below fit 32 bits (nsig * T < 2^31)
  const float* const phB = phase + (size_t)band * nsig * T;
  float av[16], pv[16];
  auto request = [&](int t0) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = min((int)threadIdx.x + 256 * u, nel - 1);
      const int pc = e >> lt, c = pc & 7;                         // pc = player * 8 + channel of the tile
      const uint32_t row = (pc & 8) ? (uint32_t)((b + B) * C + min(j0 + c, C - 1)) : (uint32_t)(b * C + min(i0 + c, C - 1));
      const uint32_t o = row * (uint32_t)T + (uint32_t)(t0 + (e & (Tc - 1)));
      av[u] = xbB[o];
      pv[u] = phB[o];
    }
  };
  request(0);
  for (int t0 = 0; t0 < T; t0 += Tc) {
    const int tn = Tc;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int e = threadIdx.x + 256 * u;
      if (e < nel) {
        float s_, c_;
        sincos_atan2(pv[u], s_, c_);
        lds[e] = (f32x4){av[u], pv[u], c_, s_};                  // (e >> lt) * Tc + (e & (Tc - 1)) = e
      }
    }
    __syncthreads();
    if (t0 + Tc < T) request(t0 + Tc);
    const f32x4* const P1 = lds + (2 * wave) * Tc;               // player 1, this wave's first channel
    const f32x4* const P2 = lds + 8 * Tc;                        // player 2, channel 0
    for (int t = lane; t < tn; t += 64) {
      const f32x4 x0 = P1[t], x1 = P1[Tc + t];
      const f32x2 a1 = {x0[0], x1[0]}, p1 = {x0[1], x1[1]}, c1 = {x0[2], x1[2]}, s1 = {x0[3], x1[3]};
      const f32x2 q1 = a1 * a1;
      const f32x2 zq1 = (q1 - mp1) * rp1, za1 = (a1 - m1) * r1;
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const f32x4 y = P2[v * Tc + t];
        const float a2 = y[0], p2 = y[1], c2 = y[2], s2 = y[3];
        const float q2 = a2 * a2;
        const float zq2 = (q2 - mp2[v]) * rp2[v], za2 = (a2 - m2[v]) * r2[v];
        const f32x2 d = p1 - p2;
        const f32x2 big = (d * 0x1p96f) * 0x1p96f;                // 0 stays 0; anything else, subnormal differences included, passes 1
        const f32x2 sg = {__builtin_amdgcn_fmed3f(big[0], -1.f, 1.f), __builtin_amdgcn_fmed3f(big[1], -1.f, 1.f)};
        const f32x2 w = (q1 + q2) * 0.5f;
        acc[v][0] += c1 * c2; acc[v][0] += s1 * s2;              // cos(p1 - p2), one fma per product
        acc[v][1] += s1 * c2; acc[v][1] -= c1 * s2;              // sin(p1 - p2)
        acc[v][2] += sg; acc[v][3] += sg * w;
        acc[v][5] += sg * d;                                     // |d|
        acc[v][6] += zq1 * zq2;
        acc[v][7] += za1 * za2;
      }
    }
  }
  const float invT = 1.0f / (float)T;
  const float df = fs / (float)T;
  // ---- coherence: mean over the T/2+1 rFFT bins of |X1 X2*|^2 / (|X1|^2 |X2|^2 + 1e-8); only in-band bins are non-zero.
  //      The in-band spectra of the tile's 16 channels are staged into LDS once (the time-domain tile is finished), and a pair walks
  //      bins [klo, khi] only. ----
  int klo = max(0, (int)ceilf(bt.lo[band] / df)), khi = min(nbin - 1, (int)floorf(bt.hi[band] / df));
  // the band test of the reference is on f = k * df evaluated in fp32 (D:551): settle the two edges with that very predicate
  while (klo > 0 && (float)(klo - 1) * df >= bt.lo[band]) --klo;
  while (klo < nbin && (float)klo * df < bt.lo[band]) ++klo;
  while (khi + 1 < nbin && (float)(khi + 1) * df <= bt.hi[band]) ++khi;
  while (khi >= 0 && (float)khi * df > bt.hi[band]) --khi;
  const int nin = max(0, khi - klo + 1);                 // <= 179 bins (0.5 .. 45 Hz at df = 0.25 Hz) x 16 channels x 8 B = 23 KB
  __syncthreads();
  // LDS of the tail: spectra [player * 8 + channel][nin] | per wave 32 x 68 floats of transposed partial sums | per wave 128 totals
  cf* const sp = (cf*)smem;
  float* const red = (float*)(smem + (size_t)16 * nbin * sizeof(cf)) + wave * IBS_RED_FLOATS;
  float* const tot = (float*)(smem + (size_t)16 * nbin * sizeof(cf)) + 4 * IBS_RED_FLOATS + wave * 128;
  for (int idx = threadIdx.x; idx < 16 * nin; idx += blockDim.x) {
    const int pc = idx / nin, k = idx - pc * nin;
    const int pl = pc >> 3, c = pc & 7;
    const int ch = pl ? min(j0 + c, C - 1) : min(i0 + c, C - 1);
    sp[idx] = spec[((size_t)(pl ? b + B : b) * C + ch) * nbin + klo + k];
  }
  // the wave's 16 pairs x 8 sums over its 64 lanes: every lane writes its 32 values of a pass as a column, lane l < 32 then adds up
  // row l (16 reads of 16 B).  128 values cost 128 4-B writes and 64 16-B reads per wave -- the butterfly (six dependent
  // ds_bpermute per value, 768 per wave, two or three in flight) was a third of the launch.
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int a = 4 * pass + (i >> 3), f = i & 7;             // pair a = u * 8 + v of the wave, feature f
      red[i * 68 + lane] = (f == 4) ? 0.f : acc[a & 7][f][a >> 3];   // slot 4 (sum of w) is closed-form, below
    }
    __syncthreads();
    if (lane < 32) {
      const f32x4* row = (const f32x4*)(red + lane * 68);
      f32x4 sacc = row[0];
#pragma unroll
      for (int q = 1; q < 16; ++q) sacc += row[q];
      tot[32 * pass + lane] = (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]);
    }
    __syncthreads();
  }
  // coherence: lane = (pair a, quarter of the bins); the quarters meet in two shuffles
  const int a = lane & 15, sub = lane >> 4;
  const int pr = wave * 16 + a, ii = pr >> 3, jj = pr & 7;
  float coh = 0.f;
  {
    const cf* f1 = sp + ii * nin;
    const cf* f2 = sp + (8 + jj) * nin;
    for (int k = sub; k < nin; k += 4) {
      const cf u = f1[k], w = f2[k];
      const cf xy = cmul(u, make_float2(w.x, -w.y));
      const float num = xy.x * xy.x + xy.y * xy.y;
      const float pxx = u.x * u.x + u.y * u.y, pyy = w.x * w.x + w.y * w.y;
      coh += num / (pxx * pyy + 1e-8f);
    }
    coh += __shfl_xor(coh, 16, 64);
    coh += __shfl_xor(coh, 32, 64);
    coh /= (float)(T / 2 + 1);
  }
  if (sub == 0 && i0 + ii < C && j0 + jj < C) {
    const f32x4 v0 = *(const f32x4*)(tot + 8 * a), v1 = *(const f32x4*)(tot + 8 * a + 4);
    float* o = conn + (((size_t)b * bt.nbands + band) * 7) * C * C + (size_t)(i0 + ii) * C + (j0 + jj);
    const size_t fs_ = (size_t)C * C;
    o[0 * fs_] = sqrtf(v0[0] * v0[0] + v0[1] * v0[1]) * invT;
    o[1 * fs_] = fabsf(v0[2] * invT);
    // sum over time of w = (q1 + q2) / 2 needs no pair loop: the per-channel mean of q = a^2 is in the statistics
    const float sw = 0.5f * (float)T * (stats[((size_t)band * nsig + (size_t)b * C + (i0 + ii)) * 4 + 2] +
                                        stats[((size_t)band * nsig + (size_t)(b + B) * C + (j0 + jj)) * 4 + 2]);
    o[2 * fs_] = fabsf(v0[3] / (sw + 1e-8f));
    o[3 * fs_] = coh;
    o[4 * fs_] = v1[2] * invT;
    o[5 * fs_] = v1[1] * invT;
    o[6 * fs_] = v1[3] * invT;
  }
}

// scalar variant (4 bands): block per (window pair b, band) -> 7 global features over (C, T)  (D:436-458)
__global__ __launch_bounds__(256) void ibs_scalar_kernel(const float* __restrict__ xb, const float* __restrict__ phase,
                                                         const cf* __restrict__ spec, float* __restrict__ feats, int B,
                                                         int C, int T, float fs, int nbin, BandTable bt, int band0, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* e1 = (float*)smem;  // [T] channel-averaged band signal, player 1
  float* e2 = e1 + T;
  __shared__ float red[8];
  const int b = blockIdx.x, bi = blockIdx.y, band = band0 + bi;
  const int nsig = 2 * B * C;
  const float* X1 = xb + ((size_t)band * nsig + (size_t)b * C) * T;
  const float* X2 = xb + ((size_t)band * nsig + (size_t)(b + B) * C) * T;
  const float* P1 = phase + ((size_t)band * nsig + (size_t)b * C) * T;
  const float* P2 = phase + ((size_t)band * nsig + (size_t)(b + B) * C) * T;
  float cs_ = 0.f, sn_ = 0.f, sg_ = 0.f, sw_ = 0.f, w_ = 0.f, d_ = 0.f, q1 = 0.f, q11 = 0.f, q2 = 0.f, q22 = 0.f, q12 = 0.f;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    float a1 = 0.f, a2 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float u = X1[(size_t)c * T + t], v = X2[(size_t)c * T + t];
      const float d = P1[(size_t)c * T + t] - P2[(size_t)c * T + t];
      float sn, cs;
      sincosf(d, &sn, &cs);
      const float sg = sgn(d), pu = u * u, pv = v * v, w = (pu + pv) * 0.5f;
      cs_ += cs; sn_ += sn; sg_ += sg; sw_ += sg * w; w_ += w; d_ += d;
      q1 += pu; q11 += pu * pu; q2 += pv; q22 += pv * pv; q12 += pu * pv;
      a1 += u; a2 += v;
    }
    e1[t] = a1 / (float)C;
    e2[t] = a2 / (float)C;
  }
  const float n = (float)C * (float)T;
  const float Scs = block_sum(cs_, red), Ssn = block_sum(sn_, red), Ssg = block_sum(sg_, red), Ssw = block_sum(sw_, red);
  const float Sw = block_sum(w_, red), Sd = block_sum(d_, red);
  const float Q1 = block_sum(q1, red), Q11 = block_sum(q11, red), Q2 = block_sum(q2, red), Q22 = block_sum(q22, red);
  const float Q12 = block_sum(q12, red);
  // time correlation of the channel averages
  float s1 = 0.f, s11 = 0.f, s2 = 0.f, s22 = 0.f, s12 = 0.f;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float u = e1[t], v = e2[t];
    s1 += u; s11 += u * u; s2 += v; s22 += v * v; s12 += u * v;
  }
  const float S1 = block_sum(s1, red), S11 = block_sum(s11, red), S2 = block_sum(s2, red), S22 = block_sum(s22, red);
  const float S12 = block_sum(s12, red);
  // coherence with channel-averaged cross / auto spectra (D:378-392)
  float coh = 0.f;
  const float df = fs / (float)T;
  for (int k = threadIdx.x; k < nbin; k += blockDim.x) {
    const float f = (float)k * df;
    if (f >= bt.lo[band] && f <= bt.hi[band]) {
      cf pxy = make_float2(0.f, 0.f);
      float pxx = 0.f, pyy = 0.f;
      for (int c = 0; c < C; ++c) {
        const cf u = spec[((size_t)b * C + c) * nbin + k], v = spec[((size_t)(b + B) * C + c) * nbin + k];
        const cf xy = cmul(u, make_float2(v.x, -v.y));
        pxy.x += xy.x; pxy.y += xy.y;
        pxx += u.x * u.x + u.y * u.y;
        pyy += v.x * v.x + v.y * v.y;
      }
      pxy.x /= C; pxy.y /= C; pxx /= C; pyy /= C;
      coh += (pxy.x * pxy.x + pxy.y * pxy.y) / (pxx * pyy + 1e-8f);
    }
  }
  const float Coh = block_sum(coh, red) / (float)(T / 2 + 1);
  if (threadIdx.x == 0) {
    float* o = feats + (size_t)b * ld + bi * 7;
    o[0] = sqrtf(Scs * Scs + Ssn * Ssn) / n;
    o[1] = fabsf(Ssg / n);
    o[2] = fabsf(Ssw / (Sw + 1e-8f));
    o[3] = Coh;
    // power correlation over the flattened (C*T) axis, unbiased std (+1e-8), mean of z1*z2
    const float m1 = Q1 / n, m2 = Q2 / n;
    const float sd1 = sqrtf(fmaxf((Q11 - n * m1 * m1) / (n - 1.f), 0.f)) + 1e-8f;
    const float sd2 = sqrtf(fmaxf((Q22 - n * m2 * m2) / (n - 1.f), 0.f)) + 1e-8f;
    o[4] = ((Q12 - n * m1 * m2) / n) / (sd1 * sd2);
    o[5] = fabsf(Sd / n);
    const float tn = (float)T, mu1 = S1 / tn, mu2 = S2 / tn;
    const float t1 = sqrtf(fmaxf((S11 - tn * mu1 * mu1) / (tn - 1.f), 0.f)) + 1e-8f;
    const float t2 = sqrtf(fmaxf((S22 - tn * mu2 * mu2) / (tn - 1.f), 0.f)) + 1e-8f;
    o[6] = ((S12 - tn * mu1 * mu2) / tn) / (t1 * t2);
  }
}

// STFT log-magnitude: x [nsig, T] -> img [nsig, F, nfr] fp32; one 64-thread block = 4 frames x up to 64 bins (n_fft = 128).
// A lane owns one frequency bin and keeps the four frames' sums in registers: per sample index n it reads its twiddle pair once
// and the four frame values as ONE broadcast 16-B read, for 8 multiply-adds (round 2: a wave per frame, 3 LDS reads per 2
// multiply-adds, LDS-bound at 359 us per launch at C = 32).  Same n-ascending fmaf chains per (frame, bin): bit-identical results.
__global__ __launch_bounds__(64) void stft_logmag_kernel(const float* __restrict__ x, const float* __restrict__ window,
                                                         float* __restrict__ img, int T, int n_fft, int hop, int F,
                                                         int nfr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* fr = (float*)smem;          // [n_fft][4] windowed frames, frame-interleaved
  f32x2* tw = (f32x2*)(fr + 4 * n_fft);   // [n_fft] (cos, -sin): one 8-B gather per sample instead of two 4-B ones and a negation
  const int sig = blockIdx.y, f0 = blockIdx.x * 4;
  for (int i = threadIdx.x; i < n_fft; i += 64) {
    float sn, cs;
    sincospif(2.0f * (float)i / (float)n_fft, &sn, &cs);
    tw[i] = (f32x2){cs, -sn};
  }
  for (int i = threadIdx.x; i < 4 * n_fft; i += 64) {
    const int fl = i / n_fft, n = i % n_fft;
    const int f = f0 + fl;
    float v = 0.f;
    if (f < nfr) {
      int t = f * hop + n - n_fft / 2;       // center=True, reflect padding
      if (t < 0) t = -t;
      if (t >= T) t = 2 * (T - 1) - t;
      v = x[(size_t)sig * T + t] * window[n];
    }
    fr[n * 4 + fl] = v;
  }
  __syncthreads();
  for (int kk = threadIdx.x; kk < F; kk += 64) {
    // (re, im) of a frame ride in the halves of one packed multiply-add: the per-(frame, bin) fmaf chains are those of the scalar form
    f32x2 ri[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    int idx = 0;
#pragma unroll 4
    for (int n = 0; n < n_fft; ++n) {
      const f32x2 w = tw[idx];
      idx = (idx + kk) & (n_fft - 1);
      const f32x4 v = *(const f32x4*)(fr + 4 * n);
#pragma unroll
      for (int q = 0; q < 4; ++q) ri[q] = __builtin_elementwise_fma((f32x2){v[q], v[q]}, w, ri[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (f0 + q < nfr) img[((size_t)sig * F + kk) * nfr + f0 + q] = logf(sqrtf(ri[q][0] * ri[q][0] + ri[q][1] * ri[q][1]) + 1e-8f);
  }
}

// dst[b, tok, e] = instance-norm over the token axis (biased var, eps 1e-5) * gamma[e] + beta[e]; also xhat (fp32)
// grid (column blocks of 256, B): a thread owns ONE column e of a window pair and requests its nb * nf <= 56 values before the first
// is used (the first form walked four columns per thread through three passes of dependent loads: 161 us at C = 32 for 44 MB).
// The additions run in the order of that form.
template <typename T>
__global__ __launch_bounds__(256) void ibs_inorm_kernel(const float* __restrict__ conn, const int* __restrict__ fidx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        T* __restrict__ out, float* __restrict__ xhat, int B, int nb, int nf, int E,
                                                        int use_norm) {
  constexpr int MAXTOK = MAX_BANDS * 7;
  const int b = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int ntok = nb * nf;
  float v[MAXTOK];
#pragma unroll
  for (int tkn = 0; tkn < MAXTOK; ++tkn) {
    v[tkn] = 0.f;
    if (tkn < ntok) {
      const int band = tkn / nf, f = fidx[tkn % nf];
      v[tkn] = conn[(((size_t)b * nb + band) * 7 + f) * E + e];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int tkn = 0; tkn < MAXTOK; ++tkn)
    if (tkn < ntok) s += v[tkn];
  const float mean = s / ntok;
  float var = 0.f;
#pragma unroll
  for (int tkn = 0; tkn < MAXTOK; ++tkn)
    if (tkn < ntok) {
      const float d = v[tkn] - mean;
      var += d * d;
    }
  const float rstd = rsqrtf(var / ntok + 1e-5f);
  const float ga = use_norm ? gamma[e] : 1.f, be = use_norm ? beta[e] : 0.f;
#pragma unroll
  for (int tkn = 0; tkn < MAXTOK; ++tkn)
    if (tkn < ntok) {
      float o = v[tkn];
      if (use_norm) {
        const float xh = (v[tkn] - mean) * rstd;
        xhat[((size_t)b * ntok + tkn) * E + e] = xh;
        o = xh * ga + be;
      }
      Elem<T>::st(out + ((size_t)b * ntok + tkn) * E + e, o);
    }
}

// GELU(erf) forward with dropout / backward:  h = drop(gelu(u));  du = dh * mask * gelu'(u)
template <typename T>
__global__ void gelu_fwd_kernel(const T* __restrict__ u, T* __restrict__ h, long long n, DropCfg dc, const eg_step_state* st) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = Elem<T>::ld(u + i);
  float g = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  if (dc.thresh) g = eg_dropout(g, dc, st->seed_lo, st->seed_hi, (uint32_t)i);
  Elem<T>::st(h + i, g);
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ u, const T* __restrict__ dh, T* __restrict__ du, long long n, DropCfg dc,
                                const eg_step_state* st) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = Elem<T>::ld(u + i);
  float g = Elem<T>::ld(dh + i);
  if (dc.thresh) g = eg_dropout(g, dc, st->seed_lo, st->seed_hi, (uint32_t)i);
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
  Elem<T>::st(du + i, g * (cdf + v * pdf));
}

// InstanceNorm affine gradients: dgamma[e] = sum_m dy[m,e] * xhat[m,e], dbeta[e] = sum_m dy[m,e].
// grid (column blocks of 64, row splits): block (cb, sp) sums rows sp, sp + nsplit, ... of its 64 columns and writes
// partial[sp][0][col] (gamma) / partial[sp][1][col] (beta); the caller sums the splits in order (eg_reduce_partials).
template <typename T>
__global__ __launch_bounds__(256) void affine_grad_kernel(const T* __restrict__ dy, const float* __restrict__ xhat,
                                                          float* __restrict__ partial, int M, int E, int nsplit) {
  __shared__ float rg[4][64], rb[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, sp = blockIdx.y;
  float g = 0.f, b = 0.f;
  if (col < E)
    for (int m = sp * 4 + rl; m < M; m += 4 * nsplit) {
      const float d = Elem<T>::ld(dy + (size_t)m * E + col);
      g = fmaf(d, xhat[(size_t)m * E + col], g);
      b += d;
    }
  rg[rl][threadIdx.x & 63] = g;
  rb[rl][threadIdx.x & 63] = b;
  __syncthreads();
  if (rl == 0 && col < E) {
    const int c = threadIdx.x & 63;
    partial[((size_t)sp * 2 + 0) * E + col] = rg[0][c] + rg[1][c] + rg[2][c] + rg[3][c];
    partial[((size_t)sp * 2 + 1) * E + col] = rb[0][c] + rb[1][c] + rb[2][c] + rb[3][c];
  }
}

}  // namespace

static int fill_bands(BandTable& bt, const float* lo, const float* hi, int nbands) {
  if (nbands < 1 || nbands > MAX_BANDS) return 1;
  bt.nbands = nbands;
  for (int i = 0; i < nbands; ++i) { bt.lo[i] = lo[i]; bt.hi[i] = hi[i]; }
  return 0;
}
static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int eg_ibs_analytic(const float* x, float* xb, float* phase, float* stats, float* spec, int nsig, int T,
                               float fs, int nbin, const float* band_lo, const float* band_hi, int nbands, void* stream) {
  EG_CHECK(x && xb && phase && stats && spec && band_lo && band_hi, "eg_ibs_analytic: null pointer");
  EG_CHECK(nsig > 0 && is_pow2(T) && T >= 64 && T <= 2048, "eg_ibs_analytic: T=%d must be a power of two in [64, 2048]", T);
  EG_CHECK(nbin > 0 && nbin <= T / 2 + 1, "eg_ibs_analytic: nbin=%d", nbin);
  BandTable bt;
  EG_CHECK(fill_bands(bt, band_lo, band_hi, nbands) == 0, "eg_ibs_analytic: nbands=%d", nbands);
  const int lds = (3 * T + T / 2) * (int)sizeof(float2);
  hipLaunchKernelGGL(ibs_analytic_kernel, dim3(nsig), dim3(256), lds, (hipStream_t)stream, x, xb, phase, stats, (cf*)spec,
                     nsig, T, fs, nbin, bt);
  EG_LAUNCH_CHECK("ibs_analytic");
  return 0;
}

extern "C" int eg_ibs_pairs(const float* xb, const float* phase, const float* stats, const float* spec, float* conn,
                            int B, int C, int T, float fs, int nbin, const float* band_lo, const float* band_hi, int nbands,
                            void* stream) {
  EG_CHECK(xb && phase && stats && spec && conn, "eg_ibs_pairs: null pointer");
  EG_CHECK(B > 0 && C > 0 && T > 0, "eg_ibs_pairs: bad shape");
  EG_CHECK(is_pow2(T) && T >= 64 && T <= 2048, "eg_ibs_pairs: T=%d must be a power of two in [64, 2048]", T);
  EG_CHECK((int64_t)2 * B * C * T < ((int64_t)1 << 31), "eg_ibs_pairs: 2 B C T = %lld elements per band exceed 32-bit offsets", (long long)2 * B * C * T);
  BandTable bt;
  EG_CHECK(fill_bands(bt, band_lo, band_hi, nbands) == 0, "eg_ibs_pairs: nbands=%d", nbands);
  const int Tc = T < IBS_TC ? T : IBS_TC;
  const int lds_time = 16 * Tc * 16;             // 16 channels x Tc quads (a, phase, cos, sin)
  const int lds_tail = 16 * nbin * 8 + (4 * IBS_RED_FLOATS + 4 * 128) * 4;   // spectra + transposed sums + totals
  const int lds = lds_time > lds_tail ? lds_time : lds_tail;
  EG_CHECK(nbin > 0 && lds <= 160 * 1024, "eg_ibs_pairs: nbin=%d needs %d bytes of LDS", nbin, lds);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)ibs_pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int tiles = ((C + 7) / 8) * ((C + 7) / 8);
  hipLaunchKernelGGL(ibs_pairs_kernel, dim3(B * nbands * tiles), dim3(256), lds, (hipStream_t)stream, xb, phase, stats,
                     (const cf*)spec, conn, B, C, T, fs, nbin, bt);
  EG_LAUNCH_CHECK("ibs_pairs");
  return 0;
}

extern "C" int eg_ibs_scalar(const float* xb, const float* phase, const float* spec, float* feats, int B, int C, int T,
                             float fs, int nbin, const float* band_lo, const float* band_hi, int nbands, int band0,
                             int nout_bands, int ld, void* stream) {
  EG_CHECK(xb && phase && spec && feats, "eg_ibs_scalar: null pointer");
  EG_CHECK(B > 0 && C > 0 && T > 0 && band0 >= 0 && nout_bands > 0 && band0 + nout_bands <= nbands, "eg_ibs_scalar: bad shape");
  EG_CHECK(ld >= 7 * nout_bands, "eg_ibs_scalar: ld=%d too small", ld);
  BandTable bt;
  EG_CHECK(fill_bands(bt, band_lo, band_hi, nbands) == 0, "eg_ibs_scalar: nbands=%d", nbands);
  hipLaunchKernelGGL(ibs_scalar_kernel, dim3(B, nout_bands), dim3(256), 2 * T * 4, (hipStream_t)stream, xb, phase,
                     (const cf*)spec, feats, B, C, T, fs, nbin, bt, band0, ld);
  EG_LAUNCH_CHECK("ibs_scalar");
  return 0;
}

extern "C" int eg_stft_logmag(const float* x, const float* window, float* img, int nsig, int T, int n_fft, int hop, int F,
                              void* stream) {
  EG_CHECK(x && window && img, "eg_stft_logmag: null pointer");
  EG_CHECK(nsig > 0 && is_pow2(n_fft) && n_fft <= 1024 && hop > 0 && F > 0 && F <= n_fft / 2 + 1 && T > n_fft / 2,
           "eg_stft_logmag: bad shape T=%d n_fft=%d hop=%d F=%d", T, n_fft, hop, F);
  const int nfr = 1 + T / hop;
  hipLaunchKernelGGL(stft_logmag_kernel, dim3((nfr + 3) / 4, nsig), dim3(64), 6 * n_fft * 4, (hipStream_t)stream, x, window,
                     img, T, n_fft, hop, F, nfr);
  EG_LAUNCH_CHECK("stft_logmag");
  return 0;
}

extern "C" int eg_ibs_inorm(const float* conn, const int* fidx, const float* gamma, const float* beta, void* out,
                            float* xhat, int B, int nbands, int nfeat, int E, int use_norm, int dtype, void* stream) {
  EG_CHECK(conn && fidx && out && B > 0 && B < 65536 && nbands > 0 && nbands <= MAX_BANDS && nfeat > 0 && nfeat <= 7 && E > 0,
           "eg_ibs_inorm: bad arguments");
  EG_CHECK(!use_norm || (gamma && beta && xhat), "eg_ibs_inorm: instance norm needs gamma, beta and xhat");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(ibs_inorm_kernel<bf16_t>, dim3((E + 255) / 256, B), dim3(256), 0, s, conn, fidx, gamma, beta, (bf16_t*)out, xhat, B,
                       nbands, nfeat, E, use_norm);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(ibs_inorm_kernel<f16_t>, dim3((E + 255) / 256, B), dim3(256), 0, s, conn, fidx, gamma, beta, (f16_t*)out, xhat, B,
                       nbands, nfeat, E, use_norm);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(ibs_inorm_kernel<float>, dim3((E + 255) / 256, B), dim3(256), 0, s, conn, fidx, gamma, beta, (float*)out, xhat, B, nbands,
                       nfeat, E, use_norm);
  else
    return eg_fail("eg_ibs_inorm: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("ibs_inorm");
  return 0;
}

extern "C" int eg_gelu_fwd(const void* u, void* h, int64_t n, int dtype, float drop_p, uint32_t drop_site,
                           const eg_step_state* state, void* stream) {
  EG_CHECK(u && h && n > 0 && n < (1ll << 32), "eg_gelu_fwd: bad arguments");
  EG_CHECK(drop_p == 0.f || state, "eg_gelu_fwd: dropout needs a step state");
  DropCfg dc = make_drop(drop_p, drop_site);
  dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)u, (bf16_t*)h, (long long)n, dc, state);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(gelu_fwd_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)u, (f16_t*)h, (long long)n, dc, state);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(gelu_fwd_kernel<float>, grid, dim3(256), 0, s, (const float*)u, (float*)h, (long long)n, dc, state);
  else
    return eg_fail("eg_gelu_fwd: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("gelu_fwd");
  return 0;
}

extern "C" int eg_gelu_bwd(const void* u, const void* dh, void* du, int64_t n, int dtype, float drop_p, uint32_t drop_site,
                           const eg_step_state* state, void* stream) {
  EG_CHECK(u && dh && du && n > 0 && n < (1ll << 32), "eg_gelu_bwd: bad arguments");
  EG_CHECK(drop_p == 0.f || state, "eg_gelu_bwd: dropout needs a step state");
  DropCfg dc = make_drop(drop_p, drop_site);
  dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)u, (const bf16_t*)dh, (bf16_t*)du,
                       (long long)n, dc, state);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(gelu_bwd_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)u, (const f16_t*)dh, (f16_t*)du,
                       (long long)n, dc, state);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(gelu_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)u, (const float*)dh, (float*)du,
                       (long long)n, dc, state);
  else
    return eg_fail("eg_gelu_bwd: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("gelu_bwd");
  return 0;
}

extern "C" int eg_affine_grad(const void* dy, const float* xhat, float* partial, int nsplit, int M, int E, int dtype,
                              void* stream) {
  EG_CHECK(dy && xhat && partial && M > 0 && E > 0 && nsplit > 0 && nsplit <= 65535, "eg_affine_grad: bad arguments");
  dim3 grid((E + 63) / 64, nsplit);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EG_BF16)
    hipLaunchKernelGGL(affine_grad_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dy, xhat, partial, M, E, nsplit);
  else if (dtype == EG_F16)
    hipLaunchKernelGGL(affine_grad_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)dy, xhat, partial, M, E, nsplit);
  else if (dtype == EG_F32)
    hipLaunchKernelGGL(affine_grad_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, xhat, partial, M, E, nsplit);
  else
    return eg_fail("eg_affine_grad: bad dtype %d", dtype);
  EG_LAUNCH_CHECK("affine_grad");
  return 0;
}
