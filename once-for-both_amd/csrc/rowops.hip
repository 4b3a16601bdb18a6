// HBM-bound row kernels: LayerNorm forward/backward (wave per token row, wavefront reductions),
// column sums (bias gradients, partial reductions) and the small weight-sized gate-folding kernels.
#include "hformat.h"

namespace {

constexpr int LN_MAXE = 16;   // elements per lane -> D <= 1024

// element index owned by (lane, slot j, e) for vector width V
template <int V> __device__ __forceinline__ int ln_col(int lane, int j, int e) { return (j * 64 + lane) * V + e; }

template <int V>
__device__ __forceinline__ void ln_load(float (&v)[LN_MAXE], const float* __restrict__ p, int D, int lane) {
#pragma unroll
  for (int j = 0; j < LN_MAXE / V; ++j) {
    const int c = ln_col<V>(lane, j, 0);
    if (V == 2) {
      float2 t = make_float2(0.f, 0.f);
      if (c < D) t = *reinterpret_cast<const float2*>(p + c);
      v[2 * j] = t.x; v[2 * j + 1] = t.y;
    } else {
      v[j] = (c < D) ? p[c] : 0.f;
    }
  }
}
template <int V>
__device__ __forceinline__ void ln_store(const float (&v)[LN_MAXE], float* __restrict__ p, int D, int lane) {
#pragma unroll
  for (int j = 0; j < LN_MAXE / V; ++j) {
    const int c = ln_col<V>(lane, j, 0);
    if (c < D) {
      if (V == 2) *reinterpret_cast<float2*>(p + c) = make_float2(v[2 * j], v[2 * j + 1]);
      else p[c] = v[j];
    }
  }
}

// Exponent of LayerNorm's H-format output WITHOUT a pass over it: |xhat_i| <= sqrt(D) for every row, so
//   |y_i| <= sqrt(D) |gamma_i| + |beta_i|   and   |y|_2 <= sqrt(D) max|gamma| + |beta|_2 .
// Every wave derives the same numbers from the gamma / beta values its lanes hold (elements beyond D are zero).
template <int NE>
__device__ __forceinline__ void ln_out_bound(const float (&g)[NE], const float (&b)[NE], int D, float& binf, float& rn2sq) {
  const float sd = sqrtf((float)D);
  float m = 0.f, gm = 0.f, bs = 0.f;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    m = fmaxf(m, sd * fabsf(g[i]) + fabsf(b[i]));
    gm = fmaxf(gm, fabsf(g[i]));
    bs += b[i] * b[i];
  }
  binf = ofb_wave_max_pos(m) * 1.0001f;
  const float r = sd * ofb_wave_max_pos(gm) + sqrtf(ofb_wave_sum(bs));
  rn2sq = r * r * 1.0002f;
}

// PF: the normalised rows also leave as H-format planes yP (operand form of the following GEMM, csrc/gemm_h.hip): the four waves of
// a block hold the four rows of one row group; they meet in LDS and every thread writes whole 8-byte plane slots of its columns.
template <int V, bool PF>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y, char* __restrict__ yP,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int D,
                                                     float eps) {
  __shared__ float tile[PF ? 4 * 64 * LN_MAXE : 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, row = blockIdx.x * 4 + w;
  if (!PF && row >= rows) return;
  float v[LN_MAXE];
  float g[LN_MAXE], b[LN_MAXE];
  ln_load<V>(g, gamma, D, lane);
  ln_load<V>(b, beta, D, lane);
  float hs = 1.f;
  if (PF) {
    float binf, rn2;
    ln_out_bound<LN_MAXE>(g, b, D, binf, rn2);
    const int e = ofb_h_exp(binf);
    hs = ofb_h_pow2(e);
    if (blockIdx.x == 0 && threadIdx.x == 0) { ofb_hhdr* h = reinterpret_cast<ofb_hhdr*>(yP); h->e = e; h->amax = binf; h->rn2sq = rn2; h->cn2sq = 0.f; }
  }
  if (row < rows) {
    ln_load<V>(v, x + (size_t)row * D, D, lane);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) s += v[i];
    const float mu = ofb_wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) {
      const int c = ln_col<V>(lane, i / V, i % V);
      const float d = (c < D) ? v[i] - mu : 0.f;
      q += d * d;
    }
    const float rs = 1.0f / sqrtf(ofb_wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) v[i] = (v[i] - mu) * rs * g[i] + b[i];
    if (y) ln_store<V>(v, y + (size_t)row * D, D, lane);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  } else {
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) v[i] = 0.f;                      // padding rows of the last row group: zero planes
  }
  if (PF) {
    const int Dp = (D + 15) & ~15;
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) {
      const int c = ln_col<V>(lane, i / V, i % V);
      if (c < Dp) tile[w * (64 * LN_MAXE) + c] = (c < D) ? v[i] : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Dp; c += 256)
      ofb_store_h4(yP + OFB_HHDR, Dp >> 4, blockIdx.x, c, tile[c] * hs, tile[64 * LN_MAXE + c] * hs, tile[2 * 64 * LN_MAXE + c] * hs,
                   tile[3 * 64 * LN_MAXE + c] * hs);
  }
}

// dx = rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat)) (+ dres); per-block partial dgamma/dbeta
// PF: dx * rowscale[row / rs_div] also leaves as H-format planes dxP (the gradient w.r.t. the previous branch's output is the dY
// operand of that branch's gradient GEMMs, DropPath factor applied), and part gets a third [D] section per block: the column sums
// of those scaled rows (that branch's last bias gradient).  The planes' exponent comes from dxP's header.amax, which
// ln_bwd_stat_kernel left there just before.
// The bound pass (ln_bwd_stat_kernel) leaves one (amax, rn2sq) pair per block in `stat` - no atomics, no memset node: thousands of
// blocks folding their maxima into two words serialise in one L2 channel (that was 30 of the pass's 43 us) - and EVERY block of the
// main kernel reduces the <= 1024 pairs itself (8 KB out of the L2); block 0 publishes the header.
// nb < 0: `stat` holds -nb single values handed over by the GEMM that produced dy (gemm_h.hip, E_RN): per output tile the maximum
// over its rows of rstd |gamma (.) dy, the tile's columns|_2.  A row's norm over ALL columns is at most sqrt(column tiles) times its
// largest tile part: rn_fac carries that factor, max |rowscale| (scanned here: one value per image) the DropPath factor - the bound
// pass over dy is not run at all.  (dres is not covered: the caller uses this form only without a separate residual gradient.)
__device__ __forceinline__ float ln_stat_gather(const float* __restrict__ stat, int nb, float* red8, ofb_hhdr* hdr,
                                                const float* __restrict__ rowscale, int n_rs, float rn_fac) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float am = 0.f, rn = 0.f;
  if (nb >= 0) {
    for (int i = threadIdx.x; i < nb; i += 256) {
      const float2 v = reinterpret_cast<const float2*>(stat)[i];
      am = fmaxf(am, v.x); rn = fmaxf(rn, v.y);
    }
  } else {
    for (int i = threadIdx.x; i < -nb; i += 256) am = fmaxf(am, stat[i]);
    if (rowscale) for (int i = threadIdx.x; i < n_rs; i += 256) rn = fmaxf(rn, fabsf(rowscale[i]));     // (rn: scratch for max |rowscale|)
  }
  am = ofb_wave_max_pos(am); rn = ofb_wave_max_pos(rn);
  if (lane == 0) { red8[w] = am; red8[4 + w] = rn; }
  __syncthreads();
  am = fmaxf(fmaxf(red8[0], red8[1]), fmaxf(red8[2], red8[3]));
  rn = fmaxf(fmaxf(red8[4], red8[5]), fmaxf(red8[6], red8[7]));
  __syncthreads();                                          // red8 is the kernels' reduction scratch again from here on
  if (nb < 0) {
    am = 1.0004f * rn_fac * (rowscale ? rn : 1.f) * am;
    rn = 1.0004f * am * am;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { hdr->e = ofb_h_exp(am); hdr->amax = am; hdr->rn2sq = rn; hdr->cn2sq = 0.f; }
  return am;
}

template <int V, bool PF>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     float* __restrict__ dx, float* __restrict__ part, int rows, int D,
                                                     char* __restrict__ dxP, const float* __restrict__ rowscale, int rs_div,
                                                     const float* __restrict__ stat, int stat_nb, float rn_fac) {
  __shared__ float red[4 * 2 * 1024];
  constexpr int NS = PF ? 3 : 2;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float g[LN_MAXE], ag[LN_MAXE], ab[LN_MAXE];
  float cacc[PF ? 4 : 1];                                  // PF: column sums of the scaled rows, columns threadIdx.x + 256 k
  ln_load<V>(g, gamma, D, lane);
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) ag[i] = ab[i] = 0.f;
#pragma unroll
  for (int k = 0; k < (PF ? 4 : 1); ++k) cacc[k] = 0.f;
  const int Dp = (D + 15) & ~15;
  float hs = 1.f;
  if (PF) hs = ofb_h_pow2(ofb_h_exp(ln_stat_gather(stat, stat_nb, red, reinterpret_cast<ofb_hhdr*>(dxP), rowscale, (rows + rs_div - 1) / rs_div, rn_fac)));
  for (int row0 = blockIdx.x * 4; row0 < rows; row0 += gridDim.x * 4) {
    const int row = row0 + w;
    float v[LN_MAXE];
    if (row < rows) {
      float d[LN_MAXE];
      ln_load<V>(v, x + (size_t)row * D, D, lane);
      ln_load<V>(d, dy + (size_t)row * D, D, lane);
      const float mu = mean[row], rs = rstd[row];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < LN_MAXE; ++i) {
        const int c = ln_col<V>(lane, i / V, i % V);
        const float xh = (c < D) ? (v[i] - mu) * rs : 0.f;
        const float dg = d[i] * g[i];
        ag[i] += d[i] * xh;
        ab[i] += d[i];
        s1 += dg;
        s2 += dg * xh;
        v[i] = xh;
        d[i] = dg;
      }
      const float c1 = ofb_wave_sum(s1) / (float)D, c2 = ofb_wave_sum(s2) / (float)D;
#pragma unroll
      for (int i = 0; i < LN_MAXE; ++i) v[i] = rs * (d[i] - c1 - v[i] * c2);
      if (dres) {
        float r[LN_MAXE];
        ln_load<V>(r, dres + (size_t)row * D, D, lane);
#pragma unroll
        for (int i = 0; i < LN_MAXE; ++i) v[i] += r[i];
      }
      ln_store<V>(v, dx + (size_t)row * D, D, lane);
    }
    if (PF) {
      // the block's four rows are one row group: scaled copies meet in LDS (the first 4 x 1024 floats of red), one thread per column
      const float sc = (row < rows) ? (rowscale ? rowscale[rs_div == 1 ? row : row / rs_div] : 1.f) : 0.f;
      __syncthreads();                                     // the previous trip's readers are done
#pragma unroll
      for (int i = 0; i < LN_MAXE; ++i) {
        const int c = ln_col<V>(lane, i / V, i % V);
        if (c < Dp) red[w * 1024 + c] = (row < rows && c < D) ? v[i] * sc : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < Dp) {
          const float a0 = red[c], a1 = red[1024 + c], a2 = red[2048 + c], a3 = red[3072 + c];
          cacc[k] += (a0 + a1) + (a2 + a3);
          ofb_store_h4(dxP + OFB_HHDR, Dp >> 4, row0 >> 2, c, a0 * hs, a1 * hs, a2 * hs, a3 * hs);
        }
      }
    }
  }
  if (PF) __syncthreads();
  // cross-wave reduction of the per-lane column partials
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) {
    const int c = ln_col<V>(lane, i / V, i % V);
    if (c < D) { red[w * 2048 + c] = ag[i]; red[w * 2048 + 1024 + c] = ab[i]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { sg += red[k * 2048 + c]; sb += red[k * 2048 + 1024 + c]; }
    part[(size_t)blockIdx.x * NS * D + c] = sg;
    part[(size_t)blockIdx.x * NS * D + D + c] = sb;
  }
  if (PF) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = threadIdx.x + 256 * k;
      if (c < D) part[(size_t)blockIdx.x * NS * D + 2 * D + c] = cacc[k];
    }
  }
}

// ---- H-format variants for even D: one WAVE per row group (4 rows), lane owns the column pairs 2 (64 j + lane) + {0, 1}, j < NJ: the
// four rows of a column sit in one lane's registers and leave as whole plane slots (two adjacent slots = one 16-byte store per
// plane), with no LDS and no block barrier.  P: the planes' base (behind the header); hs = 2^e ----
template <int NJ>
__device__ __forceinline__ void ln_p_store(char* __restrict__ P, int ncb, int rg, int lane, int Dp, const float (&v)[4][2 * NJ], float hs) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (j * 64 + lane) * 2;
    if (c < Dp) {
      unsigned h[4], l[4];                                  // [column e][row pair]
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        ofb_hsplit_pair(v[0][2 * j + e] * hs, v[1][2 * j + e] * hs, h[2 * e], l[2 * e]);
        ofb_hsplit_pair(v[2][2 * j + e] * hs, v[3][2 * j + e] * hs, h[2 * e + 1], l[2 * e + 1]);
      }
      char* slot = P + ((size_t)rg * ncb + (c >> 4)) * OFB_HGRAN + (c & 15) * 8;
      *reinterpret_cast<uint4*>(slot) = make_uint4(h[0], h[1], h[2], h[3]);
      *reinterpret_cast<uint4*>(slot + 128) = make_uint4(l[0], l[1], l[2], l[3]);
    }
  }
}
// (every lane reads an in-bounds address - lanes past the row's end re-read its first pair and drop it -: no branch around the load.  With
//  the loads behind `if (c < D)` hipcc put a vmcnt(0) at every join and the rows of a group arrived in four round trips instead of one)
template <int NJ>
__device__ __forceinline__ void ln_p_load(float (&v)[2 * NJ], const float* __restrict__ p, int D, int lane) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (j * 64 + lane) * 2;
    const bool in = c < D;
    const float2 t = *reinterpret_cast<const float2*>(p + (in ? c : 0));
    v[2 * j] = in ? t.x : 0.f; v[2 * j + 1] = in ? t.y : 0.f;
  }
}
// the same for rows that are not read again in this pass (saved activations and incoming gradients in the backward); the widest rows
// keep the branch (the redirected offsets cost the registers that decide their occupancy)
template <int NJ>
__device__ __forceinline__ void ln_p_load_nt(float (&v)[2 * NJ], const float* __restrict__ p, int D, int lane) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (j * 64 + lane) * 2;
    const bool in = c < D;
    if constexpr (NJ <= 3) {
      const ofb_f32x2 q = OFB_NT_LOAD(reinterpret_cast<const ofb_f32x2*>(p + (in ? c : 0)));
      v[2 * j] = in ? q[0] : 0.f; v[2 * j + 1] = in ? q[1] : 0.f;
    } else {
      ofb_f32x2 q = {0.f, 0.f};
      if (in) q = OFB_NT_LOAD(reinterpret_cast<const ofb_f32x2*>(p + c));
      v[2 * j] = q[0]; v[2 * j + 1] = q[1];
    }
  }
}

template <int NJ>
__global__ __launch_bounds__(256) void ln_fwd_p_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y, char* __restrict__ yP,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int rows, int D, float eps) {
  constexpr int NE = 2 * NJ;
  // (the wave index as a scalar: row bases then live in SGPRs and the row loads are base + one per-lane offset + immediates)
  const int lane = threadIdx.x & 63, rg = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (4 * rg >= rows) return;
  float g[NE], b[NE], v[4][NE];
  // the four rows are requested FIRST, gamma / beta and the bound of the output behind them (one memory round trip for the lot: the
  // kernel is one pass of short waves - whatever a wave waits for before its rows are under way is added to the whole launch)
#pragma unroll
  for (int r = 0; r < 4; ++r) ln_p_load<NJ>(v[r], x + (size_t)min(4 * rg + r, rows - 1) * D, D, lane);
  __builtin_amdgcn_sched_barrier(0);                       // (left alone hipcc sinks the row loads below the bound's wait for gamma / beta)
  ln_p_load<NJ>(g, gamma, D, lane);
  ln_p_load<NJ>(b, beta, D, lane);
  float binf, rn2;
  ln_out_bound<NE>(g, b, D, binf, rn2);
  const int he = ofb_h_exp(binf);
  if (blockIdx.x == 0 && threadIdx.x == 0) { ofb_hhdr* h = reinterpret_cast<ofb_hhdr*>(yP); h->e = he; h->amax = binf; h->rn2sq = rn2; h->cn2sq = 0.f; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (4 * rg + r >= rows) {                              // padding rows of the last row group: zero planes
#pragma unroll
      for (int i = 0; i < NE; ++i) v[r][i] = 0.f;
    }
  }
  // the four rows' statistics side by side: four independent reduction chains per step instead of eight chains one after the other
  // (per row the same operations in the same order as before)
  float mu[4], rs[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) s += v[r][i];
    mu[r] = s;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) mu[r] = ofb_wave_sum(mu[r]) / (float)D;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int c = ((i >> 1) * 64 + lane) * 2 + (i & 1);
      const float d = (c < D) ? v[r][i] - mu[r] : 0.f;
      q += d * d;
    }
    rs[r] = q;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) rs[r] = 1.0f / sqrtf(ofb_wave_sum(rs[r]) / (float)D + eps);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * rg + r;
    if (row >= rows) continue;                             // padding rows of the last row group stay zero
#pragma unroll
    for (int i = 0; i < NE; ++i) v[r][i] = (v[r][i] - mu[r]) * rs[r] * g[i] + b[i];     // columns >= D: g = b = 0 -> 0
    if (y) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = (j * 64 + lane) * 2;
        if (c < D) *reinterpret_cast<float2*>(y + (size_t)row * D + c) = make_float2(v[r][2 * j], v[r][2 * j + 1]);
      }
    }
    if (lane == 0) { mean[row] = mu[r]; rstd[row] = rs[r]; }
  }
  const int Dp = (D + 15) & ~15;
  ln_p_store<NJ>(yP + OFB_HHDR, Dp >> 4, rg, lane, Dp, v, ofb_h_pow2(he));
}

template <int NJ>
__global__ __launch_bounds__(256) void ln_bwd_p_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ dres,
                                                       float* __restrict__ dx, float* __restrict__ part, int rows, int D,
                                                       char* __restrict__ dxP, const float* __restrict__ rowscale, int rs_div,
                                                       const float* __restrict__ stat, int stat_nb, float rn_fac) {
  constexpr int NE = 2 * NJ;
  __shared__ float red[4 * 3 * 128 * NJ];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: row bases in SGPRs)
  const int Dp = (D + 15) & ~15;
  float g[NE], ag[NE], ab[NE], ac[NE];
  ln_p_load<NJ>(g, gamma, D, lane);
  const float hs = ofb_h_pow2(ofb_h_exp(ln_stat_gather(stat, stat_nb, red, reinterpret_cast<ofb_hhdr*>(dxP), rowscale, (rows + rs_div - 1) / rs_div, rn_fac)));
#pragma unroll
  for (int i = 0; i < NE; ++i) ag[i] = ab[i] = ac[i] = 0.f;
  // A row group's x and dy rows, statistics and row factors are requested together, before anything is used (12 KB in flight per wave
  // at D = 384; row by row it was 3 KB, with the row factor's own round trip behind every row) - round 6, scripts/lab/ln_bench.py.
  // Wider rows keep fewer rows in flight (registers).
  constexpr int RH = NJ <= 3 ? 4 : 1;
  for (int rg = blockIdx.x * 4 + w; 4 * rg < rows; rg += gridDim.x * 4) {
    float xv[4][NE], dv[4][NE], mu4[4], rs4[4], sc4[4];
    auto request = [&](int r0) {
#pragma unroll
      for (int q = r0; q < r0 + RH; ++q) {
        const int row = min(4 * rg + q, rows - 1);          // (rows past the end re-read the last one; they are dropped below)
        ln_p_load_nt<NJ>(xv[q], x + (size_t)row * D, D, lane);
        ln_p_load_nt<NJ>(dv[q], dy + (size_t)row * D, D, lane);
        mu4[q] = mean[row]; rs4[q] = rstd[row];
        sc4[q] = rowscale ? rowscale[rs_div == 1 ? row : row / rs_div] : 1.f;
      }
    };
    request(0);
    float o[4][NE];                                        // dx * rowscale of the group's rows
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r > 0 && r % RH == 0) request(r);
      const int row = 4 * rg + r;
      if (row >= rows) {
#pragma unroll
        for (int i = 0; i < NE; ++i) o[r][i] = 0.f;
        continue;
      }
      float (&v)[NE] = xv[r];
      float (&d)[NE] = dv[r];
      float rr[NE];
      if (NJ <= 3 && dres) ln_p_load<NJ>(rr, dres + (size_t)row * D, D, lane);          // (in flight under the row's two reductions)
      const float mu = mu4[r], rs = rs4[r];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int c = ((i >> 1) * 64 + lane) * 2 + (i & 1);
        const float xh = (c < D) ? (v[i] - mu) * rs : 0.f;
        const float dg = d[i] * g[i];
        ag[i] += d[i] * xh;
        ab[i] += d[i];
        s1 += dg;
        s2 += dg * xh;
        v[i] = xh;
        d[i] = dg;
      }
      const float c1 = ofb_wave_sum(s1) / (float)D, c2 = ofb_wave_sum(s2) / (float)D;
#pragma unroll
      for (int i = 0; i < NE; ++i) v[i] = rs * (d[i] - c1 - v[i] * c2);
      if (dres) {
        if (NJ > 3) ln_p_load<NJ>(rr, dres + (size_t)row * D, D, lane);     // (wide rows: registers)
#pragma unroll
        for (int i = 0; i < NE; ++i) v[i] += rr[i];
      }
      const float sc = sc4[r];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = (j * 64 + lane) * 2;
        if (c < D) *reinterpret_cast<float2*>(dx + (size_t)row * D + c) = make_float2(v[2 * j], v[2 * j + 1]);
        o[r][2 * j] = (c < D) ? v[2 * j] * sc : 0.f;
        o[r][2 * j + 1] = (c < D) ? v[2 * j + 1] * sc : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) ac[i] += (o[0][i] + o[1][i]) + (o[2][i] + o[3][i]);
    ln_p_store<NJ>(dxP + OFB_HHDR, Dp >> 4, rg, lane, Dp, o, hs);
  }
  // cross-wave reduction of the per-lane column partials: dgamma | dbeta | column sums of the scaled rows
  constexpr int SEC = 128 * NJ;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int c = ((i >> 1) * 64 + lane) * 2 + (i & 1);
    red[(w * 3 + 0) * SEC + c] = ag[i]; red[(w * 3 + 1) * SEC + c] = ab[i]; red[(w * 3 + 2) * SEC + c] = ac[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
#pragma unroll
    for (int sct = 0; sct < 3; ++sct) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) sum += red[(k * 3 + sct) * SEC + c];
      part[(size_t)blockIdx.x * 3 * D + sct * D + c] = sum;
    }
  }
}

// Bound of LayerNorm backward's scaled output BEFORE it is computed (the exponent of its H-format planes): dx = rstd P(gamma * dy)
// with P an orthogonal projection (it removes the components along 1 and xhat), so per row
//   |dx|_inf <= |dx|_2 <= rstd |gamma * dy|_2   (+ the residual gradient's own max / norm), times the row's DropPath factor.
// header.amax / header.rn2sq (zeroed by a memset node) take the maxima over the rows.  NJ > 0: even D, one wave per 4-row group, the
// four rows' float2 loads in flight together (an HBM-bound pass over dy: 8-10 us for [25216][384]); NJ == 0: any D, a wave per row.
template <int NJ>
__global__ __launch_bounds__(256) void ln_bwd_stat_kernel(const float* __restrict__ dy, const float* __restrict__ gamma,
                                                          const float* __restrict__ rstd, const float* __restrict__ dres,
                                                          const float* __restrict__ rowscale, int rs_div, int rows, int D,
                                                          float* __restrict__ stat) {
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float am = 0.f, rn = 0.f;
  if constexpr (NJ > 0) {
    constexpr int NE = 2 * NJ;
    float g[NE];
    ln_p_load<NJ>(g, gamma, D, lane);
    for (int rg = blockIdx.x * 4 + w; 4 * rg < rows; rg += gridDim.x * 4) {
      float d[4][NE], rr[4][NE];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * rg + r;
        if (row < rows) {
          ln_p_load<NJ>(d[r], dy + (size_t)row * D, D, lane);
          if (dres) ln_p_load<NJ>(rr[r], dres + (size_t)row * D, D, lane);
        } else {
#pragma unroll
          for (int i = 0; i < NE; ++i) d[r][i] = 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * rg + r;
        if (row >= rows) continue;
        float ss = 0.f, rm = 0.f, rs2 = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
          const float t = d[r][i] * g[i];
          ss += t * t;
          if (dres) { rm = fmaxf(rm, fabsf(rr[r][i])); rs2 += rr[r][i] * rr[r][i]; }
        }
        const float n2 = rstd[row] * sqrtf(ofb_wave_sum(ss));
        const float sc = rowscale ? fabsf(rowscale[rs_div == 1 ? row : row / rs_div]) : 1.f;
        if (dres) { rm = ofb_wave_max_pos(rm); rs2 = sqrtf(ofb_wave_sum(rs2)); }
        am = fmaxf(am, sc * (n2 + rm));
        const float r2 = sc * (n2 + rs2);
        rn = fmaxf(rn, r2 * r2);
      }
    }
  } else {
    for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
      const float* d = dy + (size_t)row * D;
      float ss = 0.f, rm = 0.f, rs2 = 0.f;
      for (int c = lane; c < D; c += 64) {
        const float t = d[c] * gamma[c];
        ss += t * t;
        if (dres) { const float r = dres[(size_t)row * D + c]; rm = fmaxf(rm, fabsf(r)); rs2 += r * r; }
      }
      const float n2 = rstd[row] * sqrtf(ofb_wave_sum(ss));
      const float sc = rowscale ? fabsf(rowscale[rs_div == 1 ? row : row / rs_div]) : 1.f;
      if (dres) { rm = ofb_wave_max_pos(rm); rs2 = sqrtf(ofb_wave_sum(rs2)); }
      am = fmaxf(am, sc * (n2 + rm));
      const float r2 = sc * (n2 + rs2);
      rn = fmaxf(rn, r2 * r2);
    }
  }
  if (lane == 0) { red[0][w] = am; red[1][w] = rn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    stat[2 * blockIdx.x] = 1.0002f * fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    stat[2 * blockIdx.x + 1] = 1.0004f * fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  }
}

// partial[slab][col] = sum over the slab's rows of x[row][col] * (rowscale ? rowscale[row / rs_div] : 1)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ld, int M, int N,
                                                     const float* __restrict__ rowscale, int rs_div,
                                                     float* __restrict__ out, int rows_per_slab) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
  float s = 0.f;
  if (col < N)
    for (int r = r0 + w; r < r1; r += 4) {
      float v = x[(size_t)r * ld + col];
      if (rowscale) v *= rowscale[r / rs_div];
      s += v;
    }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && col < N) out[(size_t)blockIdx.y * N + col] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// Many column-sum jobs in ONE launch (the LayerNorm parameter gradients of a whole backward pass: 25 x [1024 partial rows][3 D]):
// block (column block, job) of 16 waves; wave w adds the job's rows w, w + 16, ... in a fixed order, then the 16 sums in wave order.
__global__ __launch_bounds__(1024) void colsum_multi_kernel(const ofb_colsum_job* __restrict__ jobs) {
  __shared__ float red[16][64];
  const ofb_colsum_job j = jobs[blockIdx.y];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  if (blockIdx.x * 64 >= j.N) return;
  float s0 = 0.f, s1 = 0.f;                                // two loads in flight per thread; rows w, w + 16, ... added in order
  if (col < j.N) {
    int r = w;
    // (eight loads in flight, added in the same order as the two-at-a-time loop below would: 1024 rows of partials were 32 dependent
    //  round trips per thread)
    for (; r + 112 < j.M; r += 128) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = j.x[(size_t)(r + 32 * u) * j.ld + col]; b[u] = j.x[(size_t)(r + 32 * u + 16) * j.ld + col]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s0 += a[u]; s1 += b[u]; }
    }
    for (; r + 16 < j.M; r += 32) { s0 += j.x[(size_t)r * j.ld + col]; s1 += j.x[(size_t)(r + 16) * j.ld + col]; }
    if (r < j.M) s0 += j.x[(size_t)r * j.ld + col];
  }
  red[w][lane] = s0 + s1;
  __syncthreads();
  if (w == 0 && col < j.N) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][lane];
    j.out[col] = s;
  }
}

__global__ void scale_rows_kernel(const float* __restrict__ W, const float* __restrict__ g, float* __restrict__ out, int N,
                                  int K) {
  // one float4 per thread when rows are 16-B multiples (token matrices), scalar otherwise
  const int64_t total = (int64_t)N * K;
  if ((K & 3) == 0) {
    const int64_t n4 = total >> 2;
    const int k4 = K >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
      f32x4 v = reinterpret_cast<const f32x4*>(W)[i];
      v *= g[i / k4];
      reinterpret_cast<f32x4*>(out)[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
      out[i] = W[i] * g[i / K];
  }
}

// Gate folded into a Linear layer, y = g[n] * (x W^T + b)[n]:  given the raw gradients of the UNGATED product
// (dWraw = dY^T x, dbraw = colsum dY) produce dW = g*dWraw, db = g*dbraw and dg[n] = <dWraw[n], W[n]> + dbraw[n] b[n].
__device__ __forceinline__ void gate_fold_bwd_body(const float* __restrict__ dWraw, const float* __restrict__ W,
                                                   const float* __restrict__ g, const float* __restrict__ dbraw,
                                                   const float* __restrict__ b, float* __restrict__ dW,
                                                   float* __restrict__ db, float* __restrict__ dg, int N, int K,
                                                   int dbraw_rows, int fold) {
  // fold > 1: the gate vector is N / fold values applied to `fold` row groups (q | k | v share one gate, layers.py:507-509): g holds the
  // tiled N values, dg the N / fold sums over the groups (added in group order) - no separate 3-way sum launch
  const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6), Nf = N / fold;
  if (j >= Nf) return;
  if (fold == 3) {
    // the three rows of a gate element side by side: their loads are independent, so the wave keeps three times the memory
    // requests in flight instead of walking the rows one after the other (the kernel is latency bound: 5 MB per call)
    float s3[3] = {0.f, 0.f, 0.f}, dbr3[3] = {0.f, 0.f, 0.f}, g3[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) g3[f] = g[j + f * Nf];
    if (dbraw) {
      for (int r = lane; r < dbraw_rows; r += 64)
#pragma unroll
        for (int f = 0; f < 3; ++f) dbr3[f] += dbraw[(size_t)r * N + j + f * Nf];
#pragma unroll
      for (int f = 0; f < 3; ++f) dbr3[f] = dbraw_rows > 1 ? ofb_wave_sum(dbr3[f]) : __shfl(dbr3[f], 0, 64);
    }
    for (int k = lane; k < K; k += 64) {
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const size_t o = (size_t)(j + f * Nf) * K + k;
        const float d = dWraw[o];
        s3[f] += d * W[o];
        dW[o] = d * g3[f];
      }
    }
    float tot = 0.f;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      const float s = ofb_wave_sum(s3[f]);
      tot += s + (b ? dbr3[f] * b[j + f * Nf] : 0.f);       // group order q, k, v
      if (lane == 0 && db) db[j + f * Nf] = dbr3[f] * g3[f];
    }
    if (lane == 0) dg[j] = tot;
    return;
  }
  float tot = 0.f;
  for (int f = 0; f < fold; ++f) {
    const int n = j + f * Nf;
    const float gn = g[n];
    // dbraw may arrive as dbraw_rows partial rows [rows][N] (per-image / per-tile column sums straight from the producing kernel):
    // they are added here (lane r, r + 64, ... in order, then the fixed wave tree) instead of by a reduction launch of their own
    float dbr = 0.f;
    if (dbraw) {
      for (int r = lane; r < dbraw_rows; r += 64) dbr += dbraw[(size_t)r * N + n];
      dbr = dbraw_rows > 1 ? ofb_wave_sum(dbr) : __shfl(dbr, 0, 64);
    }
    float s = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float d = dWraw[(size_t)n * K + k];
      s += d * W[(size_t)n * K + k];
      dW[(size_t)n * K + k] = d * gn;
    }
    s = ofb_wave_sum(s);
    if (lane == 0) {
      tot += s + (b ? dbr * b[n] : 0.f);
      if (db) db[n] = dbr * gn;
    }
  }
  if (lane == 0) dg[j] = tot;
}

__global__ __launch_bounds__(256) void gate_fold_bwd_kernel(const float* __restrict__ dWraw, const float* __restrict__ W,
                                                            const float* __restrict__ g, const float* __restrict__ dbraw,
                                                            const float* __restrict__ b, float* __restrict__ dW,
                                                            float* __restrict__ db, float* __restrict__ dg, int N, int K,
                                                            int dbraw_rows, int fold) {
  gate_fold_bwd_body(dWraw, W, g, dbraw, b, dW, db, dg, N, K, dbraw_rows, fold);
}
}  // namespace

namespace {
int ln_fwd_launch(const float* x, const float* gamma, const float* beta, float* y, char* yP, float* mean, float* rstd, int rows, int D,
                  float eps, hipStream_t s) {
  const dim3 grid(ofb_cdiv(rows, 4));
  ofb_prof_pre(2, s, (yP ? 12.0 : 8.0) * rows * (double)D);
  if (yP && D % 2 == 0) {
    const dim3 gp(ofb_cdiv(rows, 16));
    const int nj = ofb_cdiv(D, 128);
    if (nj <= 2) hipLaunchKernelGGL(ln_fwd_p_kernel<2>, gp, dim3(256), 0, s, x, gamma, beta, y, yP, mean, rstd, rows, D, eps);
    else if (nj <= 3) hipLaunchKernelGGL(ln_fwd_p_kernel<3>, gp, dim3(256), 0, s, x, gamma, beta, y, yP, mean, rstd, rows, D, eps);
    else if (nj <= 6) hipLaunchKernelGGL(ln_fwd_p_kernel<6>, gp, dim3(256), 0, s, x, gamma, beta, y, yP, mean, rstd, rows, D, eps);
    else hipLaunchKernelGGL(ln_fwd_p_kernel<8>, gp, dim3(256), 0, s, x, gamma, beta, y, yP, mean, rstd, rows, D, eps);
  } else if (yP) {
    hipLaunchKernelGGL((ln_fwd_kernel<1, true>), grid, dim3(256), 0, s, x, gamma, beta, y, yP, mean, rstd, rows, D, eps);
  } else {
    if (D % 2 == 0) hipLaunchKernelGGL((ln_fwd_kernel<2, false>), grid, dim3(256), 0, s, x, gamma, beta, y, yP, mean, rstd, rows, D, eps);
    else hipLaunchKernelGGL((ln_fwd_kernel<1, false>), grid, dim3(256), 0, s, x, gamma, beta, y, yP, mean, rstd, rows, D, eps);
  }
  ofb_prof_post(2, s);
  return ofb_launch_status();
}
}  // namespace

extern "C" int ofb_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                 int32_t rows, int32_t D, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || D <= 0) return OFB_EINVAL;
  if (D > 64 * LN_MAXE) return OFB_ELIMIT;
  return ln_fwd_launch(x, gamma, beta, y, nullptr, mean, rstd, rows, D, eps, (hipStream_t)stream);
}

// y (optional) and the same rows as H-format planes y_h[rows][D] (ofb_hformat_bytes(rows, D) bytes, header included).  Row groups
// 0 .. ceil(rows / 4) - 1 are written whole (padding rows and columns as zeros); the caller zeroes what is left of the last 16-row
// group (rows % 16 in 1..12).
extern "C" int ofb_layernorm_fwd_h(const float* x, const float* gamma, const float* beta, float* y, void* y_p, float* mean,
                                   float* rstd, int32_t rows, int32_t D, float eps, void* stream) {
  if (!x || !gamma || !beta || !y_p || !mean || !rstd || rows <= 0 || D <= 0) return OFB_EINVAL;
  if (D > 64 * LN_MAXE) return OFB_ELIMIT;
  return ln_fwd_launch(x, gamma, beta, y, (char*)y_p, mean, rstd, rows, D, eps, (hipStream_t)stream);
}

extern "C" int32_t ofb_layernorm_bwd_blocks(int32_t rows) { return rows >= 4096 ? 1024 : ofb_cdiv(rows, 4); }

namespace {
int ln_bwd_launch(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* dres, float* dx,
                  float* partials, int rows, int D, char* dxP, const float* rowscale, int rs_div, hipStream_t s,
                  const float* rn = nullptr, int n_rn = 0, float rn_fac = 1.f) {
  const dim3 grid(ofb_layernorm_bwd_blocks(rows));
  if (!rowscale || rs_div <= 0) rs_div = 1;                 // the kernels divide by it unconditionally; without a rowscale any value was legal
  ofb_prof_pre(3, s, (dxP ? 24.0 : 16.0) * rows * (double)D);
  // per-block maxima of the bound pass: the last 8 KB of the plane buffer (its slack past the matrix: never read as values)
  const float* stat = dxP ? reinterpret_cast<float*>(dxP + ofb_hformat_bytes(rows, D) - 8192) : nullptr;
  int nb = 0;
  if (dxP && rn) {                                          // the producing GEMM left the bound's ingredients: no pass over dy
    stat = rn;
    nb = -n_rn;
  } else if (dxP) {
    float* stat = reinterpret_cast<float*>(dxP + ofb_hformat_bytes(rows, D) - 8192);
    const int nj = ofb_cdiv(D, 128);
    if (D % 2 == 0) {
      nb = ofb_cdiv(rows, 16) < 1024 ? ofb_cdiv(rows, 16) : 1024;   // (measured on [25216][384]: 2048 / 1024 / 512 / 256 blocks = 17 / 13 / 17 / 21 us)
      if (nj <= 2) hipLaunchKernelGGL(ln_bwd_stat_kernel<2>, dim3(nb), dim3(256), 0, s, dy, gamma, rstd, dres, rowscale, rs_div, rows, D, stat);
      else if (nj <= 3) hipLaunchKernelGGL(ln_bwd_stat_kernel<3>, dim3(nb), dim3(256), 0, s, dy, gamma, rstd, dres, rowscale, rs_div, rows, D, stat);
      else if (nj <= 6) hipLaunchKernelGGL(ln_bwd_stat_kernel<6>, dim3(nb), dim3(256), 0, s, dy, gamma, rstd, dres, rowscale, rs_div, rows, D, stat);
      else hipLaunchKernelGGL(ln_bwd_stat_kernel<8>, dim3(nb), dim3(256), 0, s, dy, gamma, rstd, dres, rowscale, rs_div, rows, D, stat);
    } else {
      nb = ofb_cdiv(rows, 4) < 1024 ? ofb_cdiv(rows, 4) : 1024;
      hipLaunchKernelGGL(ln_bwd_stat_kernel<0>, dim3(nb), dim3(256), 0, s, dy, gamma, rstd, dres, rowscale, rs_div, rows, D, stat);
    }
  }
  if (dxP && D % 2 == 0) {
    const int nj = ofb_cdiv(D, 128);
    if (nj <= 2) hipLaunchKernelGGL(ln_bwd_p_kernel<2>, grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, partials, rows, D, dxP, rowscale, rs_div, stat, nb, rn_fac);
    else if (nj <= 3) hipLaunchKernelGGL(ln_bwd_p_kernel<3>, grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, partials, rows, D, dxP, rowscale, rs_div, stat, nb, rn_fac);
    else if (nj <= 6) hipLaunchKernelGGL(ln_bwd_p_kernel<6>, grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, partials, rows, D, dxP, rowscale, rs_div, stat, nb, rn_fac);
    else hipLaunchKernelGGL(ln_bwd_p_kernel<8>, grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, partials, rows, D, dxP, rowscale, rs_div, stat, nb, rn_fac);
  } else if (dxP) {
    hipLaunchKernelGGL((ln_bwd_kernel<1, true>), grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, partials, rows, D, dxP, rowscale, rs_div, stat, nb, rn_fac);
  } else {
    if (D % 2 == 0) hipLaunchKernelGGL((ln_bwd_kernel<2, false>), grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, partials, rows, D, dxP, rowscale, rs_div, stat, nb, rn_fac);
    else hipLaunchKernelGGL((ln_bwd_kernel<1, false>), grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, partials, rows, D, dxP, rowscale, rs_div, stat, nb, rn_fac);
  }
  ofb_prof_post(3, s);
  return ofb_launch_status();
}
}  // namespace

extern "C" int ofb_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                 const float* dres, float* dx, float* partials, int32_t rows, int32_t D, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !partials || rows <= 0 || D <= 0) return OFB_EINVAL;
  if (D > 64 * LN_MAXE) return OFB_ELIMIT;
  return ln_bwd_launch(dy, x, gamma, mean, rstd, dres, dx, partials, rows, D, nullptr, nullptr, 1, (hipStream_t)stream);
}

// Same, and dx * rowscale[row / rs_div] (rowscale optional) also as H-format planes dx_h[rows][D]; partials is then
// [ofb_layernorm_bwd_blocks(rows)][3][D]: dgamma | dbeta | column sums of the scaled dx rows.
extern "C" int ofb_layernorm_bwd_h(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                   const float* dres, float* dx, float* partials, void* dx_p, const float* rowscale, int32_t rs_div,
                                   int32_t rows, int32_t D, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !partials || !dx_p || rows <= 0 || D <= 0) return OFB_EINVAL;
  if (rowscale && rs_div <= 0) return OFB_EINVAL;
  if (D > 64 * LN_MAXE) return OFB_ELIMIT;
  return ln_bwd_launch(dy, x, gamma, mean, rstd, dres, dx, partials, rows, D, (char*)dx_p, rowscale, rs_div, (hipStream_t)stream);
}

// The same with the bound of the result taken from rn[n_rn] (ofb_gemm_h's rn_out of the GEMM that produced dy; rn_fac = sqrt of its
// column tiles) instead of a pass over dy.  dy must be the ONLY gradient of the LayerNorm's output (no dres).
extern "C" int ofb_layernorm_bwd_h_rn(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                      float* dx, float* partials, void* dx_p, const float* rowscale, int32_t rs_div, int32_t rows,
                                      int32_t D, const float* rn, int32_t n_rn, float rn_fac, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !partials || !dx_p || !rn || n_rn <= 0 || !(rn_fac >= 1.f) || rows <= 0 || D <= 0)
    return OFB_EINVAL;
  if (rowscale && rs_div <= 0) return OFB_EINVAL;
  if (D > 64 * LN_MAXE) return OFB_ELIMIT;
  return ln_bwd_launch(dy, x, gamma, mean, rstd, nullptr, dx, partials, rows, D, (char*)dx_p, rowscale, rs_div, (hipStream_t)stream, rn,
                       n_rn, rn_fac);
}

extern "C" int32_t ofb_colsum_slabs(int32_t M, int32_t N) {
  if (M <= 512) return 1;               // short inputs (second stages, per-tile / per-block partials): one launch, no scratch
  const int colblocks = ofb_cdiv(N, 64);
  int slabs = ofb_cdiv(2048, colblocks);
  if (slabs > ofb_cdiv(M, 16)) slabs = ofb_cdiv(M, 16);
  return slabs < 1 ? 1 : slabs;
}

// out[N] = column sums of x[M][ld]; scratch must hold ofb_colsum_slabs(M,N)*N floats.
extern "C" int ofb_colsum(const float* x, int32_t ld, int32_t M, int32_t N, const float* rowscale, int32_t rs_div,
                          float* out, float* scratch, void* stream) {
  if (!x || !out || M <= 0 || N <= 0 || ld < N || (rowscale && rs_div <= 0)) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int slabs = ofb_colsum_slabs(M, N);
  const int colblocks = ofb_cdiv(N, 64);
  if (slabs == 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3(colblocks, 1), dim3(256), 0, s, x, ld, M, N, rowscale, rs_div, out, M);
    return ofb_launch_status();
  }
  if (!scratch) return OFB_EINVAL;
  const int rps = ofb_cdiv(M, slabs);
  hipLaunchKernelGGL(colsum_kernel, dim3(colblocks, slabs), dim3(256), 0, s, x, ld, M, N, rowscale, rs_div, scratch, rps);
  hipLaunchKernelGGL(colsum_kernel, dim3(colblocks, 1), dim3(256), 0, s, (const float*)scratch, N, slabs, N,
                     (const float*)nullptr, 1, out, slabs);
  return ofb_launch_status();
}

extern "C" int ofb_scale_rows(const float* W, const float* g, float* out, int32_t N, int32_t K, void* stream) {
  if (!W || !g || !out || N <= 0 || K <= 0) return OFB_EINVAL;
  if ((K & 3) == 0 && (!ofb_aligned16(W) || !ofb_aligned16(out))) return OFB_EINVAL;
  const int64_t total = (K & 3) == 0 ? (int64_t)N * K / 4 : (int64_t)N * K;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, g, out, N, K);
  return ofb_launch_status();
}

extern "C" int ofb_gate_fold_bwd(const float* dWraw, const float* W, const float* g, const float* dbraw, int32_t dbraw_rows,
                                 const float* b, float* dW, float* db, float* dg, int32_t N, int32_t K, int32_t fold, void* stream) {
  if (!dWraw || !W || !g || !dW || !dg || N <= 0 || K <= 0 || (dbraw && dbraw_rows <= 0) || fold <= 0 || N % fold) return OFB_EINVAL;
  hipLaunchKernelGGL(gate_fold_bwd_kernel, dim3(ofb_cdiv(N / fold, 4)), dim3(256), 0, (hipStream_t)stream, dWraw, W, g, dbraw, b,
                     dW, db, dg, N, K, dbraw_rows, fold);
  return ofb_launch_status();
}


// out[N] = column sums of x[M][ld] for every job, one launch; max_N = the widest job
extern "C" int ofb_colsum_multi(const ofb_colsum_job* jobs_dev, int32_t n_jobs, int32_t max_N, void* stream) {
  if (!jobs_dev || n_jobs <= 0 || n_jobs > 65535 || max_N <= 0) return OFB_EINVAL;
  hipLaunchKernelGGL(colsum_multi_kernel, dim3(ofb_cdiv(max_N, 64), n_jobs), dim3(1024), 0, (hipStream_t)stream, jobs_dev);
  return ofb_launch_status();
}
