// f32 GEMM with fused epilogues.  Block tile 128x128x16, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles (64 accumulator
// VGPRs).  Two matrix engines share everything but the LDS image and the inner product (OFB_GEMM_BF16X6):
//
//  1 (default)  f32 operands are split EXACTLY into three bf16 values (x = hi + mid + lo, round-to-nearest residual chain)
//               while they are staged to LDS, and every K-step issues the six leading product terms (mid*mid, hi*lo, lo*hi,
//               hi*mid, mid*hi, hi*hi; the dropped ones are <= 2^-24 relative) on v_mfma_f32_32x32x16_bf16 with f32
//               accumulation.  Measured error equals that of an f32 fma chain (1.3e-7 of sum|a*b| at K = 384..1536); the
//               bf16 matrix pipe has 16x the f32-MFMA rate, so six terms still run ~1.6x faster than the f32 MFMA form.
//               LDS image per operand tile: 3 planes x 2 k-halves x [128 rows][8 bf16] (fragment = one ds_read_b128).
//  0            v_mfma_f32_32x32x2_f32 on f32 tiles staged [mn][k] (row pitch 20 floats): bit-exact f32 fma chains.
//
// Operands go global -> registers -> LDS with a one-tile prefetch; K-contiguous operands are written row-wise,
// MN-contiguous ones are transposed on the way in, so the three storage combinations the Linear layers need
// (x@W^T, dY@W, dY^T@X) share one inner loop.
//
// Scheduling (hybrid stream-K): W = CUs x 3 persistent workgroups.  Output tiles that fill whole rounds of W are
// computed data-parallel with the epilogue fused; the R = tiles mod W remaining tiles are cut along K into W equal
// runs of K-iterations (a run touches at most two tiles), written as raw partial tiles to a workspace and summed
// by a small fix-up kernel that applies the same epilogue.  Every workgroup therefore issues the same number of
// MFMAs (within one K-iteration) whatever the tile count, and weight gradients (few tiles, long K = all tokens) need
// no separate split-K path.  Partials are summed in a fixed order: results are run-to-run deterministic.
#include "ofb_common.h"
#include <type_traits>

#define BM 128
#define BN 128
#ifndef BK
#define BK 16
#endif
#define LDP (BK + 4)   // LDS row pitch in floats ([mn][k] layout): 80-B / 144-B rows keep b128 fragment reads conflict-free
#define NLD (BM * BK / 4 / 256)   // float4 loads per thread and operand tile
#ifndef GEMM_WAVES_PER_SIMD
#define GEMM_WAVES_PER_SIMD 2
#endif
#ifndef OFB_GEMM_BF16X6
#define OFB_GEMM_BF16X6 1
#endif
#if OFB_GEMM_BF16X6
static_assert(BK == 16, "the bf16x6 engine stages one 32x32x16 MFMA K-step per tile");
#define SPL_BLK (BM * 16 + 16)       // bytes of one [128 rows][8 bf16] block (+16: the two k-halves land on different banks)
#define SPL_PLANE (2 * SPL_BLK)      // k-halves
#define SPL_OPER (3 * SPL_PLANE)     // hi / mid / lo planes
#define LDS_OPER_FLOATS (SPL_OPER / 4)
#else
#define LDS_OPER_FLOATS (BM * LDP)
#endif

namespace {

struct TileRegs { f32x4 v[NLD]; };

// K-contiguous storage X[o*ld + k]: 512 float4 per tile, 2 per thread (o = idx>>2, kq = idx&3) -> one ds_write_b128 each.
template <bool VEC, bool GUARD>
__device__ __forceinline__ void load_kc(TileRegs& r, const float* __restrict__ X, int ld, int o0, int O, int k0, int kend,
                                        int t) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = t + 256 * i, o = o0 + idx / (BK / 4), k = k0 + ((idx % (BK / 4)) << 2);
    if (!GUARD) {      // full tiles only (M, N multiples of 128, K multiple of BK): no bounds checks at all
      r.v[i] = *reinterpret_cast<const f32x4*>(X + (size_t)o * ld + k);
      continue;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (o < O) {
      const float* p = X + (size_t)o * ld + k;
      if (VEC) {
        if (k < kend) v = *reinterpret_cast<const f32x4*>(p);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k + j < kend) v[j] = p[j];
      }
    }
    r.v[i] = v;
  }
}
#if OFB_GEMM_BF16X6
typedef __bf16 ofb_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ofb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ofb_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {      // v_cvt_pk_bf16_f32: a -> low half, b -> high half (RNE)
  ofb_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ofb_bf16x2));
}
// x = hi + mid + lo with each part a bf16 (24 significant bits in total: exact for finite f32 in the normal range);
// two values at a time so every step is one packed instruction.
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = pack_bf16(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  mid = pack_bf16(ra, rb);
  lo = pack_bf16(ra - __uint_as_float(mid << 16), rb - __uint_as_float(mid & 0xffff0000u));
}
// K-contiguous tile: thread item (row = idx/4, k = 4*(idx%4) .. +3) -> 4 bf16 (8 B) per plane at [k>>3][row][k&7]
__device__ __forceinline__ void store_kc(const TileRegs& r, float* __restrict__ S, int t) {
  char* base = reinterpret_cast<char*>(S);
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = t + 256 * i, row = idx >> 2, kq = idx & 3;
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair(r.v[i][0], r.v[i][1], h0, m0, l0);
    split_pair(r.v[i][2], r.v[i][3], h1, m1, l1);
    char* p = base + (kq >> 1) * SPL_BLK + row * 16 + (kq & 1) * 8;
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + SPL_PLANE) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(p + 2 * SPL_PLANE) = make_uint2(l0, l1);
  }
}
#else
__device__ __forceinline__ void store_kc(const TileRegs& r, float* __restrict__ S, int t) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = t + 256 * i;
    *reinterpret_cast<f32x4*>(&S[(idx / (BK / 4)) * LDP + ((idx % (BK / 4)) << 2)]) = r.v[i];
  }
}
#endif
#if OFB_GEMM_BF16X6
// MN-contiguous storage X[k*ld + o]: thread t takes rows o = 4*(t/8) .. +3 at the k PAIR (2*(t%8), 2*(t%8)+1): v[0] = even k,
// v[1] = odd k, so that each output row's two values pack into one bf16x2 LDS word per plane (the transpose costs
// 12 ds_write_b32 per thread and tile instead of 24 ds_write_b16).
template <bool VEC, bool GUARD>
__device__ __forceinline__ void load_mc(TileRegs& r, const float* __restrict__ X, int ld, int o0, int O, int k0, int kend,
                                        int t, const float* __restrict__ kscale, int ks_div) {
  static_assert(NLD == 2, "pair loader assumes 2 float4 per thread");
  const int o = o0 + ((t >> 3) << 2);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = k0 + 2 * (t & 7) + i;
    if (!GUARD) {
      f32x4 u = *reinterpret_cast<const f32x4*>(X + (size_t)k * ld + o);
      if (kscale) u *= kscale[ks_div == 1 ? k : k / ks_div];
      r.v[i] = u;
      continue;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k < kend) {
      const float* p = X + (size_t)k * ld + o;
      if (VEC) {
        if (o < O) v = *reinterpret_cast<const f32x4*>(p);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (o + j < O) v[j] = p[j];
      }
      if (kscale) v *= kscale[ks_div == 1 ? k : k / ks_div];
    }
    r.v[i] = v;
  }
}
__device__ __forceinline__ void store_mc(const TileRegs& r, float* __restrict__ S, int t) {
  const int kp = t & 7, row0 = (t >> 3) << 2;
  char* base = reinterpret_cast<char*>(S) + (kp >> 2) * SPL_BLK + (kp & 3) * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned h, m, l;
    split_pair(r.v[0][j], r.v[1][j], h, m, l);
    char* p = base + (row0 + j) * 16;
    *reinterpret_cast<unsigned*>(p) = h;
    *reinterpret_cast<unsigned*>(p + SPL_PLANE) = m;
    *reinterpret_cast<unsigned*>(p + 2 * SPL_PLANE) = l;
  }
}
#else
// MN-contiguous storage X[k*ld + o]: kk = idx&15, oq = idx>>4 (consecutive lanes take consecutive k rows so that the
// transposing b32 writes S[(4oq+j)][kk] hit 32 distinct banks).
template <bool VEC, bool GUARD>
__device__ __forceinline__ void load_mc(TileRegs& r, const float* __restrict__ X, int ld, int o0, int O, int k0, int kend,
                                        int t, const float* __restrict__ kscale, int ks_div) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = t + 256 * i, k = k0 + (idx % BK), o = o0 + ((idx / BK) << 2);
    if (!GUARD) {
      f32x4 u = *reinterpret_cast<const f32x4*>(X + (size_t)k * ld + o);
      if (kscale) u *= kscale[ks_div == 1 ? k : k / ks_div];
      r.v[i] = u;
      continue;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k < kend) {
      const float* p = X + (size_t)k * ld + o;
      if (VEC) {
        if (o < O) v = *reinterpret_cast<const f32x4*>(p);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (o + j < O) v[j] = p[j];
      }
      if (kscale) v *= kscale[ks_div == 1 ? k : k / ks_div];
    }
    r.v[i] = v;
  }
}
__device__ __forceinline__ void store_mc(const TileRegs& r, float* __restrict__ S, int t) {
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = t + 256 * i, kk = idx % BK, o = (idx / BK) << 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) S[(o + j) * LDP + kk] = r.v[i][j];
  }
}
#endif

struct Plan { int mt, nt, ntiles, I, W, full_rounds, R, q, S, qs; };

__host__ __device__ inline Plan make_plan(int M, int N, int K, int W) {
  Plan p;
  p.mt = (M + BM - 1) / BM;
  p.nt = (N + BN - 1) / BN;
  p.ntiles = p.mt * p.nt;
  p.I = (K + BK - 1) / BK;
  p.W = W;
  p.full_rounds = p.ntiles / W;
  p.R = p.ntiles - p.full_rounds * W;
  // After >= 3 full rounds a remainder that fills at least half a round runs as one more (partly idle) data-parallel round:
  // the idle share (< 1/8 of the launch) costs less than the partial-tile traffic and the fix-up launch of a streamed tail.
  if (p.full_rounds >= 3 && 2 * p.R >= W) { p.full_rounds += 1; p.R = 0; }
  p.q = p.R ? (int)(((long long)p.R * p.I + W - 1) / W) : 0;      // K-iterations per workgroup in the streamed tail (q <= I)
  // Split-major tail: when the W workgroups divide (almost) evenly over the R tail tiles, cut every tile's K range into the
  // same S pieces and give workgroup v piece v / R of tile v % R.  Workgroups that run side by side on one XCD then walk the
  // SAME K rows of different tiles and share them in its L2 (weight gradients: every token row of dY / X is needed by all
  // tiles), where the flattened runs above place neighbours on different K ranges of one tile and nothing is shared.
  p.S = 0; p.qs = 0;
  if (p.R > 0) {
    const int S = W / p.R;
    if (S >= 2 && (W - S * p.R) * 10 <= W) { p.S = S; p.qs = (p.I + S - 1) / S; }
  }
  return p;
}


// tile id -> (row block, column tile).  OFB_GEMM_PANEL = c > 0: the column tiles are walked in panels of c (row-major inside a
// panel), so the workgroups that run side by side on one XCD share c weight slices instead of all nt of them.
#ifndef OFB_GEMM_PANEL
#define OFB_GEMM_PANEL 0
#endif
__host__ __device__ __forceinline__ void tile_coord(const Plan& p, int tile, int& m0, int& n0) {
#if OFB_GEMM_PANEL > 0
  const int per = p.mt * OFB_GEMM_PANEL;
  const int panel = tile / per, within = tile - panel * per;
  const int c = min(OFB_GEMM_PANEL, p.nt - panel * OFB_GEMM_PANEL);
  m0 = (within / c) * BM; n0 = (panel * OFB_GEMM_PANEL + within % c) * BN;
#else
  m0 = (tile / p.nt) * BM; n0 = (tile % p.nt) * BN;
#endif
}

// One unit of work: K-iterations [it0, it1) of output tile `tile`; slot < 0 -> full tile, fused epilogue to C;
// slot >= 0 -> raw partial tile to workspace[slot].
struct Seg { int m0, n0, it0, it1, slot; bool ok; };

template <bool TAIL>
__device__ __forceinline__ Seg get_seg(const Plan p, int v, int idx) {
  Seg s;
  s.m0 = s.n0 = s.it0 = s.it1 = 0; s.slot = -1; s.ok = false;
  if (!TAIL) {
    if (idx >= p.full_rounds) return s;
    const int tile = v + idx * p.W;
    if (tile >= p.ntiles) return s;
    tile_coord(p, tile, s.m0, s.n0); s.it0 = 0; s.it1 = p.I; s.ok = true;
    return s;
  }
  const int part = idx;
  if (part > 1 || p.R == 0) return s;
  if (p.S > 0) {                          // split-major: one piece per workgroup, slot = v
    if (part != 0) return s;
    const int sp = v / p.R, tl = v - sp * p.R;
    const int i0 = sp * p.qs, i1 = min(i0 + p.qs, p.I);
    if (sp >= p.S || i0 >= i1) return s;
    const int tile = p.full_rounds * p.W + tl;
    tile_coord(p, tile, s.m0, s.n0); s.it0 = i0; s.it1 = i1; s.slot = v; s.ok = true;
    return s;
  }
  // R * I < W * I <= 768 * (K/16): fits 32 bits for every K the host accepts (checked in ofb_gemm_f32)
  const int beg = v * p.q, tot = p.R * p.I;
  const int end = min(beg + p.q, tot);
  if (beg >= end) return s;
  const int a = beg / p.I, it0 = beg - a * p.I;
  const int n0 = end - beg;
  const int first = min(p.I - it0, n0);
  int tl, i0, i1;
  if (part == 0) { tl = a; i0 = it0; i1 = it0 + first; }
  else {
    if (n0 - first <= 0) return s;
    tl = a + 1; i0 = 0; i1 = n0 - first;
  }
  const int tile = p.full_rounds * p.W + tl;
  tile_coord(p, tile, s.m0, s.n0); s.it0 = i0; s.it1 = i1; s.slot = 2 * v + part; s.ok = true;
  return s;
}

// v = alpha*acc (+bias)(*colscale); act; (*rowscale); (+resid)   -- shared by the fused epilogue and the fix-up kernel
__device__ __forceinline__ float epilogue_value(float alpha, int act, float* __restrict__ aux, int ldaux, float accv, int row,
                                                int col, float bias, float cs, float rsv, float rv, float av) {
  float v = (accv * alpha + bias) * cs;
  if (act == OFB_ACT_GELU) {
    if (aux) aux[(size_t)row * ldaux + col] = v;
    v = ofb_gelu(v);
  } else if (act == OFB_ACT_DGELU) {
    v *= ofb_dgelu(av);
  }
  return v * rsv + rv;
}

// TAIL = false: the full rounds (tile = v, v + W, ...; fused epilogue).  TAIL = true: the streamed remainder (<= 2 runs of
// K-iterations per workgroup, raw partial tiles to the workspace).  Same main loop; launched back to back.
template <bool A_KC, bool B_KC, bool VEC, bool GUARD, bool FULL_EPI, bool TAIL, bool DEFER>
__global__ __launch_bounds__(256, GEMM_WAVES_PER_SIMD) void gemm_f32_kernel(const ofb_gemm_args g, const Plan p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * LDS_OPER_FLOATS];
  float* As = lds;                          // [2 buffers] of one staged A tile
  float* Bs = lds + 2 * LDS_OPER_FLOATS;    // [2 buffers] of one staged B tile

  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
  const int v = ofb_xcd_remap(blockIdx.x, p.W);     // consecutive v share an XCD (and thus A/B panels in its L2)
  const int wm0 = (w >> 1) * 64, wn0 = (w & 1) * 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileRegs ra, rb;
  auto gload = [&](const Seg& sg, int it) __attribute__((always_inline)) {
    const int k0 = it * BK;
    if (A_KC) load_kc<VEC, GUARD>(ra, g.A, g.lda, sg.m0, g.M, k0, g.K, t);
    else load_mc<VEC, GUARD>(ra, g.A, g.lda, sg.m0, g.M, k0, g.K, t, g.kscale, g.ks_div);
    if (B_KC) load_kc<VEC, GUARD>(rb, g.B, g.ldb, sg.n0, g.N, k0, g.K, t);
    else load_mc<VEC, GUARD>(rb, g.B, g.ldb, sg.n0, g.N, k0, g.K, t, nullptr, 1);
  };
  // fused bias gradient (weight-gradient launches): column sums of the stored A (= dY) for the workgroups that own the
  // first column tile.
  float bsum = 0.f;
#if OFB_GEMM_BF16X6
  f32x4 bsum4 = {0.f, 0.f, 0.f, 0.f};    // rows 4*(t/8) .. +3, this thread's k pairs; reduced over the 8 pair-lanes at unit end
#endif
  auto lstore = [&](int buf, int tile_n0) __attribute__((always_inline)) {
#if OFB_GEMM_BF16X6
    if (TAIL && !A_KC && g.a_colsum && tile_n0 == 0) bsum4 += ra.v[0] + ra.v[1];
#endif
    if (A_KC) store_kc(ra, As + buf * LDS_OPER_FLOATS, t); else store_mc(ra, As + buf * LDS_OPER_FLOATS, t);
    if (B_KC) store_kc(rb, Bs + buf * LDS_OPER_FLOATS, t); else store_mc(rb, Bs + buf * LDS_OPER_FLOATS, t);
  };
#if OFB_GEMM_BF16X6
  auto compute = [&](int buf) __attribute__((always_inline)) {
    // lane (row l31, k-half h) reads its 8 bf16 of each plane with one b128; A and B share the k <-> (half, j) map.
    const char* a_s = reinterpret_cast<const char*>(As + buf * LDS_OPER_FLOATS) + h * SPL_BLK + (wm0 + l31) * 16;
    const char* b_s = reinterpret_cast<const char*>(Bs + buf * LDS_OPER_FLOATS) + h * SPL_BLK + (wn0 + l31) * 16;
    ofb_bf16x8 af[2][3], bf[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        af[i][pl] = *reinterpret_cast<const ofb_bf16x8*>(a_s + pl * SPL_PLANE + i * 32 * 16);
        bf[i][pl] = *reinterpret_cast<const ofb_bf16x8*>(b_s + pl * SPL_PLANE + i * 32 * 16);
      }
    // six product terms, smallest first: (mid,mid) (hi,lo) (lo,hi) (hi,mid) (mid,hi) (hi,hi)
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][TA[q]], bf[j][TB[q]], acc[i][j], 0, 0, 0);
  };
#else
  auto compute = [&](int buf) __attribute__((always_inline)) {
    // fragment reads: lane (row l31, half h) takes k = 8q + 4h + j (q < BK/8; j = 0..3) with one b128 per q; A and B
    // use the same k <-> (q, h, j) map, so each MFMA step (q, j) multiplies matching k's.
    const float* a_s = As + buf * BM * LDP + (wm0 + l31) * LDP + 4 * h;
    const float* b_s = Bs + buf * BN * LDP + (wn0 + l31) * LDP + 4 * h;
    f32x4 af[2][BK / 8], bf[2][BK / 8];
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      af[0][q] = *reinterpret_cast<const f32x4*>(a_s + 8 * q);
      af[1][q] = *reinterpret_cast<const f32x4*>(a_s + 32 * LDP + 8 * q);
      bf[0][q] = *reinterpret_cast<const f32x4*>(b_s + 8 * q);
      bf[1][q] = *reinterpret_cast<const f32x4*>(b_s + 32 * LDP + 8 * q);
    }
#pragma unroll
    for (int q = 0; q < BK / 8; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][q][j], bf[0][q][j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][q][j], bf[1][q][j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][q][j], bf[0][q][j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][q][j], bf[1][q][j], acc[1][1], 0, 0, 0);
      }
  };
#endif

  int sidx = 0;
  Seg cur = get_seg<TAIL>(p, v, 0);
  if (!cur.ok) return;
  // f32 engine: the sums are taken from the staged f32 A tile (thread t < 128 owns output row t); the split engine adds
  // the registers up while staging them (lstore) because its LDS image is no longer f32.
  auto colsum_acc = [&](int buf) {
#if !OFB_GEMM_BF16X6
    if (TAIL && !A_KC && g.a_colsum && cur.n0 == 0 && t < BM) {
      const float* row = As + buf * BM * LDP + t * LDP;
#pragma unroll
      for (int q = 0; q < BK / 4; ++q) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(row + 4 * q);
        bsum += (u[0] + u[1]) + (u[2] + u[3]);
      }
    }
#endif
  };
  gload(cur, cur.it0);
  lstore(0, cur.n0);
  __syncthreads();
  int buf = 0;

  if constexpr (DEFER) {
    // ---- full rounds, full tiles, >= 17 K-iterations: DEFERRED, DISTRIBUTED epilogue -------------------------------
    // A finished tile's accumulators move to `pacc` and are written out 4 rows x (mi, ni) at a time during the first 16
    // K-iterations of the NEXT tile (that stretch of the K loop is unrolled so every group index is static): side
    // inputs are requested before the MFMA block and consumed after it, stores drain while the next MFMA block runs.
    // The epilogue's HBM traffic is thereby spread under the matrix work instead of arriving as one burst per round
    // during which every co-resident workgroup idles its MFMA pipe.
    f32x16 pacc[2][2];
    bool pend = false;
    int pm0 = 0, pn0 = 0;
    float pbias0 = 0.f, pbias1 = 0.f, pcs0 = 1.f, pcs1 = 1.f;
    f32x4 e_rv = {0.f, 0.f, 0.f, 0.f}, e_av = {0.f, 0.f, 0.f, 0.f}, e_rs = {1.f, 1.f, 1.f, 1.f};
    auto epi_load = [&](auto G) __attribute__((always_inline)) {          // request the side inputs of group G
      constexpr int gi = decltype(G)::value, ni = gi >> 3, mi = (gi >> 2) & 1, rg = gi & 3;
      if (!FULL_EPI) return;
      const int col = pn0 + wn0 + 32 * ni + l31, rbase = pm0 + wm0 + 32 * mi + 4 * h + 8 * rg;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int row = rbase + r4;
        if (g.resid) e_rv[r4] = g.resid[(size_t)row * g.ldr + col];
        if (g.act == OFB_ACT_DGELU) e_av[r4] = g.aux[(size_t)row * g.ldaux + col];
        if (g.rowscale) e_rs[r4] = g.rowscale[g.rs_div == 1 ? row : row / g.rs_div];
      }
    };
    auto epi_store = [&](auto G) __attribute__((always_inline)) {         // finish and store group G
      constexpr int gi = decltype(G)::value, ni = gi >> 3, mi = (gi >> 2) & 1, rg = gi & 3;
      const int col = pn0 + wn0 + 32 * ni + l31, rbase = pm0 + wm0 + 32 * mi + 4 * h + 8 * rg;
      const float bias = ni ? pbias1 : pbias0, cs = ni ? pcs1 : pcs0;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int row = rbase + r4;
        const float a = pacc[mi][ni][4 * rg + r4];
        if (FULL_EPI) g.C[(size_t)row * g.ldc + col] = epilogue_value(g.alpha, g.act, g.aux, g.ldaux, a, row, col, bias, cs, e_rs[r4], e_rv[r4], e_av[r4]);
        else g.C[(size_t)row * g.ldc + col] = (a * g.alpha + bias) * cs;
      }
    };
    auto kstep = [&](int it, auto G) __attribute__((always_inline)) {     // one K-iteration (not the tile's last) + group G
      if (pend) epi_load(G);
      gload(cur, it + 1);
      compute(buf);
      __builtin_amdgcn_sched_barrier(0);            // every MFMA of this K-tile stays ahead of the wait / LDS refill
      lstore(buf ^ 1, cur.n0);
      if (pend) epi_store(G);
      __syncthreads();
      buf ^= 1;
    };
#define OFB_IC(n) std::integral_constant<int, n>{}
    while (true) {
      kstep(0, OFB_IC(0));   kstep(1, OFB_IC(1));   kstep(2, OFB_IC(2));   kstep(3, OFB_IC(3));
      kstep(4, OFB_IC(4));   kstep(5, OFB_IC(5));   kstep(6, OFB_IC(6));   kstep(7, OFB_IC(7));
      kstep(8, OFB_IC(8));   kstep(9, OFB_IC(9));   kstep(10, OFB_IC(10)); kstep(11, OFB_IC(11));
      kstep(12, OFB_IC(12)); kstep(13, OFB_IC(13)); kstep(14, OFB_IC(14)); kstep(15, OFB_IC(15));
      for (int it = 16; it + 1 < p.I; ++it) {                              // the rest of the K loop (host guarantees I >= 17)
        gload(cur, it + 1);
        compute(buf);
        __builtin_amdgcn_sched_barrier(0);
        lstore(buf ^ 1, cur.n0);
        __syncthreads();
        buf ^= 1;
      }
      const Seg nxt = get_seg<false>(p, v, sidx + 1);
      const bool has_next = nxt.ok;
      if (has_next) gload(nxt, 0);
      compute(buf);
      __builtin_amdgcn_sched_barrier(0);
      // this tile becomes the pending one
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          pacc[mi][ni] = acc[mi][ni];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        }
      pend = true;
      pm0 = cur.m0; pn0 = cur.n0;
      {
        const int c0 = pn0 + wn0 + l31;
        pbias0 = g.bias ? g.bias[c0] : 0.f; pbias1 = g.bias ? g.bias[c0 + 32] : 0.f;
        pcs0 = g.colscale ? g.colscale[c0] : 1.f; pcs1 = g.colscale ? g.colscale[c0 + 32] : 1.f;
      }
      if (!has_next) break;
      lstore(buf ^ 1, nxt.n0);
      __syncthreads();
      buf ^= 1;
      ++sidx;
      cur = nxt;
    }
    // the worker's last tile: plain epilogue
#define OFB_FLUSH(n) epi_load(OFB_IC(n)); epi_store(OFB_IC(n));
    OFB_FLUSH(0) OFB_FLUSH(1) OFB_FLUSH(2) OFB_FLUSH(3) OFB_FLUSH(4) OFB_FLUSH(5) OFB_FLUSH(6) OFB_FLUSH(7)
    OFB_FLUSH(8) OFB_FLUSH(9) OFB_FLUSH(10) OFB_FLUSH(11) OFB_FLUSH(12) OFB_FLUSH(13) OFB_FLUSH(14) OFB_FLUSH(15)
#undef OFB_FLUSH
#undef OFB_IC
    return;
  }

  while (true) {
    // all K-iterations of this unit but the last: prefetch the next K-tile of the same unit
    for (int it = cur.it0; it + 1 < cur.it1; ++it) {
      gload(cur, it + 1);
      // pin the prefetch ahead of the MFMA block: left alone, hipcc lets the loaded tile share VGPRs with the fragments and
      // sinks the global loads behind the last MFMAs, exposing their whole latency in front of the LDS refill
#ifndef LAB_NO_LOAD_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      colsum_acc(buf);      // its LDS reads / adds are issued ahead of (and overlap) the MFMA block
      compute(buf);
      // keep every MFMA of this K-tile ahead of the vmcnt wait / LDS refill / barrier
      __builtin_amdgcn_sched_barrier(0);
      lstore(buf ^ 1, cur.n0);
      __syncthreads();
      buf ^= 1;
    }
    // last K-iteration: the NEXT unit's first K-tile goes in flight before this unit's stores
    const Seg nxt = get_seg<TAIL>(p, v, sidx + 1);
    const bool has_next = nxt.ok;
    if (has_next) gload(nxt, nxt.it0);
#ifndef LAB_NO_LOAD_PIN
    __builtin_amdgcn_sched_barrier(0);
#endif
    colsum_acc(buf);
    compute(buf);
    __builtin_amdgcn_sched_barrier(0);
    if (TAIL) {
#if OFB_GEMM_BF16X6
      if (!A_KC && g.a_colsum && cur.n0 == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {                 // sum over the 8 k-pair lanes (lane bits 0..2) that share these 4 rows
          float s4 = bsum4[j];
          s4 += __shfl_xor(s4, 1);
          s4 += __shfl_xor(s4, 2);
          s4 += __shfl_xor(s4, 4);
          if ((t & 7) == 0) g.workspace[(size_t)2 * p.W * (BM * BN) + (size_t)cur.slot * BM + ((t >> 3) << 2) + j] = s4;
          bsum4[j] = 0.f;
        }
      }
#else
      if (!A_KC && g.a_colsum && cur.n0 == 0 && t < BM) {
        g.workspace[(size_t)2 * p.W * (BM * BN) + (size_t)cur.slot * BM + t] = bsum;
        bsum = 0.f;
      }
#endif
      // raw partial tile -> workspace[slot][128][128] (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
      float* ws = g.workspace + (size_t)cur.slot * (BM * BN);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            ws[(wm0 + 32 * mi + 4 * h + (r & 3) + 8 * (r >> 2)) * BN + wn0 + 32 * ni + l31] = acc[mi][ni][r];
            acc[mi][ni][r] = 0.f;
          }
        }
    } else if (!GUARD) {
      // unguarded build (every tile is full): no per-element guards, so loads / stores issue back to back behind ONE wait
      // (hipcc otherwise brackets every guarded store with s_waitcnt vmcnt(0), serialising 64 round trips per wave)
      float biasv[2], csv[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = cur.n0 + wn0 + 32 * ni + l31;
        biasv[ni] = g.bias ? g.bias[col] : 0.f;
        csv[ni] = g.colscale ? g.colscale[col] : 1.f;
      }
      if (FULL_EPI) {
        // side inputs (residual / saved pre-activation / per-row scale) are requested one whole 32x32 block (16 values per
        // lane) ahead of the block being finished, so ~32 loads per lane are in flight instead of 4: the epilogue is bound
        // by memory latency, not bandwidth, and used to expose one round trip per 4 rows.
        // `side` carries the saved pre-activation for the dGELU form and the residual otherwise (a launch that wants both
        // reads its residual inside `finish`); the per-row scales of a 32-row band serve both of its column blocks.
        const bool dg = g.act == OFB_ACT_DGELU;
        const float* sp = dg ? g.aux : g.resid;
        const int lds_ = dg ? g.ldaux : g.ldr;
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        if (!sp && !g.rowscale) {
          // nothing to fetch (bias / gate / GELU forms): finish and stream out all 64 values, no staging, no waits
          auto stream_out = [&](auto Bk) __attribute__((always_inline)) {
            constexpr int bk = decltype(Bk)::value, mi = bk >> 1, ni = bk & 1;
            const int col = cur.n0 + wn0 + 32 * ni + l31, rbase = cur.m0 + wm0 + 32 * mi + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = rbase + (r & 3) + 8 * (r >> 2);
              g.C[(size_t)row * g.ldc + col] = epilogue_value(g.alpha, g.act, g.aux, g.ldaux, acc[mi][ni][r], row, col, biasv[ni], csv[ni],
                                                              1.f, 0.f, 0.f);
              acc[mi][ni][r] = 0.f;
            }
          };
          stream_out(I0{}); stream_out(I1{}); stream_out(I2{}); stream_out(I3{});
        } else {
          // all four blocks' side inputs are requested before the first is finished: one exposed round trip per tile
          f32x16 side[4], rsv[2];
          auto request = [&](auto Bk) __attribute__((always_inline)) {
            constexpr int bk = decltype(Bk)::value, mi = bk >> 1, ni = bk & 1;
            const int col = cur.n0 + wn0 + 32 * ni + l31, rbase = cur.m0 + wm0 + 32 * mi + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = rbase + (r & 3) + 8 * (r >> 2);
              side[bk][r] = sp ? sp[(size_t)row * lds_ + col] : 0.f;
              if (ni == 0) rsv[mi][r] = g.rowscale ? g.rowscale[g.rs_div == 1 ? row : row / g.rs_div] : 1.f;
            }
          };
          auto finish = [&](auto Bk) __attribute__((always_inline)) {
            constexpr int bk = decltype(Bk)::value, mi = bk >> 1, ni = bk & 1;
            const int col = cur.n0 + wn0 + 32 * ni + l31, rbase = cur.m0 + wm0 + 32 * mi + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = rbase + (r & 3) + 8 * (r >> 2);
              const float rvv = dg ? (g.resid ? g.resid[(size_t)row * g.ldr + col] : 0.f) : side[bk][r];
              g.C[(size_t)row * g.ldc + col] = epilogue_value(g.alpha, g.act, g.aux, g.ldaux, acc[mi][ni][r], row, col, biasv[ni], csv[ni],
                                                              rsv[mi][r], rvv, dg ? side[bk][r] : 0.f);
              acc[mi][ni][r] = 0.f;
            }
          };
          request(I0{}); request(I1{}); request(I2{}); request(I3{});
          __builtin_amdgcn_sched_barrier(0);
          finish(I0{}); finish(I1{}); finish(I2{}); finish(I3{});
        }
      } else {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int col = cur.n0 + wn0 + 32 * ni + l31;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const int rbase = cur.m0 + wm0 + 32 * mi + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              g.C[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * g.ldc + col] = (acc[mi][ni][r] * g.alpha + biasv[ni]) * csv[ni];
              acc[mi][ni][r] = 0.f;
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = cur.n0 + wn0 + 32 * ni + l31;
        const bool colok = col < g.N;
        float bias = 0.f, cs = 1.f;
        if (colok) {
          if (g.bias) bias = g.bias[col];
          if (g.colscale) cs = g.colscale[col];
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int rbase = cur.m0 + wm0 + 32 * mi + 4 * h;
          if (FULL_EPI) {
            // per group of 4 rows: gather the side inputs first (independent loads in flight together), then compute + store
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              f32x4 rv, av, rsv;     // vector types: register-resident with the static indices below
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                const int row = rbase + r4 + 8 * rg;
                const bool ok = colok && row < g.M;
                rv[r4] = (g.resid && ok) ? g.resid[(size_t)row * g.ldr + col] : 0.f;
                av[r4] = (g.act == OFB_ACT_DGELU && ok) ? g.aux[(size_t)row * g.ldaux + col] : 0.f;
                rsv[r4] = (g.rowscale && row < g.M) ? g.rowscale[g.rs_div == 1 ? row : row / g.rs_div] : 1.f;
              }
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                const int row = rbase + r4 + 8 * rg;
                if (colok && row < g.M)
                  g.C[(size_t)row * g.ldc + col] = epilogue_value(g.alpha, g.act, g.aux, g.ldaux, acc[mi][ni][4 * rg + r4], row, col, bias, cs, rsv[r4], rv[r4], av[r4]);
              }
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = rbase + (r & 3) + 8 * (r >> 2);
              if (colok && row < g.M) g.C[(size_t)row * g.ldc + col] = (acc[mi][ni][r] * g.alpha + bias) * cs;
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        }
      }
    }
    if (!has_next) break;
    lstore(buf ^ 1, nxt.n0);
    __syncthreads();
    buf ^= 1;
    ++sidx;
    cur = nxt;
  }
}

// Sums the partial tiles of each streamed tail tile in workgroup order and applies the epilogue.
// grid (R, 32), 128 threads: block (r, part) handles rows [4*part, 4*part+4) of tail tile r, one float4 (4 columns) per thread.
// The kernel is bound by loads in flight, not bandwidth (weight gradients: a few dozen tail tiles with ~14-30 contributors
// each): 16-byte loads, eight contributors per trip, 32 blocks per tile.  The sums stay in contributor order (deterministic).
#define FIX_PARTS 32
#define FIX_ROWS (BM / FIX_PARTS)
#define FIX_THREADS (FIX_ROWS * BN / 4)
__global__ __launch_bounds__(FIX_THREADS) void gemm_fixup_kernel(const ofb_gemm_args g, const Plan p) {
  const int r = blockIdx.x, part = blockIdx.y, t = threadIdx.x;
  const int tile = p.full_rounds * p.W + r;
  int m0, n0;
  tile_coord(p, tile, m0, n0);
  const int lo = r * p.I, hi = lo + p.I;         // this tile's run of flattened K-iterations
  // contributors: split-major -> pieces i = 0 .. n-1 in slots i*R + r; flattened -> workgroups v0 .. v1 (slot 2v+1 when the
  // workgroup's run started in the previous tile, else 2v)
  const int v0 = p.S ? 0 : lo / p.q, v1 = p.S ? (p.I + p.qs - 1) / p.qs - 1 : (hi - 1) / p.q;
  auto slot_of = [&](int v) { return p.S ? v * p.R + r : ((v * p.q < lo) ? 2 * v + 1 : 2 * v); };
  const int lrow = FIX_ROWS * part + t / (BN / 4), c4 = (t % (BN / 4)) * 4;
  const size_t roff = (size_t)lrow * BN + c4;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  int v = v0;
  for (; v + 7 <= v1; v += 8) {
    f32x4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const f32x4*>(g.workspace + (size_t)slot_of(v + u) * (BM * BN) + roff);
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += x[u];
  }
  for (; v <= v1; ++v) sum += *reinterpret_cast<const f32x4*>(g.workspace + (size_t)slot_of(v) * (BM * BN) + roff);
  if (g.a_colsum && n0 == 0 && part < BM / FIX_THREADS) {
    const int rr = part * FIX_THREADS + t;
    if (m0 + rr < g.M) {
      float bs = 0.f;
      for (int u = v0; u <= v1; ++u) bs += g.workspace[(size_t)2 * p.W * (BM * BN) + (size_t)slot_of(u) * BM + rr];
      g.a_colsum[m0 + rr] = bs;
    }
  }
  const int row = m0 + lrow, col0 = n0 + c4;
  if (row >= g.M || col0 >= g.N) return;
  const float rsv = g.rowscale ? g.rowscale[g.rs_div == 1 ? row : row / g.rs_div] : 1.f;
  const bool full4 = col0 + 3 < g.N;
  if (full4 && !g.bias && !g.colscale && !g.resid && g.act == OFB_ACT_NONE && !g.rowscale && g.alpha == 1.0f && (g.ldc & 3) == 0 &&
      ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0)) {
    *reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + col0) = sum;       // plain weight gradient: one 16-byte store
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int col = col0 + k;
    if (col >= g.N) break;
    const float bias = g.bias ? g.bias[col] : 0.f, cs = g.colscale ? g.colscale[col] : 1.f;
    const float rv = g.resid ? g.resid[(size_t)row * g.ldr + col] : 0.f;
    const float av = (g.act == OFB_ACT_DGELU) ? g.aux[(size_t)row * g.ldaux + col] : 0.f;
    g.C[(size_t)row * g.ldc + col] = epilogue_value(g.alpha, g.act, g.aux, g.ldaux, sum[k], row, col, bias, cs, rsv, rv, av);
  }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int64_t count, float* __restrict__ out,
                                     int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    float s = accumulate ? out[i] : 0.f;
    for (int z = 0; z < splits; ++z) s += ws[(size_t)z * count + i];
    out[i] = s;
  }
}

template <bool A_KC, bool B_KC, bool VEC, bool GUARD>
void launch2(const ofb_gemm_args& g, const Plan& p, bool full, hipStream_t s) {
  const dim3 grid(p.W);
  if (p.full_rounds > 0) {
    if constexpr (!GUARD) {
      // deferred, distributed epilogue: needs 16 K-iterations of the next tile to hide under, and the register room of
      // the row-wise staging (with the split engine only the x @ W^T form keeps pending + live accumulators spill-free)
#ifndef OFB_GEMM_DEFER
#define OFB_GEMM_DEFER 0
#endif
      constexpr bool CAN_DEFER = OFB_GEMM_DEFER && (OFB_GEMM_BF16X6 ? (A_KC && B_KC) : A_KC);
      bool deferred = false;
      if constexpr (CAN_DEFER) {
        if (p.I >= 17) {
          deferred = true;
          if (full) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, VEC, false, true, false, true>), grid, dim3(256), 0, s, g, p);
          else hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, VEC, false, false, false, true>), grid, dim3(256), 0, s, g, p);
        }
      }
      if (!deferred) {
        if (full) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, VEC, false, true, false, false>), grid, dim3(256), 0, s, g, p);
        else hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, VEC, false, false, false, false>), grid, dim3(256), 0, s, g, p);
      }
    } else {
      if (full) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, VEC, GUARD, true, false, false>), grid, dim3(256), 0, s, g, p);
      else hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, VEC, GUARD, false, false, false>), grid, dim3(256), 0, s, g, p);
    }
  }
  if (p.R > 0) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, VEC, GUARD, false, true, false>), grid, dim3(256), 0, s, g, p);
}

template <bool A_KC, bool B_KC>
int launch(const ofb_gemm_args& g, const Plan& p, bool vec, hipStream_t s) {
  const bool full = g.act != OFB_ACT_NONE || g.rowscale || g.resid;
  // unguarded kernels need every tile full: M, N multiples of 128 and K a multiple of the K-step
  const bool guard = !vec || (g.M % BM) || (g.N % BN) || (g.K % BK);
  if (!vec) launch2<A_KC, B_KC, false, true>(g, p, full, s);
  else if (guard) launch2<A_KC, B_KC, true, true>(g, p, full, s);
  else launch2<A_KC, B_KC, true, false>(g, p, full, s);
  return ofb_launch_status();
}

int worker_count() {
  static int W = 0;
  if (W == 0) {
    hipDeviceProp_t prop;
    int devid = 0;
    W = 256 * GEMM_WAVES_PER_SIMD;
    if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess && prop.multiProcessorCount > 0)
      W = prop.multiProcessorCount * GEMM_WAVES_PER_SIMD;
  }
  return W;
}

Plan plan_for(const ofb_gemm_args& g) {
  int W = worker_count();
  const int tiles = ofb_cdiv(g.M, BM) * ofb_cdiv(g.N, BN);
  const long long iters = (long long)tiles * ofb_cdiv(g.K, BK);
  if (iters < W) W = (int)iters;                    // tiny problems: one K-iteration per workgroup
  return make_plan(g.M, g.N, g.K, W);
}

}  // namespace

extern "C" int64_t ofb_gemm_workspace_bytes(const ofb_gemm_args* args) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
  const Plan p = plan_for(*args);
  return p.R ? (int64_t)2 * p.W * (BM * BN + BM) * (int64_t)sizeof(float) : 0;
}

extern "C" int32_t ofb_gemm_is_streamed(const ofb_gemm_args* args) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
  return plan_for(*args).full_rounds == 0 ? 1 : 0;
}

extern "C" int ofb_gemm_f32(const ofb_gemm_args* args, void* stream) {
  if (!args) return OFB_EINVAL;
  const ofb_gemm_args& g = *args;
  if (!g.A || !g.B || !g.C || g.M <= 0 || g.N <= 0 || g.K <= 0) return OFB_EINVAL;
  if (g.a_kc == 0 && g.b_kc == 1) return OFB_ELIMIT;           // A^T * B^T is not on the path
  if (g.kscale && (g.a_kc != 0 || g.ks_div <= 0)) return OFB_EINVAL;
  if (g.rowscale && g.rs_div <= 0) return OFB_EINVAL;
  if (g.act == OFB_ACT_DGELU && !g.aux) return OFB_EINVAL;
  if (g.a_colsum && g.a_kc != 0) return OFB_EINVAL;
  // minimum leading dimensions for the declared storage
  if (g.lda < (g.a_kc ? g.K : g.M) || g.ldb < (g.b_kc ? g.K : g.N) || g.ldc < g.N) return OFB_EINVAL;
  const Plan p = plan_for(g);
  if ((long long)p.W * p.I > 0x7fffffffLL / 2) return OFB_ELIMIT;
  if (g.a_colsum && p.full_rounds > 0) return OFB_ELIMIT;   // fused column sums ride on the streamed tail only (see ofb_gemm_is_streamed)
  if (p.R && (!g.workspace || g.workspace_bytes < ofb_gemm_workspace_bytes(args))) return OFB_EINVAL;
  // vector (16-B) staging needs aligned bases, ld % 4 == 0 and a contiguous extent that is a multiple of 4
  bool vec = ofb_aligned16(g.A) && ofb_aligned16(g.B) && (g.lda % 4 == 0) && (g.ldb % 4 == 0);
  vec = vec && ((g.a_kc ? g.K : g.M) % 4 == 0) && ((g.b_kc ? g.K : g.N) % 4 == 0);
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(0, s, 2.0 * g.M * g.N * (double)g.K);
  int rc;
  if (g.a_kc && g.b_kc) rc = launch<true, true>(g, p, vec, s);
  else if (g.a_kc) rc = launch<true, false>(g, p, vec, s);
  else rc = launch<false, false>(g, p, vec, s);
  if (rc == 0 && p.R) {
    hipLaunchKernelGGL(gemm_fixup_kernel, dim3(p.R, FIX_PARTS), dim3(FIX_THREADS), 0, s, g, p);
    rc = ofb_launch_status();
  }
  ofb_prof_post(0, s);
  return rc;
}

extern "C" int ofb_splitk_reduce(const float* workspace, int32_t splits, int64_t count, float* out, int32_t accumulate,
                                 void* stream) {
  if (!workspace || !out || splits <= 0 || count <= 0) return OFB_EINVAL;
  const int blocks = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, workspace, splits, count, out,
                     accumulate);
  return ofb_launch_status();
}
