// LAB ONLY (not part of libofb_hip.so since round 3): the round-1 in-loop split engine (f32 operands split into three bf16 planes
// while a tile is staged), kept as the A/B reference for the P-format engine (csrc/gemm_p.hip).  Build on its own:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I once-for-both_amd/csrc scripts/lab/gemm_split_engine_r1.hip \
//         once-for-both_amd/csrc/prof.hip -o scripts/lab/bin/libofb_split.so
#include "ofb_common.h"
typedef struct ofb_gemm_args {
  const float* A; const float* B; float* C;
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  int32_t a_kc, b_kc;
  float alpha;
  const float* bias;
  const float* colscale;
  const float* rowscale; int32_t rs_div;
  const float* resid; int32_t ldr;
  float* aux; int32_t ldaux;
  int32_t act;
  const float* kscale; int32_t ks_div;
  float* workspace; int64_t workspace_bytes;
  float* a_colsum;
} ofb_gemm_args;
extern "C" int64_t ofb_gemm_workspace_bytes(const ofb_gemm_args* args);
extern "C" int32_t ofb_gemm_is_streamed(const ofb_gemm_args* args);
extern "C" int ofb_gemm_f32(const ofb_gemm_args* args, void* stream);
// f32 GEMM with fused epilogues on the bf16 matrix pipe at fp32 accuracy.
//
// f32 operands are split EXACTLY into three bf16 values (x = hi + mid + lo, round-to-nearest residual chain) while they are
// staged to LDS, and every K-step issues the six leading product terms (mid*mid, hi*lo, lo*hi, hi*mid, mid*hi, hi*hi; the
// dropped ones are <= 2^-24 relative) on v_mfma_f32_32x32x16_bf16 with f32 accumulation.  Measured error equals that of an f32
// fma chain (1.3e-7 of sum|a*b| at K = 384..1536).  LDS image per operand tile: 3 planes x 2 k-halves x [rows][8 bf16]
// (fragment = one ds_read_b128).
//
// Operands go global -> registers -> LDS with a one-tile prefetch; K-contiguous operands are written row-wise, MN-contiguous
// ones are transposed on the way in, so the three storage combinations the Linear layers need (x@W^T, dY@W, dY^T@X) share one
// inner loop.
//
// Two tile configurations of the same kernel (template parameter TC):
//   T128: 128x128x16 block tile, 4 waves (2x2), wave tile 64x64 (64 accumulator VGPRs), two workgroups per CU;
//   T256: 256x256x16 block tile, 8 waves (2x4), wave tile 128x64 (128 accumulator VGPRs), one workgroup per CU.
// The inner loop is bound by VALU ISSUE, not by the matrix pipe: per wave and K-step T128 carries ~100 VALU (73 of them the
// split, the rest addressing) + 12 ds_read_b128 + 6 ds_write2 + 4 global loads beside 24 MFMAs (768 matrix-pipe cycles against
// ~700 issue cycles of everything else, two waves per SIMD).  T256 stages the same 16 values per thread and K-step but feeds 48
// MFMAs with them, which lifts the main loop from ~160 to ~220 TFLOP/s (lab: scripts/lab/gemm_w_lab.hip); it is used where its
// 256-wide tiles fit the shape (ofb_gemm_f32: choose_tile).
//
// Scheduling (hybrid stream-K): W persistent workgroups (CUs x workgroups per CU).  Output tiles that fill whole rounds of W
// are computed data-parallel with the epilogue fused; the R = tiles mod W remaining tiles are cut along K into W equal runs of
// K-iterations (a run touches at most two tiles), written as raw partial tiles to a workspace and summed by a small fix-up
// kernel that applies the same epilogue.  Every workgroup therefore issues the same number of MFMAs (within one K-iteration)
// whatever the tile count, and weight gradients (few tiles, long K = all tokens) need no separate split-K path.  Partials are
// summed in a fixed order: results are run-to-run deterministic.
#include "ofb_common.h"
#include "gemm_plan.h"
#include <type_traits>

#define BK 16

namespace {

template <int BM_, int BN_, int WM_, int WN_, int WG_PER_CU_>
struct Tile {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, WG_PER_CU = WG_PER_CU_;
  static constexpr int NT = 64 * WM * WN;                       // threads
  static constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN);  // 32x32 blocks per wave
  static constexpr int WPS = WG_PER_CU * WM * WN / 4;           // waves per SIMD (launch bound)
  static constexpr int BLK_A = BM * 16 + 16, PLANE_A = 2 * BLK_A, OPER_A = 3 * PLANE_A;   // +16: the k-halves land on different banks
  static constexpr int BLK_B = BN * 16 + 16, PLANE_B = 2 * BLK_B, OPER_B = 3 * PLANE_B;
  static constexpr int LDS_BYTES = 2 * (OPER_A + OPER_B);
  static_assert(BM * BK / 4 == 2 * NT && BN * BK / 4 == 2 * NT, "the loaders stage exactly two float4 per thread and operand");
};
#ifndef OFB_GEMM_WG_PER_CU
#define OFB_GEMM_WG_PER_CU 2
#endif
using T128 = Tile<128, 128, 2, 2, OFB_GEMM_WG_PER_CU>;
using T256 = Tile<256, 256, 2, 4, 1>;

struct TileRegs { f32x4 v[2]; };

#ifdef OFB_GEMM_STAMPS
// lab only (scripts/lab/stamp_gemm.py): in-kernel cycle stamps of one wave of two workgroups, 5 per K-iteration
__device__ unsigned long long ofb_gemm_stamps[2 * 4096];
#define OFB_STAMP(k) do { if (stamp_on && nst < 4090) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ofb_gemm_stamps[stamp_base + nst++] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define OFB_STAMP(k) do { } while (0)
#endif

// K-contiguous storage X[o*ld + k]: 2 float4 per thread (o = idx>>2, kq = idx&3).
template <int NT, bool VEC, bool GUARD>
__device__ __forceinline__ void load_kc(TileRegs& r, const float* __restrict__ X, int ld, int o0, int O, int k0, int kend,
                                        int t) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = t + NT * i, o = o0 + idx / (BK / 4), k = k0 + ((idx % (BK / 4)) << 2);
    if (!GUARD) {      // full tiles only (M, N multiples of the tile, K multiple of BK): no bounds checks at all
      // wave-uniform base (SGPR pair, advanced per K-step by scalar adds) + a 32-bit per-thread byte offset that is constant
      // for the whole launch: global_load ... v_off, s[base] - no vector address arithmetic in the K loop
      const char* ub = reinterpret_cast<const char*>(X + (size_t)o0 * ld + k0);
      const unsigned voff = (unsigned)((idx / (BK / 4)) * ld + ((idx % (BK / 4)) << 2)) * 4u;
      r.v[i] = *reinterpret_cast<const f32x4*>(ub + voff);
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
// (tried: the residual of a packed half in ONE instruction, v_dot2_f32_bf16(pair, (-1, 0) | (0, -1), x) - exact, 7 instead of 9
// VALU per pair.  Through __builtin_amdgcn_fdot2_f32_bf16 hipcc selects the destructive v_dot2c form and adds a copy per value, so
// the count does not drop; as inline asm the compiler no longer sees that a DOT result may not be read by another VALU instruction
// for 3 wait states (the hardware does not interlock it: garbage).  scripts/lab/dot2_check.hip)
// K-contiguous tile: thread item (row = idx/4, k = 4*(idx%4) .. +3) -> 4 bf16 (8 B) per plane at [k>>3][row][k&7]
template <int NT, int BLK, int PLANE>
__device__ __forceinline__ void store_kc(const TileRegs& r, char* __restrict__ base, int t) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = t + NT * i, row = idx >> 2, kq = idx & 3;
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair(r.v[i][0], r.v[i][1], h0, m0, l0);
    split_pair(r.v[i][2], r.v[i][3], h1, m1, l1);
    char* p = base + (kq >> 1) * BLK + row * 16 + (kq & 1) * 8;
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(p + 2 * PLANE) = make_uint2(l0, l1);
  }
}
// MN-contiguous storage X[k*ld + o]: thread t takes rows o = 4*(t/8) .. +3 at the k PAIR (2*(t%8), 2*(t%8)+1): v[0] = even k,
// v[1] = odd k, so that each output row's two values pack into one bf16x2 LDS word per plane (the transpose costs
// 12 ds_write_b32 per thread and tile instead of 24 ds_write_b16).  NT threads cover NT/2 rows.
template <bool VEC, bool GUARD>
__device__ __forceinline__ void load_mc(TileRegs& r, const float* __restrict__ X, int ld, int o0, int O, int k0, int kend,
                                        int t, const float* __restrict__ kscale, int ks_div) {
  const int o = o0 + ((t >> 3) << 2);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = k0 + 2 * (t & 7) + i;
    if (!GUARD) {
      const char* ub = reinterpret_cast<const char*>(X + (size_t)k0 * ld + o0);
      const unsigned voff = (unsigned)((2 * (t & 7) + i) * ld + ((t >> 3) << 2)) * 4u;
      f32x4 u = *reinterpret_cast<const f32x4*>(ub + voff);
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
template <int BLK, int PLANE>
__device__ __forceinline__ void store_mc(const TileRegs& r, char* __restrict__ lds_oper, int t) {
  const int kp = t & 7, row0 = (t >> 3) << 2;
  char* base = lds_oper + (kp >> 2) * BLK + (kp & 3) * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned h, m, l;
    split_pair(r.v[0][j], r.v[1][j], h, m, l);
    char* p = base + (row0 + j) * 16;
    *reinterpret_cast<unsigned*>(p) = h;
    *reinterpret_cast<unsigned*>(p + PLANE) = m;
    *reinterpret_cast<unsigned*>(p + 2 * PLANE) = l;
  }
}

using ofb_plan::Plan; using ofb_plan::make_plan; using ofb_plan::tile_coord; using ofb_plan::Seg; using ofb_plan::get_seg;

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
template <class TC, bool A_KC, bool B_KC, bool VEC, bool GUARD, bool FULL_EPI, bool TAIL>
__global__ __launch_bounds__(TC::NT, TC::WPS) void gemm_f32_kernel(const ofb_gemm_args g, const Plan p) {
  constexpr int BM = TC::BM, BN = TC::BN, MI = TC::MI, NI = TC::NI, NT = TC::NT;
  __shared__ __attribute__((aligned(16))) char lds[TC::LDS_BYTES];
  char* const As = lds;                       // [2 buffers] of one staged A tile
  char* const Bs = lds + 2 * TC::OPER_A;      // [2 buffers] of one staged B tile

  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
  const int v = ofb_xcd_remap(blockIdx.x, p.W);     // consecutive v share an XCD (and thus A/B panels in its L2)
  const int wm0 = (w / TC::WN) * (32 * MI), wn0 = (w % TC::WN) * (32 * NI);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileRegs ra, rb;
  auto gload = [&](const Seg& sg, int it) __attribute__((always_inline)) {
    const int k0 = it * BK;
    if (A_KC) load_kc<NT, VEC, GUARD>(ra, g.A, g.lda, sg.m0, g.M, k0, g.K, t);
    else load_mc<VEC, GUARD>(ra, g.A, g.lda, sg.m0, g.M, k0, g.K, t, g.kscale, g.ks_div);
    if (B_KC) load_kc<NT, VEC, GUARD>(rb, g.B, g.ldb, sg.n0, g.N, k0, g.K, t);
    else load_mc<VEC, GUARD>(rb, g.B, g.ldb, sg.n0, g.N, k0, g.K, t, nullptr, 1);
  };
  // fused bias gradient (weight-gradient launches): column sums of the stored A (= dY) for the workgroups that own the first
  // column tile, added up from the staging registers: rows 4*(t/8) .. +3, this thread's k pairs; reduced over the 8 pair-lanes
  // at unit end
  f32x4 bsum4 = {0.f, 0.f, 0.f, 0.f};
  auto lstore = [&](int buf, int tile_n0) __attribute__((always_inline)) {
    if (TAIL && !A_KC && g.a_colsum && tile_n0 == 0) bsum4 += ra.v[0] + ra.v[1];
    if (A_KC) store_kc<NT, TC::BLK_A, TC::PLANE_A>(ra, As + buf * TC::OPER_A, t);
    else store_mc<TC::BLK_A, TC::PLANE_A>(ra, As + buf * TC::OPER_A, t);
    if (B_KC) store_kc<NT, TC::BLK_B, TC::PLANE_B>(rb, Bs + buf * TC::OPER_B, t);
    else store_mc<TC::BLK_B, TC::PLANE_B>(rb, Bs + buf * TC::OPER_B, t);
  };
  // MA x NA = the 32x32 blocks of this wave's tile that hold any valid output (guarded launches on ragged shapes: the valid region
  // of an edge tile is a prefix of rows / columns, and multiplying padding is pure wasted energy on a power-limited chip)
  auto mma_blocks = [&](int buf, auto MAc, auto NAc) __attribute__((always_inline)) {
    constexpr int MA = decltype(MAc)::value, NA = decltype(NAc)::value;
    // lane (row l31, k-half h) reads its 8 bf16 of each plane with one b128; A and B share the k <-> (half, j) map.
    const char* a_s = As + buf * TC::OPER_A + h * TC::BLK_A + (wm0 + l31) * 16;
    const char* b_s = Bs + buf * TC::OPER_B + h * TC::BLK_B + (wn0 + l31) * 16;
    ofb_bf16x8 af[MA][3], bf[NA][3];
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) af[i][pl] = *reinterpret_cast<const ofb_bf16x8*>(a_s + pl * TC::PLANE_A + i * 32 * 16);
#pragma unroll
    for (int j = 0; j < NA; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bf[j][pl] = *reinterpret_cast<const ofb_bf16x8*>(b_s + pl * TC::PLANE_B + j * 32 * 16);
    // six product terms, smallest first: (mid,mid) (hi,lo) (lo,hi) (hi,mid) (mid,hi) (hi,hi)
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int j = 0; j < NA; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][TA[q]], bf[j][TB[q]], acc[i][j], 0, 0, 0);
  };
  int ma_act = MI, na_act = NI;         // active blocks of the current unit (set per unit in guarded T128 launches)
  auto set_active = [&](const Seg& sg) __attribute__((always_inline)) {
    if constexpr (GUARD && MI == 2 && NI == 2) {
      const int rows = g.M - sg.m0 - wm0, cols = g.N - sg.n0 - wn0;       // valid rows / columns from this wave's corner on
      ma_act = rows <= 0 ? 0 : (rows <= 32 ? 1 : 2);
      na_act = cols <= 0 ? 0 : (cols <= 32 ? 1 : 2);
    }
  };
  auto compute = [&](int buf) __attribute__((always_inline)) {
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    if constexpr (GUARD && MI == 2 && NI == 2) {
      if (ma_act == 2 && na_act == 2) mma_blocks(buf, C2{}, C2{});
      else if (ma_act == 0 || na_act == 0) { }
      else if (ma_act == 2) mma_blocks(buf, C2{}, C1{});
      else if (na_act == 2) mma_blocks(buf, C1{}, C2{});
      else mma_blocks(buf, C1{}, C1{});
    } else {
      mma_blocks(buf, std::integral_constant<int, MI>{}, std::integral_constant<int, NI>{});
    }
  };

#ifdef OFB_GEMM_STAMPS
  const bool stamp_on = (blockIdx.x == 8 || blockIdx.x == 8 + 256) && t == 0 && !TAIL;
  const int stamp_base = (blockIdx.x == 8) ? 0 : 4096;
  int nst = 0;
#endif
  int sidx = 0;
  Seg cur = get_seg<TAIL>(p, v, 0);
  if (!cur.ok) return;
  set_active(cur);
  gload(cur, cur.it0);
  lstore(0, cur.n0);
  __syncthreads();
  int buf = 0;

  while (true) {
    // all K-iterations of this unit but the last: prefetch the next K-tile of the same unit
    for (int it = cur.it0; it + 1 < cur.it1; ++it) {
      OFB_STAMP(0);
      gload(cur, it + 1);
      // pin the prefetch ahead of the MFMA block: left alone, hipcc lets the loaded tile share VGPRs with the fragments and
      // sinks the global loads behind the last MFMAs, exposing their whole latency in front of the LDS refill
      __builtin_amdgcn_sched_barrier(0);
      OFB_STAMP(1);
      compute(buf);
      // keep every MFMA of this K-tile ahead of the vmcnt wait / LDS refill / barrier
      __builtin_amdgcn_sched_barrier(0);
      OFB_STAMP(2);
#ifdef OFB_GEMM_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      OFB_STAMP(3);
#endif
      lstore(buf ^ 1, cur.n0);
      OFB_STAMP(4);
      __syncthreads();
      buf ^= 1;
    }
    // last K-iteration: the NEXT unit's first K-tile goes in flight before this unit's stores
    const Seg nxt = get_seg<TAIL>(p, v, sidx + 1);
    const bool has_next = nxt.ok;
    if (has_next) gload(nxt, nxt.it0);
    __builtin_amdgcn_sched_barrier(0);
    compute(buf);
    __builtin_amdgcn_sched_barrier(0);
    if (TAIL) {
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
      // raw partial tile -> workspace[slot][BM][BN] (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
      float* ws = g.workspace + (size_t)cur.slot * (BM * BN);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            ws[(wm0 + 32 * mi + 4 * h + (r & 3) + 8 * (r >> 2)) * BN + wn0 + 32 * ni + l31] = acc[mi][ni][r];
            acc[mi][ni][r] = 0.f;
          }
        }
    } else if (!GUARD) {
      // unguarded build (every tile is full): no per-element guards, so loads / stores issue back to back behind ONE wait
      // (hipcc otherwise brackets every guarded store with s_waitcnt vmcnt(0), serialising 64 round trips per wave)
      float biasv[NI], csv[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = cur.n0 + wn0 + 32 * ni + l31;
        biasv[ni] = g.bias ? g.bias[col] : 0.f;
        csv[ni] = g.colscale ? g.colscale[col] : 1.f;
      }
      if (FULL_EPI) {
        // side inputs (residual / saved pre-activation / per-row scale) of a whole 32-row band (NI blocks, 16 values per lane
        // each) are requested one band ahead of the band being finished, so ~32 loads per lane are in flight instead of 4: the
        // epilogue is bound by memory latency, not bandwidth.  `side` carries the saved pre-activation for the dGELU form and the
        // residual otherwise (a launch that wants both reads its residual inside `finish`).
        const bool dg = g.act == OFB_ACT_DGELU;
        const float* sp = dg ? g.aux : g.resid;
        const int lds_ = dg ? g.ldaux : g.ldr;
        if (!sp && !g.rowscale) {
          // nothing to fetch (bias / gate / GELU forms): finish and stream out, no staging, no waits
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              const int col = cur.n0 + wn0 + 32 * ni + l31, rbase = cur.m0 + wm0 + 32 * mi + 4 * h;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                g.C[(size_t)row * g.ldc + col] = epilogue_value(g.alpha, g.act, g.aux, g.ldaux, acc[mi][ni][r], row, col, biasv[ni], csv[ni],
                                                                1.f, 0.f, 0.f);
                acc[mi][ni][r] = 0.f;
              }
            }
        } else {
          f32x16 side[2][NI], rsv[2];
          auto request = [&](auto Mi) __attribute__((always_inline)) {
            constexpr int mi = decltype(Mi)::value;
            const int rbase = cur.m0 + wm0 + 32 * mi + 4 * h;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              const int col = cur.n0 + wn0 + 32 * ni + l31;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                side[mi & 1][ni][r] = sp ? sp[(size_t)row * lds_ + col] : 0.f;
                if (ni == 0) rsv[mi & 1][r] = g.rowscale ? g.rowscale[g.rs_div == 1 ? row : row / g.rs_div] : 1.f;
              }
            }
          };
          auto finish = [&](auto Mi) __attribute__((always_inline)) {
            constexpr int mi = decltype(Mi)::value;
            const int rbase = cur.m0 + wm0 + 32 * mi + 4 * h;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              const int col = cur.n0 + wn0 + 32 * ni + l31;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                const float rvv = dg ? (g.resid ? g.resid[(size_t)row * g.ldr + col] : 0.f) : side[mi & 1][ni][r];
                g.C[(size_t)row * g.ldc + col] = epilogue_value(g.alpha, g.act, g.aux, g.ldaux, acc[mi][ni][r], row, col, biasv[ni], csv[ni],
                                                                rsv[mi & 1][r], rvv, dg ? side[mi & 1][ni][r] : 0.f);
                acc[mi][ni][r] = 0.f;
              }
            }
          };
          using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
          using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
          static_assert(MI == 2 || MI == 4, "epilogue bands are unrolled for 2 or 4 row blocks per wave");
          request(I0{}); request(I1{});
          __builtin_amdgcn_sched_barrier(0);
          finish(I0{});
          if constexpr (MI == 4) {
            request(I2{});
            __builtin_amdgcn_sched_barrier(0);
            finish(I1{});
            request(I3{});
            __builtin_amdgcn_sched_barrier(0);
            finish(I2{}); finish(I3{});
          } else {
            finish(I1{});
          }
        }
      } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int col = cur.n0 + wn0 + 32 * ni + l31;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
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
      for (int ni = 0; ni < NI; ++ni) {
        const int col = cur.n0 + wn0 + 32 * ni + l31;
        const bool colok = col < g.N;
        float bias = 0.f, cs = 1.f;
        if (colok) {
          if (g.bias) bias = g.bias[col];
          if (g.colscale) cs = g.colscale[col];
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
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
    set_active(cur);
  }
}

// Sums the partial tiles of each streamed tail tile in workgroup order and applies the epilogue.
// grid (R, BM / 4), BN threads: block (r, part) handles rows [4*part, 4*part+4) of tail tile r, one float4 (4 columns) per
// thread; 16-byte loads, eight contributors per trip.  The sums stay in contributor order (deterministic).
#define FIX_ROWS 4
template <class TC>
__global__ __launch_bounds__(TC::BN) void gemm_fixup_kernel(const ofb_gemm_args g, const Plan p) {
  constexpr int BM = TC::BM, BN = TC::BN, FIX_THREADS = FIX_ROWS * BN / 4;
  static_assert(FIX_THREADS == BN && BM <= BN, "one block per 4 rows; the first block also sums the BM fused bias partials");
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
  if (g.a_colsum && n0 == 0 && part == 0 && t < BM && m0 + t < g.M) {
    float bs = 0.f;
    for (int u = v0; u <= v1; ++u) bs += g.workspace[(size_t)2 * p.W * (BM * BN) + (size_t)slot_of(u) * BM + t];
    g.a_colsum[m0 + t] = bs;
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

template <class TC, bool A_KC, bool B_KC, bool VEC, bool GUARD>
void launch2(const ofb_gemm_args& g, const Plan& p, bool full, hipStream_t s) {
  const dim3 grid(p.W), block(TC::NT);
  if (p.full_rounds > 0) {
    if (full) hipLaunchKernelGGL((gemm_f32_kernel<TC, A_KC, B_KC, VEC, GUARD, true, false>), grid, block, 0, s, g, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<TC, A_KC, B_KC, VEC, GUARD, false, false>), grid, block, 0, s, g, p);
  }
  if (p.R > 0) hipLaunchKernelGGL((gemm_f32_kernel<TC, A_KC, B_KC, VEC, GUARD, false, true>), grid, block, 0, s, g, p);
}

template <class TC, bool A_KC, bool B_KC>
int launch(const ofb_gemm_args& g, const Plan& p, bool vec, hipStream_t s) {
  const bool full = g.act != OFB_ACT_NONE || g.rowscale || g.resid;
  // unguarded kernels need every tile full: M, N multiples of the tile and K a multiple of the K-step
  const bool guard = !vec || (g.M % TC::BM) || (g.N % TC::BN) || (g.K % BK);
  if (!vec) launch2<TC, A_KC, B_KC, false, true>(g, p, full, s);
  else if (guard) launch2<TC, A_KC, B_KC, true, true>(g, p, full, s);
  else launch2<TC, A_KC, B_KC, true, false>(g, p, full, s);
  return ofb_launch_status();
}

int cu_count() {
  static int n = 0;
  if (n == 0) {
    hipDeviceProp_t prop;
    int devid = 0;
    n = 256;
    if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess && prop.multiProcessorCount > 0)
      n = prop.multiProcessorCount;
  }
  return n;
}

// Tile choice.  T256 feeds twice the MFMAs per staged value, and its bare main loop is 1.35x faster at long K with an even tile
// count (lab: 223 vs ~165 TFLOP/s at 32768 x 2048 x 2048), but on the step's shapes it LOSES (same-box A/B, DeiT-S 35.1 vs 31.5
// ms/step, DeiT-B 54.2 vs 53.3): K is 384..1536, so a 256x256 tile lives for only 24..96 K-steps and its prologue / epilogue are
// not hidden by a second workgroup on the CU, its stream-K partial tiles are 256 KB each, and 594 tiles over 256 workers round
// badly.  The default build (OFB_GEMM_TILE 0) therefore uses T256 for pad-free weight gradients only (see use_t256);
// -DOFB_GEMM_TILE=128 / 256 force one configuration, -DOFB_GEMM_TILE=1 enables the wider per-shape choice below.
#ifndef OFB_GEMM_TILE
#define OFB_GEMM_TILE 0
#endif
bool use_t256(const ofb_gemm_args& g) {
#if OFB_GEMM_TILE == 0
  // default: weight gradients (dY^T X: a handful of output tiles, K = all tokens) whose output is a whole number of 256x256 tiles.
  // Their K loop is thousands of iterations long, which is where T256's leaner main loop pays: DeiT-B bs 64 dW1 (3072 x 768 x 12608)
  // 354 vs 397 us.  DeiT-S never qualifies (384 = 1.5 tiles).
  return !g.a_kc && !g.b_kc && (g.M % 256 == 0) && (g.N % 256 == 0);
#elif OFB_GEMM_TILE == 128
  return false;
#elif OFB_GEMM_TILE == 256
  return true;
#else
  if (!g.a_kc) return false;
  const long long mp = ofb_cdiv(g.M, 256) * 256LL, np = ofb_cdiv(g.N, 256) * 256LL;
  if (mp * np * 8 > (long long)g.M * g.N * 9) return false;                 // more than 1/8 of the MFMA work would be padding
  return (mp / 256) * (np / 256) >= cu_count();
#endif
}

template <class TC>
Plan plan_for(const ofb_gemm_args& g) {
  int W = cu_count() * TC::WG_PER_CU;
  const int tiles = ofb_cdiv(g.M, TC::BM) * ofb_cdiv(g.N, TC::BN);
  const long long iters = (long long)tiles * ofb_cdiv(g.K, BK);
  if (iters < W) W = (int)iters;                    // tiny problems: one K-iteration per workgroup
  return make_plan(g.M, g.N, g.K, W, TC::BM, TC::BN);
}
Plan plan_any(const ofb_gemm_args& g, bool& t256) {
  t256 = use_t256(g);
#if OFB_GEMM_TILE == 128
  return plan_for<T128>(g);
#else
  return t256 ? plan_for<T256>(g) : plan_for<T128>(g);
#endif
}

// T256 for the weight-gradient storage form only (the default build does not instantiate its other forms)
int run_t256_wgrad(const ofb_gemm_args& g, const Plan& p, bool vec, hipStream_t s);

template <class TC>
int run(const ofb_gemm_args& g, const Plan& p, bool vec, hipStream_t s) {
  int rc;
  if (g.a_kc && g.b_kc) rc = launch<TC, true, true>(g, p, vec, s);
  else if (g.a_kc) rc = launch<TC, true, false>(g, p, vec, s);
  else rc = launch<TC, false, false>(g, p, vec, s);
  if (rc == 0 && p.R) {
    hipLaunchKernelGGL(gemm_fixup_kernel<TC>, dim3(p.R, TC::BM / FIX_ROWS), dim3(TC::BN), 0, s, g, p);
    rc = ofb_launch_status();
  }
  return rc;
}

int run_t256_wgrad(const ofb_gemm_args& g, const Plan& p, bool vec, hipStream_t s) {
  int rc = launch<T256, false, false>(g, p, vec, s);
  if (rc == 0 && p.R) {
    hipLaunchKernelGGL(gemm_fixup_kernel<T256>, dim3(p.R, T256::BM / FIX_ROWS), dim3(T256::BN), 0, s, g, p);
    rc = ofb_launch_status();
  }
  return rc;
}

}  // namespace

extern "C" int64_t ofb_gemm_workspace_bytes(const ofb_gemm_args* args) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
  bool t256;
  const Plan p = plan_any(*args, t256);
  return p.R ? (int64_t)2 * p.W * ((int64_t)p.bm * p.bn + p.bm) * (int64_t)sizeof(float) : 0;
}

extern "C" int32_t ofb_gemm_is_streamed(const ofb_gemm_args* args) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
  bool t256;
  return plan_any(*args, t256).full_rounds == 0 ? 1 : 0;
}

extern "C" int ofb_gemm_f32(const ofb_gemm_args* args, void* stream) {
  if (!args) return OFB_EINVAL;
  const ofb_gemm_args& g = *args;
  if (!g.A || !g.B || !g.C || g.M <= 0 || g.N <= 0 || g.K <= 0) return OFB_EINVAL;
  if (g.a_kc == 0 && g.b_kc == 1) return OFB_ELIMIT;           // A^T * B^T is not on the path
  if (g.kscale && (g.a_kc != 0 || g.ks_div <= 0)) return OFB_EINVAL;
  if (g.rowscale && g.rs_div <= 0) return OFB_EINVAL;
  if (g.act < OFB_ACT_NONE || g.act > OFB_ACT_DGELU) return OFB_EINVAL;   // the save-derivative forms exist in ofb_gemm_p only
  if (g.act == OFB_ACT_DGELU && !g.aux) return OFB_EINVAL;
  if (g.a_colsum && g.a_kc != 0) return OFB_EINVAL;
  // minimum leading dimensions for the declared storage
  if (g.lda < (g.a_kc ? g.K : g.M) || g.ldb < (g.b_kc ? g.K : g.N) || g.ldc < g.N) return OFB_EINVAL;
  bool t256;
  const Plan p = plan_any(g, t256);
  if ((long long)p.W * p.I > 0x7fffffffLL / 2) return OFB_ELIMIT;
  if (g.a_colsum && p.full_rounds > 0) return OFB_ELIMIT;   // fused column sums ride on the streamed tail only (see ofb_gemm_is_streamed)
  if (p.R && (!g.workspace || g.workspace_bytes < ofb_gemm_workspace_bytes(args))) return OFB_EINVAL;
  // vector (16-B) staging needs aligned bases, ld % 4 == 0 and a contiguous extent that is a multiple of 4
  bool vec = ofb_aligned16(g.A) && ofb_aligned16(g.B) && (g.lda % 4 == 0) && (g.ldb % 4 == 0);
  vec = vec && ((g.a_kc ? g.K : g.M) % 4 == 0) && ((g.b_kc ? g.K : g.N) % 4 == 0);
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(0, s, 2.0 * g.M * g.N * (double)g.K);
#if OFB_GEMM_TILE == 128
  const int rc = run<T128>(g, p, vec, s);
#elif OFB_GEMM_TILE == 0
  const int rc = t256 ? run_t256_wgrad(g, p, vec, s) : run<T128>(g, p, vec, s);
#else
  const int rc = t256 ? run<T256>(g, p, vec, s) : run<T128>(g, p, vec, s);
#endif
  ofb_prof_post(0, s);
  return rc;
}

#ifdef OFB_GEMM_STAMPS
extern "C" int ofb_diag_gemm_stamps(unsigned long long* out_host) {      /* lab only, not part of the ABI */
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(ofb_gemm_stamps), sizeof(unsigned long long) * 2 * 4096);
}
#endif

