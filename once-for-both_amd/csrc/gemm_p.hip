// f32-accurate GEMM on operands that are ALREADY split into three bf16 planes ("P-format"), staged by LDS-DMA.
//
// Why: the exact 3-way split x = hi + mid + lo that lets the bf16 matrix pipe deliver fp32 accuracy (six MFMA terms per product,
// gemm.hip) costs ~9 VALU per value pair.  Done inside the K loop it is repeated for every tile that touches an operand
// (activations 3-12x, weights ~200x) and it is what bounds that loop (VALU issue + VGPR->LDS stores beside the MFMAs, on a
// power-limited chip).  Here every producer on the path (LayerNorm, GEMM epilogues, attention, the optimizer for weights) emits
// the planes ONCE, and the GEMM's K loop holds nothing but LDS-DMA, fragment reads and MFMAs.
//
// P-format of a matrix X[R][C]  (ofb_hip.h: ofb_pformat_bytes / ofb_to_pformat):
//   granules of 4 rows x 16 columns, 384 B each, stored [ceil(R/16)*4][ceil(C/16)]; inside a granule
//   [plane hi | mid | lo][c % 16][r % 4] bf16 (128 B per plane).  Rows >= R and columns >= C of the last granules hold zeros.
// One layout serves both consumers of an activation / weight:
//   mode KC (reduction along C, the MFMA row/column index is R): fragment = two ds_read_b64_tr_b16 (the hardware 4x16 transpose)
//   mode KR (reduction along R, the MFMA row/column index is C): fragment = two ds_read_b64 (4 consecutive r are contiguous)
// and a 32x32 accumulator block (lane = column, 4 consecutive rows per register group) stores P-format with 8-byte stores that
// fill whole 128-B lines, so a GEMM epilogue can feed the next GEMM directly.
//
// Kernel: 256x256 tile, 8 waves (2 x 4), wave tile 128x64 = 4x2 blocks of v_mfma_f32_32x32x16_bf16, K step 16 (48 MFMAs per
// wave), three 48-KB LDS stages filled by global_load_lds_dwordx4 three steps ahead (inline asm: through the builtin hipcc
// drains every LDS-DMA with vmcnt(0) before the next ds_read), one barrier per K step placed BETWEEN the two halves of the step
// with the fragment reads of the next half-step issued in the MFMA gaps of the current one.  Scheduling is the hybrid stream-K
// of gemm_plan.h (data-parallel rounds + K-split tail + deterministic fix-up).
#include "ofb_common.h"
#include "gemm_plan.h"
#include <type_traits>

namespace {

using ofb_plan::Plan; using ofb_plan::make_plan; using ofb_plan::tile_coord; using ofb_plan::Seg; using ofb_plan::get_seg;

typedef __bf16 pbf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef short ps16x4 __attribute__((ext_vector_type(4)));
typedef short ps16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x2 __attribute__((ext_vector_type(2)));
#define OFB_LDSP(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int BM = 256, BN = 256, WN = 4, MI = 4, NI = 2, NT = 512, NST = 3;
constexpr int A_BYTES = BM * 16 * 6, B_BYTES = BN * 16 * 6, STAGE = A_BYTES + B_BYTES;       // 24 KB + 24 KB per K16 step
constexpr int GRAN = 384;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {          // v_cvt_pk_bf16_f32: a -> low half, b -> high half (RNE)
  pf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pbf16x2));
}
// x = hi + mid + lo, each a bf16 (24 significant bits in total: exact for finite f32 in the normal range); two values at a time
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = pk_bf16(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  mid = pk_bf16(ra, rb);
  lo = pk_bf16(ra - __uint_as_float(mid << 16), rb - __uint_as_float(mid & 0xffff0000u));
}
// four consecutive rows of one column -> the 8-byte column slot of each plane slab
__device__ __forceinline__ void store_p4(char* slot, float v0, float v1, float v2, float v3) {
  unsigned h0, m0, l0, h1, m1, l1;
  split_pair(v0, v1, h0, m0, l0);
  split_pair(v2, v3, h1, m1, l1);
  *reinterpret_cast<uint2*>(slot) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(slot + 128) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(slot + 256) = make_uint2(l0, l1);
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// ---- f32 <-> P-format ---------------------------------------------------------------------------------------------
// X[R][C] row-major (ld) -> P; value = X * rowscale[r / rs_div] (optional); zero padding up to (Rp, Cp)
__global__ void to_pformat_kernel(const float* __restrict__ X, int R, int C, int ld, char* __restrict__ P, int ncb,
                                  const float* __restrict__ rowscale, int rs_div) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, rg = blockIdx.y;
  if (c >= ncb * 16) return;
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = 4 * rg + t;
    float x = (r < R && c < C) ? X[(size_t)r * ld + c] : 0.f;
    if (rowscale && r < R) x *= rowscale[rs_div == 1 ? r : r / rs_div];
    v[t] = x;
  }
  store_p4(P + ((size_t)rg * ncb + (c >> 4)) * GRAN + (c & 15) * 8, v[0], v[1], v[2], v[3]);
}
// P -> X[R][C] (hi + mid + lo is exact)
__global__ void from_pformat_kernel(const char* __restrict__ P, int ncb, float* __restrict__ X, int R, int C, int ld) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, rg = blockIdx.y;
  if (c >= C) return;
  const char* slot = P + ((size_t)rg * ncb + (c >> 4)) * GRAN + (c & 15) * 8;
  const uint2 h = *reinterpret_cast<const uint2*>(slot), m = *reinterpret_cast<const uint2*>(slot + 128),
              l = *reinterpret_cast<const uint2*>(slot + 256);
  const float v[4] = {bf16_lo(h.x) + (bf16_lo(m.x) + bf16_lo(l.x)), bf16_hi(h.x) + (bf16_hi(m.x) + bf16_hi(l.x)),
                      bf16_lo(h.y) + (bf16_lo(m.y) + bf16_lo(l.y)), bf16_hi(h.y) + (bf16_hi(m.y) + bf16_hi(l.y))};
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (4 * rg + t < R) X[(size_t)(4 * rg + t) * ld + c] = v[t];
}

// ---- the GEMM -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz(int tg) { return ((tg >> 1) & 3) << 1; }

// v = alpha*acc (+bias)(*colscale); act; (*rowscale); (+resid)   -- shared by the fused epilogue and the fix-up kernel
__device__ __forceinline__ float epi_value(const ofb_gemm_p_args& g, float accv, int row, int col, float bias, float cs, bool ok) {
  float v = (accv * g.alpha + bias) * cs;
  if (g.act == OFB_ACT_GELU) {
    if (g.aux && ok) g.aux[(size_t)row * g.ldaux + col] = v;
    v = ofb_gelu(v);
  } else if (g.act == OFB_ACT_DGELU) {
    v *= ofb_dgelu(ok ? g.aux[(size_t)row * g.ldaux + col] : 0.f);
  }
  if (g.rowscale) v *= ok ? g.rowscale[g.rs_div == 1 ? row : row / g.rs_div] : 1.f;
  if (g.resid) v += ok ? g.resid[(size_t)row * g.ldr + col] : 0.f;
  return v;
}

template <bool A_KC, bool B_KC, bool TAIL>
__global__ __launch_bounds__(NT, 2) void gemm_p_kernel(const ofb_gemm_p_args g, const Plan p) {
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE];
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, h = lane >> 5;
  const int wm0 = (w / WN) * (32 * MI), wn0 = (w % WN) * (32 * NI);
  const int a_ncb = g.a_ncb, b_ncb = g.b_ncb;

  // LDS-DMA source offsets (bytes, relative to the tile / stage base) of this wave's three 1-KB pieces per operand
  //   KC: piece = (block j of 32 rows, plane): lane -> granule tg = l>>3 of the block, 16-B chunk cp = (l&7) ^ swz(tg); the XOR on
  //       the SOURCE chunk (LDS stays lane-linear) makes the transposed fragment reads bank-conflict free
  //   KR: the stage image is a linear copy of [4 row groups][BN/16 granules], cut into 1-KB pieces
  unsigned a_off[3], b_off[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int piece = w + 8 * q;
    const int j = piece / 3, pl = piece % 3, tg = lane >> 3, cp = (lane & 7) ^ swz(tg);
    const int bl = piece * 1024 + lane * 16;
    a_off[q] = A_KC ? (unsigned)(((8 * j + tg) * a_ncb) * GRAN + pl * 128 + cp * 16)
                    : (unsigned)(((bl / (BM / 16 * GRAN)) * a_ncb) * GRAN + bl % (BM / 16 * GRAN));
    b_off[q] = B_KC ? (unsigned)(((8 * j + tg) * b_ncb) * GRAN + pl * 128 + cp * 16)
                    : (unsigned)(((bl / (BN / 16 * GRAN)) * b_ncb) * GRAN + bl % (BN / 16 * GRAN));
  }
  // fragment read offsets (bytes inside an operand's stage image), two 8-byte reads per fragment
  int a_r0, a_r1, b_r0, b_r1;
  {
    const int gq = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, hh = gq >> 1, tg = 4 * (gq & 1) + pp;
    const int c0 = 8 * hh + q, c1 = c0 + 4;
    const int kc0 = (8 * tg + ((c0 >> 1) ^ swz(tg))) * 16 + (c0 & 1) * 8, kc1 = (8 * tg + ((c1 >> 1) ^ swz(tg))) * 16 + (c1 & 1) * 8;
    const int kr_a0 = ((2 * h) * (BM / 16) + (l31 >> 4)) * GRAN + (l31 & 15) * 8, kr_a1 = kr_a0 + (BM / 16) * GRAN;
    const int kr_b0 = ((2 * h) * (BN / 16) + (l31 >> 4)) * GRAN + (l31 & 15) * 8, kr_b1 = kr_b0 + (BN / 16) * GRAN;
    a_r0 = A_KC ? kc0 : kr_a0; a_r1 = A_KC ? kc1 : kr_a1;
    b_r0 = B_KC ? kc0 : kr_b0; b_r1 = B_KC ? kc1 : kr_b1;
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // six LDS-DMA per wave and stage, invisible to hipcc's wait-count bookkeeping: counted by hand below (vmcnt(12/6/0)).
  // Pieces of one wave sit 8 KB apart in the stage image (A: w, w+8, w+16; B follows A at +24 KB = 3 x 8 KB).
  const unsigned lds0 = (unsigned)(size_t)OFB_LDSP(lds) + (unsigned)w * 1024u;
  auto issue = [&](int buf, const char* a_src, const char* b_src) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %8\n\t"
        "s_add_u32 m0, m0, 0x2000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %8\n\t"
        "s_add_u32 m0, m0, 0x2000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %8\n\t"
        "s_add_u32 m0, m0, 0x2000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, %9\n\t"
        "s_add_u32 m0, m0, 0x2000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, %9\n\t"
        "s_add_u32 m0, m0, 0x2000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, %9\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds0 + (unsigned)buf * (unsigned)STAGE), "v"(a_off[0]), "v"(a_off[1]), "v"(a_off[2]), "v"(b_off[0]), "v"(b_off[1]),
          "v"(b_off[2]), "s"(a_src), "s"(b_src)
        : "memory", "scc");
  };
  auto frag = [&](const char* base, int r0, int r1, bool kc, int blk, int plane) __attribute__((always_inline)) -> pbf16x8 {
    ps16x4 lo4, hi4;
    if (kc) {
      const char* q = base + (blk * 3 + plane) * 1024;
      lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4*)(q + r0));
      hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4*)(q + r1));
    } else {
      const char* q = base + (2 * blk) * GRAN + plane * 128;
      lo4 = *reinterpret_cast<const ps16x4*>(q + r0);
      hi4 = *reinterpret_cast<const ps16x4*>(q + r1);
    }
    ps16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(pbf16x8, v);
  };
  pbf16x8 a01[2][3], a23[2][3], bb[2][NI][3];
  auto rdA = [&](pbf16x8 (&dst)[2][3], int buf, int blk0) __attribute__((always_inline)) {
    const char* la = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[i][pl] = frag(la, a_r0, a_r1, A_KC, (wm0 >> 5) + blk0 + i, pl);
  };
  auto rdB = [&](pbf16x8 (&dst)[NI][3], int buf) __attribute__((always_inline)) {
    const char* lb = lds + buf * STAGE + A_BYTES;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[j][pl] = frag(lb, b_r0, b_r1, B_KC, (wn0 >> 5) + j, pl);
  };
  // six product terms, smallest first: (mid,mid) (hi,lo) (lo,hi) (hi,mid) (mid,hi) (hi,hi)
  constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#define OFB_MMA_HALF(AF, BF, BLK0)                                                                                                  \
  _Pragma("unroll") for (int q = 0; q < 6; ++q) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) \
      acc[BLK0 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[i][TA[q]], BF[j][TB[q]], acc[BLK0 + i][j], 0, 0, 0);
#define OFB_INTERLEAVE(NM, ND)                                                               \
  _Pragma("unroll") for (int z_ = 0; z_ < NM; ++z_) {                                        \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
    __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);                                      \
  }

  const int v = ofb_xcd_remap(blockIdx.x, p.W);     // consecutive v share an XCD (and thus operand panels in its L2)
  int sidx = 0;
  Seg cur = get_seg<TAIL>(p, v, 0);
  if (!cur.ok) return;
  while (true) {
    const int nk = cur.it1 - cur.it0;
    const size_t a_step = A_KC ? GRAN : (size_t)4 * a_ncb * GRAN, b_step = B_KC ? GRAN : (size_t)4 * b_ncb * GRAN;
    const char* a_base = (const char*)g.A + (A_KC ? (size_t)(cur.m0 / 4) * a_ncb * GRAN : (size_t)(cur.m0 / 16) * GRAN) + cur.it0 * a_step;
    const char* b_base = (const char*)g.B + (B_KC ? (size_t)(cur.n0 / 4) * b_ncb * GRAN : (size_t)(cur.n0 / 16) * GRAN) + cur.it0 * b_step;
    __builtin_amdgcn_s_barrier();                        // every wave is past the previous unit's LDS reads
    issue(0, a_base, b_base);
    if (nk > 1) issue(1, a_base + a_step, b_base + b_step);
    if (nk > 2) issue(2, a_base + 2 * a_step, b_base + 2 * b_step);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rdA(a01, 0, 0);
    rdB(bb[0], 0);
    auto step = [&](int i, int buf, auto PAR) __attribute__((always_inline)) {
      constexpr int par = decltype(PAR)::value;
      const int nbuf = buf + 1 == NST ? 0 : buf + 1;
      // first half: row blocks 0,1 x B(i); the reads of row blocks 2,3 ride in the MFMA gaps
      __builtin_amdgcn_sched_barrier(0);
      rdA(a23, buf, 2);
      OFB_MMA_HALF(a01, bb[par], 0)
      OFB_INTERLEAVE(24, 1)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // this wave is done reading buf(i)
      if (i + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // own pieces of stage i+1 landed (i+2 may be in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (i + 3 < nk) issue(buf, a_base + (size_t)(i + 3) * a_step, b_base + (size_t)(i + 3) * b_step);
      // second half: row blocks 2,3 x B(i); the reads of step i+1 (row blocks 0,1 and B) ride in the gaps (after the last step
      // they fetch a stale buffer that nothing consumes)
      __builtin_amdgcn_sched_barrier(0);
      rdA(a01, nbuf, 0);
      rdB(bb[par ^ 1], nbuf);
      OFB_MMA_HALF(a23, bb[par], 2)
      OFB_INTERLEAVE(12, 2)
      OFB_INTERLEAVE(12, 1)
      __builtin_amdgcn_sched_barrier(0);
    };
    int buf = 0, i = 0;
    for (; i + 1 < nk; i += 2) {
      step(i, buf, std::integral_constant<int, 0>{});
      buf = buf + 1 == NST ? 0 : buf + 1;
      step(i + 1, buf, std::integral_constant<int, 1>{});
      buf = buf + 1 == NST ? 0 : buf + 1;
    }
    if (i < nk) step(i, buf, std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // the trailing (unused) fragment reads

    {
      // Fused epilogue through LDS (the stage buffers are free now): the accumulators of one 128-row half of the tile are parked as
      // T[col][128 rows (+4 pad)] f32 - a lane's four consecutive rows of a column are one ds_write_b128 - and all 8 waves finish
      // that half together, one (4-row group, column) item per thread and trip: f32 loads / stores are coalesced along the columns,
      // the P-format store is the item's three 8-byte plane slots.  The body exists once (a direct epilogue over 128 accumulator
      // values per lane does not unroll and spills).
      constexpr int TROW = 132;                                     // floats per column of T: conflict-free b128 writes and reads
      float* T = reinterpret_cast<float*>(lds);
      const int rp_out = (g.M + 15) & ~15;
      __builtin_amdgcn_s_barrier();                                 // every wave has finished its fragment reads
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (w / WN == half) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
              for (int gq = 0; gq < 4; ++gq) {
                f32x4 q4 = {acc[mi][ni][4 * gq], acc[mi][ni][4 * gq + 1], acc[mi][ni][4 * gq + 2], acc[mi][ni][4 * gq + 3]};
                *reinterpret_cast<f32x4*>(T + (wn0 + 32 * ni + l31) * TROW + 32 * mi + 8 * gq + 4 * h) = q4;
                acc[mi][ni][4 * gq] = 0.f; acc[mi][ni][4 * gq + 1] = 0.f; acc[mi][ni][4 * gq + 2] = 0.f; acc[mi][ni][4 * gq + 3] = 0.f;
              }
        }
        __syncthreads();
        const int col = cur.n0 + (t & 255);
        const bool colok = col < g.N;
        const float bias = (!TAIL && g.bias && colok) ? g.bias[col] : 0.f, cs = (!TAIL && g.colscale && colok) ? g.colscale[col] : 1.f;
        for (int it = 0; it < 16; ++it) {
          const int rgl = 2 * it + (t >> 8);                        // 4-row group inside the half (0..31)
          const int row0 = cur.m0 + 128 * half + 4 * rgl;
          const f32x4 q4 = *reinterpret_cast<const f32x4*>(T + (t & 255) * TROW + 4 * rgl);
          if (TAIL) {                                               // raw partial tile -> workspace[slot][BM][BN]
            float* ws = g.workspace + (size_t)cur.slot * (BM * BN) + (size_t)(128 * half + 4 * rgl) * BN + (t & 255);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) ws[tt * BN] = q4[tt];
            continue;
          }
          if (row0 >= rp_out) break;
          float pv[4];
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
            const int row = row0 + tt;
            const bool ok = colok && row < g.M;
            const float val = epi_value(g, q4[tt], row, col, bias, cs, ok);
            if (g.C && ok) g.C[(size_t)row * g.ldc + col] = val;
            pv[tt] = ok ? val : 0.f;
          }
          if (g.Cp && col < g.c_ncb * 16)
            store_p4((char*)g.Cp + ((size_t)(row0 >> 2) * g.c_ncb + (col >> 4)) * GRAN + (col & 15) * 8, pv[0], pv[1], pv[2], pv[3]);
        }
        if (half == 0) __syncthreads();                             // T is rewritten by the other half (the next unit starts with a barrier)
      }
    }
    const Seg nxt = get_seg<TAIL>(p, v, ++sidx);
    if (!nxt.ok) break;
    cur = nxt;
  }
}

// Sums the partial tiles of each streamed tail tile in a fixed contributor order and applies the epilogue.
// grid (R, BM / 4), BN threads: block (r, rg) handles rows [4*rg, 4*rg+4) of tail tile r, one column per thread.
__global__ __launch_bounds__(BN) void gemm_p_fixup_kernel(const ofb_gemm_p_args g, const Plan p) {
  const int r = blockIdx.x, rgl = blockIdx.y, t = threadIdx.x;
  const int tile = p.full_rounds * p.W + r;
  int m0, n0;
  tile_coord(p, tile, m0, n0);
  const int lo = r * p.I, hi = lo + p.I;
  const int v0 = p.S ? 0 : lo / p.q, v1 = p.S ? (p.I + p.qs - 1) / p.qs - 1 : (hi - 1) / p.q;
  auto slot_of = [&](int v) { return p.S ? v * p.R + r : ((v * p.q < lo) ? 2 * v + 1 : 2 * v); };
  float sum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int v = v0; v <= v1; ++v) {
    const float* ws = g.workspace + (size_t)slot_of(v) * (BM * BN) + (size_t)(4 * rgl) * BN + t;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) sum[tt] += ws[tt * BN];
  }
  const int row0 = m0 + 4 * rgl, col = n0 + t;
  const bool colok = col < g.N;
  const float bias = (g.bias && colok) ? g.bias[col] : 0.f, cs = (g.colscale && colok) ? g.colscale[col] : 1.f;
  float pv[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int row = row0 + tt;
    const bool ok = colok && row < g.M;
    const float val = epi_value(g, sum[tt], row, col, bias, cs, ok);
    if (g.C && ok) g.C[(size_t)row * g.ldc + col] = val;
    pv[tt] = ok ? val : 0.f;
  }
  if (g.Cp && row0 < ((g.M + 15) & ~15) && col < g.c_ncb * 16)
    store_p4((char*)g.Cp + ((size_t)(row0 >> 2) * g.c_ncb + (col >> 4)) * GRAN + (col & 15) * 8, pv[0], pv[1], pv[2], pv[3]);
}

int p_cu_count() {
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

Plan plan_p(const ofb_gemm_p_args& g) {
  int W = p_cu_count();                              // one 144-KB workgroup per CU
  const int tiles = ofb_cdiv(g.M, BM) * ofb_cdiv(g.N, BN);
  const long long iters = (long long)tiles * ofb_cdiv(g.K, 16);
  if (iters < W) W = (int)iters;
  return make_plan(g.M, g.N, g.K, W, BM, BN, 16);
}

template <bool A_KC, bool B_KC>
int launch_p(const ofb_gemm_p_args& g, const Plan& p, hipStream_t s) {
  const dim3 grid(p.W), block(NT);
  if (p.full_rounds > 0) hipLaunchKernelGGL((gemm_p_kernel<A_KC, B_KC, false>), grid, block, 0, s, g, p);
  if (p.R > 0) {
    hipLaunchKernelGGL((gemm_p_kernel<A_KC, B_KC, true>), grid, block, 0, s, g, p);
    hipLaunchKernelGGL(gemm_p_fixup_kernel, dim3(p.R, BM / 4), dim3(BN), 0, s, g, p);
  }
  return ofb_launch_status();
}

}  // namespace

extern "C" int64_t ofb_pformat_bytes(int32_t R, int32_t C) {
  if (R <= 0 || C <= 0) return 0;
  // row groups up to the next 256-row tile boundary (mode KC reads whole tiles) + 16 granules of slack behind the last row
  // group (mode KR reads whole 256-column tiles); the slack is never initialised and only ever feeds accumulators that are
  // not stored
  const int64_t ncb = (C + 15) / 16, rgs = (int64_t)((R + 255) / 256) * 64;
  return (rgs * ncb + 16) * GRAN;
}

extern "C" int ofb_to_pformat(const float* X, int32_t R, int32_t C, int32_t ld, void* P, const float* rowscale, int32_t rs_div,
                              void* stream) {
  if (!X || !P || R <= 0 || C <= 0 || ld < C) return OFB_EINVAL;
  if (rowscale && rs_div <= 0) return OFB_EINVAL;
  const int ncb = (C + 15) / 16, rgs = ((R + 15) / 16) * 4;
  hipLaunchKernelGGL(to_pformat_kernel, dim3((ncb * 16 + 255) / 256, rgs), dim3(256), 0, (hipStream_t)stream, X, R, C, ld, (char*)P,
                     ncb, rowscale, rs_div);
  return ofb_launch_status();
}

extern "C" int ofb_from_pformat(const void* P, int32_t R, int32_t C, float* X, int32_t ld, void* stream) {
  if (!X || !P || R <= 0 || C <= 0 || ld < C) return OFB_EINVAL;
  const int ncb = (C + 15) / 16, rgs = (R + 3) / 4;
  hipLaunchKernelGGL(from_pformat_kernel, dim3((C + 255) / 256, rgs), dim3(256), 0, (hipStream_t)stream, (const char*)P, ncb, X, R,
                     C, ld);
  return ofb_launch_status();
}

extern "C" int64_t ofb_gemm_p_workspace_bytes(const ofb_gemm_p_args* args) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
  const Plan p = plan_p(*args);
  return p.R ? (int64_t)2 * p.W * BM * BN * (int64_t)sizeof(float) : 0;
}

extern "C" int ofb_gemm_p(const ofb_gemm_p_args* args, void* stream) {
  if (!args) return OFB_EINVAL;
  const ofb_gemm_p_args& g = *args;
  if (!g.A || !g.B || (!g.C && !g.Cp) || g.M <= 0 || g.N <= 0 || g.K <= 0) return OFB_EINVAL;
  if (g.a_kc == 0 && g.b_kc == 1) return OFB_ELIMIT;           // A^T * B^T is not on the path
  if (g.rowscale && g.rs_div <= 0) return OFB_EINVAL;
  if (g.act == OFB_ACT_DGELU && !g.aux) return OFB_EINVAL;
  if (g.C && g.ldc < g.N) return OFB_EINVAL;
  if (g.Cp && g.c_ncb < (g.N + 15) / 16) return OFB_EINVAL;
  // granule columns of each operand's P matrix must cover its extent along that axis
  if (g.a_ncb < ((g.a_kc ? g.K : g.M) + 15) / 16 || g.b_ncb < ((g.b_kc ? g.K : g.N) + 15) / 16) return OFB_EINVAL;
  if (!ofb_aligned16(g.A) || !ofb_aligned16(g.B) || (g.Cp && !ofb_aligned16(g.Cp))) return OFB_EINVAL;
  const Plan p = plan_p(g);
  if ((long long)p.W * p.I > 0x7fffffffLL / 2) return OFB_ELIMIT;
  if (p.R && (!g.workspace || g.workspace_bytes < ofb_gemm_p_workspace_bytes(args))) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(0, s, 2.0 * g.M * g.N * (double)g.K);
  int rc;
  if (g.a_kc && g.b_kc) rc = launch_p<true, true>(g, p, s);
  else if (g.a_kc) rc = launch_p<true, false>(g, p, s);
  else rc = launch_p<false, false>(g, p, s);
  ofb_prof_post(0, s);
  return rc;
}
