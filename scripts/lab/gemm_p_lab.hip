// Lab: "design P" -- f32-accurate GEMM whose operands arrive ALREADY split into three bf16 planes ("P-format"), staged by
// LDS-DMA (global_load_lds_dwordx4), no VALU split and no VGPR->LDS stores in the loop.
//
// P-format of a matrix X[R][C]: granules of 4 rows x 16 columns, 384 B each, stored [R/4][C/16]; inside a granule
// [plane hi|mid|lo][c % 16][r % 4] bf16 (128 B per plane).  One layout serves both consumers of an activation / weight:
//   mode KC (reduction along C, operand rows = R): fragments by ds_read_b64_tr_b16 (the 4x16 transpose block IS a plane slab row set)
//   mode KR (reduction along R, operand rows = C): fragments by two ds_read_b64 (4 consecutive r are contiguous)
// and a 32x32 MFMA accumulator block (lane = column, 4 consecutive rows per register group) stores it with 8-byte stores that
// fill whole 128-B lines.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/gemm_p_lab.hip -o /tmp/gemm_p_lab && /tmp/gemm_p_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef NST
#define NST 3            // LDS stages
#endif
#ifndef PIPE
#define PIPE 1
#endif
#include <type_traits>
constexpr int BM = 256, BN = 256, WM = 2, WN = 4, MI = 4, NI = 2, NT = 512;
constexpr int A_BYTES = BM * 16 * 6, B_BYTES = BN * 16 * 6, STAGE = A_BYTES + B_BYTES;       // 24 KB + 24 KB per K16 step
constexpr int GL_A = A_BYTES / 1024 / 8, GL_B = B_BYTES / 1024 / 8;                         // glds per wave and stage (3 + 3)

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

// ---- f32 -> P-format ----------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pk(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = pk(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = pk(ra, rb);
  l = pk(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}
// X[R][C] row-major (ld) -> P[Rp/4][Cp/16][3][16][4]; pads with zeros
__global__ void to_pformat(const float* __restrict__ X, int R, int C, int ld, char* __restrict__ P, int Rp, int Cp) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, rg = blockIdx.y;
  if (c >= Cp) return;
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { const int r = 4 * rg + t; v[t] = (r < R && c < C) ? X[(size_t)r * ld + c] : 0.f; }
  unsigned h0, m0, l0, h1, m1, l1;
  split2(v[0], v[1], h0, m0, l0);
  split2(v[2], v[3], h1, m1, l1);
  char* g = P + ((size_t)rg * (Cp / 16) + (c >> 4)) * 384 + (c & 15) * 8;
  *reinterpret_cast<uint2*>(g) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(g + 128) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(g + 256) = make_uint2(l0, l1);
}

// ---- the GEMM -----------------------------------------------------------------------------------------------------
// C[M][N] (f32 row-major) = A * B^T-ish: A operand rows = M index, B operand rows = N index, both P-format.
//   A_KC: A's P matrix is [R = M][C = K] (ncb = Kp/16);  else [R = K][C = M] (ncb = Mp/16)
//   B_KC: B's P matrix is [R = N][C = K];                else [R = K][C = N]
template <bool KC>
struct Oper {
  // per-lane constant source offset for glds instruction `q` of this wave (q = 0..2), and the per-stage / per-tile strides
  // KC: instruction (j = block of 32 rows, p = plane): lane -> (tg = l>>3, cp = (l&7) ^ swz(tg))
  // KR: linear 1-KB pieces of the [tq][cb][plane][128] image
};

__device__ __forceinline__ int swz(int tg) { return ((tg >> 1) & 3) << 1; }

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(NT, 2) void gemm_p_kernel(const char* __restrict__ Ap, int a_ncb, const char* __restrict__ Bp, int b_ncb,
                                                       float* __restrict__ C, int ldc, int M, int N, int K, int W) {
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE];
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, h = lane >> 5;
  const int wm0 = (w / WN) * (32 * MI), wn0 = (w % WN) * (32 * NI);
  const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN, ntiles = mt * nt, nk = K / 16;

  // ---- glds source offsets (bytes, relative to the tile/stage base), per wave-instruction q = 0..2 -> piece id = w + 8 q
  unsigned a_off[3], b_off[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int piece = w + 8 * q;                          // 0..23: 1-KB piece of the 24-KB operand image
    if (A_KC) {
      const int j = piece / 3, p = piece % 3, tg = lane >> 3, cp = (lane & 7) ^ swz(tg);
      a_off[q] = (unsigned)(((8 * j + tg) * a_ncb) * 384 + p * 128 + cp * 16);
    } else {
      const int b = piece * 1024 + lane * 16, tq = b / (BM / 16 * 384), rem = b % (BM / 16 * 384);
      a_off[q] = (unsigned)((tq * a_ncb) * 384 + rem);
    }
    if (B_KC) {
      const int j = piece / 3, p = piece % 3, tg = lane >> 3, cp = (lane & 7) ^ swz(tg);
      b_off[q] = (unsigned)(((8 * j + tg) * b_ncb) * 384 + p * 128 + cp * 16);
    } else {
      const int b = piece * 1024 + lane * 16, tq = b / (BN / 16 * 384), rem = b % (BN / 16 * 384);
      b_off[q] = (unsigned)((tq * b_ncb) * 384 + rem);
    }
  }
  // ---- fragment read offsets (bytes inside an operand's stage image)
  // KC: two tr reads; group g = lane>>4, i = lane&15, q = i>>2 (k row), pp = i&3 (token group)
  int a_r0, a_r1, b_r0, b_r1;
  {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, hh = g >> 1, tg = 4 * (g & 1) + pp;
    const int c0 = 8 * hh + q, c1 = c0 + 4;
    const int kc0 = (8 * tg + ((c0 >> 1) ^ swz(tg))) * 16 + (c0 & 1) * 8, kc1 = (8 * tg + ((c1 >> 1) ^ swz(tg))) * 16 + (c1 & 1) * 8;
    // KR: n = lane&31, h = lane>>5: tq = 2h, 2h+1
    const int kr_a0 = ((2 * h) * (BM / 16) + (l31 >> 4)) * 384 + (l31 & 15) * 8, kr_a1 = kr_a0 + (BM / 16) * 384;
    const int kr_b0 = ((2 * h) * (BN / 16) + (l31 >> 4)) * 384 + (l31 & 15) * 8, kr_b1 = kr_b0 + (BN / 16) * 384;
    a_r0 = A_KC ? kc0 : kr_a0; a_r1 = A_KC ? kc1 : kr_a1;
    b_r0 = B_KC ? kc0 : kr_b0; b_r1 = B_KC ? kc1 : kr_b1;
  }

  f32x16 acc[MI][NI];

  // LDS-DMA issue in inline asm: through the builtin hipcc treats every LDS-DMA as a store that may alias the fragment reads and
  // drains it with s_waitcnt vmcnt(0) in front of the first ds_read of the K-step (no pipelining at all).  In asm the loads are
  // invisible to its bookkeeping; their completion is counted by hand (6 per wave and stage, vmcnt(6) / vmcnt(0) below).
  // Pieces of one wave sit 8 KB apart in the stage image (A: w, w+8, w+16; B follows A at +24 KB = 3 x 8 KB).
  const unsigned lds0 = (unsigned)(size_t)LDSP(lds) + (unsigned)w * 1024u;
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
  auto frag = [&](const char* base, int r0, int r1, bool kc, int blk, int plane) __attribute__((always_inline)) -> bf16x8 {
    s16x4 lo4, hi4;
    if (kc) {
      const char* p = base + (blk * 3 + plane) * 1024;
      lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + r0));
      hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + r1));
    } else {
      const char* p = base + (2 * blk) * 384 + plane * 128;
      lo4 = *reinterpret_cast<const s16x4*>(p + r0);
      hi4 = *reinterpret_cast<const s16x4*>(p + r1);
    }
    s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* la = lds + buf * STAGE;
    const char* lb = la + A_BYTES;
    bf16x8 af[MI][3], bf[NI][3];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[j][p] = frag(lb, b_r0, b_r1, B_KC, (wn0 >> 5) + j, p);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) af[i][p] = frag(la, a_r0, a_r1, A_KC, (wm0 >> 5) + i, p);
    // six product terms, smallest first: (mid,mid) (hi,lo) (lo,hi) (hi,mid) (mid,hi) (hi,hi)
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][TA[q]], bf[j][TB[q]], acc[i][j], 0, 0, 0);
  };

  const int v = xcd_remap(blockIdx.x, W);
#if PIPE
  // software-pipelined K loop: fragments of the NEXT half-step are read while the MFMAs of the current one run, one barrier per
  // K16 step placed between the two halves (24 MFMAs on either side), LDS-DMA three stages ahead.
  bf16x8 a01[2][3], a23[2][3], bb[2][NI][3];
  auto rdA = [&](bf16x8 (&dst)[2][3], int buf, int blk0) __attribute__((always_inline)) {
    const char* la = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[i][p] = frag(la, a_r0, a_r1, A_KC, (wm0 >> 5) + blk0 + i, p);
  };
  auto rdB = [&](bf16x8 (&dst)[NI][3], int buf) __attribute__((always_inline)) {
    const char* lb = lds + buf * STAGE + A_BYTES;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[j][p] = frag(lb, b_r0, b_r1, B_KC, (wn0 >> 5) + j, p);
  };
  constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#define MMA_HALF(AF, BF, BLK0)                                                                                              \
  _Pragma("unroll") for (int q = 0; q < 6; ++q) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) \
      acc[BLK0 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[i][TA[q]], BF[j][TB[q]], acc[BLK0 + i][j], 0, 0, 0);
#define INTERLEAVE(NM, ND)                                                                   \
  _Pragma("unroll") for (int z_ = 0; z_ < NM; ++z_) {                                        \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
    __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);                                      \
  }
  for (int tile = v; tile < ntiles; tile += W) {
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* a_base = A_KC ? Ap + (size_t)(m0 / 4) * a_ncb * 384 : Ap + (size_t)(m0 / 16) * 384;
    const char* b_base = B_KC ? Bp + (size_t)(n0 / 4) * b_ncb * 384 : Bp + (size_t)(n0 / 16) * 384;
    const size_t a_step = A_KC ? 384 : (size_t)4 * a_ncb * 384, b_step = B_KC ? 384 : (size_t)4 * b_ncb * 384;
    __builtin_amdgcn_s_barrier();                        // every wave is past the previous tile's LDS reads
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
      // ---- first half: A blocks 0,1 x B(i); the reads of A blocks 2,3 ride in the MFMA gaps
      __builtin_amdgcn_sched_barrier(0);
      rdA(a23, buf, 2);
      MMA_HALF(a01, bb[par], 0)
      INTERLEAVE(24, 1)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // this wave is done reading buf(i)
      if (i + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // own pieces of stage i+1 landed (i+2 may be in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (i + 3 < nk) issue(buf, a_base + (size_t)(i + 3) * a_step, b_base + (size_t)(i + 3) * b_step);
      // ---- second half: A blocks 2,3 x B(i); the reads of step i+1 (A blocks 0,1 and B) ride in the gaps
      __builtin_amdgcn_sched_barrier(0);
      rdA(a01, nbuf, 0);
      rdB(bb[par ^ 1], nbuf);
      MMA_HALF(a23, bb[par], 2)
      INTERLEAVE(12, 2)
      INTERLEAVE(12, 1)
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the trailing (unused) fragment reads
#else
  for (int tile = v; tile < ntiles; tile += W) {
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // tile bases and per-stage strides
    const char* a_base = A_KC ? Ap + (size_t)(m0 / 4) * a_ncb * 384 : Ap + (size_t)(m0 / 16) * 384;
    const char* b_base = B_KC ? Bp + (size_t)(n0 / 4) * b_ncb * 384 : Bp + (size_t)(n0 / 16) * 384;
    const size_t a_step = A_KC ? 384 : (size_t)4 * a_ncb * 384, b_step = B_KC ? 384 : (size_t)4 * b_ncb * 384;
    // all waves are past the previous tile's LDS reads (they finished their MFMAs' operands) before anyone refills
    __builtin_amdgcn_s_barrier();
    issue(0, a_base, b_base);
    if (nk > 1) issue(1, a_base + a_step, b_base + b_step);
    int buf = 0;
    for (int i = 0; i < nk; ++i) {
      if (i + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (i + 2 < nk) {
        const int nb = buf + 2 >= NST ? buf + 2 - NST : buf + 2;
        issue(nb, a_base + (size_t)(i + 2) * a_step, b_base + (size_t)(i + 2) * b_step);
      }
      compute(buf);
      buf = buf + 1 == NST ? 0 : buf + 1;
    }
#endif
    // epilogue: plain f32 stores (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn0 + 32 * j + l31, rbase = m0 + wm0 + 32 * i + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (row < M && col < N) C[(size_t)row * ldc + col] = acc[i][j][r];
        }
      }
  }
}

// ---- host ---------------------------------------------------------------------------------------------------------
static void fill(std::vector<float>& v, unsigned seed) {
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 32768.0f - 1.0f + ((s >> 24) & 0xff) * 1e-6f; }
}
static int rup(int a, int b) { return (a + b - 1) / b * b; }

template <bool A_KC, bool B_KC>
static void run_case(const char* name, int M, int N, int K, int iters) {
  // host matrices in "logical" form: A[M][K], B[N][K]
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  fill(hA, 1); fill(hB, 2);
  // device f32 storage: KC -> [rows][K]; KR -> [K][rows]
  std::vector<float> sA(hA.size()), sB(hB.size());
  if (A_KC) sA = hA; else for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) sA[(size_t)k * M + m] = hA[(size_t)m * K + k];
  if (B_KC) sB = hB; else for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) sB[(size_t)k * N + n] = hB[(size_t)n * K + k];
  const int Mp = rup(M, BM), Np = rup(N, BN), Kp = rup(K, 16);
  // P matrices: KC: R = rows (pad to tile), C = K;  KR: R = K, C = rows (pad to tile)
  const int aR = A_KC ? Mp : Kp, aC = A_KC ? Kp : Mp, bR = B_KC ? Np : Kp, bC = B_KC ? Kp : Np;
  float *dA, *dB, *dC; char *pA, *pB;
  CHECK(hipMalloc(&dA, sA.size() * 4)); CHECK(hipMalloc(&dB, sB.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
  CHECK(hipMalloc(&pA, (size_t)aR * aC * 6)); CHECK(hipMalloc(&pB, (size_t)bR * bC * 6));
  CHECK(hipMemcpy(dA, sA.data(), sA.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, sB.data(), sB.size() * 4, hipMemcpyHostToDevice));
  {
    const int r = A_KC ? M : K, c = A_KC ? K : M;
    hipLaunchKernelGGL(to_pformat, dim3((aC + 255) / 256, aR / 4), dim3(256), 0, 0, dA, r, c, c, pA, aR, aC);
    const int r2 = B_KC ? N : K, c2 = B_KC ? K : N;
    hipLaunchKernelGGL(to_pformat, dim3((bC + 255) / 256, bR / 4), dim3(256), 0, 0, dB, r2, c2, c2, pB, bR, bC);
  }
  CHECK(hipDeviceSynchronize());
  const int tiles = (Mp / BM) * (Np / BN);
  const int W = std::min(tiles, 256);
  auto launch = [&]() {
    hipLaunchKernelGGL((gemm_p_kernel<A_KC, B_KC>), dim3(W), dim3(NT), 0, 0, pA, aC / 16, pB, bC / 16, dC, N, M, N, Kp, W);
  };
  launch();
  CHECK(hipDeviceSynchronize());
  // check a sample of outputs against fp64
  std::vector<float> hC((size_t)M * N);
  CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double maxrel = 0;
  for (int s = 0; s < 4000; ++s) {
    const int m = (int)((s * 2654435761u) % (unsigned)M), n = (int)((s * 40503u + 17) % (unsigned)N);
    double ref = 0, asum = 0;
    for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k]; ref += p; asum += fabs(p); }
    maxrel = std::max(maxrel, fabs(hC[(size_t)m * N + n] - ref) / (asum + 1e-30));
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch();
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, tf = 2.0 * M * N * (double)K / (us * 1e-6) / 1e12;
  printf("%-4s M %6d N %5d K %5d tiles %4d: %8.1f us  %7.1f TFLOP/s (f32-equivalent)  max err %.2e of sum|ab|\n", name, M, N, K, tiles, us, tf, maxrel);
  CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC)); CHECK(hipFree(pA)); CHECK(hipFree(pB));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  // small correctness cases first (ragged M), then timing shapes with whole rounds of 256 tiles
  run_case<true, true>("nt", 300, 256, 64, 2);
  run_case<true, false>("nn", 300, 256, 64, 2);
  run_case<false, false>("tn", 256, 512, 80, 2);
  run_case<true, true>("nt", 32768, 2048, 384, iters);      // 1024 tiles = 4 rounds
  run_case<true, true>("nt", 32768, 1024, 1536, iters);     // 512 tiles = 2 rounds
  run_case<true, false>("nn", 32768, 2048, 384, iters);
  run_case<false, false>("tn", 1536, 512, 25216, iters);    // 12 tiles: no split-K here, just the long-K main loop on 12 CUs
  run_case<true, true>("nt", 25216, 1536, 384, iters);      // fc1 (594 tiles: 2.32 rounds)
  run_case<true, true>("nt", 25216, 384, 1536, iters);      // fc2 (198 tiles)
  return 0;
}
