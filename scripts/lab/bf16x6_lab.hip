// Lab: f32 GEMM C[M][N] = A[M][K] * B[N][K]^T on bf16 MFMA with an exact 3-way bf16 split of both operands and the 6
// leading product terms (hh, hm, mh, hl, lh, mm), f32 accumulate.  Standalone feasibility / rate probe.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/bf16x6_lab.hip -o /tmp/bf16x6_lab && /tmp/bf16x6_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef TERMS
#define TERMS 6
#endif
#ifndef PAD2
#define PAD2 0
#endif
#ifndef SCHED
#define SCHED 0
#endif
#ifndef ORDER
#define ORDER 1
#endif
#ifndef ABL
#define ABL 0
#endif
#ifndef STAMP
#define STAMP 0
#endif
#ifndef PRIO
#define PRIO 0
#endif
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int BLK = 128 * 16 + 32;                 // bytes of one [128 rows][8 bf16] block (+pad)
constexpr int PLANE = 4 * BLK;                    // (ks, half) blocks
constexpr int OPER = 3 * PLANE;

__device__ __forceinline__ unsigned pk(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// exact-ish 3-way split of a pair: returns packed (hi, mid, lo) bf16 pairs
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = pk(a, b);
#if ABL == 1
  m = h; l = h; return;
#endif
  float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = pk(ra, rb);
  float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
  l = pk(sa, sb);
}

__device__ __forceinline__ void stage(const float4 (&r)[4], char* lds_oper, int t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = t + 256 * i, row = f >> 3, kq = f & 7;
    const int blk = kq >> 1, j0 = (kq & 1) * 4;
    unsigned h0, m0, l0, h1, m1, l1;
    split2(r[i].x, r[i].y, h0, m0, l0);
    split2(r[i].z, r[i].w, h1, m1, l1);
    char* p = lds_oper + blk * BLK + row * 16 + j0 * 2;
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(p + 2 * PLANE) = make_uint2(l0, l1);
  }
}

__device__ __forceinline__ void gload(float4 (&r)[4], const float* __restrict__ X, int ld, int row0, int k0, int t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = t + 256 * i, row = f >> 3, kq = f & 7;
    r[i] = *reinterpret_cast<const float4*>(X + (size_t)(row0 + row) * ld + k0 + 4 * kq);
  }
}

__global__ __launch_bounds__(256, 2) void gemm_bf16x6(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                      int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) char lds[2 * OPER];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, hf = lane >> 5;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int mt = M / BM;
  const int m0 = (blockIdx.x % mt) * BM, n0 = (blockIdx.x / mt) * BN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra[4], rb[4];
  gload(ra, A, K, m0, 0, t);
  gload(rb, B, K, n0, 0, t);
  char* la = lds;
  char* lb = lds + OPER;
  for (int k0 = 0; k0 < K; k0 += BK) {
    stage(ra, la, t);
    stage(rb, lb, t);
    __syncthreads();
    if (k0 + BK < K) {
      gload(ra, A, K, m0, k0 + BK, t);
      gload(rb, B, K, n0, k0 + BK, t);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fa[i][p] = *reinterpret_cast<const bf16x8*>(la + p * PLANE + (ks * 2 + hf) * BLK + (wm + 32 * i + l31) * 16);
          fb[i][p] = *reinterpret_cast<const bf16x8*>(lb + p * PLANE + (ks * 2 + hf) * BLK + (wn + 32 * i + l31) * 16);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // small terms first
#if TERMS >= 6
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
#endif
#if TERMS >= 3
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
#endif
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hf, col = n0 + wn + 32 * j + l31;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
}


// ---- v2: BK=16, double-buffered LDS, one barrier per k-tile, split of tile k+1 interleaved with the MFMAs of tile k ----
constexpr int BLK2 = 128 * 16 + PAD2;              // one [128 rows][8 bf16] block
constexpr int PLANE2 = 2 * BLK2;                   // halves
constexpr int OPER2 = 3 * PLANE2;
constexpr int STAGE2 = 2 * OPER2;

__device__ __forceinline__ void gload2(float4 (&r)[2], const float* __restrict__ X, int ld, int row0, int k0, int t) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = t + 256 * i, row = f >> 2, kq = f & 3;
    r[i] = *reinterpret_cast<const float4*>(X + (size_t)(row0 + row) * ld + k0 + 4 * kq);
  }
}
__device__ __forceinline__ void stage2(const float4 (&r)[2], char* lds_oper, int t) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = t + 256 * i, row = f >> 2, kq = f & 3;
    const int half = kq >> 1, j0 = (kq & 1) * 4;
    unsigned h0, m0, l0, h1, m1, l1;
    split2(r[i].x, r[i].y, h0, m0, l0);
    split2(r[i].z, r[i].w, h1, m1, l1);
    char* p = lds_oper + half * BLK2 + row * 16 + j0 * 2;
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + PLANE2) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(p + 2 * PLANE2) = make_uint2(l0, l1);
  }
}

__device__ unsigned long long g_stamp[4 * 4096];
__global__ __launch_bounds__(256, 2) void gemm_v2(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                  int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE2];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, hf = lane >> 5;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int mt = M / BM, nt = N / BN;
#if ORDER == 0
  const int m0 = (blockIdx.x % mt) * BM, n0 = (blockIdx.x / mt) * BN;
#else
  // XCD-aware: workgroup b runs on XCD b%8; give every XCD one contiguous run of n-fastest tiles (A tile shared in its L2)
  const int chunk = (mt * nt + 7) / 8;
  const int tile = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
  if (blockIdx.x / 8 >= chunk || tile >= mt * nt) return;
  const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
#endif
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra[2], rb[2];
  gload2(ra, A, K, m0, 0, t);
  gload2(rb, B, K, n0, 0, t);
  stage2(ra, lds, t);
  stage2(rb, lds + OPER2, t);
  if (K > 16) {
    gload2(ra, A, K, m0, 16, t);
    gload2(rb, B, K, n0, 16, t);
  }
  __syncthreads();
  const int nk = K / 16;
#if STAMP
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = lds + (kt & 1) * STAGE2;
    char* nxt = lds + ((kt & 1) ^ 1) * STAGE2;
    bf16x8 fa[2][3], fb[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[i][p] = *reinterpret_cast<const bf16x8*>(cur + p * PLANE2 + hf * BLK2 + (wm + 32 * i + l31) * 16);
        fb[i][p] = *reinterpret_cast<const bf16x8*>(cur + OPER2 + p * PLANE2 + hf * BLK2 + (wn + 32 * i + l31) * 16);
      }
#if ABL != 3
    if (kt + 1 < nk) {
      stage2(ra, nxt, t);
      stage2(rb, nxt + OPER2, t);
    }
#endif
    if (kt + 2 < nk && ABL != 2) {
      gload2(ra, A, K, m0, 16 * (kt + 2), t);
      gload2(rb, B, K, n0, 16 * (kt + 2), t);
    }
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#if PRIO
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(PRIO);
#endif
#pragma unroll
    for (int q = 6 - TERMS; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][TA[q]], fb[j][TB[q]], acc[i][j], 0, 0, 0);
#if PRIO
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
#endif
#if SCHED
#pragma unroll
    for (int g = 0; g < 4 * TERMS; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, SCHED, 0);   // VALU
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 DS write
    }
#endif
    __syncthreads();
  }
#if STAMP
  if (t == 0 && blockIdx.x < 4096) {
    g_stamp[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
    g_stamp[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
#endif
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hf, col = n0 + wn + 32 * j + l31;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
}

// ---- v3: like v2 but prefetch distance 2 (two register sets), loop peeled so that the split / LDS writes of tile k+1 and
// the MFMAs of tile k are in ONE basic block and can be interleaved by sched_group_barrier ----
template <bool STAGE, bool LOAD>
__device__ __forceinline__ void body3(f32x16 (&acc)[2][2], char* cur, char* nxt, float4 (&ua)[2], float4 (&ub)[2], float4 (&la)[2],
                                      float4 (&lb)[2], const float* __restrict__ A, const float* __restrict__ B, int K, int m0, int n0,
                                      int kload, int t, int wm, int wn, int l31, int hf) {
  bf16x8 fa[2][3], fb[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      fa[i][p] = *reinterpret_cast<const bf16x8*>(cur + p * PLANE2 + hf * BLK2 + (wm + 32 * i + l31) * 16);
      fb[i][p] = *reinterpret_cast<const bf16x8*>(cur + OPER2 + p * PLANE2 + hf * BLK2 + (wn + 32 * i + l31) * 16);
    }
  if (LOAD) {
    gload2(la, A, K, m0, kload, t);
    gload2(lb, B, K, n0, kload, t);
  }
  if (STAGE) {
    stage2(ua, nxt, t);
    stage2(ub, nxt + OPER2, t);
  }
  constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
  for (int q = 6 - TERMS; q < 6; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][TA[q]], fb[j][TB[q]], acc[i][j], 0, 0, 0);
#if SCHED
  if (STAGE) {
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);     // fragment reads first
    if (LOAD) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);   // then the global prefetch
#pragma unroll
    for (int g = 0; g < 4 * TERMS; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, SCHED, 0);   // VALU
      if (g % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
    }
  }
#endif
  __syncthreads();
}

__global__ __launch_bounds__(256, 2) void gemm_v3(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                  int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE2];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, hf = lane >> 5;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int mt = M / BM, nt = N / BN;
  const int chunk = (mt * nt + 7) / 8;
  const int tile = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
  if (blockIdx.x / 8 >= chunk || tile >= mt * nt) return;
  const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 a0[2], b0[2], a1[2], b1[2];
  gload2(a0, A, K, m0, 0, t);
  gload2(b0, B, K, n0, 0, t);
  gload2(a1, A, K, m0, 16, t);
  gload2(b1, B, K, n0, 16, t);
  stage2(a0, lds, t);
  stage2(b0, lds + OPER2, t);
  __syncthreads();
  char* L0 = lds;
  char* L1 = lds + STAGE2;
  const int nk = K / 16;                 // even, >= 4
  // invariant at kt (even): LDS L0 = tile kt, regs a1/b1 = tile kt+1
  int kt = 0;
  for (; kt + 4 <= nk; kt += 2) {
    body3<true, true>(acc, L0, L1, a1, b1, a0, b0, A, B, K, m0, n0, 16 * (kt + 2), t, wm, wn, l31, hf);   // stage kt+1, load kt+2 -> a0
    body3<true, true>(acc, L1, L0, a0, b0, a1, b1, A, B, K, m0, n0, 16 * (kt + 3), t, wm, wn, l31, hf);   // stage kt+2, load kt+3 -> a1
  }
  body3<true, false>(acc, L0, L1, a1, b1, a0, b0, A, B, K, m0, n0, 0, t, wm, wn, l31, hf);                // kt = nk-2: stage nk-1
  body3<false, false>(acc, L1, L0, a0, b0, a1, b1, A, B, K, m0, n0, 0, t, wm, wn, l31, hf);               // kt = nk-1
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hf, col = n0 + wn + 32 * j + l31;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
}

// ---- v4: ping-pong.  One 512-thread workgroup = two 4-wave groups, one wave of each per SIMD.  Group g owns output tile
// (2p + g, n); both share the B tile.  In even phases group 0 issues MFMAs while group 1 splits / stages, in odd phases the
// roles swap, so every SIMD's matrix pipe always has exactly one wave feeding it and the staging VALU runs beside it.
constexpr int V4_A0 = 0, V4_A1 = OPER2, V4_B = 2 * OPER2;      // LDS: A0 | A1 | B[2]

__device__ __forceinline__ void gload_rows(float4& r, const float* __restrict__ X, int ld, int row, int k0, int kq) {
  r = *reinterpret_cast<const float4*>(X + (size_t)row * ld + k0 + 4 * kq);
}
__device__ __forceinline__ void stage_one(const float4& r, char* lds_oper, int row, int kq) {
  const int half = kq >> 1, j0 = (kq & 1) * 4;
  unsigned h0, m0, l0, h1, m1, l1;
  split2(r.x, r.y, h0, m0, l0);
  split2(r.z, r.w, h1, m1, l1);
  char* p = lds_oper + half * BLK2 + row * 16 + j0 * 2;
  *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(p + PLANE2) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(p + 2 * PLANE2) = make_uint2(l0, l1);
}

__global__ __launch_bounds__(512, 2) void gemm_v4(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                  int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) char lds[4 * OPER2];
  const int t = threadIdx.x, grp = t >> 8, tg = t & 255, lane = t & 63, w = tg >> 6, l31 = lane & 31, hf = lane >> 5;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int mt = M / BM, nt = N / BN, mp = (mt + 1) / 2;
  const int chunk = (mp * nt + 7) / 8;
  const int tile = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
  if (blockIdx.x / 8 >= chunk || tile >= mp * nt) return;
  const int mtile = 2 * (tile / nt) + grp, n0 = (tile % nt) * BN;
  const bool valid = mtile < mt;
  const int m0 = (valid ? mtile : mt - 1) * BM;          // an unpaired last tile: group 1 shadows a valid tile and does not store
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // staging items of this thread: A_g rows (tg>>2) and (tg>>2)+64, k-quad tg&3; B half g: row 64*grp + (tg>>2), k-quad tg&3
  const int srow = tg >> 2, kq = tg & 3;
  float4 ra0, ra1, rb;
  char* const LA = lds + (grp ? V4_A1 : V4_A0);
  const int nk = K / 16;
  auto loadA = [&](int k) { gload_rows(ra0, A, K, m0 + srow, 16 * k, kq); gload_rows(ra1, A, K, m0 + srow + 64, 16 * k, kq); };
  auto loadB = [&](int k) { gload_rows(rb, B, K, n0 + 64 * grp + srow, 16 * k, kq); };
  auto stageA = [&]() { stage_one(ra0, LA, srow, kq); stage_one(ra1, LA, srow + 64, kq); };
  auto stageB = [&](int k) { stage_one(rb, lds + V4_B + (k & 1) * OPER2, 64 * grp + srow, kq); };
  auto compute = [&](int k) {
    const char* cb = lds + V4_B + (k & 1) * OPER2;
    bf16x8 fa[2][3], fb[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[i][p] = *reinterpret_cast<const bf16x8*>(LA + p * PLANE2 + hf * BLK2 + (wm + 32 * i + l31) * 16);
        fb[i][p] = *reinterpret_cast<const bf16x8*>(cb + p * PLANE2 + hf * BLK2 + (wn + 32 * i + l31) * 16);
      }
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int q = 6 - TERMS; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][TA[q]], fb[j][TB[q]], acc[i][j], 0, 0, 0);
  };
  // prologue
  if (grp == 0) { loadA(0); loadB(0); stageA(); stageB(0); }
  else { loadB(0); stageB(0); loadA(0); if (nk > 1) loadB(1); }
  __syncthreads();
#if STAMP
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int k = 0; k < nk; ++k) {
    // even phase: group 0 computes k, group 1 stages A1(k) and its half of B(k+1)
    if (grp == 0) {
      if (k + 1 < nk) { loadA(k + 1); loadB(k + 1); }
      compute(k);
    } else {
      stageA();
      if (k + 1 < nk) stageB(k + 1);
    }
    __syncthreads();
    // odd phase: group 1 computes k, group 0 stages A0(k+1) and its half of B(k+1)
    if (grp == 0) {
      if (k + 1 < nk) { stageA(); stageB(k + 1); }
    } else {
      if (k + 1 < nk) { loadA(k + 1); if (k + 2 < nk) loadB(k + 2); }
      compute(k);
    }
    __syncthreads();
  }
#if STAMP
  if (t == 0 && blockIdx.x < 4096) {
    g_stamp[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
    g_stamp[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
#endif
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hf, col = n0 + wn + 32 * j + l31;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
}

// ---- v5: ping-pong as v4 but each phase covers KS k-steps of 16 (longer MFMA bursts amortise the barrier + fragment-read head)
#ifndef KS
#define KS 2
#endif
constexpr int OPER5 = KS * OPER2;                                  // one operand tile of KS k-steps: [ks][plane][half][row][8]
constexpr int V5_A0 = 0, V5_A1 = OPER5, V5_B = 2 * OPER5;          // LDS: A0 | A1 | B[2]

__global__ __launch_bounds__(512, 2) void gemm_v5(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                  int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, grp = t >> 8, tg = t & 255, lane = t & 63, w = tg >> 6, l31 = lane & 31, hf = lane >> 5;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int mt = M / BM, nt = N / BN, mp = (mt + 1) / 2;
  const int chunk = (mp * nt + 7) / 8;
  const int tile = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
  if (blockIdx.x / 8 >= chunk || tile >= mp * nt) return;
  const int mtile = 2 * (tile / nt) + grp, n0 = (tile % nt) * BN;
  const bool valid = mtile < mt;
  const int m0 = (valid ? mtile : mt - 1) * BM;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int srow = tg >> 2, kq = tg & 3;
  float4 ra0[KS], ra1[KS], rb[KS];
  char* const LA = lds + (grp ? V5_A1 : V5_A0);
  const int nk = K / (16 * KS);                       // phases pairs
  auto loadA = [&](int k) {
#pragma unroll
    for (int s = 0; s < KS; ++s) { gload_rows(ra0[s], A, K, m0 + srow, 16 * (KS * k + s), kq); gload_rows(ra1[s], A, K, m0 + srow + 64, 16 * (KS * k + s), kq); }
  };
  auto loadB = [&](int k) {
#pragma unroll
    for (int s = 0; s < KS; ++s) gload_rows(rb[s], B, K, n0 + 64 * grp + srow, 16 * (KS * k + s), kq);
  };
  auto stageA = [&]() {
#pragma unroll
    for (int s = 0; s < KS; ++s) { stage_one(ra0[s], LA + s * OPER2, srow, kq); stage_one(ra1[s], LA + s * OPER2, srow + 64, kq); }
  };
  auto stageB = [&](int k) {
#pragma unroll
    for (int s = 0; s < KS; ++s) stage_one(rb[s], lds + V5_B + (k & 1) * OPER5 + s * OPER2, 64 * grp + srow, kq);
  };
  auto compute = [&](int k) {
#if PRIO
    __builtin_amdgcn_s_setprio(PRIO);
#endif
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const char* ca = LA + s * OPER2;
      const char* cb = lds + V5_B + (k & 1) * OPER5 + s * OPER2;
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fa[i][p] = *reinterpret_cast<const bf16x8*>(ca + p * PLANE2 + hf * BLK2 + (wm + 32 * i + l31) * 16);
          fb[i][p] = *reinterpret_cast<const bf16x8*>(cb + p * PLANE2 + hf * BLK2 + (wn + 32 * i + l31) * 16);
        }
      constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
      for (int q = 6 - TERMS; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][TA[q]], fb[j][TB[q]], acc[i][j], 0, 0, 0);
    }
#if PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  if (grp == 0) { loadA(0); loadB(0); stageA(); stageB(0); }
  else { loadB(0); stageB(0); loadA(0); if (nk > 1) loadB(1); }
  __syncthreads();
#if STAMP
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int k = 0; k < nk; ++k) {
    if (grp == 0) {
      if (k + 1 < nk) { loadA(k + 1); loadB(k + 1); }
      compute(k);
    } else {
      stageA();
      if (k + 1 < nk) stageB(k + 1);
    }
    __syncthreads();
    if (grp == 0) {
      if (k + 1 < nk) { stageA(); stageB(k + 1); }
    } else {
      if (k + 1 < nk) { loadA(k + 1); if (k + 2 < nk) loadB(k + 2); }
      compute(k);
    }
    __syncthreads();
  }
#if STAMP
  if (t == 0 && blockIdx.x < 4096) {
    g_stamp[4 * blockIdx.x] = (__builtin_amdgcn_s_memtime() - c0) / KS;
    g_stamp[4 * blockIdx.x + 1] = (__builtin_amdgcn_s_memrealtime() - r0) / KS;
  }
#endif
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hf, col = n0 + wn + 32 * j + l31;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
}

// ---- v6: dual-tile ping-pong.  Two independent 128x128 tiles per 512-thread workgroup (group g = tile g), every phase ends
// in a workgroup barrier; group 0 computes in even phases and stages in odd ones, group 1 the other way round.  Single-
// buffered LDS per group: a group stages tile-step k+1 right after it finished reading step k.
constexpr int OPER6 = KS * OPER2;

__global__ __launch_bounds__(512, 2) void gemm_v6(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                  int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, grp = t >> 8, tg = t & 255, lane = t & 63, w = tg >> 6, l31 = lane & 31, hf = lane >> 5;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int mt = M / BM, nt = N / BN, ntl = mt * nt, np = (ntl + 1) / 2;
  const int chunk = (np + 7) / 8;
  const int pair = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
  if (blockIdx.x / 8 >= chunk || pair >= np) return;
  const int tile_raw = 2 * pair + grp;
  const bool valid = tile_raw < ntl;
  const int tile = valid ? tile_raw : ntl - 1;
  const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int srow = tg >> 2, kq = tg & 3;
  float4 ra0[KS], ra1[KS], rb0[KS], rb1[KS];
  char* const LA = lds + grp * 2 * OPER6;
  char* const LB = LA + OPER6;
  const int nk = K / (16 * KS);
  auto load = [&](int k) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k0 = 16 * (KS * k + s);
      gload_rows(ra0[s], A, K, m0 + srow, k0, kq); gload_rows(ra1[s], A, K, m0 + srow + 64, k0, kq);
      gload_rows(rb0[s], B, K, n0 + srow, k0, kq); gload_rows(rb1[s], B, K, n0 + srow + 64, k0, kq);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      stage_one(ra0[s], LA + s * OPER2, srow, kq); stage_one(ra1[s], LA + s * OPER2, srow + 64, kq);
      stage_one(rb0[s], LB + s * OPER2, srow, kq); stage_one(rb1[s], LB + s * OPER2, srow + 64, kq);
    }
  };
  auto compute = [&]() {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const char* ca = LA + s * OPER2;
      const char* cb = LB + s * OPER2;
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fa[i][p] = *reinterpret_cast<const bf16x8*>(ca + p * PLANE2 + hf * BLK2 + (wm + 32 * i + l31) * 16);
          fb[i][p] = *reinterpret_cast<const bf16x8*>(cb + p * PLANE2 + hf * BLK2 + (wn + 32 * i + l31) * 16);
        }
      constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
      for (int q = 6 - TERMS; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][TA[q]], fb[j][TB[q]], acc[i][j], 0, 0, 0);
    }
  };
  // prologue: both groups stage step 0; group 1 then waits one extra phase
  load(0); stage();
  if (nk > 1) load(1);
  __syncthreads();
#if STAMP
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
  // phases p = 0 .. 2 nk: group 0 computes step p/2 in even phases and stages step p/2 + 1 in odd ones; group 1 computes step
  // (p-1)/2 in odd phases and stages the next step in the following even phase.
  for (int p = 0; p <= 2 * nk; ++p) {
    const int ph = p - grp;                      // this group's own phase clock
    if (ph >= 0 && ph < 2 * nk) {
      const int k = ph >> 1;
      if ((ph & 1) == 0) {
        compute();
      } else if (k + 1 < nk) {
        stage();                                 // step k+1 (loaded during compute)
        if (k + 2 < nk) load(k + 2);
      }
    }
    __syncthreads();
  }
#if STAMP
  if (t == 0 && blockIdx.x < 4096) {
    g_stamp[4 * blockIdx.x] = (__builtin_amdgcn_s_memtime() - c0) / KS;
    g_stamp[4 * blockIdx.x + 1] = (__builtin_amdgcn_s_memrealtime() - r0) / KS;
  }
#endif
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hf, col = n0 + wn + 32 * j + l31;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
}

#ifndef KERNEL
#define KERNEL gemm_bf16x6
#endif
#ifndef NTHREADS
#define NTHREADS 256
#endif
#ifndef DYN_LDS
#define DYN_LDS 0
#endif
#define STR2(x) #x
#define STR(x) STR2(x)
#define NAME STR(KERNEL) " pad=" STR(PAD2) " sched=" STR(SCHED) " order=" STR(ORDER) " abl=" STR(ABL) " prio=" STR(PRIO)
int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 25216, N = argc > 2 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 384;
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hA) v = rnd() * (1.0f + 3.0f * rnd() * rnd());
  for (auto& v : hB) v = rnd() * 0.05f;
  float *dA, *dB, *dC;
  hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  dim3 grid(((M / BM) * (N / BN) + 7) / 8 * 8);
  if (DYN_LDS) hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, DYN_LDS);
  for (int i = 0; i < 1500; ++i) hipLaunchKernelGGL(KERNEL, grid, dim3(NTHREADS), DYN_LDS, 0, dA, dB, dC, M, N, K);   // clock ramp
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 30;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(KERNEL, grid, dim3(NTHREADS), DYN_LDS, 0, dA, dB, dC, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<float> hC((size_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, worst32 = 0;
  for (int sidx = 0; sidx < 400; ++sidx) {
    const int i = (sidx * 7919) % M, j = (sidx * 104729) % N;
    double ref = 0, mag = 0; float f32 = 0.f;
    for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k]; ref += p; mag += fabs(p); f32 = fmaf(hA[(size_t)i * K + k], hB[(size_t)j * K + k], f32); }
    worst = fmax(worst, fabs(hC[(size_t)i * N + j] - ref) / mag);
    worst32 = fmax(worst32, fabs((double)f32 - ref) / mag);
  }
#if STAMP
  {
    std::vector<unsigned long long> st(4 * 4096);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamp), st.size() * 8);
    std::vector<double> cyc, clk;
    const int nwg = (M / BM) * (N / BN) < 4096 ? (M / BM) * (N / BN) : 4096;
    for (int i = 0; i < nwg; ++i) if (st[4 * i + 1]) { cyc.push_back((double)st[4 * i] / (K / 16)); clk.push_back((double)st[4 * i] / st[4 * i + 1] * 0.1); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    printf("   in-kernel: median %.0f cycles per k16-iteration per workgroup, clock %.2f GHz (median over %zu workgroups)\n", cyc[cyc.size() / 2], clk[clk.size() / 2], cyc.size());
  }
#endif
  printf("%s TERMS=%d M=%d N=%d K=%d: %.3f ms  %.1f TF (algorithmic)  max|err|/sum|ab| = %.2e (f32 fma chain: %.2e)\n", NAME, TERMS, M, N, K, ms,
         2.0 * M * N * K / ms / 1e9, worst, worst32);
  return 0;
}
