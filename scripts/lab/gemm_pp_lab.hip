// Lab: "design PP" -- barrier-staggered ping-pong main loop for the f32 GEMM on the bf16 matrix pipe (6-term exact split).
//   One 512-thread workgroup per CU = two 4-wave groups, one wave of each group on every SIMD.  Each group owns its own
//   128x128 output tile (wave tile 64x64) and its own single LDS image of split planes (A + B, 24 KB).  A group's K-step has
//   four phases, every phase ends in a workgroup barrier, and group 1 runs two phases behind group 0:
//        group 0:  M1(t)   M2(t)   W(t+1)  R(t+1) | M1(t+1) ...
//        group 1:  W(t)    R(t)    M1(t)   M2(t)  | W(t+1)  ...
//     M1/M2 : 12 MFMAs each (row block 0 / 1 of the wave tile) on fragments that are ALREADY in registers
//     W     : (tile-end stores,) split the prefetched f32 registers into planes and write them to the group's LDS image
//     R     : read the 12 fragments of that image into registers; issue the global loads of a K-step two ahead (register ring)
//   so the matrix pipe of every SIMD always has exactly one wave feeding it, nothing in a staging phase waits for memory that was
//   requested less than ~1.5 K-steps earlier, and no MFMA waits for an LDS read.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/gemm_pp_lab.hip -o scripts/lab/bin/gemm_pp_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef STAMPS
#define STAMPS 0
#endif
#ifndef ABLATE
#define ABLATE 0          // bit 0: no split VALU; bit 1: no global loads in the loop; bit 2: no MFMA
#endif

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int BLK = BM * 16 + 16;          // one [128 rows][8 bf16] block (+16 B: the two k-halves land on different banks)
constexpr int PLANE = 2 * BLK;
constexpr int OPER = 3 * PLANE;
constexpr int GROUP_LDS = 2 * OPER;

__device__ __forceinline__ unsigned pk(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = pk(a, b);
#if ABLATE & 1
  m = h; l = h; return;
#endif
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = pk(ra, rb);
  l = pk(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

struct Regs { f32x4 a[2], b[2]; };

__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                         float* __restrict__ C, int ldc, int M, int N, int K, int W, unsigned long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(16))) char lds[2 * GROUP_LDS];
  const int t = threadIdx.x;
  const int grp = __builtin_amdgcn_readfirstlane(t >> 8);
  const int tg = t & 255, lane = tg & 63, w = tg >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm0 = (w >> 1) * 64, wn0 = (w & 1) * 64;
  const int mt = M / BM, nt = N / BN, ntiles = mt * nt, npairs = (ntiles + 1) / 2, KS = K / BK;
  const int v = xcd_remap(blockIdx.x, W);
  const int my_pairs = (npairs - v + W - 1) / W;
  if (my_pairs <= 0) return;
  const int total = my_pairs * KS;
  char* const LA = lds + grp * GROUP_LDS;
  char* const LB = LA + OPER;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  Regs ring[2];
  int nstamp = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
#if STAMPS
    if (blockIdx.x == 8 && tg == 0 && nstamp < 96) stamps[grp * 96 + nstamp++] = __builtin_amdgcn_s_memtime();
#endif
  };
  bf16x8 af[2][3], bf[2][3];

  // the flattened K-step j of this group: pair j / KS, k-step j % KS
  auto locate = [&](int j, int& m0, int& n0, int& ks, bool& valid) __attribute__((always_inline)) {
    j = min(j, total - 1);
    const int pi = j / KS;
    ks = j - pi * KS;
    int tile = 2 * (v + pi * W) + grp;
    valid = tile < ntiles;
    tile = min(tile, ntiles - 1);
    m0 = (tile / nt) * BM; n0 = (tile % nt) * BN;
  };
  auto gload = [&](Regs& r, int j) __attribute__((always_inline)) {
    int m0, n0, ks; bool valid;
    locate(j, m0, n0, ks, valid);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tg + 256 * i, row = idx >> 2, kq = idx & 3;
      r.a[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + row) * lda + ks * BK + kq * 4);
      r.b[i] = *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + row) * ldb + ks * BK + kq * 4);
    }
  };
  auto stage_one = [&](const f32x4& x, char* oper, int idx) __attribute__((always_inline)) {
    const int row = idx >> 2, kq = idx & 3;
    unsigned h0, m0, l0, h1, m1, l1;
    split2(x[0], x[1], h0, m0, l0);
    split2(x[2], x[3], h1, m1, l1);
    char* p = oper + (kq >> 1) * BLK + row * 16 + (kq & 1) * 8;
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(p + 2 * PLANE) = make_uint2(l0, l1);
  };
  auto stage = [&](const Regs& r) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      stage_one(r.a[i], LA, tg + 256 * i);
      stage_one(r.b[i], LB, tg + 256 * i);
    }
  };
  auto ldfrag = [&]() __attribute__((always_inline)) {
    const char* a_s = LA + h * BLK + (wm0 + l31) * 16;
    const char* b_s = LB + h * BLK + (wn0 + l31) * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        af[i][pl] = *reinterpret_cast<const bf16x8*>(a_s + pl * PLANE + i * 32 * 16);
        bf[i][pl] = *reinterpret_cast<const bf16x8*>(b_s + pl * PLANE + i * 32 * 16);
      }
  };
  auto mma_half = [&](int i) __attribute__((always_inline)) {
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#if ABLATE & 4
        { if (q == 0) acc[i][j][0] += __builtin_bit_cast(f32x4, af[i][0])[0] + __builtin_bit_cast(f32x4, bf[j][1])[1]; }
#else
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][TA[q]], bf[j][TB[q]], acc[i][j], 0, 0, 0);
#endif
  };
  // the tile whose last K-step was multiplied as flattened step j: store it (called at the head of a W phase)
  auto store_if_done = [&](int j) __attribute__((always_inline)) {
    int m0, n0, ks; bool valid;
    locate(j, m0, n0, ks, valid);
    if (ks != KS - 1) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int col = n0 + wn0 + 32 * jj + l31, rbase = m0 + wm0 + 32 * i + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (valid) C[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * ldc + col] = acc[i][jj][r];
          acc[i][jj][r] = 0.f;
        }
      }
  };

  if (grp == 0) {
    // prologue: tile 0 staged and its fragments read; tiles 1, 2 in flight
    gload(ring[0], 0);
    gload(ring[1], 1);
    stage(ring[0]);
    if (!(ABLATE & 2)) gload(ring[0], 2);
    PP_BARRIER(); stamp();
    ldfrag();
    PP_BARRIER(); stamp();
    for (int it = 0; it < total; it += 2) {
      // ---- even K-step `it`: next tile it+1 sits in ring[1]
      mma_half(0);
      PP_BARRIER(); stamp();
      mma_half(1);
      PP_BARRIER(); stamp();
      store_if_done(it);
      stage(ring[1]);                                   // W(it+1)
      PP_BARRIER(); stamp();
      ldfrag();                                         // R(it+1)
      if (!(ABLATE & 2)) gload(ring[1], it + 3);
      PP_BARRIER(); stamp();
      if (it + 1 >= total) break;
      // ---- odd K-step it+1: next tile it+2 sits in ring[0]
      mma_half(0);
      PP_BARRIER(); stamp();
      mma_half(1);
      PP_BARRIER(); stamp();
      store_if_done(it + 1);
      stage(ring[0]);                                   // W(it+2)
      PP_BARRIER(); stamp();
      ldfrag();                                         // R(it+2)
      if (!(ABLATE & 2)) gload(ring[0], it + 4);
      PP_BARRIER(); stamp();
    }
  } else {
    gload(ring[0], 0);
    gload(ring[1], 1);
    PP_BARRIER(); stamp();
    PP_BARRIER(); stamp();
    for (int it = 0; it < total; it += 2) {
      if (it > 0) store_if_done(it - 1);
      stage(ring[0]);                                   // W(it)
      PP_BARRIER(); stamp();
      ldfrag();                                         // R(it)
      if (!(ABLATE & 2)) gload(ring[0], it + 2);
      PP_BARRIER(); stamp();
      mma_half(0);
      PP_BARRIER(); stamp();
      mma_half(1);
      PP_BARRIER(); stamp();
      if (it + 1 >= total) break;
      store_if_done(it);
      stage(ring[1]);                                   // W(it+1)
      PP_BARRIER(); stamp();
      ldfrag();                                         // R(it+1)
      if (!(ABLATE & 2)) gload(ring[1], it + 3);
      PP_BARRIER(); stamp();
      mma_half(0);
      PP_BARRIER(); stamp();
      mma_half(1);
      PP_BARRIER(); stamp();
    }
    store_if_done(total - 1);
  }
}

struct Shape { int M, N, K; };

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const Shape shapes[] = {{25216, 1536, 384}, {25216, 384, 1536}, {25216, 1152, 384}, {25216, 384, 384}, {32768, 2048, 512}};
  for (const Shape& s : shapes) {
    const int M = s.M, N = s.N, K = s.K;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    srand(1);
    for (auto& x : hA) x = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
    for (auto& x : hB) x = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
    float *dA, *dB, *dC;
    CHECK(hipMalloc(&dA, hA.size() * 4)); CHECK(hipMalloc(&dB, hB.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(dC, 0, (size_t)M * N * 4));
    const int W = ncu;
    unsigned long long* dS; CHECK(hipMalloc(&dS, 192 * 8)); CHECK(hipMemset(dS, 0, 192 * 8));
    auto launch = [&]() { gemm_pp_kernel<<<W, 512>>>(dA, K, dB, K, dC, N, M, N, K, W, dS); };
    launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)M * N);
    CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    srand(7);
    for (int sidx = 0; sidx < 6000; ++sidx) {
      const int m = (sidx < 256) ? M - 1 - sidx : rand() % M, n = rand() % N;
      double ref = 0, mag = 0;
      for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k]; ref += p; mag += fabs(p); }
      worst = std::max(worst, fabs(hC[(size_t)m * N + n] - ref) / mag);
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipEventRecord(e0));
    const int reps = 30;
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    const int ntiles = (M / BM) * (N / BN);
    printf("PP ablate=%d  M=%d N=%d K=%d : %8.1f us  %7.1f TFLOP/s alg (%7.1f bf16 issued)  pairs %d (%.2f rounds)  worst err/sum|ab| %.2e\n", ABLATE, M, N, K,
           us, tf, tf * 6, (ntiles + 1) / 2, (double)((ntiles + 1) / 2) / W, worst);
#if STAMPS
    { unsigned long long hs[192]; CHECK(hipMemcpy(hs, dS, sizeof(hs), hipMemcpyDeviceToHost));
      for (int g = 0; g < 2; ++g) { printf("  group %d phase cycles:", g); for (int i = 40; i < 72; ++i) printf(" %llu", hs[g * 96 + i + 1] - hs[g * 96 + i]); printf("\n"); } }
#endif
    fflush(stdout);
    CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC));
  }
  return 0;
}
