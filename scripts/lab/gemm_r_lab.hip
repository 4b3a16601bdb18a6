// Lab: "design R" main loop for the f32 GEMM on the bf16 matrix pipe (6-term exact split).
//   A (activations, K-contiguous f32): raw f32 tile -> LDS by LDS-DMA (global_load_lds, swizzled on the source address),
//                                      split into three bf16 planes IN REGISTERS after the fragment read.
//   B (weights): pre-split ONCE into a tile-major image of bf16 planes in HBM, copied to LDS by LDS-DMA.
// No VGPR staging, no ds_write, one barrier per K-step, next K-step's DMA in flight under the MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/gemm_r_lab.hip -o scripts/lab/bin/gemm_r_lab
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int BM = 128, BK = 16;
#ifndef STAGES
#define STAGES 2
#endif
#ifndef ABL
#define ABL 0            // bit 0: no split VALU; bit 1: no DMA inside the loop; bit 2: no MFMA
#endif
#ifndef FRAGPF
#define FRAGPF 0            // 1: fragments of K-step t+1 are read into a second register set while K-step t multiplies (needs STAGES 3)
#endif
#ifndef WPS
#define WPS 2            // waves per SIMD = workgroups per CU (4-wave workgroups)
#endif

__device__ __forceinline__ unsigned pk(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = pk(a, b);
#if ABL & 1
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

// image[rb][ks][plane][half][128 rows][8 bf16] of X(row, k) = X[row*rs + k*cs]; rows >= R and k >= K are zero.
__global__ void split_image_kernel(const float* __restrict__ X, int R, int K, long rs, long cs, int RBs, int KS, char* __restrict__ img) {
  const long total = (long)RBs * KS * 2 * 128;     // one item = (rb, ks, half, row): 8 values -> 16 B in each plane
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
    const int row = it & 127, half = (it >> 7) & 1;
    const long c = it >> 8;
    const int ks = c % KS, rb = c / KS;
    const int gr = rb * 128 + row, k0 = ks * 16 + half * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (gr < R && k0 + j < K) ? X[gr * rs + (k0 + j) * cs] : 0.f;
    u32x4 hh, mm, ll;
#pragma unroll
    for (int j = 0; j < 4; ++j) { unsigned h, m, l; split2(v[2 * j], v[2 * j + 1], h, m, l); hh[j] = h; mm[j] = m; ll[j] = l; }
    char* base = img + (((long)rb * KS + ks) * 6) * 2048 + half * 2048 + row * 16;
    *reinterpret_cast<u32x4*>(base) = hh;
    *reinterpret_cast<u32x4*>(base + 2 * 2048) = mm;
    *reinterpret_cast<u32x4*>(base + 4 * 2048) = ll;
  }
}

template <int BN>
struct Cfg {
  static constexpr int A_BYTES = BM * BK * 4;
  static constexpr int B_BLK = BN * 16 + 16;
  static constexpr int B_BYTES = 6 * B_BLK;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int NI = BN / 64;
  static constexpr int RB = BN / 128;
};

#define LDSP(p) ((void __attribute__((address_space(3)))*)(p))

template <int BN>
__global__ __launch_bounds__(256, WPS) void gemm_r_kernel(const float* __restrict__ A, int lda, const char* __restrict__ Bimg, float* __restrict__ C,
                                                          int ldc, int M, int N, int K, int W) {
  using cfg = Cfg<BN>;
  constexpr int NI = cfg::NI, RB = cfg::RB;
  __shared__ __attribute__((aligned(16))) char lds[STAGES * cfg::STAGE];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
  const int v = xcd_remap(blockIdx.x, W);
  const int wm0 = (w >> 1) * 64, wn0 = (w & 1) * (BN / 2);
  const int mt = M / BM, nt = (N + BN - 1) / BN, ntiles = mt * nt, KS = K / BK;
  const int my_tiles = (ntiles - v + W - 1) / W;
  if (my_tiles <= 0) return;
  const int total = my_tiles * KS;

  f32x16 acc[2][NI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DMA issue for flattened iteration `it` (tile index it / KS of this workgroup, K-step it % KS) into stage buffer `buf`
  auto issue = [&](int it, int buf) __attribute__((always_inline)) {
    const int ti = it / KS, ks = it - ti * KS;
    const int tile = v + ti * W;
    const int m0 = (tile / nt) * BM, nb = tile % nt;
    char* sb = lds + buf * cfg::STAGE;
    // A: 8 wave-instructions of 16 rows x 64 B; lane -> (row, slot); slot holds chunk slot ^ ((row >> 2) & 3)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = w * 2 + i, row = q * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
      const float* src = A + (size_t)(m0 + row) * lda + ks * BK + c * 4;
      __builtin_amdgcn_global_load_lds((const void*)src, LDSP(sb + q * 1024), 16, 0, 0);
    }
    char* bb = sb + cfg::A_BYTES;
#pragma unroll
    for (int i = 0; i < 3 * RB; ++i) {
      const int q = w * (3 * RB) + i, rbi = q / 12, rem = q % 12, blk = rem >> 1, rh = rem & 1;
      const char* src = Bimg + ((((size_t)(nb * RB + rbi) * KS + ks) * 6 + blk) * 2048) + rh * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((const void*)src, LDSP(bb + blk * cfg::B_BLK + rbi * 2048 + rh * 1024), 16, 0, 0);
    }
  };

  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* sb = lds + buf * cfg::STAGE;
    const char* bb = sb + cfg::A_BYTES;
    bf16x8 af[2][3], bf[NI][3];
    f32x4 a_lo[2], a_hi[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int R = wm0 + i * 32 + l31, s = (R >> 2) & 3;
      a_lo[i] = *reinterpret_cast<const f32x4*>(sb + R * 64 + (((2 * h) ^ s) << 4));
      a_hi[i] = *reinterpret_cast<const f32x4*>(sb + R * 64 + (((2 * h + 1) ^ s) << 4));
    }
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        bf[j][pl] = *reinterpret_cast<const bf16x8*>(bb + (pl * 2 + h) * cfg::B_BLK + (wn0 + j * 32 + l31) * 16);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 hh, mm, ll;
      unsigned a, b, c;
      split2(a_lo[i][0], a_lo[i][1], a, b, c); hh[0] = a; mm[0] = b; ll[0] = c;
      split2(a_lo[i][2], a_lo[i][3], a, b, c); hh[1] = a; mm[1] = b; ll[1] = c;
      split2(a_hi[i][0], a_hi[i][1], a, b, c); hh[2] = a; mm[2] = b; ll[2] = c;
      split2(a_hi[i][2], a_hi[i][3], a, b, c); hh[3] = a; mm[3] = b; ll[3] = c;
      af[i][0] = __builtin_bit_cast(bf16x8, hh);
      af[i][1] = __builtin_bit_cast(bf16x8, mm);
      af[i][2] = __builtin_bit_cast(bf16x8, ll);
    }
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#if ABL & 4
          { if (q == 0) acc[i][j][0] += __builtin_bit_cast(f32x4, af[i][0])[0] + __builtin_bit_cast(f32x4, af[i][1])[1] + __builtin_bit_cast(f32x4, af[i][2])[2] + __builtin_bit_cast(f32x4, bf[j][0])[0] + __builtin_bit_cast(f32x4, bf[j][1])[1] + __builtin_bit_cast(f32x4, bf[j][2])[2]; }
#else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][TA[q]], bf[j][TB[q]], acc[i][j], 0, 0, 0);
#endif
  };

#if FRAGPF
  // ---- fragment prefetch: register double buffer, DMA two K-steps ahead, one barrier per K-step --------------------------
  constexpr int PERF = 2 + 3 * RB;
  f32x4 ra_lo[2][2], ra_hi[2][2];
  bf16x8 rbf[2][NI][3];
  auto ldfrag = [&](int buf, int S) __attribute__((always_inline)) {
    const char* sb = lds + buf * cfg::STAGE;
    const char* bb = sb + cfg::A_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int R = wm0 + i * 32 + l31, s = (R >> 2) & 3;
      ra_lo[S][i] = *reinterpret_cast<const f32x4*>(sb + R * 64 + (((2 * h) ^ s) << 4));
      ra_hi[S][i] = *reinterpret_cast<const f32x4*>(sb + R * 64 + (((2 * h + 1) ^ s) << 4));
    }
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        rbf[S][j][pl] = *reinterpret_cast<const bf16x8*>(bb + (pl * 2 + h) * cfg::B_BLK + (wn0 + j * 32 + l31) * 16);
  };
  auto mma = [&](int S) __attribute__((always_inline)) {
    bf16x8 af[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      u32x4 hh, mm, ll;
      unsigned a, b, c;
      split2(ra_lo[S][i][0], ra_lo[S][i][1], a, b, c); hh[0] = a; mm[0] = b; ll[0] = c;
      split2(ra_lo[S][i][2], ra_lo[S][i][3], a, b, c); hh[1] = a; mm[1] = b; ll[1] = c;
      split2(ra_hi[S][i][0], ra_hi[S][i][1], a, b, c); hh[2] = a; mm[2] = b; ll[2] = c;
      split2(ra_hi[S][i][2], ra_hi[S][i][3], a, b, c); hh[3] = a; mm[3] = b; ll[3] = c;
      af[i][0] = __builtin_bit_cast(bf16x8, hh);
      af[i][1] = __builtin_bit_cast(bf16x8, mm);
      af[i][2] = __builtin_bit_cast(bf16x8, ll);
    }
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][TA[q]], rbf[S][j][TB[q]], acc[i][j], 0, 0, 0);
  };
  auto waitp = [&]() __attribute__((always_inline)) {
    if (PERF == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  };
  issue(0, 0);
  issue(min(1, total - 1), 1);
  waitp();
  __builtin_amdgcn_s_barrier();
  ldfrag(0, 0);
  int ks = 0, ti = 0, buf = 0;
  auto body = [&](int it, int S) __attribute__((always_inline)) {
    int b1 = buf + 1; if (b1 >= 3) b1 -= 3;
    int b2 = buf + 2; if (b2 >= 3) b2 -= 3;
    issue(min(it + 2, total - 1), b2);
    waitp();                                          // tile it+1 has landed (this wave's pieces)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    ldfrag(b1, S ^ 1);
    mma(S);
    __builtin_amdgcn_sched_barrier(0);
    buf = b1;
    if (++ks == KS) {
      ks = 0;
      const int tile = v + ti * W;
      ++ti;
      const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int col = n0 + wn0 + 32 * j + l31, rbase = m0 + wm0 + 32 * i + 4 * h;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (col < N) C[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * ldc + col] = acc[i][j][r];
            acc[i][j][r] = 0.f;
          }
        }
    }
  };
  int it = 0;
  for (; it + 1 < total; it += 2) { body(it, 0); body(it + 1, 1); }
  if (it < total) body(it, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif STAGES == 2
  issue(0, 0);
  int ks = 0, ti = 0;
  for (int it = 0; it < total; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 < total && !(ABL & 2)) issue(it + 1, (it + 1) & 1);
    compute(it & 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (++ks == KS) {
      ks = 0;
      const int tile = v + ti * W;
      ++ti;
      const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int col = n0 + wn0 + 32 * j + l31, rbase = m0 + wm0 + 32 * i + 4 * h;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (col < N) C[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * ldc + col] = acc[i][j][r];
            acc[i][j][r] = 0.f;
          }
        }
    }
  }
#else
  // 3 stages: two K-steps of DMA in flight; per-thread DMA count per stage = 2 + 3*RB
  constexpr int PER = 2 + 3 * RB;
  issue(0, 0);
  if (total > 1) issue(1, 1);
  int ks = 0, ti = 0, buf = 0;
  for (int it = 0; it < total; ++it) {
    if (it + 1 < total) {
      if (PER == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    int nb2 = buf + 2; if (nb2 >= 3) nb2 -= 3;
    if (it + 2 < total && !(ABL & 2)) issue(it + 2, nb2);
    compute(buf);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (++buf == 3) buf = 0;
    if (++ks == KS) {
      ks = 0;
      const int tile = v + ti * W;
      ++ti;
      const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int col = n0 + wn0 + 32 * j + l31, rbase = m0 + wm0 + 32 * i + 4 * h;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (col < N) C[(size_t)(rbase + (r & 3) + 8 * (r >> 2)) * ldc + col] = acc[i][j][r];
            acc[i][j][r] = 0.f;
          }
        }
    }
  }
#endif
}

struct Shape { int M, N, K; };

template <int BN>
void run(const Shape& s, int ncu, const float* dA, const float* dB, float* dC, char* dImg, const std::vector<float>& hA, const std::vector<float>& hB) {
  const int M = s.M, N = s.N, K = s.K;
  const int nt = (N + BN - 1) / BN, RBs = nt * (BN / 128), KS = K / 16;
  split_image_kernel<<<1024, 256>>>(dB, N, K, K, 1, RBs, KS, dImg);
  CHECK(hipGetLastError());
  const int W = ncu * WPS;
  CHECK(hipMemset(dC, 0, (size_t)M * N * 4));
  auto launch = [&]() { gemm_r_kernel<BN><<<W, 256>>>(dA, K, dImg, dC, N, M, N, K, W); };
  launch();
  CHECK(hipDeviceSynchronize());
  // check sampled entries against f64
  std::vector<float> hC((size_t)M * N);
  CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  srand(7);
  for (int sidx = 0; sidx < 4000; ++sidx) {
    const int m = rand() % M, n = rand() % N;
    double ref = 0, mag = 0;
    for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k]; ref += p; mag += fabs(p); }
    worst = std::max(worst, fabs(hC[(size_t)m * N + n] - ref) / mag);
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("R fp=%d abl=%d BN=%d stages=%d wps=%d  M=%d N=%d K=%d : %8.1f us  %7.1f TFLOP/s alg (%7.1f bf16 issued)  worst err/sum|ab| %.2e\n", FRAGPF, ABL, BN, STAGES, WPS, M, N, K, us, tf,
         tf * 6, worst);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, ncu);
  const Shape shapes[] = {{25216, 1536, 384}, {25216, 384, 1536}, {25216, 1152, 384}, {25216, 384, 384}};
  for (const Shape& s : shapes) {
    std::vector<float> hA((size_t)s.M * s.K), hB((size_t)s.N * s.K);
    srand(1);
    for (auto& x : hA) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& x : hB) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *dA, *dB, *dC;
    char* dImg;
    CHECK(hipMalloc(&dA, hA.size() * 4)); CHECK(hipMalloc(&dB, hB.size() * 4)); CHECK(hipMalloc(&dC, (size_t)s.M * s.N * 4));
    CHECK(hipMalloc(&dImg, (size_t)(s.N + 512) * s.K * 6 + 65536));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    run<128>(s, ncu, dA, dB, dC, dImg, hA, hB);
    run<256>(s, ncu, dA, dB, dC, dImg, hA, hB);
    CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC)); CHECK(hipFree(dImg));
  }
  return 0;
}
