// Lab: "design W" -- the product kernel's staging (global -> registers -> exact 3-way bf16 split -> LDS planes, one K-step of
// register prefetch, two LDS buffers, one barrier per K-step) on a 256x256 workgroup tile with 8 waves (2 x 4), wave tile 128x64.
// Every staged value now feeds 2x the MFMAs of the 128x128 / 4-wave kernel: the inner loop of that kernel is bound by VALU issue
// (~100 VALU + 24 MFMA per wave and K-step), this one carries ~100 VALU per 48 MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/gemm_w_lab.hip -o scripts/lab/bin/gemm_w_lab
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

#ifndef WM
#define WM 2             // waves along M (wave tile rows = BMT / WM)
#endif
#ifndef MI
#define MI 4             // 32-row blocks per wave
#endif
#ifndef NI
#define NI 2             // 32-column blocks per wave
#endif
constexpr int NWAVES = 8, WN = NWAVES / WM;
constexpr int BMT = 32 * MI * WM, BNT = 32 * NI * WN, BK = 16;
constexpr int NT = 64 * NWAVES;
constexpr int ABLK = BMT * 16 + 16, APLANE = 2 * ABLK, AOPER = 3 * APLANE;
constexpr int BBLK = BNT * 16 + 16, BPLANE = 2 * BBLK, BOPER = 3 * BPLANE;
constexpr int STAGE = AOPER + BOPER;
constexpr int NLA = BMT * 4 / NT, NLB = BNT * 4 / NT;          // float4 loads per thread and K-step

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
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

__global__ __launch_bounds__(NT, 2) void gemm_w_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                       float* __restrict__ C, int ldc, int M, int N, int K, int W) {
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm0 = (w / WN) * (32 * MI), wn0 = (w % WN) * (32 * NI);
  const int mt = (M + BMT - 1) / BMT, nt = (N + BNT - 1) / BNT, ntiles = mt * nt, KS = K / BK;
  const int v = xcd_remap(blockIdx.x, W);

  f32x16 acc[MI][NI];
  f32x4 ra[NLA], rb[NLB];
  auto gload = [&](int m0, int n0, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int idx = t + NT * i, row = idx >> 2, kq = idx & 3;
      ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)min(m0 + row, M - 1) * lda + ks * BK + kq * 4);
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int idx = t + NT * i, row = idx >> 2, kq = idx & 3;
      rb[i] = *reinterpret_cast<const f32x4*>(B + (size_t)min(n0 + row, N - 1) * ldb + ks * BK + kq * 4);
    }
  };
  auto stage_one = [&](const f32x4& x, char* oper, int blk, int plane, int idx) __attribute__((always_inline)) {
    const int row = idx >> 2, kq = idx & 3;
    unsigned h0, m0, l0, h1, m1, l1;
    split2(x[0], x[1], h0, m0, l0);
    split2(x[2], x[3], h1, m1, l1);
    char* p = oper + (kq >> 1) * blk + row * 16 + (kq & 1) * 8;
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + plane) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(p + 2 * plane) = make_uint2(l0, l1);
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
    char* sa = lds + buf * STAGE;
    char* sb = sa + AOPER;
#pragma unroll
    for (int i = 0; i < NLA; ++i) stage_one(ra[i], sa, ABLK, APLANE, t + NT * i);
#pragma unroll
    for (int i = 0; i < NLB; ++i) stage_one(rb[i], sb, BBLK, BPLANE, t + NT * i);
  };
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* a_s = lds + buf * STAGE + h * ABLK + (wm0 + l31) * 16;
    const char* b_s = lds + buf * STAGE + AOPER + h * BBLK + (wn0 + l31) * 16;
    bf16x8 af[MI][3], bf[NI][3];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8*>(a_s + pl * APLANE + i * 32 * 16);
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8*>(b_s + pl * BPLANE + j * 32 * 16);
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][TA[q]], bf[j][TB[q]], acc[i][j], 0, 0, 0);
  };

  int buf = 0;
  for (int tile = v; tile < ntiles; tile += W) {
    const int m0 = (tile / nt) * BMT, n0 = (tile % nt) * BNT;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    gload(m0, n0, 0);
    lstore(buf);
    __syncthreads();
    for (int ks = 0; ks + 1 < KS; ++ks) {
      gload(m0, n0, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(buf);
      __builtin_amdgcn_sched_barrier(0);
      lstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    compute(buf);
    __builtin_amdgcn_sched_barrier(0);
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
    buf ^= 1;
    __syncthreads();
  }
}

struct Shape { int M, N, K; };

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const Shape shapes[] = {{25216, 1536, 384}, {25216, 384, 1536}, {25216, 1152, 384}, {25216, 384, 384}, {32768, 2048, 512}, {32768, 2048, 2048}};
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
    auto launch = [&]() { gemm_w_kernel<<<W, NT>>>(dA, K, dB, K, dC, N, M, N, K, W); };
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
    for (int i = 0; i < 30; ++i) launch();
    CHECK(hipEventRecord(e0));
    const int reps = 30;
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    const int ntiles = ((M + BMT - 1) / BMT) * ((N + BNT - 1) / BNT);
    printf("W tile %dx%d (wave %dx%d)  M=%d N=%d K=%d : %8.1f us  %7.1f TFLOP/s alg (%7.1f bf16 issued)  tiles %d (%.2f rounds)  worst err/sum|ab| %.2e\n", BMT, BNT,
           32 * MI, 32 * NI, M, N, K, us, tf, tf * 6, ntiles, (double)ntiles / W, worst);
    fflush(stdout);
    CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC));
  }
  return 0;
}
