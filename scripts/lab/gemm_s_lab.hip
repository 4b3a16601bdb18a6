// Lab: "design S" main loop for the f32 GEMM on the bf16 matrix pipe (6-term exact split) -- ONE wave per SIMD, deep pipeline.
//   workgroup = 4 waves (2x2), wave tile (32*MI) x (32*NI), one workgroup per CU with the whole 512-register file
//   A (activations, K-contiguous f32): raw f32 tile -> LDS by LDS-DMA (swizzled on the source address); fragments are read one
//     K-step ahead into registers and split into three bf16 planes between the MFMAs.
//   B (weights): pre-split once into a tile-major image of bf16 planes in HBM, copied to LDS by LDS-DMA.
//   3 LDS stages (DMA two K-steps ahead), register double buffer for the fragments, one barrier per K-step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/gemm_s_lab.hip -o scripts/lab/bin/gemm_s_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int BK = 16;
#ifndef ABL
#define ABL 0            // bit 0: no split VALU; bit 1: no DMA inside the loop; bit 2: no MFMA
#endif
#ifndef SGB
#define SGB 0            // 1: sched_group_barrier interleave pattern
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
  const long total = (long)RBs * KS * 2 * 128;
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

#define LDSP(p) ((void __attribute__((address_space(3)))*)(p))

template <int MI, int NI>
__global__ __launch_bounds__(256, 1) void gemm_s_kernel(const float* __restrict__ A, int lda, const char* __restrict__ Bimg, float* __restrict__ C,
                                                        int ldc, int M, int N, int K, int W) {
  constexpr int BM = 64 * MI, BN = 64 * NI, RBN = BN / 128;
  constexpr int A_BYTES = BM * 64, B_BLK = BN * 16 + 16, STAGE = A_BYTES + 6 * B_BLK;
  constexpr int PER = MI + 3 * RBN;                  // DMA instructions per thread and K-step
  static_assert(NI % 2 == 0, "B row blocks are 128 wide");
  __shared__ __attribute__((aligned(16))) char lds[3 * STAGE];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
  const int v = xcd_remap(blockIdx.x, W);
  const int wm0 = (w >> 1) * (32 * MI), wn0 = (w & 1) * (32 * NI);
  const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN, ntiles = mt * nt, KS = K / BK;
  const int my_tiles = (ntiles - v + W - 1) / W;
  if (my_tiles <= 0) return;
  const int total = my_tiles * KS;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 raw[MI][2];
  bf16x8 bfr[2][NI][3];

  auto issue = [&](int it_, int buf) __attribute__((always_inline)) {
    const int it = min(it_, total - 1);
    const int ti = it / KS, ks = it - ti * KS;
    const int tile = v + ti * W;
    const int m0 = (tile / nt) * BM, nb = tile % nt;
    char* sb = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int q = w * MI + i, row = q * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
      const float* src = A + (size_t)min(m0 + row, M - 1) * lda + ks * BK + c * 4;
      __builtin_amdgcn_global_load_lds((const void*)src, LDSP(sb + q * 1024), 16, 0, 0);
    }
    char* bb = sb + A_BYTES;
#pragma unroll
    for (int i = 0; i < 3 * RBN; ++i) {
      const int q = w * (3 * RBN) + i, rbi = q / 12, rem = q % 12, blk = rem >> 1, rh = rem & 1;
      const char* src = Bimg + ((((size_t)(nb * RBN + rbi) * KS + ks) * 6 + blk) * 2048) + rh * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((const void*)src, LDSP(bb + blk * B_BLK + rbi * 2048 + rh * 1024), 16, 0, 0);
    }
  };

  auto ld_a = [&](int buf, int i) __attribute__((always_inline)) {
    const char* sb = lds + buf * STAGE;
    const int R = wm0 + i * 32 + l31, s = (R >> 2) & 3;
    raw[i][0] = *reinterpret_cast<const f32x4*>(sb + R * 64 + (((2 * h) ^ s) << 4));
    raw[i][1] = *reinterpret_cast<const f32x4*>(sb + R * 64 + (((2 * h + 1) ^ s) << 4));
  };
  auto ld_b = [&](int buf, auto Sc) __attribute__((always_inline)) {
    constexpr int S = decltype(Sc)::value;
    const char* bb = lds + buf * STAGE + A_BYTES;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        bfr[S][j][pl] = *reinterpret_cast<const bf16x8*>(bb + (pl * 2 + h) * B_BLK + (wn0 + j * 32 + l31) * 16);
  };

  auto mma = [&](auto Sc, int nbuf) __attribute__((always_inline)) {
    constexpr int S = decltype(Sc)::value;
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      bf16x8 af[3];
      {
        u32x4 hh, mm, ll;
        unsigned a, b, c;
        split2(raw[i][0][0], raw[i][0][1], a, b, c); hh[0] = a; mm[0] = b; ll[0] = c;
        split2(raw[i][0][2], raw[i][0][3], a, b, c); hh[1] = a; mm[1] = b; ll[1] = c;
        split2(raw[i][1][0], raw[i][1][1], a, b, c); hh[2] = a; mm[2] = b; ll[2] = c;
        split2(raw[i][1][2], raw[i][1][3], a, b, c); hh[3] = a; mm[3] = b; ll[3] = c;
        af[0] = __builtin_bit_cast(bf16x8, hh);
        af[1] = __builtin_bit_cast(bf16x8, mm);
        af[2] = __builtin_bit_cast(bf16x8, ll);
      }
      if (!(ABL & 8)) ld_a(nbuf, i);          // rolling: this row block's registers take the NEXT K-step's values as soon as they are split
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#if ABL & 4
          { if (q == 0) acc[i][j][0] += __builtin_bit_cast(f32x4, af[0])[0] + __builtin_bit_cast(f32x4, af[1])[1] + __builtin_bit_cast(f32x4, af[2])[2] + __builtin_bit_cast(f32x4, bfr[S][j][0])[0] + __builtin_bit_cast(f32x4, bfr[S][j][1])[1] + __builtin_bit_cast(f32x4, bfr[S][j][2])[2]; }
#else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA[q]], bfr[S][j][TB[q]], acc[i][j], 0, 0, 0);
#endif
    }
  };

  auto store_tile = [&](int ti) __attribute__((always_inline)) {
    const int tile = v + ti * W;
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn0 + 32 * j + l31, rbase = m0 + wm0 + 32 * i + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (col < N && row < M && (!(ABL & 32) || acc[i][j][r] == 12345.678f)) C[(size_t)row * ldc + col] = acc[i][j][r];
          acc[i][j][r] = 0.f;
        }
      }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  int ks = 0, ti = 0, buf = 0;     // buf = stage holding tile `it`
  auto body = [&](int it, auto Cur, auto Nxt) __attribute__((always_inline)) {
    int b1 = buf + 1; if (b1 >= 3) b1 -= 3;
    int b2 = buf + 2; if (b2 >= 3) b2 -= 3;
    if (!(ABL & 2)) issue(it + 2, b2);
    if (PER == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (PER == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (PER == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (!(ABL & 8)) ld_b(b1, Nxt); else if (decltype(Nxt)::value == 1) { for (int j = 0; j < NI; ++j) for (int pl = 0; pl < 3; ++pl) bfr[1][j][pl] = bfr[0][j][pl]; }
    mma(Cur, b1);
#if SGB
    // per MFMA: a couple of VALU (the split of the next row block), and the next K-step's fragment reads spread over the block
#pragma unroll
    for (int g = 0; g < MI * 6 * NI; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
      if (g % 4 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    buf = b1;
    if (++ks == KS) { ks = 0; store_tile(ti); ++ti; }
  };

  // prologue
  issue(0, 0);
  issue(1, 1);
  if (PER == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (PER == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (PER == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  ld_b(0, S0{});
#pragma unroll
  for (int i = 0; i < MI; ++i) ld_a(0, i);
  int it = 0;
  for (; it + 1 < total; it += 2) {
    body(it, S0{}, S1{});
    body(it + 1, S1{}, S0{});
  }
  if (it < total) body(it, S0{}, S1{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

struct Shape { int M, N, K; };

template <int MI, int NI>
void run(const Shape& s, int ncu, const float* dA, const float* dB, float* dC, char* dImg, const std::vector<float>& hA, const std::vector<float>& hB) {
  constexpr int BM = 64 * MI, BN = 64 * NI;
  const int M = s.M, N = s.N, K = s.K;
  const int nt = (N + BN - 1) / BN, RBs = nt * (BN / 128), KS = K / 16;
  split_image_kernel<<<1024, 256>>>(dB, N, K, K, 1, RBs, KS, dImg);
  CHECK(hipGetLastError());
  const int W = ncu;
  CHECK(hipMemset(dC, 0, (size_t)M * N * 4));
  auto launch = [&]() { gemm_s_kernel<MI, NI><<<W, 256>>>(dA, K, dImg, dC, N, M, N, K, W); };
  launch();
  CHECK(hipDeviceSynchronize());
  std::vector<float> hC((size_t)M * N);
  CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  srand(7);
  for (int sidx = 0; sidx < 4000; ++sidx) {
    const int m = (sidx < 64) ? M - 1 - sidx : rand() % M, n = rand() % N;
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
  const int mt = (M + BM - 1) / BM;
  printf("S abl=%d sgb=%d tile %dx%d  M=%d N=%d K=%d : %8.1f us  %7.1f TFLOP/s alg (%7.1f bf16 issued)  tiles %d (%.2f rounds)  worst err/sum|ab| %.2e\n", ABL, SGB,
         BM, BN, M, N, K, us, tf, tf * 6, mt * nt, (double)mt * nt / W, worst);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const Shape shapes[] = {{25216, 1536, 384}, {25216, 384, 1536}, {25216, 1152, 384}, {25216, 384, 384}, {32768, 2048, 512}};
  for (const Shape& s : shapes) {
    std::vector<float> hA((size_t)s.M * s.K), hB((size_t)s.N * s.K);
    srand(1);
    for (auto& x : hA) x = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
    for (auto& x : hB) x = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
    float *dA, *dB, *dC;
    char* dImg;
    CHECK(hipMalloc(&dA, hA.size() * 4)); CHECK(hipMalloc(&dB, hB.size() * 4)); CHECK(hipMalloc(&dC, (size_t)s.M * s.N * 4));
    CHECK(hipMalloc(&dImg, (size_t)(s.N + 512) * s.K * 6 + 65536));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    run<4, 4>(s, ncu, dA, dB, dC, dImg, hA, hB);
    run<2, 4>(s, ncu, dA, dB, dC, dImg, hA, hB);
    run<4, 2>(s, ncu, dA, dB, dC, dImg, hA, hB);
    CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC)); CHECK(hipFree(dImg));
  }
  return 0;
}
