// Lab: f32 GEMM C[M][N] = A[M][K] * B[N][K]^T on bf16 MFMA with an exact 3-way bf16 split of both operands and the 6
// leading product terms (hh, hm, mh, hl, lh, mm), f32 accumulate.  Standalone feasibility / rate probe.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/bf16x6_lab.hip -o /tmp/bf16x6_lab && /tmp/bf16x6_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

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
#pragma unroll
    for (int q = 6 - TERMS; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][TA[q]], fb[j][TB[q]], acc[i][j], 0, 0, 0);
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

#ifndef KERNEL
#define KERNEL gemm_bf16x6
#endif
#define STR2(x) #x
#define STR(x) STR2(x)
#define NAME STR(KERNEL) " pad=" STR(PAD2) " sched=" STR(SCHED) " order=" STR(ORDER) " abl=" STR(ABL)
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
  for (int i = 0; i < 1500; ++i) hipLaunchKernelGGL(KERNEL, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);   // clock ramp
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 30;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(KERNEL, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
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
  printf("%s TERMS=%d M=%d N=%d K=%d: %.3f ms  %.1f TF (algorithmic)  max|err|/sum|ab| = %.2e (f32 fma chain: %.2e)\n", NAME, TERMS, M, N, K, ms,
         2.0 * M * N * K / ms / 1e9, worst, worst32);
  return 0;
}
