// Micro-lab: which ingredient of a GEMM main loop costs f32-MFMA issue rate? (diagnostic only, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDP 20
template <int LDSREAD, int BARRIER, int LDSWRITE, int GLOAD>
__global__ __launch_bounds__(256, 3) void lab(const float* __restrict__ src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * 128 * LDP];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
  for (int i = t; i < 2 * 2 * 128 * LDP; i += 256) lds[i] = (float)(i % 7) * 0.125f;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 af[2][2], bf[2][2];
  for (int q = 0; q < 2; ++q) for (int m = 0; m < 2; ++m) { af[m][q] = f32x4{1.f, 2.f, 3.f, 4.f}; bf[m][q] = f32x4{.5f, .25f, .125f, 1.f}; }
  f32x4 g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
  const float* gp = src + (size_t)blockIdx.x * 4096 + t * 4;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (GLOAD == 2) {   // GEMM-shaped: 16 rows x 64 B per wave-instruction from a [25216 x 384] matrix, K swept 16 at a time
      const int tile = (blockIdx.x + (it / 24) * gridDim.x) % 1773, k0 = (it % 24) * 16;
      const float* ap = src + ((size_t)((tile / 9) * 128 + (t >> 2)) * 384 + k0 + ((t & 3) << 2));
      const float* bp = src + (size_t)25216 * 384 + ((size_t)((tile % 9) * 128 + (t >> 2)) * 384 + k0 + ((t & 3) << 2));
      g0 = *reinterpret_cast<const f32x4*>(ap) + *reinterpret_cast<const f32x4*>(ap + 64 * 384);
      g1 = *reinterpret_cast<const f32x4*>(bp) + *reinterpret_cast<const f32x4*>(bp + 64 * 384);
    } else if (GLOAD) { g0 = *reinterpret_cast<const f32x4*>(gp + (it & 3) * 1024); g1 = *reinterpret_cast<const f32x4*>(gp + ((it + 1) & 3) * 1024 + 2048); }
    if (LDSREAD) {
      const float* a_s = lds + buf * 128 * LDP + ((w >> 1) * 64 + l31) * LDP + 4 * h;
      const float* b_s = lds + 2 * 128 * LDP + buf * 128 * LDP + ((w & 1) * 64 + l31) * LDP + 4 * h;
      for (int q = 0; q < 2; ++q) {
        af[0][q] = *reinterpret_cast<const f32x4*>(a_s + 8 * q);
        af[1][q] = *reinterpret_cast<const f32x4*>(a_s + 32 * LDP + 8 * q);
        bf[0][q] = *reinterpret_cast<const f32x4*>(b_s + 8 * q);
        bf[1][q] = *reinterpret_cast<const f32x4*>(b_s + 32 * LDP + 8 * q);
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][q][j], bf[0][q][j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][q][j], bf[1][q][j], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][q][j], bf[0][q][j], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][q][j], bf[1][q][j], acc[3], 0, 0, 0);
      }
    if (LDSWRITE) {
      f32x4 v0 = GLOAD ? g0 : f32x4{1, 2, 3, (float)it}, v1 = GLOAD ? g1 : f32x4{2, 3, 4, (float)it};
      float* d = lds + (buf ^ 1) * 128 * LDP;
      *reinterpret_cast<f32x4*>(&d[(t >> 2) * LDP + ((t & 3) << 2)]) = v0;
      *reinterpret_cast<f32x4*>(&d[((t + 256) >> 2) * LDP + ((t & 3) << 2)]) = v1;
      *reinterpret_cast<f32x4*>(&d[2 * 128 * LDP + (t >> 2) * LDP + ((t & 3) << 2)]) = v1;
      *reinterpret_cast<f32x4*>(&d[2 * 128 * LDP + ((t + 256) >> 2) * LDP + ((t & 3) << 2)]) = v0;
    }
    if (BARRIER) __syncthreads();
  }
  float s = g0[0] + g1[1];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}
template <int A, int B, int C, int D>
void run(const char* name, const float* src, float* out, int blocks) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((lab<A, B, C, D>), dim3(blocks), dim3(256), 0, 0, src, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((lab<A, B, C, D>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
  hipEventRecord(e1, 0); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * iters * 32 * 4096.0;
  printf("%-44s blocks %4d: %7.2f ms  %6.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
}
int main() {
  float *src, *out; hipMalloc(&src, (size_t)(25216 + 1152 + 256) * 384 * 4); hipMalloc(&out, 64);
  hipMemset(src, 0, (size_t)(25216 + 1152 + 256) * 384 * 4);
  for (int blocks : {256, 512, 768}) {
    run<0, 0, 0, 0>("mfma only", src, out, blocks);
    run<1, 0, 0, 0>("+ lds reads", src, out, blocks);
    run<1, 1, 0, 0>("+ lds reads + barrier", src, out, blocks);
    run<1, 1, 1, 0>("+ lds reads + barrier + lds writes", src, out, blocks);
    run<1, 1, 1, 1>("+ lds reads + barrier + lds writes + gload", src, out, blocks);
    run<1, 1, 1, 2>("+ lds rd + barrier + lds wr + GEMM-shaped gload", src, out, blocks);
  }
  return 0;
}
