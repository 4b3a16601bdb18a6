// Lab (round 4): sustained rate of bf16 vs f16 MFMA (32x32x16 and 16x16x32) on RANDOM operands held in registers (power-limited
// regime), 64x64 accumulators per wave, 4 waves per workgroup, WPS workgroups per CU.  Question: does the f16 multiplier array
// (11-bit significands) hold a lower clock than the bf16 one (8-bit) on random data?
// hipcc --offload-arch=gfx950 -O3 mfma_f16_vs_bf16_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int F16> struct MM;
template <> struct MM<0> {
  typedef bf16x8 V;
  static __device__ __forceinline__ f32x16 m32(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x4 m16(V a, V b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MM<1> {
  typedef f16x8 V;
  static __device__ __forceinline__ f32x16 m32(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x4 m16(V a, V b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

template <int F16, int SHAPE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const uint4* __restrict__ data, float* __restrict__ out, int iters) {
  typedef typename MM<F16>::V V;
  const int t = blockIdx.x * 256 + threadIdx.x;
  V a[6], b[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    a[i] = __builtin_bit_cast(V, data[(t * 12 + i) & 0xffff]);
    b[i] = __builtin_bit_cast(V, data[(t * 12 + 6 + i) & 0xffff]);
  }
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = MM<F16>::m32(a[(q + i) % 6], b[(q * 2 + i) % 6], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = MM<F16>::m16(a[(q + i) % 6], b[(q * 2 + i) % 6], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  out[t] = s;
}

template <int F16, int SHAPE, int WPS>
double run(const uint4* d, float* o, int ncu, const char* what) {
  const int iters = 4000, blocks = ncu * WPS;
  k<F16, SHAPE, WPS><<<blocks, 256>>>(d, o, 200);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) k<F16, SHAPE, WPS><<<blocks, 256>>>(d, o, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double macs = (SHAPE == 32 ? 24.0 * 16384 : 48.0 * 8192) * iters * 5.0 * blocks * 4;
  const double tf = 2 * macs / (ms * 1e-3) / 1e12;
  printf("%s %s shape %dx%d wps %d: %.1f TFLOP/s\n", what, F16 ? "f16 " : "bf16", SHAPE, SHAPE, WPS, tf);
  return tf;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  std::vector<uint32_t> h(65536 * 4);
  uint4* d; float* o;
  CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMalloc(&o, (size_t)ncu * 2 * 256 * 4));
  // interleave the two dtypes round by round (same box, same thermal state)
  for (int round = 0; round < 3; ++round)
    for (int f16 = 0; f16 < 2; ++f16)
      for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (auto& x : h) {
          if (mode == 0) x = 0;
          else {
            uint32_t v = 0;
            for (int hlf = 0; hlf < 2; ++hlf) {
              // bf16 in [-1,1): sign | exponent 119..126 | 7-bit mantissa; f16 in [-1,1): sign | exponent 7..14 | 10-bit mantissa
              const uint32_t e = f16 ? (uint32_t)(((rand() & 1) << 15) | ((7 + rand() % 8) << 10) | (rand() & 1023))
                                     : (uint32_t)(((rand() & 1) << 15) | ((119 + rand() % 8) << 7) | (rand() & 127));
              v |= e << (16 * hlf);
            }
            x = v;
          }
        }
        CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        const char* what = mode ? "random" : "zeros ";
        if (f16) { run<1, 32, 1>(d, o, ncu, what); run<1, 16, 1>(d, o, ncu, what); run<1, 32, 2>(d, o, ncu, what); run<1, 16, 2>(d, o, ncu, what); }
        else { run<0, 32, 1>(d, o, ncu, what); run<0, 16, 1>(d, o, ncu, what); run<0, 32, 2>(d, o, ncu, what); run<0, 16, 2>(d, o, ncu, what); }
      }
  return 0;
}
