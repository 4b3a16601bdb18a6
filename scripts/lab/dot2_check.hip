// Lab: does v_dot2_f32_bf16(pair, (-1,0)|(0,-1), x) return x - (float)pair.half exactly?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = x[2 * i], b = x[2 * i + 1];
  f32x2 v = {a, b};
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  float ra, rb;
  asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(ra) : "v"(hi), "s"(0x0000bf80u), "v"(a));
  asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(rb) : "v"(hi), "s"(0xbf800000u), "v"(b));
  out[4 * i] = ra; out[4 * i + 1] = rb;
  out[4 * i + 2] = a - __uint_as_float(hi << 16); out[4 * i + 3] = b - __uint_as_float(hi & 0xffff0000u);
}
int main() {
  const int n = 1 << 16;
  float* h = (float*)malloc(2 * n * 4);
  srand(3);
  for (int i = 0; i < 2 * n; ++i) { float s = ldexpf((float)rand() / RAND_MAX * 2 - 1, rand() % 60 - 40); h[i] = s; }
  float *dx, *dout; hipMalloc(&dx, 2 * n * 4); hipMalloc(&dout, 4 * n * 4);
  hipMemcpy(dx, h, 2 * n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  float* o = (float*)malloc(4 * n * 4);
  hipMemcpy(o, dout, 4 * n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 2; ++j)
      if (memcmp(&o[4 * i + j], &o[4 * i + 2 + j], 4) != 0) { if (bad < 8) printf("x=(%g,%g) half %d: dot2 %g  ref %g\n", h[2 * i], h[2 * i + 1], j, o[4 * i + j], o[4 * i + 2 + j]); ++bad; }
  printf("mismatches: %d of %d\n", bad, 2 * n);
  return 0;
}
