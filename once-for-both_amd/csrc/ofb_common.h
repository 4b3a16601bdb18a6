// Shared device/host helpers for libofb_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ofb_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define OFB_WAVE 64

static inline int ofb_launch_status() { return (int)hipGetLastError(); }
static inline bool ofb_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int ofb_cdiv(int a, int b) { return (a + b - 1) / b; }

// profiling hook (prof.hip): brackets a launch with events when enabled
void ofb_prof_pre(int tag, hipStream_t s, double work);
void ofb_prof_post(int tag, hipStream_t s);

__device__ __forceinline__ float ofb_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// d/dx [0.5 x (1 + erf(x/sqrt2))] = 0.5 (1 + erf(x/sqrt2)) + x * exp(-x^2/2) / sqrt(2 pi)
__device__ __forceinline__ float ofb_dgelu(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

__device__ __forceinline__ float ofb_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float ofb_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware remap (blocks b and b+8 share an XCD under round-robin dispatch): gives every
// XCD a contiguous run of logical tile ids so neighbouring tiles share operand panels in one L2.
__device__ __forceinline__ int ofb_xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}
