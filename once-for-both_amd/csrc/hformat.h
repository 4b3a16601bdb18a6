// H-format: a matrix as TWO f16 planes of a power-of-two scaled copy, the operand form of the f32-class GEMM / attention engines.
//
//   X[R][C] * 2^e = h1 + h2 (+ eps),  h1 = fl16(X 2^e), h2 = fl16(X 2^e - h1):  two 11-bit significands, the second one signed
//   against the first: 23 significant bits, |eps| <= 2^-23 |X 2^e| (RMS ~2^-25) while h2 is a normal f16; e is ONE exponent per
//   tensor, chosen by the producer from an upper bound b >= max|X| so that b 2^e lies in [2^14, 2^15) (f16 overflows at 65504).
//   Elements >= 2^-18 b keep that relative accuracy, smaller ones keep an ABSOLUTE accuracy of 2^-39 b (f16 subnormal spacing) - a
//   floor 2^15 times below what an f32 of the size of b resolves.  A product a b is issued as three v_mfma_f32_*_f16 terms
//   (h2 h1, h1 h2, h1 h1) with f32 accumulation; the dropped h2 h2 term is <= 2^-22 |a b| in the worst case and a zero-mean
//   2^-25 |a b| in RMS (the residuals' signs are those of rounding errors).  Per product: worst case 2^-21, RMS ~2^-24 - the size of
//   f32's own product rounding - and the accumulation is the f32 pipe's.  Measured against fp64 the engine is at or below a
//   k-ordered f32 fma chain on every operand class of tests/test_gpu_accuracy_class.py.
//
// Buffer layout (ofb_hformat_bytes): [header 256 B][granules].  Granule = 4 rows x 16 columns, 256 B: [plane h1 | h2][c % 16][r % 4]
// f16 (128 B per plane), stored [ceil(R/16)*4][ncb = ceil(C/16)].  Rows >= R / columns >= C inside the last granules are ZERO.
// Header (device memory, written by the producer kernel, read by the consumers' kernels - never by the host):
//   int32 e      planes hold X 2^e
//   f32   amax   upper bound of max|X| (exact where the producer measured it)
//   f32   rn2sq  upper bound of max_r sum_c X[r][c]^2 (0 = unknown: consumers use C amax^2)
//   f32   cn2sq  upper bound of max_c sum_r X[r][c]^2 (0 = unknown: consumers use R amax^2)
// The norms feed the Cauchy-Schwarz bound |sum_k a_k b_k| <= |a| |b| with which a GEMM that WRITES H-format chooses its output's e
// before the first tile is finished (csrc/gemm_h.hip: gemm_h_bound_kernel).
#pragma once
#include "ofb_common.h"

#define OFB_HGRAN 256
#define OFB_HHDR 256
#define OFB_H_EMAX 60            /* |e| <= 60: 2^-(ea + eb) stays a normal f32 */

typedef _Float16 ofb_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ofb_f16x8 __attribute__((ext_vector_type(8)));

struct ofb_hhdr { int32_t e; float amax, rn2sq, cn2sq; };

// exponent for an upper bound b of max|X|: b 2^e in [2^14, 2^15); b == 0 / non-finite / tiny: clamped
__host__ __device__ __forceinline__ int ofb_h_exp(float b) {
  if (!(b > 0.f) || !(b < 3.0e38f)) return 0;
  int ex;
  frexpf(b, &ex);                                  // b = m 2^ex, m in [0.5, 1)
  int e = 15 - ex;                                 // b 2^e = m 2^15 in [2^14, 2^15)
  return e > OFB_H_EMAX ? OFB_H_EMAX : (e < -OFB_H_EMAX ? -OFB_H_EMAX : e);
}
__device__ __forceinline__ float ofb_h_pow2(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }   // |e| <= 126

__device__ __forceinline__ unsigned ofb_pk_f16(float a, float b) {       // RNE, a -> low half
  ofb_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ofb_f16x2));
}
__device__ __forceinline__ float ofb_f16_lo(unsigned w) { return (float)__builtin_bit_cast(ofb_f16x2, w)[0]; }
__device__ __forceinline__ float ofb_f16_hi(unsigned w) { return (float)__builtin_bit_cast(ofb_f16x2, w)[1]; }
// two ALREADY SCALED values -> their h1 / h2 words
__device__ __forceinline__ void ofb_hsplit_pair(float a, float b, unsigned& h1, unsigned& h2) {
  h1 = ofb_pk_f16(a, b);
  h2 = ofb_pk_f16(a - ofb_f16_lo(h1), b - ofb_f16_hi(h1));
}
// rows 4 rg .. 4 rg + 3 of column c (already scaled) -> the 8-byte column slot of each plane; P = the planes' base (behind the header)
__device__ __forceinline__ void ofb_store_h4(char* __restrict__ planes, int ncb, int rg, int c, float v0, float v1, float v2, float v3) {
  char* slot = planes + ((size_t)rg * ncb + (c >> 4)) * OFB_HGRAN + (c & 15) * 8;
  unsigned a0, b0, a1, b1;
  ofb_hsplit_pair(v0, v1, a0, b0);
  ofb_hsplit_pair(v2, v3, a1, b1);
  *reinterpret_cast<uint2*>(slot) = make_uint2(a0, a1);
  *reinterpret_cast<uint2*>(slot + 128) = make_uint2(b0, b1);
}
__device__ __forceinline__ char* ofb_h_planes(void* P) { return (char*)P + OFB_HHDR; }
__device__ __forceinline__ const char* ofb_h_planes(const void* P) { return (const char*)P + OFB_HHDR; }
__device__ __forceinline__ const ofb_hhdr* ofb_h_hdr(const void* P) { return (const ofb_hhdr*)P; }

// max of non-negative floats through their bit patterns (ordered like unsigned integers).  (fmaxf drops NaN operands, so a NaN never
// reaches a bound; the NaN element itself converts to a NaN f16 and stays loud in the planes.)
// The current value is read first: thousands of blocks fold their maxima into ONE word, the atomics on it serialise in its L2 channel
// (1576 blocks x 2 words cost the LayerNorm-backward bound pass 30 of its 43 us), and after the first few arrivals almost nobody raises it.
__device__ __forceinline__ void ofb_atomic_max_pos(float* addr, float v) {
  unsigned* a = reinterpret_cast<unsigned*>(addr);
  if (__float_as_uint(v) > __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a, __float_as_uint(v));
}
__device__ __forceinline__ float ofb_wave_max_pos(float v) {
  OFB_WAVE_REDUCE(v, fmaxf)                                  // (ofb_common.h: the xor butterfly without the LDS crossbar)
  return v;
}
