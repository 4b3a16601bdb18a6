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
__device__ __forceinline__ bool ofb_aligned16_dev(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// profiling hook (prof.hip): brackets a launch with events when enabled
void ofb_prof_pre(int tag, hipStream_t s, double work);
void ofb_prof_post(int tag, hipStream_t s);

typedef float ofb_f32x2 __attribute__((ext_vector_type(2)));

// Streaming hints: data that is read for the last time in a pass, or written for a reader that is a whole pass away (what the
// forward saves for the backward), goes past the caches as "non-temporal" so that it does not evict what the next kernels re-read
// (measured on the fc1 product: GELU' written non-temporal 167 -> 153 us).  -DOFB_LAB_NO_NT builds the plain forms for an A/B.
#ifdef OFB_LAB_NO_NT
#define OFB_NT_LOAD(p) (*(p))
#define OFB_NT_STORE(v, p) (*(p) = (v))
#else
#define OFB_NT_LOAD(p) __builtin_nontemporal_load(p)
#define OFB_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif

// GELU(erf) pieces from ONE exponential: Phi(x) = 0.5 (1 + erf(x / sqrt2)) via the Abramowitz-Stegun 7.1.26 erfc form
// (|error| <= 1.5e-7 on erf; measured 4e-7 max abs error on gelu / gelu' in fp32, below torch's own fp32 gelu error of
// 1.2e-6 at |x| ~ 8), phi(x) = exp(-x^2/2) / sqrt(2 pi).  ~16 VALU + v_exp + v_rcp instead of libm erff's ~45.
__device__ __forceinline__ void ofb_gelu_parts(float x, float& Phi, float& phi) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);       // v_rcp_f32 (1 ulp); __frcp_rn is a 12-instruction IEEE division
  const float e = __expf(-0.5f * x * x);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float half_erfc = 0.5f * poly * e;
  Phi = (x >= 0.f) ? 1.0f - half_erfc : half_erfc;
  phi = e * 0.39894228040143267794f;
}
__device__ __forceinline__ float ofb_gelu(float x) {
  float Phi, phi;
  ofb_gelu_parts(x, Phi, phi);
  return x * Phi;
}
// d/dx [x Phi(x)] = Phi(x) + x phi(x)
__device__ __forceinline__ float ofb_dgelu(float x) {
  float Phi, phi;
  ofb_gelu_parts(x, Phi, phi);
  return Phi + x * phi;
}

// Whole-wave reductions (every lane gets the result; call them with all 64 lanes active).  The tree is the xor butterfly 32, 16, 8, 4,
// 2, 1 - v[i] = v[i] (+) v[i ^ o] - and the results are BIT-identical to the `__shfl_xor` form it replaces (the operation is
// commutative, every lane combines the same two values at every level), but no step goes through the LDS crossbar (ds_bpermute_b32:
// an address VGPR, an lgkmcnt round trip of >= 100 cycles per level, six levels in a row): xor 32 / 16 are the gfx950 half / row
// exchanges v_permlane32_swap / v_permlane16_swap of (v, v) - afterwards the pair holds (own, partner) or (partner, own) -, xor 8 is
// row_ror:8, xor 4 two bank-masked row shifts, xor 2 / 1 quad permutes (DPP).  Measured on the LayerNorm kernels, round 6.
#ifdef OFB_LAB_SHFL_REDUCE
#define OFB_WAVE_REDUCE(v, OP)                                                                                  \
  _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) v = OP(v, __shfl_xor(v, o, 64));
#else
#define OFB_DPP_F(x, ctrl, bank) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), (ctrl), 0xF, (bank), false))
#define OFB_WAVE_REDUCE(v, OP)                                                                                  \
  {                                                                                                             \
    const unsigned ub_ = __builtin_bit_cast(unsigned, v);                                                       \
    const auto r32_ = __builtin_amdgcn_permlane32_swap(ub_, ub_, false, false);                                 \
    v = OP(__builtin_bit_cast(float, (unsigned)r32_[0]), __builtin_bit_cast(float, (unsigned)r32_[1]));        \
    const unsigned uc_ = __builtin_bit_cast(unsigned, v);                                                       \
    const auto r16_ = __builtin_amdgcn_permlane16_swap(uc_, uc_, false, false);                                 \
    v = OP(__builtin_bit_cast(float, (unsigned)r16_[0]), __builtin_bit_cast(float, (unsigned)r16_[1]));        \
    v = OP(v, OFB_DPP_F(v, 0x128, 0xF));                       /* row_ror:8 = lane ^ 8 */                        \
    {                                                          /* lane ^ 4: banks 0, 2 take lane + 4 (row_shl:4), banks 1, 3 lane - 4 (row_shr:4) */ \
      int t_ = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x104, 0xF, 0x5, false);                         \
      t_ = __builtin_amdgcn_update_dpp(t_, __builtin_bit_cast(int, v), 0x114, 0xF, 0xA, false);                 \
      v = OP(v, __builtin_bit_cast(float, t_));                                                                 \
    }                                                                                                           \
    v = OP(v, OFB_DPP_F(v, 0x4E, 0xF));                        /* quad_perm [2,3,0,1] = lane ^ 2 */              \
    v = OP(v, OFB_DPP_F(v, 0xB1, 0xF));                        /* quad_perm [1,0,3,2] = lane ^ 1 */              \
  }
#endif
// single butterfly levels v = OP(v, v[lane ^ o]) without the LDS crossbar (OP commutative; all 64 lanes active): the pieces of OFB_WAVE_REDUCE
#ifdef OFB_LAB_SHFL_REDUCE
#define OFB_XOR_STEP(v, OP, o) v = OP(v, __shfl_xor(v, o, 64))
#else
#define OFB_XOR_STEP(v, OP, o)                                                                                  \
  {                                                                                                             \
    if constexpr ((o) == 1) v = OP(v, OFB_DPP_F(v, 0xB1, 0xF));                                                 \
    else if constexpr ((o) == 2) v = OP(v, OFB_DPP_F(v, 0x4E, 0xF));                                            \
    else if constexpr ((o) == 4) {                                                                              \
      int t_ = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x104, 0xF, 0x5, false);                         \
      t_ = __builtin_amdgcn_update_dpp(t_, __builtin_bit_cast(int, v), 0x114, 0xF, 0xA, false);                 \
      v = OP(v, __builtin_bit_cast(float, t_));                                                                 \
    } else if constexpr ((o) == 8) v = OP(v, OFB_DPP_F(v, 0x128, 0xF));                                         \
    else if constexpr ((o) == 16) {                                                                             \
      const unsigned u_ = __builtin_bit_cast(unsigned, v);                                                      \
      const auto r_ = __builtin_amdgcn_permlane16_swap(u_, u_, false, false);                                   \
      v = OP(__builtin_bit_cast(float, (unsigned)r_[0]), __builtin_bit_cast(float, (unsigned)r_[1]));           \
    } else {                                                                                                    \
      const unsigned u_ = __builtin_bit_cast(unsigned, v);                                                      \
      const auto r_ = __builtin_amdgcn_permlane32_swap(u_, u_, false, false);                                   \
      v = OP(__builtin_bit_cast(float, (unsigned)r_[0]), __builtin_bit_cast(float, (unsigned)r_[1]));           \
    }                                                                                                           \
  }
#endif
__device__ __forceinline__ float ofb_add_(float a, float b) { return a + b; }
__device__ __forceinline__ float ofb_wave_sum(float v) {
  OFB_WAVE_REDUCE(v, ofb_add_)
  return v;
}
__device__ __forceinline__ float ofb_wave_max(float v) {
  OFB_WAVE_REDUCE(v, fmaxf)
  return v;
}

// Bijective XCD-aware remap (blocks b and b+8 share an XCD under round-robin dispatch): gives every
// XCD a contiguous run of logical tile ids so neighbouring tiles share operand panels in one L2.
__device__ __forceinline__ int ofb_xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}
