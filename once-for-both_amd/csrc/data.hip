// Input side of the training step, all HBM-bound byte / elementwise work:
//   * Mixup / CutMix on the resident batch, in place, pair (b, B-1-b) handled by one thread so no copy of the flipped batch
//     is made (timm Mixup as used by engine.py:35-36,99-100; search.py:651-655; finetune.py:310);
//   * the smoothed / mixed soft targets and SoftTargetCrossEntropy forward + gradient (search.py:581-583; finetune.py:390);
//   * RandomResizedCrop + horizontal flip + ToTensor + Normalize of the reference's build_transform (datasets.py:127-163)
//     from decoded uint8 HWC images: a separable, antialiased (PIL-style) triangle / cubic resampler in two passes.
#include "ofb_common.h"

// torch rounds every elementwise product and sum separately; HIP contracts a*b + c into an FMA by default (and its __fmul_rn
// is a plain multiply), so contraction is switched off for this whole file: the mixes and the normalisation are bit-exact.
#pragma clang fp contract(off)
// ... and each product is additionally pinned in a register (hipcc still fused the pair above into v_fma / v_fmac)
__device__ __forceinline__ float mul_rn(float a, float b) {
  float p = a * b;
  asm volatile("" : "+v"(p));
  return p;
}

namespace {

// ---------------------------------------------------------------------------------------------------------------
// Mixup / CutMix
// ---------------------------------------------------------------------------------------------------------------
// new_a = a*lam_a + b*(1-lam_a) outside CutMix samples; inside a CutMix sample's box the partner's ORIGINAL value.
// Products and the sum are rounded separately (torch: x.mul_(lam).add_(x.flip(0).mul_(1-lam))), hence no FMA.
__device__ __forceinline__ float mix_one(float self, float other, const ofb_mix_param& p, int y, int x) {
  if (p.use_cutmix) return (y >= p.yl && y < p.yh && x >= p.xl && x < p.xh) ? other : self;
  if (p.lam == 1.0f) return self;
  return mul_rn(self, p.lam) + mul_rn(other, p.one_minus_lam);
}

__global__ __launch_bounds__(256) void mixup_kernel(float* __restrict__ x, const ofb_mix_param* __restrict__ params, int B, int C, int H,
                                                    int W) {
  const int pair = blockIdx.y, i = pair, j = B - 1 - pair;
  const ofb_mix_param pi = params[i], pj = params[j];
  const int64_t chw = (int64_t)C * H * W;
  float* xi = x + (int64_t)i * chw;
  float* xj = x + (int64_t)j * chw;
  const bool vec = (W & 3) == 0;
  if (vec) {
    const int64_t n4 = chw >> 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
      const int64_t o = e << 2;
      const int xx = (int)(o % W), yy = (int)((o / W) % H);
      f32x4 a = *reinterpret_cast<f32x4*>(xi + o), b = *reinterpret_cast<f32x4*>(xj + o), na, nb;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        na[k] = mix_one(a[k], b[k], pi, yy, xx + k);
        nb[k] = mix_one(b[k], a[k], pj, yy, xx + k);
      }
      *reinterpret_cast<f32x4*>(xi + o) = na;
      if (i != j) *reinterpret_cast<f32x4*>(xj + o) = nb;
    }
  } else {
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < chw; o += (int64_t)gridDim.x * 256) {
      const int xx = (int)(o % W), yy = (int)((o / W) % H);
      const float a = xi[o], b = xj[o];
      xi[o] = mix_one(a, b, pi, yy, xx);
      if (i != j) xj[o] = mix_one(b, a, pj, yy, xx);
    }
  }
}

// out[b][c] = y1*lam_b + y2*(1-lam_b), y1 = smoothed one-hot of labels[b], y2 of labels[B-1-b] (timm mixup_target)
__global__ __launch_bounds__(256) void mix_targets_kernel(const int64_t* __restrict__ labels, const ofb_mix_param* __restrict__ params,
                                                          float* __restrict__ out, int B, int Cn, float on_value, float off_value) {
  const int b = blockIdx.x;
  const int y1 = (int)labels[b], y2 = (int)labels[B - 1 - b];
  const float lam = params[b].lam, oml = params[b].one_minus_lam;
  for (int c = threadIdx.x; c < Cn; c += 256) {
    const float v1 = (c == y1) ? on_value : off_value, v2 = (c == y2) ? on_value : off_value;
    out[(size_t)b * Cn + c] = mul_rn(v1, lam) + mul_rn(v2, oml);
  }
}

// SoftTargetCrossEntropy: loss = mean_b sum_c -t[b][c] log_softmax(x[b])[c]; grad = (p * sum_c t - t) / B
__global__ __launch_bounds__(256) void soft_ce_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                      float* __restrict__ row_loss, float* __restrict__ grad, int Bn, int Cn) {
  __shared__ float red[4];
  __shared__ float bc[3];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* x = logits + (size_t)b * Cn;
  const float* tg = target + (size_t)b * Cn;
  float m = -INFINITY;
  for (int c = t; c < Cn; c += 256) m = fmaxf(m, x[c]);
  m = ofb_wave_max(m);
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f, st = 0.f, stx = 0.f;
  for (int c = t; c < Cn; c += 256) {
    se += expf(x[c] - m);
    st += tg[c];
    stx += tg[c] * (x[c] - m);
  }
  float vals[3] = {se, st, stx};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float v = ofb_wave_sum(vals[k]);
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    if (t == 0) bc[k] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
  }
  const float lse_m = logf(bc[0]);                 // lse - m
  if (t == 0) row_loss[b] = bc[1] * lse_m - bc[2]; // sum_c t (lse - x) = sum t * (lse - m) - sum t (x - m)
  const float invB = 1.0f / (float)Bn, sumt = bc[1];
  for (int c = t; c < Cn; c += 256) {
    const float p = expf(x[c] - m - lse_m);
    grad[(size_t)b * Cn + c] = (p * sumt - tg[c]) * invB;
  }
}

__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  s = ofb_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

// ---------------------------------------------------------------------------------------------------------------
// RandomResizedCrop (+flip) resampler, PIL semantics: output pixel o of a pass covers input centre
// (o + 0.5) * scale + box0 with the filter stretched by max(scale, 1) (antialias), weights normalised to 1; the
// 8-bit intermediate and result are rounded half up and clipped like ImagingResample's clip8.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float filt(float x, int cubic) {
  x = fabsf(x);
  if (!cubic) return x < 1.f ? 1.f - x : 0.f;
  const float a = -0.5f;                                   // PIL bicubic_filter
  if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
  return 0.f;
}
struct Taps { int lo, n; float scale, fscale, center0, inv_fs; };
__device__ __forceinline__ Taps taps_for(int o, int in_size, float box0, float box_len, int out_size, int cubic) {
  Taps t;
  t.scale = box_len / (float)out_size;
  t.fscale = fmaxf(t.scale, 1.f);
  const float support = (cubic ? 2.f : 1.f) * t.fscale;
  t.center0 = box0 + ((float)o + 0.5f) * t.scale;
  int lo = (int)(t.center0 - support + 0.5f), hi = (int)(t.center0 + support + 0.5f);
  if (lo < 0) lo = 0;
  if (hi > in_size) hi = in_size;
  t.lo = lo; t.n = hi - lo; t.inv_fs = 1.f / t.fscale;
  return t;
}
__device__ __forceinline__ float clip8(float v) {
  v = floorf(v + 0.5f);
  return fminf(fmaxf(v, 0.f), 255.f);
}

// pass 1 (horizontal): tmp[b][r][ox][c] (uint8) for the rows r of the vertical pass's support band
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, const ofb_crop_param* __restrict__ params,
                                                       uint8_t* __restrict__ tmp, int S, int max_rows) {
  const int b = blockIdx.y;
  const ofb_crop_param p = params[b];
  const uint8_t* img = src + p.offset;
  // rows needed by the vertical pass: PIL computes the band from the vertical coefficients' bounds
  const Taps t0 = taps_for(0, p.src_h, (float)p.top, (float)p.height, S, p.cubic), t1 = taps_for(S - 1, p.src_h, (float)p.top, (float)p.height, S, p.cubic);
  const int r0 = t0.lo, nrows = t1.lo + t1.n - r0;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < nrows * S; e += gridDim.x * 256) {
    const int r = e / S, ox = e - r * S;
    const Taps t = taps_for(ox, p.src_w, (float)p.left, (float)p.width, S, p.cubic);
    float ws = 0.f, acc[3] = {0.f, 0.f, 0.f};
    const uint8_t* row = img + ((size_t)(r0 + r) * p.src_w + t.lo) * 3;
    for (int k = 0; k < t.n; ++k) {
      const float w = filt(((float)(t.lo + k) - t.center0 + 0.5f) * t.inv_fs, p.cubic);
      ws += w;
      acc[0] += w * (float)row[3 * k]; acc[1] += w * (float)row[3 * k + 1]; acc[2] += w * (float)row[3 * k + 2];
    }
    const float inv = ws != 0.f ? 1.f / ws : 0.f;
    uint8_t* o = tmp + (((size_t)b * max_rows + r) * S + ox) * 3;
    o[0] = (uint8_t)clip8(acc[0] * inv); o[1] = (uint8_t)clip8(acc[1] * inv); o[2] = (uint8_t)clip8(acc[2] * inv);
  }
}

// pass 2 (vertical) + flip + ToTensor + Normalize: out[b][c][oy][ox] = ((v/255) - mean[c]) / std[c]
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* __restrict__ tmp, const ofb_crop_param* __restrict__ params,
                                                            float* __restrict__ out, uint8_t* __restrict__ out_u8, int S, int max_rows,
                                                            float m0, float m1, float m2, float s0, float s1, float s2) {
  const int b = blockIdx.y;
  const ofb_crop_param p = params[b];
  const Taps tfirst = taps_for(0, p.src_h, (float)p.top, (float)p.height, S, p.cubic);
  const int r0 = tfirst.lo;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  for (int e = blockIdx.x * 256 + threadIdx.x; e < S * S; e += gridDim.x * 256) {
    const int oy = e / S, ox = e - oy * S;
    const Taps t = taps_for(oy, p.src_h, (float)p.top, (float)p.height, S, p.cubic);
    float ws = 0.f, acc[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < t.n; ++k) {
      const float w = filt(((float)(t.lo + k) - t.center0 + 0.5f) * t.inv_fs, p.cubic);
      const uint8_t* px = tmp + (((size_t)b * max_rows + (t.lo + k - r0)) * S + ox) * 3;
      ws += w;
      acc[0] += w * (float)px[0]; acc[1] += w * (float)px[1]; acc[2] += w * (float)px[2];
    }
    const float inv = ws != 0.f ? 1.f / ws : 0.f;
    const int dx = p.flip ? S - 1 - ox : ox;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v8 = clip8(acc[c] * inv);
      if (out_u8) out_u8[(((size_t)b * 3 + c) * S + oy) * S + dx] = (uint8_t)v8;
      if (out) out[(((size_t)b * 3 + c) * S + oy) * S + dx] = __fdiv_rn(__fsub_rn(__fdiv_rn(v8, 255.f), mean[c]), sd[c]);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// RandomErasing, mode 'pixel' (timm RandomErasing as built by datasets.build_transform: reprob 0.25, one rectangle per image,
// every pixel of it replaced by N(0, 1) noise).  The rectangles are drawn on the host like the library does; the noise comes from
// a counter-based generator so that no noise tensor crosses HBM: Philox4x32-10 keyed by (seed, sample), counter = element index
// / 4, Box-Muller on the four 32-bit words.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0, 1)

__global__ __launch_bounds__(256) void erase_kernel(float* __restrict__ x, const ofb_erase_param* __restrict__ params, int C, int H, int W,
                                                    unsigned seed_lo, unsigned seed_hi) {
  const int b = blockIdx.y;
  const ofb_erase_param p = params[b];
  if (p.h <= 0 || p.w <= 0) return;
  const int n = C * p.h * p.w, groups = (n + 3) >> 2;
  for (int gidx = blockIdx.x * 256 + threadIdx.x; gidx < groups; gidx += gridDim.x * 256) {
    unsigned r[4];
    philox4x32_10((unsigned)gidx, 0u, (unsigned)b, 0u, seed_lo, seed_hi, r);
    float z[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float rad = sqrtf(-2.0f * logf(u01(r[2 * k]))), ang = 6.28318530717958647692f * u01(r[2 * k + 1]);
      z[2 * k] = rad * cosf(ang); z[2 * k + 1] = rad * sinf(ang);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = 4 * gidx + k;
      if (e >= n) break;
      const int c = e / (p.h * p.w), rem = e - c * p.h * p.w, yy = rem / p.w, xx = rem - yy * p.w;
      x[(((size_t)b * C + c) * H + p.top + yy) * W + p.left + xx] = z[k];
    }
  }
}

}  // namespace

extern "C" int ofb_random_erase(float* x, const ofb_erase_param* params_dev, int32_t B, int32_t C, int32_t H, int32_t W, uint64_t seed,
                                void* stream) {
  if (!x || !params_dev || B <= 0 || C <= 0 || H <= 0 || W <= 0) return OFB_EINVAL;
  hipLaunchKernelGGL(erase_kernel, dim3(32, B), dim3(256), 0, (hipStream_t)stream, x, params_dev, C, H, W, (unsigned)seed,
                     (unsigned)(seed >> 32));
  return ofb_launch_status();
}

extern "C" int ofb_mixup_batch(float* x, const ofb_mix_param* params_dev, int32_t B, int32_t C, int32_t H, int32_t W, void* stream) {
  if (!x || !params_dev || B <= 0 || C <= 0 || H <= 0 || W <= 0) return OFB_EINVAL;
  if (!ofb_aligned16(x)) return OFB_EINVAL;
  const int64_t chw = (int64_t)C * H * W;
  int bx = (int)((chw / 4 + 255) / 256);
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(mixup_kernel, dim3(bx, (B + 1) / 2), dim3(256), 0, (hipStream_t)stream, x, params_dev, B, C, H, W);
  return ofb_launch_status();
}

extern "C" int ofb_mixup_targets(const int64_t* labels, const ofb_mix_param* params_dev, float* out, int32_t B, int32_t num_classes,
                                 float on_value, float off_value, void* stream) {
  if (!labels || !params_dev || !out || B <= 0 || num_classes <= 0) return OFB_EINVAL;
  hipLaunchKernelGGL(mix_targets_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, labels, params_dev, out, B, num_classes, on_value,
                     off_value);
  return ofb_launch_status();
}

extern "C" int ofb_soft_cross_entropy(const float* logits, const float* target, float* row_loss, float* loss, float* grad, int32_t B,
                                      int32_t C, void* stream) {
  if (!logits || !target || !row_loss || !loss || !grad || B <= 0 || C <= 0) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(soft_ce_kernel, dim3(B), dim3(256), 0, s, logits, target, row_loss, grad, B, C);
  hipLaunchKernelGGL(mean_rows_kernel, dim3(1), dim3(256), 0, s, (const float*)row_loss, B, loss);
  return ofb_launch_status();
}

extern "C" int64_t ofb_crop_resize_scratch_bytes(int32_t B, int32_t out_size, int32_t max_src_h) {
  if (B <= 0 || out_size <= 0 || max_src_h <= 0) return 0;
  return (int64_t)B * max_src_h * out_size * 3;
}

extern "C" int ofb_crop_resize_norm(const uint8_t* src, const ofb_crop_param* params_dev, int32_t B, int32_t out_size, int32_t max_src_h,
                                    const float* mean3, const float* std3, float* out, uint8_t* out_u8, uint8_t* scratch, void* stream) {
  if (!src || !params_dev || !scratch || (!out && !out_u8) || !mean3 || !std3 || B <= 0 || out_size <= 0 || max_src_h <= 0) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int bx = (max_src_h * out_size + 255) / 256;
  if (bx > 128) bx = 128;
  hipLaunchKernelGGL(resize_h_kernel, dim3(bx, B), dim3(256), 0, s, src, params_dev, scratch, out_size, max_src_h);
  int bv = (out_size * out_size + 255) / 256;
  if (bv > 128) bv = 128;
  hipLaunchKernelGGL(resize_v_norm_kernel, dim3(bv, B), dim3(256), 0, s, (const uint8_t*)scratch, params_dev, out, out_u8, out_size,
                     max_src_h, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return ofb_launch_status();
}
