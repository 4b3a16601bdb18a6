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

// ---------------------------------------------------------------------------------------------------------------
// RandAugment (timm rand_augment_transform('rand-m9-mstd0.5-inc1'), the reference's default --aa, search.py:123) on uint8 CHW images:
// one op per image and layer, Pillow semantics (ImageOps / ImageEnhance / Image.transform), checked against Pillow in the tests.
//   1 AutoContrast  2 Equalize  3 Invert  4 Posterize(iarg bits)  5 Solarize(iarg threshold)  6 SolarizeAdd(iarg add, threshold 128)
//   7 Color  8 Contrast  9 Brightness  10 Sharpness (farg factor: Image.blend(degenerate, image, factor))
//   11 Affine (m[6], iarg = resample 0 nearest / 2 bilinear / 3 bicubic; fill = fill_rgb): Rotate / ShearX / ShearY / TranslateX / Y
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }   // PIL RGB -> L

// per image: histogram of every channel (int32 [3][256]) and the sum of L
__global__ __launch_bounds__(256) void aug_stats_kernel(const uint8_t* __restrict__ img, int HW, int* __restrict__ hist, unsigned long long* __restrict__ lsum) {
  __shared__ int h[768];
  __shared__ unsigned long long ls[4];
  const int b = blockIdx.x, t = threadIdx.x;
  for (int i = t; i < 768; i += 256) h[i] = 0;
  __syncthreads();
  const uint8_t* p = img + (size_t)b * 3 * HW;
  unsigned long long acc = 0;
  for (int i = t; i < HW; i += 256) {
    const int r = p[i], g = p[HW + i], bl = p[2 * HW + i];
    atomicAdd(&h[r], 1); atomicAdd(&h[256 + g], 1); atomicAdd(&h[512 + bl], 1);
    acc += (unsigned long long)luma(r, g, bl);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((t & 63) == 0) ls[t >> 6] = acc;
  __syncthreads();
  for (int i = t; i < 768; i += 256) hist[(size_t)b * 768 + i] = h[i];
  if (t == 0) lsum[b] = ls[0] + ls[1] + ls[2] + ls[3];
}

__device__ __forceinline__ uint8_t blend8(int degenerate, int image, float alpha) {      // PIL ImagingBlend
  const float temp = (float)degenerate + mul_rn(alpha, (float)(image - degenerate));
  if (alpha >= 0.f && alpha <= 1.f) return (uint8_t)temp;
  if (temp <= 0.f) return 0;
  if (temp >= 255.f) return 255;
  return (uint8_t)temp;
}
__device__ __forceinline__ int xclip(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }
// Pillow Geometry.c BICUBIC (the transform filter, not the resize filter): v2 + d(-v1 + v3) + d^2(2(v1 - v2) + v3 - v4) + d^3(-v1 + v2 - v3 + v4)
__device__ __forceinline__ double pil_cubic(double v1, double v2, double v3, double v4, double d) {
  const double p1 = v2, p2 = -v1 + v3, p3 = 2 * (v1 - v2) + v3 - v4, p4 = -v1 + v2 - v3 + v4;
  return p1 + d * (p2 + d * (p3 + d * p4));
}

__global__ __launch_bounds__(256) void aug_apply_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const ofb_aug_op* __restrict__ ops,
                                                        const int* __restrict__ hist, const unsigned long long* __restrict__ lsum, int H, int W) {
  __shared__ uint8_t lut[768];
  __shared__ int s_mean;
  const int b = blockIdx.y, t = threadIdx.x, HW = H * W;
  const ofb_aug_op op = ops[b];
  const uint8_t* src = in + (size_t)b * 3 * HW;
  uint8_t* dst = out + (size_t)b * 3 * HW;
  const bool use_lut = op.op >= 1 && op.op <= 6;
  if (use_lut) {
    if (op.op == 1 || op.op == 2) {
      if (t < 3) {
        const int* h = hist + (size_t)b * 768 + 256 * t;
        uint8_t* l = lut + 256 * t;
        if (op.op == 1) {                                   // ImageOps.autocontrast, cutoff 0
          int lo = 0, hi = 255;
          while (lo < 256 && h[lo] == 0) ++lo;
          while (hi >= 0 && h[hi] == 0) --hi;
          if (hi <= lo) { for (int i = 0; i < 256; ++i) l[i] = (uint8_t)i; }
          else {
            const double scale = 255.0 / (double)(hi - lo), offset = -(double)lo * scale;
            for (int i = 0; i < 256; ++i) { int v = (int)((double)i * scale + offset); l[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
          }
        } else {                                            // ImageOps.equalize
          int nz = 0, last = 0; long long total = 0;
          for (int i = 0; i < 256; ++i) if (h[i]) { ++nz; last = h[i]; total += h[i]; }
          const long long step = nz <= 1 ? 0 : (total - last) / 255;
          if (step == 0) { for (int i = 0; i < 256; ++i) l[i] = (uint8_t)i; }
          else {
            long long n = step / 2;
            for (int i = 0; i < 256; ++i) { const long long q = n / step; l[i] = (uint8_t)(q > 255 ? 255 : q); n += h[i]; }   // Image.point clips the table to 8 bits
          }
        }
      }
    } else {
      for (int i = t; i < 768; i += 256) {
        const int v = i & 255;
        int o = v;
        if (op.op == 3) o = 255 - v;
        else if (op.op == 4) o = v & ~((1 << (8 - op.iarg)) - 1);
        else if (op.op == 5) o = v < op.iarg ? v : 255 - v;
        else if (op.op == 6) o = v < 128 ? min(255, v + op.iarg) : v;
        lut[i] = (uint8_t)o;
      }
    }
  }
  if (op.op == 8 && t == 0) s_mean = (int)((double)lsum[b] / (double)HW + 0.5);      // int(ImageStat.Stat(L).mean[0] + 0.5)
  __syncthreads();
  const float k1 = 1.f / 13.f, k5 = 5.f / 13.f;              // ImageFilter.SMOOTH, kernel / 13 in float32
  for (int i = blockIdx.x * 256 + t; i < HW; i += gridDim.x * 256) {
    const int y = i / W, x = i - y * W;
    int px[3] = {src[i], src[HW + i], src[2 * HW + i]};
    int o[3];
    if (op.op == 0) { o[0] = px[0]; o[1] = px[1]; o[2] = px[2]; }
    else if (use_lut) { o[0] = lut[px[0]]; o[1] = lut[256 + px[1]]; o[2] = lut[512 + px[2]]; }
    else if (op.op == 7) { const int g = luma(px[0], px[1], px[2]); for (int c = 0; c < 3; ++c) o[c] = blend8(g, px[c], op.farg); }
    else if (op.op == 8) { for (int c = 0; c < 3; ++c) o[c] = blend8(s_mean, px[c], op.farg); }
    else if (op.op == 9) { for (int c = 0; c < 3; ++c) o[c] = blend8(0, px[c], op.farg); }
    else if (op.op == 10) {
      for (int c = 0; c < 3; ++c) {
        int d = px[c];
        if (y > 0 && y < H - 1 && x > 0 && x < W - 1) {       // ImagingFilter3x3: rows y+1, y, y-1; borders copied
          const uint8_t* pc = src + (size_t)c * HW;
          float ss = 0.5f;
          ss += mul_rn((float)pc[i + W - 1], k1) + mul_rn((float)pc[i + W], k1) + mul_rn((float)pc[i + W + 1], k1);
          ss += mul_rn((float)pc[i - 1], k1) + mul_rn((float)pc[i], k5) + mul_rn((float)pc[i + 1], k1);
          ss += mul_rn((float)pc[i - W - 1], k1) + mul_rn((float)pc[i - W], k1) + mul_rn((float)pc[i - W + 1], k1);
          d = ss <= 0.f ? 0 : (ss >= 255.f ? 255 : (int)ss);
        }
        o[c] = blend8(d, px[c], op.farg);
      }
    } else {                                                 // 11: Image.transform(size, AFFINE, m, resample, fillcolor)
      const double xin0 = op.m[0] * (x + 0.5) + op.m[1] * (y + 0.5) + op.m[2], yin0 = op.m[3] * (x + 0.5) + op.m[4] * (y + 0.5) + op.m[5];
      const bool inside = xin0 >= 0.0 && xin0 < (double)W && yin0 >= 0.0 && yin0 < (double)H;
      for (int c = 0; c < 3; ++c) {
        const uint8_t* pc = src + (size_t)c * HW;
        int v = op.fill[c];
        if (inside) {
          if (op.iarg == 0) {
            v = pc[(int)yin0 * W + (int)xin0];
          } else if (op.iarg == 2) {
            const double xin = xin0 - 0.5, yin = yin0 - 0.5;
            const int xf = (int)floor(xin), yf = (int)floor(yin);
            const double dx = xin - xf, dy = yin - yf;
            const int x0 = xclip(xf, W), x1 = xclip(xf + 1, W), y0 = xclip(yf, H), y1 = xclip(yf + 1, H);
            const double v1 = pc[y0 * W + x0] + dx * ((double)pc[y0 * W + x1] - pc[y0 * W + x0]);
            const double v2 = pc[y1 * W + x0] + dx * ((double)pc[y1 * W + x1] - pc[y1 * W + x0]);
            const double r = v1 + dy * (v2 - v1);
            v = (int)r;
          } else {
            const double xin = xin0 - 0.5, yin = yin0 - 0.5;
            const int xf = (int)floor(xin), yf = (int)floor(yin);
            const double dx = xin - xf, dy = yin - yf;
            double rows[4];
            for (int ky = 0; ky < 4; ++ky) {
              const uint8_t* pr = pc + xclip(yf - 1 + ky, H) * W;
              rows[ky] = pil_cubic(pr[xclip(xf - 1, W)], pr[xclip(xf, W)], pr[xclip(xf + 1, W)], pr[xclip(xf + 2, W)], dx);
            }
            const double r = pil_cubic(rows[0], rows[1], rows[2], rows[3], dy);
            v = r <= 0.0 ? 0 : (r >= 255.0 ? 255 : (int)r);
          }
        }
        o[c] = v;
      }
    }
    dst[i] = (uint8_t)o[0]; dst[HW + i] = (uint8_t)o[1]; dst[2 * HW + i] = (uint8_t)o[2];
  }
}

// ToTensor + Normalize of uint8 CHW pixels
__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n_per_chan, int64_t total, float m0, float m1,
                                                           float m2, float s0, float s1, float s2) {
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)((i / n_per_chan) % 3);
    out[i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)in[i], 255.f), mean[c]), sd[c]);
  }
}

}  // namespace

extern "C" int ofb_randaug_layer(const uint8_t* in, uint8_t* out, const ofb_aug_op* ops_dev, int32_t B, int32_t H, int32_t W, int32_t* hist_scratch,
                                 uint64_t* lsum_scratch, void* stream) {
  if (!in || !out || in == out || !ops_dev || !hist_scratch || !lsum_scratch || B <= 0 || H <= 2 || W <= 2) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(aug_stats_kernel, dim3(B), dim3(256), 0, s, in, H * W, hist_scratch, (unsigned long long*)lsum_scratch);
  hipLaunchKernelGGL(aug_apply_kernel, dim3(16, B), dim3(256), 0, s, in, out, ops_dev, (const int*)hist_scratch, (const unsigned long long*)lsum_scratch, H, W);
  return ofb_launch_status();
}

extern "C" int ofb_normalize_u8(const uint8_t* in, float* out, int32_t B, int32_t H, int32_t W, const float* mean3, const float* std3, void* stream) {
  if (!in || !out || !mean3 || !std3 || B <= 0 || H <= 0 || W <= 0) return OFB_EINVAL;
  const int64_t total = (int64_t)B * 3 * H * W;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(normalize_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, (int64_t)H * W, total, mean3[0], mean3[1], mean3[2], std3[0],
                     std3[1], std3[2]);
  return ofb_launch_status();
}

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
