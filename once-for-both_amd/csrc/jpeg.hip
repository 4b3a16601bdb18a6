// JPEG decode for the input side of the step (reference datasets.py:90-125 reads ImageNet through torchvision's
// ImageFolder + PIL's default_loader, i.e. libjpeg-turbo).  Hybrid split, chosen by where each stage parallelises:
//   host (csrc/jpeg_host.cpp, plain C++): marker parsing and the ENTROPY stage - Huffman decoding is a serial bit stream per image, so it
//        runs on the loader's CPU threads (ofb_jpeg_parse / ofb_jpeg_decode_coefficients: re-entrant, no global state) and hands
//        over de-zigzagged int16 coefficient blocks;
//   device: dequantisation + the 8x8 inverse DCT (one thread per block), then chroma upsampling + YCbCr -> RGB (one thread per
//        pixel) for a whole batch of images per launch, writing the decoded uint8 HWC pixels DeviceTransform starts from.
// Arithmetic restates libjpeg(-turbo)'s default decompression path bit for bit, because that is what the reference's pixels are
// made by: jidctint.c (JDCT_ISLOW: 13-bit constants, two passes), jdsample.c fancy ("triangle") upsampling h2v1 / h2v2 / h1v2
// with edge replication, jdcolor.c fixed-point YCbCr -> RGB.  Scope: baseline / extended-sequential Huffman JPEG (SOF0 / SOF1),
// 8-bit, 1 (grayscale -> RGB) or 3 (YCbCr) components, any scan structure, restart intervals.  Progressive (SOF2), arithmetic
// coding, CMYK and 12-bit files are rejected with OFB_ELIMIT (the loader may hand those few files to another decoder).
#include "ofb_common.h"
#include <string.h>

// ---------------------------------------------------------------------------------------------------------------------------
namespace {

#define JCONST_BITS 13
#define JPASS1_BITS 2
#define JFIX_0_298631336 2446
#define JFIX_0_390180644 3196
#define JFIX_0_541196100 4433
#define JFIX_0_765366865 6270
#define JFIX_0_899976223 7373
#define JFIX_1_175875602 9633
#define JFIX_1_501321110 12299
#define JFIX_1_847759065 15137
#define JFIX_1_961570560 16069
#define JFIX_2_053119869 16819
#define JFIX_2_562915447 20995
#define JFIX_3_072711026 25172
typedef long long jlong;                                   // libjpeg's JLONG is 64 bits wide on LP64 hosts: no overflow on odd streams
__device__ __forceinline__ int jdescale(jlong x, int n) { return (int)((x + ((jlong)1 << (n - 1))) >> n); }

// jidctint.c jpeg_idct_islow on one 8-vector: in = dequantised values of a column (pass 1) or a workspace row (pass 2)
__device__ __forceinline__ void idct8(const int (&in)[8], int (&out)[8], int shift) {
  jlong z2 = in[2], z3 = in[6];
  jlong z1 = (z2 + z3) * JFIX_0_541196100;
  const jlong tmp2 = z1 + z3 * (-JFIX_1_847759065);
  const jlong tmp3 = z1 + z2 * JFIX_0_765366865;
  z2 = in[0]; z3 = in[4];
  const jlong tmp0 = (z2 + z3) << JCONST_BITS;
  const jlong tmp1 = (z2 - z3) << JCONST_BITS;
  const jlong tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  jlong t0 = in[7], t1 = in[5], t2 = in[3], t3 = in[1];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
  jlong z4 = t1 + t3;
  const jlong z5 = (z3 + z4) * JFIX_1_175875602;
  t0 *= JFIX_0_298631336; t1 *= JFIX_2_053119869; t2 *= JFIX_3_072711026; t3 *= JFIX_1_501321110;
  z1 *= -JFIX_0_899976223; z2 *= -JFIX_2_562915447; z3 *= -JFIX_1_961570560; z4 *= -JFIX_0_390180644;
  z3 += z5; z4 += z5;
  t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
  out[0] = jdescale(tmp10 + t3, shift); out[7] = jdescale(tmp10 - t3, shift);
  out[1] = jdescale(tmp11 + t2, shift); out[6] = jdescale(tmp11 - t2, shift);
  out[2] = jdescale(tmp12 + t1, shift); out[5] = jdescale(tmp12 - t1, shift);
  out[3] = jdescale(tmp13 + t0, shift); out[4] = jdescale(tmp13 - t0, shift);
}

// one thread per 8x8 block: dequantise, columns then rows (jidctint.c), +128, clamp -> plane[comp] (blocks_h*8 rows, blocks_w*8 bytes)
__global__ __launch_bounds__(64) void jpeg_idct_kernel(const ofb_jpeg_job* __restrict__ jobs, const int16_t* __restrict__ coef,
                                                       uint8_t* __restrict__ planes) {
  const ofb_jpeg_job& J = jobs[blockIdx.z];
  const int c = blockIdx.y;
  if (c >= J.ncomp) return;
  const int nb = J.blocks_w[c] * J.blocks_h[c], b = blockIdx.x * 64 + threadIdx.x;
  if (b >= nb) return;
  const int16_t* in = coef + J.coef_off[c] + (int64_t)b * 64;
  int ws[64];
#pragma unroll
  for (int col = 0; col < 8; ++col) {
    int v[8], o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = (int)in[8 * r + col] * (int)J.quant[c][8 * r + col];
    idct8(v, o, JCONST_BITS - JPASS1_BITS);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[8 * r + col] = o[r];
  }
  const int bx = b % J.blocks_w[c], by = b / J.blocks_w[c], pitch = J.blocks_w[c] * 8;
  uint8_t* out = planes + J.plane_off[c] + (int64_t)(by * 8) * pitch + bx * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int v[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = ws[8 * r + k];
    idct8(v, o, JCONST_BITS + JPASS1_BITS + 3);
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lo |= (unsigned)min(255, max(0, o[k] + 128)) << (8 * k);
      hi |= (unsigned)min(255, max(0, o[4 + k] + 128)) << (8 * k);
    }
    *reinterpret_cast<uint2*>(out + (int64_t)r * pitch) = make_uint2(lo, hi);
  }
}

// sample of component c at full-resolution pixel (y, x): jdsample.c fancy upsampling (edge samples replicated)
__device__ __forceinline__ int jpeg_sample(const ofb_jpeg_job& J, const uint8_t* __restrict__ planes, int c, int y, int x) {
  const uint8_t* p = planes + J.plane_off[c];
  const int pitch = J.blocks_w[c] * 8;
  const int rh = J.hmax / J.hs[c], rv = J.vmax / J.vs[c];
  if (rh == 1 && rv == 1) return p[(int64_t)y * pitch + x];
  const int dw = (J.width * J.hs[c] + J.hmax - 1) / J.hmax, dh = (J.height * J.vs[c] + J.vmax - 1) / J.vmax;
  if (rh == 2 && rv == 1) {                               // h2v1_fancy_upsample
    const int ix = x >> 1, nb = (x & 1) ? min(ix + 1, dw - 1) : max(ix - 1, 0);
    const uint8_t* row = p + (int64_t)y * pitch;
    return (3 * row[ix] + row[nb] + ((x & 1) ? 2 : 1)) >> 2;
  }
  const int iy = y >> 1, fy = (y & 1) ? min(iy + 1, dh - 1) : max(iy - 1, 0);
  const uint8_t* r0 = p + (int64_t)iy * pitch;
  const uint8_t* r1 = p + (int64_t)fy * pitch;
  if (rh == 1) return (3 * r0[x] + r1[x] + ((y & 1) ? 2 : 1)) >> 2;           // h1v2_fancy_upsample
  const int ix = x >> 1, nb = (x & 1) ? min(ix + 1, dw - 1) : max(ix - 1, 0);    // h2v2_fancy_upsample
  const int cs = 3 * r0[ix] + r1[ix], cn = 3 * r0[nb] + r1[nb];
  return (3 * cs + cn + ((x & 1) ? 7 : 8)) >> 4;
}

// one thread per pixel: upsample + jdcolor.c ycc_rgb_convert -> out (HWC uint8 RGB)
__global__ __launch_bounds__(256) void jpeg_color_kernel(const ofb_jpeg_job* __restrict__ jobs, const uint8_t* __restrict__ planes,
                                                         uint8_t* __restrict__ out) {
  const ofb_jpeg_job& J = jobs[blockIdx.z];
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= J.width || y >= J.height) return;
  const int yy = jpeg_sample(J, planes, 0, y, x);
  int r = yy, g = yy, b = yy;
  if (J.ncomp == 3) {
    const int cb = jpeg_sample(J, planes, 1, y, x) - 128, cr = jpeg_sample(J, planes, 2, y, x) - 128;
    // FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554; ONE_HALF = 32768
    r = yy + ((91881 * cr + 32768) >> 16);
    b = yy + ((116130 * cb + 32768) >> 16);
    g = yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    r = min(255, max(0, r)); g = min(255, max(0, g)); b = min(255, max(0, b));
  }
  uint8_t* o = out + J.out_off + ((int64_t)y * J.width + x) * 3;
  o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b;
}

}  // namespace

// jobs_dev[n]: per image the geometry (ofb_jpeg_info fields), offsets of its coefficients / component planes / output pixels;
// max_blocks = the largest blocks_w * blocks_h of any component, max_w / max_h = the largest image
extern "C" int ofb_jpeg_decode_pixels(const ofb_jpeg_job* jobs_dev, int32_t n_images, int32_t max_blocks, int32_t max_w, int32_t max_h,
                                      const int16_t* coef_dev, uint8_t* planes_dev, uint8_t* out_dev, void* stream) {
  if (!jobs_dev || !coef_dev || !planes_dev || !out_dev || n_images <= 0 || n_images > 65535 || max_blocks <= 0 || max_w <= 0 ||
      max_h <= 0 || max_h > 65535 * 4)
    return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3(ofb_cdiv(max_blocks, 64), 3, n_images), dim3(64), 0, s, jobs_dev, coef_dev, planes_dev);
  hipLaunchKernelGGL(jpeg_color_kernel, dim3(ofb_cdiv(max_w, 64), ofb_cdiv(max_h, 4), n_images), dim3(256), 0, s, jobs_dev,
                     (const uint8_t*)planes_dev, out_dev);
  return ofb_launch_status();
}
