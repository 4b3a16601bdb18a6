// JPEG decode for the input side of the step (reference datasets.py:90-125 reads ImageNet through torchvision's
// ImageFolder + PIL's default_loader, i.e. libjpeg-turbo).  Hybrid split, chosen by where each stage parallelises:
//   host (this file, plain C++): marker parsing and the ENTROPY stage - Huffman decoding is a serial bit stream per image, so it
//        runs on the loader's CPU threads (ofb_jpeg_parse / ofb_jpeg_decode_coefficients: re-entrant, no global state) and hands
//        over de-zigzagged int16 coefficient blocks;
//   device: dequantisation + the 8x8 inverse DCT (one thread per block), then chroma upsampling + YCbCr -> RGB (one thread per
//        pixel) for a whole batch of images per launch, writing the decoded uint8 HWC pixels DeviceTransform starts from.
// Arithmetic restates libjpeg(-turbo)'s default decompression path bit for bit, because that is what the reference's pixels are
// made by: jidctint.c (JDCT_ISLOW: 13-bit constants, two passes), jdsample.c fancy ("triangle") upsampling h2v1 / h2v2 / h1v2
// with edge replication, jdcolor.c fixed-point YCbCr -> RGB.  Scope: baseline / extended-sequential Huffman JPEG (SOF0 / SOF1),
// 8-bit, 1 (grayscale -> RGB) or 3 (YCbCr) components, any scan structure, restart intervals.  Progressive (SOF2), arithmetic
// coding, CMYK and 12-bit files are rejected with OFB_ELIMIT (the loader may hand those few files to another decoder).
#include <atomic>
#include <thread>
#include <vector>
#include "ofb_common.h"
#include <string.h>

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
  bool present = false;
  // canonical decode (JPEG Annex F.2.2.3): per code length the largest code and the index of its first value
  int32_t maxcode[18];
  int32_t valptr[17];
  int32_t mincode[17];
  uint8_t vals[256];
  uint8_t look_nbits[256];      // 8-bit prefix lookup: code length (0 = longer than 8 bits)
  uint8_t look_val[256];
};

bool build_table(const uint8_t* bits /* [1..16] at bits[0..15] */, const uint8_t* vals, int nvals, HuffTable& t) {
  int total = 0;
  for (int i = 0; i < 16; ++i) total += bits[i];
  if (total != nvals || total > 256) return false;
  memcpy(t.vals, vals, (size_t)nvals);
  int code = 0, k = 0;
  memset(t.look_nbits, 0, sizeof(t.look_nbits));
  for (int l = 1; l <= 16; ++l) {
    t.valptr[l] = k;
    t.mincode[l] = code;
    for (int i = 0; i < bits[l - 1]; ++i, ++k, ++code) {
      if (l <= 8) {
        const int first = code << (8 - l), n = 1 << (8 - l);
        if (first + n > 256) return false;
        for (int j = 0; j < n; ++j) { t.look_nbits[first + j] = (uint8_t)l; t.look_val[first + j] = vals[k]; }
      }
    }
    t.maxcode[l] = bits[l - 1] ? code - 1 : -1;
    if (code > (1 << l)) return false;
    code <<= 1;
  }
  t.maxcode[17] = 0x7fffffff;
  t.present = true;
  return true;
}

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t acc = 0;      // bits left-aligned at the top
  int nbits = 0;
  bool hit_marker = false;
  void fill() {
    while (nbits <= 56) {
      int byte = 0;
      if (!hit_marker && p < end) {
        byte = *p;
        if (byte == 0xff) {
          if (p + 1 < end && p[1] == 0x00) { p += 2; }
          else { hit_marker = true; byte = 0; }          // a marker: feed zeros (a well-formed stream never consumes them)
        } else {
          ++p;
        }
      }
      acc |= (uint64_t)byte << (56 - nbits);
      nbits += 8;
    }
  }
  inline int peek(int n) { return (int)(acc >> (64 - n)); }
  inline void skip(int n) { acc <<= n; nbits -= n; }
  inline int get(int n) { if (n == 0) return 0; if (nbits < n) fill(); const int v = peek(n); skip(n); return v; }
  void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

inline int decode_symbol(BitReader& br, const HuffTable& t) {
  if (br.nbits < 16) br.fill();
  const int look = br.peek(8);
  const int nb = t.look_nbits[look];
  if (nb) { br.skip(nb); return t.look_val[look]; }
  int code = br.peek(9), l = 9;
  while (l <= 16 && code > t.maxcode[l]) { ++l; code = br.peek(l); }
  if (l > 16) return -1;
  br.skip(l);
  return t.vals[t.valptr[l] + code - t.mincode[l]];
}
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

struct Parsed {
  ofb_jpeg_info info;
  int comp_id[3], comp_tq[3];
  uint16_t qt[4][64];
  bool qt_present[4] = {false, false, false, false};
  HuffTable dc[4], ac[4];
  int restart_interval = 0;
};

// walks the marker segments up to the first SOS (frame header, quantisation tables, component list)
int parse_headers(const uint8_t* d, int64_t n, Parsed& P) {
  if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return OFB_EINVAL;
  memset(&P.info, 0, sizeof(P.info));
  int64_t pos = 2;
  bool have_sof = false;
  while (pos + 4 <= n) {
    if (d[pos] != 0xff) return OFB_EINVAL;
    while (pos < n && d[pos] == 0xff) ++pos;             // fill bytes
    if (pos >= n) return OFB_EINVAL;
    const int m = d[pos++];
    if (m == 0xd9) break;
    if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
    if (pos + 2 > n) return OFB_EINVAL;
    const int len = be16(d + pos);
    if (len < 2 || pos + len > n) return OFB_EINVAL;
    const uint8_t* s = d + pos + 2;
    const int sl = len - 2;
    if (m == 0xc0 || m == 0xc1) {
      if (sl < 6) return OFB_EINVAL;
      if (s[0] != 8) return OFB_ELIMIT;
      P.info.height = be16(s + 1); P.info.width = be16(s + 3); P.info.ncomp = s[5];
      if (P.info.width <= 0 || P.info.height <= 0) return OFB_ELIMIT;   // DNL-defined heights are not supported
      if (P.info.ncomp != 1 && P.info.ncomp != 3) return OFB_ELIMIT;
      if (sl < 6 + 3 * P.info.ncomp) return OFB_EINVAL;
      for (int c = 0; c < P.info.ncomp; ++c) {
        P.comp_id[c] = s[6 + 3 * c];
        P.info.hs[c] = s[7 + 3 * c] >> 4; P.info.vs[c] = s[7 + 3 * c] & 15;
        P.comp_tq[c] = s[8 + 3 * c];
        if (P.info.hs[c] < 1 || P.info.hs[c] > 4 || P.info.vs[c] < 1 || P.info.vs[c] > 4 || P.comp_tq[c] > 3) return OFB_EINVAL;
      }
      have_sof = true;
    } else if (m == 0xc2 || (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) {
      return OFB_ELIMIT;                                  // progressive / lossless / arithmetic-coded frames
    } else if (m == 0xcc) {
      return OFB_ELIMIT;
    } else if (m == 0xdb) {
      int q = 0;
      while (q < sl) {
        const int pq = s[q] >> 4, tq = s[q] & 15;
        if (tq > 3 || pq > 1) return OFB_EINVAL;
        if (q + 1 + 64 * (pq + 1) > sl) return OFB_EINVAL;
        for (int i = 0; i < 64; ++i) P.qt[tq][kZigzag[i]] = pq ? (uint16_t)be16(s + q + 1 + 2 * i) : s[q + 1 + i];
        P.qt_present[tq] = true;
        q += 1 + 64 * (pq + 1);
      }
    } else if (m == 0xc4) {
      int q = 0;
      while (q + 17 <= sl) {
        const int tc = s[q] >> 4, th = s[q] & 15;
        if (tc > 1 || th > 3) return OFB_EINVAL;
        int nv = 0;
        for (int i = 0; i < 16; ++i) nv += s[q + 1 + i];
        if (q + 17 + nv > sl) return OFB_EINVAL;
        if (!build_table(s + q + 1, s + q + 17, nv, tc ? P.ac[th] : P.dc[th])) return OFB_EINVAL;
        q += 17 + nv;
      }
    } else if (m == 0xdd) {
      if (sl < 2) return OFB_EINVAL;
      P.restart_interval = be16(s);
    } else if (m == 0xda) {
      return have_sof ? OFB_OK : OFB_EINVAL;
    }
    pos += len;
  }
  return OFB_EINVAL;                                      // no scan
}

void finish_info(Parsed& P) {
  ofb_jpeg_info& I = P.info;
  I.hmax = I.vmax = 1;
  for (int c = 0; c < I.ncomp; ++c) { I.hmax = I.hs[c] > I.hmax ? I.hs[c] : I.hmax; I.vmax = I.vs[c] > I.vmax ? I.vs[c] : I.vmax; }
  I.mcu_x = (I.width + 8 * I.hmax - 1) / (8 * I.hmax);
  I.mcu_y = (I.height + 8 * I.vmax - 1) / (8 * I.vmax);
  I.coef_count = 0;
  for (int c = 0; c < I.ncomp; ++c) {
    I.blocks_w[c] = I.mcu_x * I.hs[c];
    I.blocks_h[c] = I.mcu_y * I.vs[c];
    I.coef_off[c] = I.coef_count;
    I.coef_count += (int64_t)I.blocks_w[c] * I.blocks_h[c] * 64;
    for (int i = 0; i < 64; ++i) I.quant[c][i] = P.qt[P.comp_tq[c]][i];
  }
}

}  // namespace

extern "C" int ofb_jpeg_parse(const uint8_t* data, int64_t nbytes, ofb_jpeg_info* info) {
  if (!data || !info || nbytes < 4) return OFB_EINVAL;
  Parsed P;
  if (int rc = parse_headers(data, nbytes, P)) return rc;
  for (int c = 0; c < P.info.ncomp; ++c)
    if (!P.qt_present[P.comp_tq[c]]) return OFB_EINVAL;
  finish_info(P);
  // the sampling layouts the device upsampler implements: luma at the maximum factors, chroma 1x1 / 2x1 / 1x2 / 2x2 below it
  if (P.info.ncomp == 3) {
    if (P.info.hs[0] != P.info.hmax || P.info.vs[0] != P.info.vmax) return OFB_ELIMIT;
    for (int c = 1; c < 3; ++c) {
      const int rh = P.info.hmax / P.info.hs[c], rv = P.info.vmax / P.info.vs[c];
      if (P.info.hmax % P.info.hs[c] || P.info.vmax % P.info.vs[c] || rh > 2 || rv > 2) return OFB_ELIMIT;
    }
  }
  *info = P.info;
  return OFB_OK;
}

// coef: info->coef_count int16, component c at coef_off[c] as [blocks_h][blocks_w][64] in natural (row-major) order; blocks that
// the scans do not cover (none in a well-formed file) stay zero
extern "C" int ofb_jpeg_decode_coefficients(const uint8_t* data, int64_t nbytes, const ofb_jpeg_info* info, int16_t* coef) {
  if (!data || !info || !coef) return OFB_EINVAL;
  Parsed P;
  if (int rc = parse_headers(data, nbytes, P)) return rc;
  finish_info(P);
  if (P.info.coef_count != info->coef_count || P.info.width != info->width || P.info.height != info->height) return OFB_EINVAL;
  memset(coef, 0, (size_t)info->coef_count * sizeof(int16_t));
  // second walk: tables may be redefined between scans, so segments are applied in stream order while the scans are decoded
  Parsed Q;
  memset(&Q.info, 0, sizeof(Q.info));
  const uint8_t* d = data;
  const int64_t n = nbytes;
  int64_t pos = 2;
  while (pos + 4 <= n) {
    if (d[pos] != 0xff) return OFB_EINVAL;
    while (pos < n && d[pos] == 0xff) ++pos;
    if (pos >= n) return OFB_EINVAL;
    const int m = d[pos++];
    if (m == 0xd9) break;
    if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
    if (pos + 2 > n) return OFB_EINVAL;
    const int len = be16(d + pos);
    if (len < 2 || pos + len > n) return OFB_EINVAL;
    const uint8_t* s = d + pos + 2;
    const int sl = len - 2;
    if (m == 0xc4) {
      int q = 0;
      while (q + 17 <= sl) {
        const int tc = s[q] >> 4, th = s[q] & 15;
        int nv = 0;
        for (int i = 0; i < 16; ++i) nv += s[q + 1 + i];
        if (tc > 1 || th > 3 || q + 17 + nv > sl) return OFB_EINVAL;
        if (!build_table(s + q + 1, s + q + 17, nv, tc ? Q.ac[th] : Q.dc[th])) return OFB_EINVAL;
        q += 17 + nv;
      }
    } else if (m == 0xdd) {
      if (sl < 2) return OFB_EINVAL;
      Q.restart_interval = be16(s);
    } else if (m == 0xda) {
      if (sl < 1) return OFB_EINVAL;
      const int ns = s[0];
      if (ns < 1 || ns > P.info.ncomp || sl < 1 + 2 * ns + 3) return OFB_EINVAL;
      int sc[3], td[3], ta[3];
      for (int i = 0; i < ns; ++i) {
        sc[i] = -1;
        for (int c = 0; c < P.info.ncomp; ++c)
          if (P.comp_id[c] == s[1 + 2 * i]) sc[i] = c;
        td[i] = s[2 + 2 * i] >> 4; ta[i] = s[2 + 2 * i] & 15;
        if (sc[i] < 0 || td[i] > 3 || ta[i] > 3 || !Q.dc[td[i]].present || !Q.ac[ta[i]].present) return OFB_EINVAL;
      }
      if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63) return OFB_ELIMIT;        // spectral selection = progressive
      // entropy-coded segment
      BitReader br;
      br.p = d + pos + len; br.end = d + n;
      int pred[3] = {0, 0, 0};
      const ofb_jpeg_info& I = P.info;
      // a single-component scan is not interleaved: its MCU is one block, over the component's own (unpadded) block grid
      const bool inter = ns > 1;
      const int c0 = sc[0];
      const int mx = inter ? I.mcu_x : (int)(((int64_t)I.width * I.hs[c0] + I.hmax - 1) / I.hmax + 7) / 8;
      const int my = inter ? I.mcu_y : (int)(((int64_t)I.height * I.vs[c0] + I.vmax - 1) / I.vmax + 7) / 8;
      int64_t mcus_left = Q.restart_interval > 0 ? Q.restart_interval : -1;
      int next_rst = 0;
      for (int y = 0; y < my; ++y) {
        for (int x = 0; x < mx; ++x) {
          if (mcus_left == 0) {
            // expect RSTn: drop the bit buffer, find the marker
            const uint8_t* q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) ++q;
            if (q + 1 >= br.end || q[1] != 0xd0 + next_rst) return OFB_EINVAL;
            br.p = q + 2; br.reset();
            next_rst = (next_rst + 1) & 7;
            pred[0] = pred[1] = pred[2] = 0;
            mcus_left = Q.restart_interval;
          }
          for (int i = 0; i < ns; ++i) {
            const int c = sc[i];
            const int nh = inter ? I.hs[c] : 1, nv = inter ? I.vs[c] : 1;
            for (int v = 0; v < nv; ++v)
              for (int h = 0; h < nh; ++h) {
                const int bx = x * nh + h, by = y * nv + v;
                if (bx >= I.blocks_w[c] || by >= I.blocks_h[c]) return OFB_EINVAL;
                int16_t* blk = coef + I.coef_off[c] + ((int64_t)by * I.blocks_w[c] + bx) * 64;
                const int t = decode_symbol(br, Q.dc[td[i]]);
                if (t < 0 || t > 15) return OFB_EINVAL;
                if (t) pred[i] += extend(br.get(t), t);
                blk[0] = (int16_t)pred[i];
                for (int k = 1; k < 64;) {
                  const int rs = decode_symbol(br, Q.ac[ta[i]]);
                  if (rs < 0) return OFB_EINVAL;
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz == 0) {
                    if (r != 15) break;
                    k += 16;
                    continue;
                  }
                  k += r;
                  if (k > 63) return OFB_EINVAL;
                  blk[kZigzag[k]] = (int16_t)extend(br.get(sz), sz);
                  ++k;
                }
              }
          }
          if (mcus_left > 0) --mcus_left;
        }
      }
      // continue the marker walk behind the entropy-coded data: the next 0xFF followed by a non-zero, non-RST byte
      const uint8_t* q = br.p;
      if (br.hit_marker && q > d) { /* q already sits on the marker's 0xFF */ }
      while (q + 1 < br.end && !(q[0] == 0xff && q[1] != 0x00 && !(q[1] >= 0xd0 && q[1] <= 0xd7) && q[1] != 0xff)) ++q;
      pos = q - d;
      continue;
    }
    pos += len;
  }
  return OFB_OK;
}

// Whole-batch host stage (what data.JpegDecoder calls): the per-file work above without a Python round trip per file.
// plan: parses every frame header and lays the batch out - infos[i], the device job records jobs[i] (absolute coefficient / plane /
// pixel offsets), totals[6] = {coefficient elements, plane bytes, pixel bytes, max blocks of a component, max width, max height}.
extern "C" int ofb_jpeg_plan_batch(const uint8_t* const* files, const int64_t* nbytes, int32_t n, ofb_jpeg_info* infos, ofb_jpeg_job* jobs,
                                   int64_t* totals) {
  if (!files || !nbytes || !infos || !jobs || !totals || n <= 0) return OFB_EINVAL;
  int64_t coef_total = 0, plane_total = 0, out_total = 0, max_blocks = 1, max_w = 1, max_h = 1;
  for (int i = 0; i < n; ++i) {
    if (int rc = ofb_jpeg_parse(files[i], nbytes[i], &infos[i])) return rc;
    const ofb_jpeg_info& I = infos[i];
    ofb_jpeg_job& J = jobs[i];
    memset(&J, 0, sizeof(J));
    J.width = I.width; J.height = I.height; J.ncomp = I.ncomp; J.hmax = I.hmax; J.vmax = I.vmax;
    for (int c = 0; c < I.ncomp; ++c) {
      J.hs[c] = I.hs[c]; J.vs[c] = I.vs[c]; J.blocks_w[c] = I.blocks_w[c]; J.blocks_h[c] = I.blocks_h[c];
      J.coef_off[c] = coef_total + I.coef_off[c];
      J.plane_off[c] = plane_total;
      const int64_t nb = (int64_t)I.blocks_w[c] * I.blocks_h[c];
      plane_total += (nb * 64 + 15) / 16 * 16;
      max_blocks = nb > max_blocks ? nb : max_blocks;
      memcpy(J.quant[c], I.quant[c], sizeof(J.quant[c]));
    }
    J.out_off = out_total;
    out_total += ((int64_t)I.height * I.width * 3 + 15) / 16 * 16;
    coef_total += (I.coef_count + 7) / 8 * 8;
    max_w = I.width > max_w ? I.width : max_w;
    max_h = I.height > max_h ? I.height : max_h;
  }
  totals[0] = coef_total; totals[1] = plane_total; totals[2] = out_total; totals[3] = max_blocks; totals[4] = max_w; totals[5] = max_h;
  return OFB_OK;
}

// entropy stage of the planned batch on `threads` host threads (a work queue over the files); coef: totals[0] int16 of staging
extern "C" int ofb_jpeg_decode_batch(const uint8_t* const* files, const int64_t* nbytes, int32_t n, const ofb_jpeg_info* infos,
                                     const ofb_jpeg_job* jobs, int16_t* coef, int32_t threads) {
  if (!files || !nbytes || !infos || !jobs || !coef || n <= 0) return OFB_EINVAL;
  std::atomic<int> next(0), status(OFB_OK);
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) return;
      const int rc = ofb_jpeg_decode_coefficients(files[i], nbytes[i], &infos[i], coef + (jobs[i].coef_off[0] - infos[i].coef_off[0]));
      if (rc != OFB_OK) { int ok = OFB_OK; status.compare_exchange_strong(ok, rc); }
    }
  };
  const int nt = threads < 1 ? 1 : (threads > n ? n : threads);
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
  return status.load();
}

// ---------------------------------------------------------------------------------------------------------------------------
// device side
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
