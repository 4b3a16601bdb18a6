// JPEG decode, HOST stage (reference datasets.py:90-125): marker parsing and the ENTROPY stage - Huffman decoding is a
// serial bit stream per image, so it runs on the loader's CPU threads (ofb_jpeg_parse / ofb_jpeg_decode_coefficients: re-entrant, no
// global state) and hands over de-zigzagged int16 coefficient blocks; the device stage (IDCT, upsampling, colour) is csrc/jpeg.hip.
// Plain C++ with no HIP dependency: the library build compiles it with hipcc like every other source, and tests/test_jpeg_fuzz.py
// builds the SAME file with g++ -fsanitize=address,undefined and feeds it truncated / corrupted files (this code reads untrusted
// bytes).  Scope: baseline / extended-sequential Huffman JPEG (SOF0 / SOF1), 8-bit, 1 (grayscale) or 3 (YCbCr) components, any scan
// structure, restart intervals.  Progressive (SOF2), arithmetic coding, CMYK / YCCK, RGB-stored (Adobe transform 0) and 12-bit files
// are rejected with OFB_ELIMIT: data.JpegDecoder routes those few files to its fallback decoder.
#include <atomic>
#include <thread>
#include <vector>
#include <stdint.h>
#include "../../include/ofb_hip.h"
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
  const int idx = t.valptr[l] + code - t.mincode[l];     // canonical codes keep this inside [0, 255]; a hostile table must not matter
  return (idx >= 0 && idx < 256) ? t.vals[idx] : -1;
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
  int adobe_transform = -1;                              // APP14 "Adobe" colour transform flag (-1: no such segment)
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
    } else if (m == 0xee) {
      // APP14 "Adobe": byte 11 = colour transform (0: the components are stored as they are - RGB for three of them -, 1: YCbCr)
      if (sl >= 12 && s[0] == 'A' && s[1] == 'd' && s[2] == 'o' && s[3] == 'b' && s[4] == 'e') P.adobe_transform = s[11];
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
  // libjpeg's colour-space guess for three components (jdapimin.c default_decompress_parms): an Adobe marker with transform 0, or -
  // without JFIF / Adobe markers - the component ids 'R', 'G', 'B', mean the planes ARE RGB: no YCbCr conversion.  The device stage
  // always converts, so those files are out of scope here (the loader routes them to its fallback decoder).
  if (P.info.ncomp == 3) {
    if (P.adobe_transform == 0) return OFB_ELIMIT;
    if (P.adobe_transform < 0 && P.comp_id[0] == 'R' && P.comp_id[1] == 'G' && P.comp_id[2] == 'B') return OFB_ELIMIT;
  }
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
