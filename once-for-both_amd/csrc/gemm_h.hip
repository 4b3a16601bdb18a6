// f32-class GEMM on operands that are ALREADY split into two f16 planes ("H-format", hformat.h), staged by LDS-DMA.
//
// Why two f16 planes: the step runs at the package power cap, so its time is its energy.  Round 1-3 computed every f32
// product as SIX v_mfma_f32_32x32x16_bf16 terms of an exact three-way bf16 split; a two-way f16 split of a power-of-two scaled copy
// carries 23 significant bits in TWO planes and needs THREE terms (h2 h1, h1 h2, h1 h1; the dropped h2 h2 is a zero-mean 2^-25 of
// the product in RMS, <= 2^-22 worst case): half the matrix-pipe work and two thirds of the operand bytes per product, at a
// per-product error of ~2^-24 RMS with the accumulation still in f32 (measured against fp64: 3-10x BELOW a k-ordered f32 fma chain
// on every operand class of tests/test_gpu_accuracy_class.py).
// f16 has 5 exponent bits, so every tensor carries ONE power-of-two exponent e in its header (hformat.h); the kernel folds
// 2^-(ea + eb) into alpha.
//
// Kernel (product configuration C128): 128 x 192 tile, 4 waves (2 x 2), wave tile 64 x 96 = 2 x 3 blocks of
// v_mfma_f32_32x32x16_f16, LDS stages of K = 32 (two K16 sub-steps: 36 MFMAs per wave and stage), TWO workgroups per CU, two 40-KB
// stages (all 160 KB of the CU's LDS) filled by global_load_lds_dwordx4 (inline asm, counted by hand), ONE barrier per stage placed
// between the two halves of its second sub-step, the fragment reads of the next half-step issued in the MFMA gaps of the current
// one; epilogue through LDS with compile-time forms.  Scheduling is the hybrid stream-K of gemm_plan.h.
#include "hformat.h"
#include "gemm_plan.h"
#include <cstdlib>
#include <type_traits>

namespace {

using ofb_plan::Plan; using ofb_plan::make_plan; using ofb_plan::tile_coord; using ofb_plan::Seg; using ofb_plan::get_seg;

typedef short hs16x4 __attribute__((ext_vector_type(4)));
typedef short hs16x8 __attribute__((ext_vector_type(8)));
#define OFB_LDSP(p) ((__attribute__((address_space(3))) void*)(p))

// lab only (scripts/lab/ablate_gemm_h.sh): -DOFB_LAB_ABLATE=n removes ONE ingredient of the 16x16x32 kernel (results are wrong) to see
// what its time is made of: 1 = no B pieces after the prologue, 2 = no LDS-DMA at all after the prologue, 3 = no MFMAs (fragment reads
// kept alive), 4 = no fragment reads after the prologue, 5 = no global stores in the wide epilogue, 6 = no GELU' (aux) stores, 7 = no
// plane stores, 8 = the GELU pieces replaced by two FMAs, 9 = no epilogue at all
#ifndef OFB_LAB_ABLATE
#define OFB_LAB_ABLATE 0
#endif
#ifndef OFB_DIRECT_PAIR
#define OFB_DIRECT_PAIR 1           /* direct epilogue: neighbouring lanes pair their plane slots into 16-byte stores (0: two 8-byte stores per lane) */
#endif
#ifndef OFB_EPI_PRIO
#define OFB_EPI_PRIO 1              /* wave priority during the direct epilogue (lab: 0 / 3) */
#endif
#ifndef OFB_H_NTERM
#define OFB_H_NTERM 3               /* 4: also h2 h2 (lab: accuracy comparison) */
#endif

// Tile configuration: WM x WN waves, wave tile (32 MI) x (32 NI), NST LDS stages of KH K16 sub-steps, WGS workgroups per CU, MF = the
// MFMA's block size: 32 = v_mfma_f32_32x32x16_f16 (one instruction per K16 sub-step), 16 = v_mfma_f32_16x16x32_f16 (one per K32 stage).
// MB16 / NB16 (MF = 16 only, 0 = derived from MI / NI): the wave tile given directly in 16-row / 16-column blocks.  The tile then
// COMPUTES BMT = 16 MB16 WM rows while it STAGES BM = BMT rounded up to 32 (the LDS image keeps its 32-row pieces); tiles step by BMT.
template <int WM_, int WN_, int MI_, int NI_, int NST_, int KH_, int WGS_, int MF_ = 32, int MB16_ = 0, int NB16_ = 0>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, MI = MI_, NI = NI_, NST = NST_, KH = KH_, WGS = WGS_, NW = WM * WN, MF = MF_;
  static constexpr int MB = MB16_ ? MB16_ : 32 * MI / MF, NB = NB16_ ? NB16_ : 32 * NI / MF;     // MFMA blocks of the wave tile
  static_assert(MF == 32 || (MF == 16 && KH == 2), "a 16x16x32 instruction spans one K32 stage");
  static_assert((MB16_ == 0 && NB16_ == 0) || MF == 16, "block-granular wave tiles are a 16x16x32 form");
  static constexpr int BMT = MF * MB * WM;                                            // rows a tile computes (= its step along M)
  static constexpr int BM = (BMT + 31) / 32 * 32, BN = MF * NB * WN, NT = 64 * NW;     // rows / columns staged
  static constexpr int A_BYTES = BM * 64 * KH, B_BYTES = BN * 64 * KH, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024;          // 1-KB LDS-DMA pieces per stage
  static constexpr int QA = A_PIECES / NW, QB = (B_PIECES + NW - 1) / NW;             // pieces per wave (the last B piece only for some waves)
  static constexpr int HA = MI / 2;                                                   // row blocks per half step (MF = 32)
  static constexpr int HR = (NST * STAGE >= (128 + 32) * (BN + 4) * 4) ? 128 : 64;    // rows of the tile parked in LDS per epilogue pass (+ a column-sum row per 4)
  static constexpr int TROW = HR + 4;
  static constexpr int NPASS = (BMT + HR - 1) / HR;                                   // epilogue passes (the last one may be short)
  static_assert(A_PIECES % NW == 0 && BN % 32 == 0 && BMT % 16 == 0, "piece schedule");
  static_assert(MF == 16 || (MI % 2 == 0 && BM % HR == 0 && (32 * MI) <= HR && HR % (32 * MI) == 0), "epilogue schedule of the 32x32 forms");
  static_assert(BN * TROW * 4 <= NST * STAGE && QA >= 1 && QA <= 8 && QB <= 8 && NST >= 2 && NST <= 4 && KH >= 1 && KH <= 2, "LDS budget / schedule");
  static_assert(NST * STAGE * WGS <= 163840, "LDS per CU");
};
using C128 = Cfg<2, 2, 2, 3, 2, 2, 2>;
//   C128F: the same tile, stages and LDS image on v_mfma_f32_16x16x32_f16 (4 x 6 blocks per wave, quarter steps of 18 MFMAs): the
//        chip holds a higher clock on this shape at equal cycles per FLOP (MI355X_MICROARCH "DVFS give-back" item 7)
using C128F = Cfg<2, 2, 2, 3, 2, 2, 2, 16>;
//   C112F: a 112 x 192 tile on the same stages (128 rows staged, 7 x 16 computed): four waves side by side, wave tile 112 x 48 = 7 x 3
//        blocks.  For products whose 128-row tiles fill between half a round and one round of the 2 x CUs workgroups (the five
//        394-tile products of the DeiT-S step): the launch ends when its slowest pair of co-resident tiles does, and 452 tiles of
//        112 rows are 7/8 of the work of 394 tiles of 128 (VERDICT r4 #1b)
using C112F = Cfg<1, 4, 2, 1, 2, 2, 2, 16, 7, 3>;        // (MI, NI = 2, 1: placeholders, the 16-blocks 7 x 3 count)
//   C96: 256 x 96, 4 waves stacked along M (the same 64 x 96 wave tile): output widths that pad badly on 192 columns (N mod 192 in
//        (0, 96]: the pruned / finetune widths 264, 480, 672, ...)
using C96 = Cfg<4, 1, 2, 3, 3, 1, 2>;     // K16 stages: two K32 stages of a 256 x 96 tile x two workgroups do not fit the LDS
// (lab, round 4: Cfg<4, 2, 2, 3, 2, 2, 1> = 256 x 192 with EIGHT waves, one workgroup per CU, runs correctly and is 0-12 % slower than two
//  independent 4-wave workgroups on every step shape: profiles/r04_gemm_8wave_tile_and_turnstiles.txt)
#ifdef OFB_GEMM_H_LAB
// lab: 128 x 128 tile, wave tile 64 x 64, three K16 stages, THREE workgroups per CU (48 KB LDS, <= 168 registers): a third workgroup to
// hide the latency of the LDS-DMA, against 20 % more staged bytes and fragment reads per MFMA
using C128S = Cfg<2, 2, 2, 2, 3, 1, 3>;
using C128K1 = Cfg<2, 2, 2, 3, 4, 1, 2>;          // lab: K16 stages, FOUR of them (three stages of lead for the LDS-DMA)
#endif
constexpr int GRAN = OFB_HGRAN;
constexpr int CS_SLAB_RG = 64;                      // row groups (256 rows) per column-sum slab

__device__ __forceinline__ void store_h4(char* slot, float v0, float v1, float v2, float v3) {
  unsigned a0, b0, a1, b1;
  ofb_hsplit_pair(v0, v1, a0, b0);
  ofb_hsplit_pair(v2, v3, a1, b1);
  *reinterpret_cast<uint2*>(slot) = make_uint2(a0, a1);
  *reinterpret_cast<uint2*>(slot + 128) = make_uint2(b0, b1);
}

// ---- statistics + conversion f32 <-> H-format ---------------------------------------------------------------------
// value = X * rowscale[r / rs_div] (optional).  Stage 1: hdr.amax = max|value|, hdr.rn2sq = max_r sum_c value^2 by atomic max
// (header zeroed by a memset node ahead of the launch); one wave per row, grid-stride.
// One wave's rows r_first, r_first + r_stride, ...: FOUR rows at a time - their loads are in flight together and their row sums are
// reduced side by side (one row at a time every row cost a memory round trip and a reduction chain of its own: the bound passes ran
// at 9 % of the HBM rate).  Per row the same sum in the same order; rows past the end take part with factor 0.
__device__ __forceinline__ void hstat_rows(const float* __restrict__ X, size_t ld, int R, int C, int r_first, int r_stride,
                                           const float* __restrict__ rowscale, int rs_div, int lane, float& am, float& rn) {
  for (int r = r_first; r < R; r += 4 * r_stride) {
    const float* row[4];
    float sc[4], ss[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rk = r + k * r_stride;
      const bool ok = rk < R;
      const int rr = ok ? rk : r;
      row[k] = X + (size_t)rr * ld;
      sc[k] = !ok ? 0.f : (rowscale ? rowscale[rs_div == 1 ? rr : rr / rs_div] : 1.f);
      ss[k] = 0.f;
    }
    for (int c = lane; c < C; c += 64) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = row[k][c] * sc[k];
        am = fmaxf(am, fabsf(v));
        ss[k] += v * v;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) ss[k] = ofb_wave_sum(ss[k]);
    rn = fmaxf(fmaxf(rn, fmaxf(ss[0], ss[1])), fmaxf(ss[2], ss[3]));
  }
}
__global__ __launch_bounds__(256) void hstat_kernel(const float* __restrict__ X, int R, int C, int ld, ofb_hhdr* __restrict__ hdr,
                                                    const float* __restrict__ rowscale, int rs_div) {
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float am = 0.f, rn = 0.f;
  hstat_rows(X, (size_t)ld, R, C, blockIdx.x * 4 + w, gridDim.x * 4, rowscale, rs_div, lane, am, rn);
  am = ofb_wave_max_pos(am);
  if (lane == 0) { red[0][w] = am; red[1][w] = rn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ofb_atomic_max_pos(&hdr->amax, fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])));
    ofb_atomic_max_pos(&hdr->rn2sq, fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3])));
  }
}
// images for the patch matrix: amax only (rn2sq stays 0 = unknown)
__global__ __launch_bounds__(256) void hstat_flat_kernel(const float* __restrict__ X, size_t n, ofb_hhdr* __restrict__ hdr) {
  __shared__ float red[4];
  float am = 0.f;
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, gsz = (size_t)gridDim.x * 256;
  if ((reinterpret_cast<uintptr_t>(X) & 15) == 0) {
    const size_t n4 = n >> 2;
    for (size_t i = gid; i < n4; i += gsz) {
      const f32x4 v = reinterpret_cast<const f32x4*>(X)[i];
      am = fmaxf(fmaxf(am, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    for (size_t i = 4 * n4 + gid; i < n; i += gsz) am = fmaxf(am, fabsf(X[i]));
  } else {
    for (size_t i = gid; i < n; i += gsz) am = fmaxf(am, fabsf(X[i]));
  }
  am = ofb_wave_max_pos(am);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
  __syncthreads();
  if (threadIdx.x == 0) ofb_atomic_max_pos(&hdr->amax, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// Stage 2: X[R][C] row-major (ld) -> planes; e from hdr.amax (every thread reads the same word; block (0,0) records e)
// (bound given: the caller's device scalar replaces the measured maximum; block (0,0) then writes the whole header)
__global__ void to_hformat_kernel(const float* __restrict__ X, int R, int C, int ld, char* __restrict__ P, int ncb,
                                  const float* __restrict__ rowscale, int rs_div, const float* __restrict__ bound) {
  ofb_hhdr* hdr = reinterpret_cast<ofb_hhdr*>(P);
  const float amax = bound ? bound[0] : hdr->amax;
  const int e = ofb_h_exp(amax);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    hdr->e = e;
    if (bound) { hdr->amax = amax; hdr->rn2sq = 0.f; hdr->cn2sq = 0.f; }
  }
  const float s = ofb_h_pow2(e);
  const int c = blockIdx.x * blockDim.x + threadIdx.x, rg = blockIdx.y;
  if (c >= ncb * 16) return;
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = 4 * rg + t;
    float x = (r < R && c < C) ? X[(size_t)r * ld + c] : 0.f;
    if (rowscale && r < R) x *= rowscale[rs_div == 1 ? r : r / rs_div];
    v[t] = x * s;
  }
  ofb_store_h4(P + OFB_HHDR, ncb, rg, c, v[0], v[1], v[2], v[3]);
}
// Patch matrix of a conv-as-GEMM straight from the images (models/layers.py:177: Conv2d(k = s = patch) == patchify + Linear): row
// (b, py, px) x column (c, i, j) = img[b][c][py * patch + i][px * patch + j], written as planes without the [B*L][C*patch^2] f32 copy
__global__ void patchify_hformat_kernel(const float* __restrict__ img, int B, int Cin, int Hh, int Ww, int patch, char* __restrict__ P,
                                        int ncb) {
  ofb_hhdr* hdr = reinterpret_cast<ofb_hhdr*>(P);
  const int e = ofb_h_exp(hdr->amax);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) hdr->e = e;
  const float s = ofb_h_pow2(e);
  const int gw = Ww / patch, L = (Hh / patch) * gw, R = B * L, Cc = Cin * patch * patch;
  const int c = blockIdx.x * blockDim.x + threadIdx.x, rg = blockIdx.y;
  if (c >= ncb * 16) return;
  // (in-bounds addresses for every lane, no branch around the four loads: they leave together)
  const int cc = min(c, Cc - 1);
  const int ch = cc / (patch * patch), rem = cc - ch * patch * patch, i = rem / patch, j = rem - i * patch;
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = 4 * rg + t, rc = min(r, R - 1);
    const int b = rc / L, l = rc - b * L, py = l / gw, px = l - py * gw;
    v[t] = img[(((size_t)b * Cin + ch) * Hh + py * patch + i) * Ww + px * patch + j];
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = (4 * rg + t < R && c < Cc) ? v[t] * s : 0.f;
  ofb_store_h4(P + OFB_HHDR, ncb, rg, c, v[0], v[1], v[2], v[3]);
}
// Many matrices in ONE launch pair (the weights of the model, once per optimizer step).  Stage 1: HM_NB blocks per job leave their
// partial maxima in scratch[job][HM_NB][2] (no atomics, no memset); stage 2: every block of a job reduces those 2 x HM_NB words.
constexpr int HM_NB = 32;
__global__ __launch_bounds__(256) void hstat_multi_kernel(const ofb_hformat_job* __restrict__ jobs, float* __restrict__ scratch) {
  __shared__ float red[2][4];
  const ofb_hformat_job j = jobs[blockIdx.y];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float am = 0.f, rn = 0.f;
  hstat_rows(j.X, (size_t)j.ld, j.R, j.C, blockIdx.x * 4 + w, HM_NB * 4, j.rowscale, 1, lane, am, rn);
  am = ofb_wave_max_pos(am);
  if (lane == 0) { red[0][w] = am; red[1][w] = rn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = scratch + ((size_t)blockIdx.y * HM_NB + blockIdx.x) * 2;
    o[0] = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    o[1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  }
}
__global__ void to_hformat_multi_kernel(const ofb_hformat_job* __restrict__ jobs, const float* __restrict__ scratch) {
  const ofb_hformat_job j = jobs[blockIdx.z];
  const int c = blockIdx.x * blockDim.x + threadIdx.x, rg = blockIdx.y, ncb = (j.C + 15) >> 4;
  if (blockIdx.x * blockDim.x >= ncb * 16 || rg >= ((j.R + 15) >> 4) * 4) return;
  float am = 0.f, rn = 0.f;
  const float* sc = scratch + (size_t)blockIdx.z * HM_NB * 2;
#pragma unroll 8
  for (int i = 0; i < HM_NB; ++i) { am = fmaxf(am, sc[2 * i]); rn = fmaxf(rn, sc[2 * i + 1]); }
  const int e = ofb_h_exp(am);
  if (blockIdx.x == 0 && rg == 0 && threadIdx.x == 0) {
    ofb_hhdr* hdr = reinterpret_cast<ofb_hhdr*>(j.P);
    hdr->e = e; hdr->amax = am; hdr->rn2sq = rn; hdr->cn2sq = 0.f;
  }
  if (c >= ncb * 16) return;
  const float s = ofb_h_pow2(e);
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = 4 * rg + t;
    float x = (r < j.R && c < j.C) ? j.X[(size_t)r * j.ld + c] : 0.f;
    if (j.rowscale && r < j.R) x *= j.rowscale[r];
    v[t] = x * s;
  }
  ofb_store_h4((char*)j.P + OFB_HHDR, ncb, rg, c, v[0], v[1], v[2], v[3]);
}
// The same conversion for a gradient whose column sums are wanted as well (bias gradients: db = colsum(dY)): one pass over dY
// writes the planes AND partial[slab][c] = sum of the slab's (scaled, NOT 2^e-scaled) rows, added in row order.
__global__ __launch_bounds__(256) void to_hformat_colsum_kernel(const float* __restrict__ X, int R, int C, int ld, char* __restrict__ P,
                                                                int ncb, int rgs, const float* __restrict__ rowscale, int rs_div,
                                                                float* __restrict__ partial, const float* __restrict__ bound, int n_bound) {
  __shared__ float red[4][64];
  ofb_hhdr* hdr = reinterpret_cast<ofb_hhdr*>(P);
  float amax = bound ? bound[0] : hdr->amax;
  if (bound && n_bound > 1) {
    // a vector of partial maxima (one word per workgroup of the producer: ofb_attention_bwd_wgmax): every block reduces the same words
    // in the same order (a few KB out of the L2) - no atomics and no memset node on the producer's side
    float m = 0.f;
    for (int i = threadIdx.x; i < n_bound; i += 256) m = fmaxf(m, bound[i]);
    m = ofb_wave_max_pos(m);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = m;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    __syncthreads();                                      // red is reused for the column sums below
  }
  const int e = ofb_h_exp(amax);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    hdr->e = e;
    if (bound) { hdr->amax = amax; hdr->rn2sq = 0.f; hdr->cn2sq = 0.f; }
  }
  const float s = ofb_h_pow2(e);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = blockIdx.x * 64 + lane, slab = blockIdx.y;
  float sum = 0.f;
  if (c < ncb * 16) {
    const int rg1 = min(rgs, (slab + 1) * CS_SLAB_RG);
    for (int rg = slab * CS_SLAB_RG + w; rg < rg1; rg += 4) {
      float v[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = 4 * rg + t;
        float x = (r < R && c < C) ? OFB_NT_LOAD(X + (size_t)r * ld + c) : 0.f;       // (the f32 gradient is not read again)
        if (rowscale && r < R) x *= rowscale[rs_div == 1 ? r : r / rs_div];
        v[t] = x;
      }
      sum += (v[0] + v[1]) + (v[2] + v[3]);
      ofb_store_h4(P + OFB_HHDR, ncb, rg, c, v[0] * s, v[1] * s, v[2] * s, v[3] * s);
    }
  }
  red[w][lane] = sum;
  __syncthreads();
  if (w == 0 && c < ncb * 16) partial[(size_t)slab * (ncb * 16) + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// planes -> X[R][C]: (h1 + h2) 2^-e
__global__ void from_hformat_kernel(const char* __restrict__ P, int ncb, float* __restrict__ X, int R, int C, int ld) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, rg = blockIdx.y;
  if (c >= C) return;
  const float inv = ofb_h_pow2(-reinterpret_cast<const ofb_hhdr*>(P)->e);
  const char* slot = P + OFB_HHDR + ((size_t)rg * ncb + (c >> 4)) * GRAN + (c & 15) * 8;
  const uint2 a = *reinterpret_cast<const uint2*>(slot), b = *reinterpret_cast<const uint2*>(slot + 128);
  const float v[4] = {ofb_f16_lo(a.x) + ofb_f16_lo(b.x), ofb_f16_hi(a.x) + ofb_f16_hi(b.x), ofb_f16_lo(a.y) + ofb_f16_lo(b.y),
                      ofb_f16_hi(a.y) + ofb_f16_hi(b.y)};
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (4 * rg + t < R) X[(size_t)(4 * rg + t) * ld + c] = v[t] * inv;
}

// partial[slab][c] = sum over the slab's rows of X[r][c]: column sums of an H-format matrix (bias gradients of tensors that exist
// only as planes).  grid (ncb / 4, slabs), 256 threads = 4 waves x 64 columns; fixed order.
__global__ __launch_bounds__(256) void colsum_h_kernel(const char* __restrict__ P, int ncb, int rgs, float* __restrict__ partial, int ld) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, cb = blockIdx.x * 4 + (lane >> 4), slab = blockIdx.y;
  const float inv = ofb_h_pow2(-reinterpret_cast<const ofb_hhdr*>(P)->e);
  float s = 0.f;
  if (cb < ncb) {
    const int rg1 = min(rgs, (slab + 1) * CS_SLAB_RG);
    for (int rg = slab * CS_SLAB_RG + w; rg < rg1; rg += 4) {
      const char* slot = P + OFB_HHDR + ((size_t)rg * ncb + cb) * GRAN + (lane & 15) * 8;
      const uint2 a = *reinterpret_cast<const uint2*>(slot), b = *reinterpret_cast<const uint2*>(slot + 128);
      s += ofb_f16_lo(a.x) + ofb_f16_lo(b.x);
      s += ofb_f16_hi(a.x) + ofb_f16_hi(b.x);
      s += ofb_f16_lo(a.y) + ofb_f16_lo(b.y);
      s += ofb_f16_hi(a.y) + ofb_f16_hi(b.y);
    }
  }
  red[w][lane] = s * inv;
  __syncthreads();
  if (w == 0 && cb < ncb) partial[(size_t)slab * ld + blockIdx.x * 64 + lane] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ---- the GEMM -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz(int tg) { return ((tg >> 1) & 3) << 1; }

// T-layout of the aux tensor (OFB_ACT_GELU_GRAD_T / OFB_ACT_MULAUX_T; include/ofb_hip.h): the saved GELU derivative travels only from
// the fc1 product's epilogue to the epilogue of the product that forms dH - two launches of this kernel with the same output shape -
// so it is kept the way a wave of the 128 x 192 tile holds its accumulators: [tile (m / 128, n / 192)][wave 2 (r / 64) + (c / 96)]
// [16 x 16 block (4 row x 6 column blocks of the wave)][lane 16 ((r % 16) / 4) + c % 16][r % 4].  One wave instruction then moves
// 1 KB of it (16 bytes per lane, contiguous) straight between memory and the accumulators' registers: no trip through LDS.  Every
// other epilogue form (parked tiles, the narrow form, the fix-up kernel) addresses the same layout element by element.
__host__ __device__ __forceinline__ size_t aux_t_index(int row, int col, int nt192) {
  const int tm = row >> 7, r = row & 63, tn = col / 192, c = col - tn * 192;
  const int wq = (((row >> 6) & 1) << 1) + (c >= 96 ? 1 : 0), cc = c >= 96 ? c - 96 : c;
  const int blk = (r >> 4) * 6 + (cc >> 4), ln = (((r & 15) >> 2) << 4) + (cc & 15);
  return ((((size_t)tm * nt192 + tn) * 4 + wq) * 24 + blk) * 256 + ln * 4 + (r & 3);
}
__host__ __device__ __forceinline__ bool act_is_t(int act) { return act == OFB_ACT_GELU_GRAD_T || act == OFB_ACT_MULAUX_T; }
__host__ __device__ __forceinline__ bool act_is_gelug(int act) { return act == OFB_ACT_GELU_GRAD || act == OFB_ACT_GELU_GRAD_T; }
__host__ __device__ __forceinline__ bool act_is_mulaux(int act) { return act == OFB_ACT_MULAUX || act == OFB_ACT_MULAUX_T; }

// v = alpha*acc (+bias)(*colscale); act; (*rowscale); (+resid)   -- the fix-up kernel's form of the fused epilogue
__device__ __forceinline__ float epi_value(const ofb_gemm_h_args& g, float alpha, float accv, int row, int col, float bias, float cs, bool ok) {
  float v = (accv * alpha + bias) * cs;
  const size_t ai = act_is_t(g.act) ? aux_t_index(row, col, (g.N + 191) / 192) : (size_t)row * g.ldaux + col;
  if (g.act == OFB_ACT_GELU) {
    if (g.aux && ok) g.aux[ai] = v;
    v = ofb_gelu(v);
  } else if (act_is_gelug(g.act)) {
    float Phi, phi;
    ofb_gelu_parts(v, Phi, phi);
    if (ok) g.aux[ai] = Phi + v * phi;
    v *= Phi;
  } else if (g.act == OFB_ACT_DGELU) {
    v *= ofb_dgelu(ok ? g.aux[ai] : 0.f);
  } else if (act_is_mulaux(g.act)) {
    v *= ok ? g.aux[ai] : 0.f;
  }
  if (g.rowscale) v *= ok ? g.rowscale[g.rs_div == 1 ? row : row / g.rs_div] : 1.f;
  if (g.resid) v += ok ? g.resid[(size_t)row * g.ldr + col] : 0.f;
  return v;
}

// Upper bound of |output| -> the H-format output's header (e) and / or cbound_out, BEFORE the product runs.  Cauchy-Schwarz over
// the reduction: |sum_k a_k b_k| <= |a|_2 |b|_2, the operand norms from their headers (exact where the producer measured them,
// else K amax^2); then the epilogue: + max|bias|, x max|colscale|, x aux_bound for the multiplying activations (|gelu(v)| <= |v|),
// x max|rowscale|.  One block.  out_bound (device scalar) given: it IS the bound.
// (every thread of the block returns the same value: fixed reduction order, so every workgroup of a GEMM that folds this into its own
// start - below - derives the same bits as the one-block kernel)
template <int NWV>
__device__ __forceinline__ float gemm_h_bound_block(const ofb_gemm_h_args& g, float* red /* [3][NWV] */) {
  const int nthr = 64 * NWV, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (g.out_bound) return g.out_bound[0];
  float mb = 0.f, mc = g.colscale ? 0.f : 1.f, mr = g.rowscale ? 0.f : 1.f;
  if (g.bias) for (int i = threadIdx.x; i < g.N; i += nthr) mb = fmaxf(mb, fabsf(g.bias[i]));
  if (g.colscale) for (int i = threadIdx.x; i < g.N; i += nthr) mc = fmaxf(mc, fabsf(g.colscale[i]));
  if (g.rowscale) for (int i = threadIdx.x; i < (g.M + g.rs_div - 1) / g.rs_div; i += nthr) mr = fmaxf(mr, fabsf(g.rowscale[i]));
  mb = ofb_wave_max_pos(mb); mc = ofb_wave_max_pos(mc); mr = ofb_wave_max_pos(mr);
  if (lane == 0) { red[w] = mb; red[NWV + w] = mc; red[2 * NWV + w] = mr; }
  __syncthreads();
  mb = red[0]; mc = red[NWV]; mr = red[2 * NWV];
#pragma unroll
  for (int i = 1; i < NWV; ++i) { mb = fmaxf(mb, red[i]); mc = fmaxf(mc, red[NWV + i]); mr = fmaxf(mr, red[2 * NWV + i]); }
  const ofb_hhdr ha = *ofb_h_hdr(g.A), hb = *ofb_h_hdr(g.B);
  const float ka = (float)g.K * ha.amax * ha.amax, kb = (float)g.K * hb.amax * hb.amax;
  const float sa = g.a_kc ? ha.rn2sq : ha.cn2sq, sb = g.b_kc ? hb.rn2sq : hb.cn2sq;
  const float na = (sa > 0.f && sa < ka) ? sa : ka, nb = (sb > 0.f && sb < kb) ? sb : kb;
  float bound = (sqrtf(na) * sqrtf(nb) * fabsf(g.alpha) * 1.0001f + mb) * mc;
  if (g.act == OFB_ACT_DGELU || g.act == OFB_ACT_MULAUX) bound *= g.aux_bound > 0.f ? g.aux_bound : 1.13f;
  return bound * mr;
}
__device__ __forceinline__ void gemm_h_bound_publish(const ofb_gemm_h_args& g, float bound) {
  if (g.Cp) {
    ofb_hhdr* h = reinterpret_cast<ofb_hhdr*>(g.Cp);
    h->e = ofb_h_exp(bound); h->amax = bound; h->rn2sq = 0.f; h->cn2sq = 0.f;
  }
  if (g.cbound_out) g.cbound_out[0] = bound;
}
__global__ __launch_bounds__(256) void gemm_h_bound_kernel(const ofb_gemm_h_args g) {
  __shared__ float red[3 * 4];
  const float bound = gemm_h_bound_block<4>(g, red);
  if (threadIdx.x == 0) gemm_h_bound_publish(g, bound);
}

#ifdef OFB_H_STAMPS
// lab only (scripts/lab/stamp_gemm_h.py): s_memtime stamps of wave 0 of every workgroup: [wg][unit][4] = unit start, K loop start,
// K loop end, epilogue end; [wg][7][0..1] = s_memrealtime at kernel start / end (100 MHz), [wg][7][2] = HW_ID
__device__ unsigned long long ofb_h_stamps[1024 * 8 * 4];
#define OFB_HSTAMP(slot) do { if (t == 0 && sidx < 7 && blockIdx.x < 1024) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ofb_h_stamps[(blockIdx.x * 8 + sidx) * 4 + slot] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define OFB_HSTAMP(slot) do { } while (0)
#endif
#define OFB_VMW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void vm_wait(int n) {           // n is wave-uniform
  switch (n) {
    OFB_VMW(0) OFB_VMW(1) OFB_VMW(2) OFB_VMW(3) OFB_VMW(4) OFB_VMW(5) OFB_VMW(6) OFB_VMW(7) OFB_VMW(8) OFB_VMW(9) OFB_VMW(10) OFB_VMW(11)
    OFB_VMW(12) OFB_VMW(13) OFB_VMW(14) OFB_VMW(15) OFB_VMW(16) OFB_VMW(17) OFB_VMW(18) OFB_VMW(19) OFB_VMW(20) OFB_VMW(21) OFB_VMW(22)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// Epilogue forms are compile-time (EPI = set of E_* bits): with run-time flags hipcc has to assume that a side-input load may
// follow an aliasing store and puts s_waitcnt vmcnt(0) between the stores of every element.
enum : int { E_C = 1, E_P = 2, E_GELU = 4, E_DGELU = 8, E_RS = 16, E_RES = 32, E_ANY = 64, E_GELUG = 128, E_MULAUX = 256, E_RN = 512, E_AUXT = 1024 };

template <class CF, bool A_KC, bool B_KC, bool TAIL, int EPI>
__global__ __launch_bounds__(CF::NT, CF::WGS) void gemm_h_kernel(const ofb_gemm_h_args g, const Plan p) {
  constexpr int BM = CF::BM, BN = CF::BN, WN = CF::WN, MI = CF::MI, NI = CF::NI, HA = CF::HA, NST = CF::NST, NW = CF::NW, KH = CF::KH;
  constexpr int STAGE = CF::STAGE, A_BYTES = CF::A_BYTES, QA = CF::QA, QB = CF::QB, HR = CF::HR, TROW = CF::TROW;
  constexpr int NBA = BM / 32, NBB = BN / 32;                 // 32-row blocks of each operand's tile
  constexpr int BMT = CF::BMT, NPASS = CF::NPASS;             // rows the tile computes (its step along M); epilogue passes
  constexpr int MF = CF::MF, MB = CF::MB, NB = CF::NB;        // MFMA block size; blocks of the wave tile
  using acc_t = std::conditional_t<MF == 32, f32x16, f32x4>;
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE];
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, h = lane >> 5;
  const int wm0 = (w / WN) * (CF::MF * CF::MB), wn0 = (w % WN) * (CF::MF * CF::NB);
  const int a_ncb = g.a_ncb, b_ncb = g.b_ncb;
  const bool blast = w + NW * (QB - 1) < CF::B_PIECES;   // this wave moves a QB-th B piece per stage (wave-uniform)
  const int n_w = QA + QB - (blast ? 0 : 1);             // LDS-DMA instructions of this wave per stage

  // LDS-DMA source offsets (bytes, relative to the tile / stage base) of this wave's 1-KB pieces (ids w, w + NW, ...)
  //   KC: piece = (K16 sub-step kh, block j of 32 rows, plane): lane -> granule tg = l>>3 of the block, 16-B chunk cp = (l&7) ^ swz(tg)
  //       of that plane's 128-B slab; the XOR on the SOURCE chunk (LDS stays lane-linear) makes the transposed reads conflict free
  //   KR: the stage image is a linear copy of [4 KH row groups][B? / 16 granules] cut into 1-KB pieces = 4 granules; in the ODD
  //       granules the two plane slabs change places (source chunk ^ 8), so that the two 16-lane groups of a fragment read (granules
  //       g, g + 1 of one plane) hit opposite 128-B halves of the 256-B bank row.  MF = 16: the two groups of a half read the SAME
  //       granule of row groups 2 apart (k = 8 gq ...), so there the slabs change places in the row groups with bit 1 set
  // Where a piece never straddles a row group of the KR image (SPLIT: every tile but the 96-column B of C96) an offset is a LANE part
  // (ONE VGPR per operand) + a wave-uniform PIECE part (added to the scalar base): ten address VGPRs fewer than a full offset per
  // piece.  Wave w then moves the consecutive pieces QA w .. and QB w .. (= two whole row groups of a KR image each, so the MF = 16
  // slab swap is the same for all pieces of a wave: folded into its lane part)
  constexpr bool SPLIT = (BM * 16) % 1024 == 0 && (BN * 16) % 1024 == 0 && CF::B_PIECES % NW == 0;
  static_assert(MF == 32 || (SPLIT && QA * 1024 == 2 * BM * 16 && QB * 1024 == 2 * BN * 16), "MF = 16: two KR row groups per wave");
  unsigned a_off[QA], b_off[QB], a_pc[QA], b_pc[QB], a_ln, b_ln;
  {
    const int tg = lane >> 3, cp = (lane & 7) ^ swz(tg);
    const int kr_ln = (lane >> 4) * GRAN + ((lane & 15) ^ (((MF == 32 ? lane >> 4 : w) & 1) << 3)) * 16;
    a_ln = A_KC ? (unsigned)(tg * a_ncb * GRAN + cp * 16) : (unsigned)kr_ln;
    b_ln = B_KC ? (unsigned)(tg * b_ncb * GRAN + cp * 16) : (unsigned)kr_ln;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int piece = SPLIT ? QA * w + q : w + NW * q, kh = piece / (2 * NBA), rem = piece % (2 * NBA), j = rem >> 1, pl = rem & 1;
      const int bl = piece * 1024 + lane * 16, rgl = bl / (BM * 16), wi = bl % (BM * 16), gr = wi >> 8;
      const int ch = ((wi >> 4) & 15) ^ (((MF == 32 ? gr : rgl >> 1) & 1) << 3);
      a_off[q] = A_KC ? (unsigned)(((8 * j + tg) * a_ncb + kh) * GRAN + pl * 128 + cp * 16)
                      : (unsigned)((rgl * a_ncb + gr) * GRAN + ch * 16);
      const int prg = (piece * 1024) / (BM * 16), pg0 = ((piece * 1024) % (BM * 16)) >> 8;       // the piece's row group / first granule
      a_pc[q] = A_KC ? (unsigned)((8 * j * a_ncb + kh) * GRAN + pl * 128) : (unsigned)((prg * a_ncb + pg0) * GRAN);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int piece = SPLIT ? QB * w + q : w + NW * q, kh = piece / (2 * NBB), rem = piece % (2 * NBB), j = rem >> 1, pl = rem & 1;
      const int bl = piece * 1024 + lane * 16, rgl = bl / (BN * 16), wi = bl % (BN * 16), gr = wi >> 8;
      const int ch = ((wi >> 4) & 15) ^ (((MF == 32 ? gr : rgl >> 1) & 1) << 3);
      b_off[q] = B_KC ? (unsigned)(((8 * j + tg) * b_ncb + kh) * GRAN + pl * 128 + cp * 16)
                      : (unsigned)((rgl * b_ncb + gr) * GRAN + ch * 16);
      const int prg = (piece * 1024) / (BN * 16), pg0 = ((piece * 1024) % (BN * 16)) >> 8;
      b_pc[q] = B_KC ? (unsigned)((8 * j * b_ncb + kh) * GRAN + pl * 128) : (unsigned)((prg * b_ncb + pg0) * GRAN);
    }
  }
  // fragment read offsets (bytes inside an operand's stage image), two 8-byte reads per fragment
  //   KC: piece (kh, blk, plane) at ((kh NB + blk) 2 + plane) KB; inside it the transposed-read addresses r0 / r1
  //   KR: row groups 4 kh + 2 h, + 1; granule 2 blk + (l31 >> 4), whose plane slabs are swapped when it is odd
  //   MF = 16 (lane -> row / column lane & 15 of a 16-block, k = 8 gq + 0..7 of the K32 stage, gq = lane >> 4):
  //   KC: the lane's K16 sub-step gq >> 1 and k half gq & 1; the 16-block b of a 32-row piece starts 512 B in and its granules take
  //       the other XOR, so the even and the odd blocks have address registers of their own ([2 (b & 1) + read])
  //   KR: row groups 2 gq, 2 gq + 1 of the stage's eight; granule b, whose plane slabs are swapped when gq is odd
  int a_r0 = 0, a_r1 = 0, b_r0 = 0, b_r1 = 0, a_q[4] = {0, 0, 0, 0}, b_q[4] = {0, 0, 0, 0};
  if constexpr (MF == 32) {
    const int gq = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, hh = gq >> 1, tg = 4 * (gq & 1) + pp;
    const int c0 = 8 * hh + q, c1 = c0 + 4;
    const int kc0 = (8 * tg + ((c0 >> 1) ^ swz(tg))) * 16 + (c0 & 1) * 8, kc1 = (8 * tg + ((c1 >> 1) ^ swz(tg))) * 16 + (c1 & 1) * 8;
    const int odd = (l31 >> 4) & 1;
    const int kr_a0 = ((2 * h) * (BM / 16) + (l31 >> 4)) * GRAN + odd * 128 + (l31 & 15) * 8, kr_a1 = kr_a0 + (BM / 16) * GRAN;
    const int kr_b0 = ((2 * h) * (BN / 16) + (l31 >> 4)) * GRAN + odd * 128 + (l31 & 15) * 8, kr_b1 = kr_b0 + (BN / 16) * GRAN;
    a_r0 = A_KC ? kc0 : kr_a0; a_r1 = A_KC ? kc1 : kr_a1;
    b_r0 = B_KC ? kc0 : kr_b0; b_r1 = B_KC ? kc1 : kr_b1;
  } else {
    const int gq = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, khl = gq >> 1, hh = gq & 1;
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        const int tg = 4 * par + pp, c = 8 * hh + q + 4 * rd;
        const int kc = (8 * tg + ((c >> 1) ^ swz(tg))) * 16 + (c & 1) * 8;
        a_q[2 * par + rd] = A_KC ? khl * (NBA * 2048) + kc : ((2 * gq + rd) * (BM / 16)) * GRAN + hh * 128 + i * 8;
        b_q[2 * par + rd] = B_KC ? khl * (NBB * 2048) + kc : ((2 * gq + rd) * (BN / 16)) * GRAN + hh * 128 + i * 8;
      }
  }
  // KR: plane 1 sits at +128 in the granules whose slabs are in place, at -128 in the swapped ones
  const int kr_flip = 128 - 256 * (((MF == 32 ? l31 : lane) >> 4) & 1);

  acc_t acc[MB][NB];                                 // zeroed at the start of every unit: nothing of it lives across an epilogue

  // LDS-DMA in inline asm (through the builtin hipcc drains every LDS-DMA with vmcnt(0) before the next ds_read): invisible to
  // its wait-count bookkeeping, counted by hand (n_w per wave and stage).  The pieces of one wave sit NW KB apart in the stage
  // image, A's first, then B's (A_BYTES = QA * NW KB).  M0 is written in the statement that uses it and restored afterwards.
  const unsigned lds0 = (unsigned)(size_t)OFB_LDSP(lds) + (unsigned)w * 1024u;
  auto dma = [&](unsigned ldsaddr, unsigned voff, const char* base) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsaddr), "v"(voff), "s"(base) : "memory");
  };
  auto issue = [&](int buf, const char* a_src, const char* b_src, int steady = 0) __attribute__((always_inline)) {
    if (OFB_LAB_ABLATE == 2 && steady) return;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep) :: "memory");
    if constexpr (SPLIT) {
      const unsigned la = lds0 - (unsigned)w * 1024u + (unsigned)buf * (unsigned)STAGE + (unsigned)(QA * w) * 1024u;
      const unsigned lb = lds0 - (unsigned)w * 1024u + (unsigned)buf * (unsigned)STAGE + (unsigned)A_BYTES + (unsigned)(QB * w) * 1024u;
#pragma unroll
      for (int q = 0; q < QA; ++q) dma(la + q * 1024, a_ln, a_src + a_pc[q]);
#pragma unroll
      for (int q = 0; q < QB; ++q) if (OFB_LAB_ABLATE != 1 || steady == 0) dma(lb + q * 1024, b_ln, b_src + b_pc[q]);
    } else {
      const unsigned l0 = lds0 + (unsigned)buf * (unsigned)STAGE;
#pragma unroll
      for (int q = 0; q < QA; ++q) dma(l0 + q * (NW * 1024), a_off[q], a_src);
#pragma unroll
      for (int q = 0; q < QB - 1; ++q) dma(l0 + (QA + q) * (NW * 1024), b_off[q], b_src);
      if (blast) dma(l0 + (QA + QB - 1) * (NW * 1024), b_off[QB - 1], b_src);
    }
    asm volatile("s_mov_b32 m0, %0" :: "s"(keep) : "memory");
  };
  // one ds_read_b64 that stays one: as a relaxed wavefront-scope atomic load it is not a candidate for hipcc's pairing of LDS loads
  // into ds_read2_b64 / ds_read2st64_b64, which move half the bytes per LDS cycle of ds_read_b64 (MI355X_MICROARCH, LDS table) and,
  // paired across blocks, need register moves to form the fragments; the compiler still counts it in lgkmcnt
  auto lds_read_b64 = [&](const char* p) __attribute__((always_inline)) -> hs16x4 {
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    const unsigned long long v = __hip_atomic_load((lds_u64*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return __builtin_bit_cast(hs16x4, v);
  };
  auto frag = [&](const char* base, int r0, int r1, bool kc, int nb, int kh, int blk, int plane) __attribute__((always_inline)) -> ofb_f16x8 {
    hs16x4 lo4, hi4;
    if (kc) {
      const char* q = base + ((kh * nb + blk) * 2 + plane) * 1024;
      lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4*)(q + r0));
      hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4*)(q + r1));
    } else {
      const char* q = base + (4 * kh * (nb * 2) + 2 * blk) * GRAN + (plane ? kr_flip : 0);
      lo4 = lds_read_b64(q + r0);
      hi4 = lds_read_b64(q + r1);
    }
    hs16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(ofb_f16x8, v);
  };
  ofb_f16x8 alo[HA][2], ahi[HA][2], bb[2][NI][2];       // A fragments of the two half steps, B fragments of this / the next K16 sub-step
  auto rdA = [&](ofb_f16x8 (&dst)[HA][2], int buf, int kh, int blk0) __attribute__((always_inline)) {
    const char* la = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < HA; ++i)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) dst[i][pl] = frag(la, a_r0, a_r1, A_KC, NBA, kh, (wm0 >> 5) + blk0 + i, pl);
  };
  auto rdB = [&](ofb_f16x8 (&dst)[NI][2], int buf, int kh) __attribute__((always_inline)) {
    const char* lb = lds + buf * STAGE + A_BYTES;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) dst[j][pl] = frag(lb, b_r0, b_r1, B_KC, NBB, kh, (wn0 >> 5) + j, pl);
  };
  // MF = 16: one fragment = 16 rows x the whole K32 stage; A fragments of the lower / upper 16-blocks of the wave tile, B fragments of
  // its left / right ones (quarter steps below)
  auto frag16 = [&](const char* base, const int (&rq)[4], bool kc, int blk, int plane) __attribute__((always_inline)) -> ofb_f16x8 {
    hs16x4 lo4, hi4;
    if (kc) {
      const char* q = base + ((blk >> 1) * 2 + plane) * 1024;
      lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4*)(q + rq[2 * (blk & 1)]));
      hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4*)(q + rq[2 * (blk & 1) + 1]));
    } else {
      const char* q = base + blk * GRAN + (plane ? kr_flip : 0);
      lo4 = lds_read_b64(q + rq[0]);
      hi4 = lds_read_b64(q + rq[1]);
    }
    hs16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(ofb_f16x8, v);
  };
  constexpr int HML = (MB + 1) / 2, HMH = MB / 2, HNL = (NB + 1) / 2, HNR = NB / 2;       // lower / upper row blocks, left / right column blocks
  ofb_f16x8 fa_lo[HML][2], fa_hi[HMH > 0 ? HMH : 1][2], fb_l[HNL][2], fb_r[HNR > 0 ? HNR : 1][2];
  // (the address registers a_q / b_q carry the stage buffer: flip16 moves them to the other one - ONE copy of the stage code serves
  //  both buffers with immediate offsets only; two copies, one per buffer, cost a register shuffle of all accumulators where they join
  //  and spilled in the copies behind the loop)
  int q_buf = 0;                                              // 0 / STAGE: what a_q / b_q currently include
  auto flip16 = [&]() __attribute__((always_inline)) {
    const int d = q_buf ? -STAGE : STAGE;
#pragma unroll
    for (int k = 0; k < 4; ++k) { a_q[k] += d; b_q[k] += d; }
    q_buf = STAGE - q_buf;
  };
  auto rdA16 = [&](auto& dst, int blk0, int n) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < (int)(sizeof(dst) / sizeof(dst[0])); ++i)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        if (i < n) dst[i][pl] = frag16(lds, a_q, A_KC, (wm0 >> 4) + blk0 + i, pl);
  };
  auto rdB16 = [&](auto& dst, int blk0, int n) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < (int)(sizeof(dst) / sizeof(dst[0])); ++j)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        if (j < n) dst[j][pl] = frag16(lds + A_BYTES, b_q, B_KC, (wn0 >> 4) + blk0 + j, pl);
  };
  // an odd number of K16 steps: the last stage's second K16 half does not exist (what lies there is another row group's data or
  // uninitialised slack); its k belongs to the lanes >= 32 of every fragment, which are cleared
  auto clear_hi = [&](auto& fr) __attribute__((always_inline)) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < (int)(sizeof(fr) / sizeof(fr[0])); ++i)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        i32x4 v = __builtin_bit_cast(i32x4, fr[i][pl]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = lane < 32 ? v[e] : 0;
        fr[i][pl] = __builtin_bit_cast(ofb_f16x8, v);
      }
  };
  // product terms, smallest first: [(h2,h2)] (h2,h1) (h1,h2) (h1,h1)
  constexpr int NTERM = OFB_H_NTERM;
  constexpr int TA[4] = {1, 1, 0, 0}, TB[4] = {1, 0, 1, 0};
  constexpr int T0 = 4 - NTERM;
  constexpr int NMF = NTERM * HA * NI, RD1 = 4 * HA, RD2 = 4 * HA + 4 * NI;   // MFMAs per half step; fragment reads riding in the first / second half
  // MF = 16: MFMAs of the four quarter steps (lower / upper rows x left / right columns); fragment reads of the A / B halves
  constexpr int NQ1 = NTERM * HML * HNL, NQ2 = NTERM * HMH * HNL, NQ3 = NTERM * HML * HNR, NQ4 = NTERM * HMH * HNR;
  constexpr int RAL = 4 * HML, RAH = 4 * HMH, RBL = 4 * HNL, RBR = 4 * HNR;
  static_assert(MF == 32 || (RAH <= 2 * NQ1 && RBR <= 2 * NQ2 && RBL <= 2 * NQ3 && RAL <= 2 * NQ4), "fragment reads per MFMA gap");
#define OFB_MMA_HALF(AF, BF, BLK0)                                                                                                   \
  _Pragma("unroll") for (int q = T0; q < 4; ++q) _Pragma("unroll") for (int i = 0; i < HA; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) \
      acc[BLK0 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF[i][TA[q]], BF[j][TB[q]], acc[BLK0 + i][j], 0, 0, 0);
#if OFB_LAB_ABLATE == 3
#define OFB_MMA_QUARTER(AF, NA_, BF, NB_, I0, J0)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < (NA_); ++i) _Pragma("unroll") for (int j = 0; j < (NB_); ++j)                                 \
      asm volatile("" :: "v"(AF[i][0]), "v"(AF[i][1]), "v"(BF[j][0]), "v"(BF[j][1]));
#else
#define OFB_MMA_QUARTER(AF, NA_, BF, NB_, I0, J0)                                                                                    \
  _Pragma("unroll") for (int q = T0; q < 4; ++q) _Pragma("unroll") for (int i = 0; i < (NA_); ++i) _Pragma("unroll") for (int j = 0; j < (NB_); ++j) \
      acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AF[i][TA[q]], BF[j][TB[q]], acc[I0 + i][J0 + j], 0, 0, 0);
#endif
#define OFB_INTERLEAVE(NM, ND)                                                               \
  _Pragma("unroll") for (int z_ = 0; z_ < (NM); ++z_) {                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
    __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);                                      \
  }
  // ND reads in total spread over NM MFMA gaps (ND <= 2 NM): the first (ND - NM) gaps carry two
#define OFB_SPREAD(NM, ND)                                                                   \
  OFB_INTERLEAVE((ND) > (NM) ? (ND) - (NM) : 0, 2)                                           \
  OFB_INTERLEAVE((ND) > (NM) ? 2 * (NM) - (ND) : (NM), 1)

  const int v = ofb_xcd_remap(blockIdx.x, p.W);     // consecutive v share an XCD (and thus operand panels in its L2)
  int sidx = 0;
  Seg cur = get_seg<TAIL>(p, v, 0, (int)blockIdx.x);
  if (!cur.ok) return;
#ifdef OFB_H_STAMPS
  if (t == 0 && blockIdx.x < 1024) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    ofb_h_stamps[(blockIdx.x * 8 + 7) * 4 + 0] = __builtin_amdgcn_s_memrealtime();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    ofb_h_stamps[(blockIdx.x * 8 + 7) * 4 + 2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    ofb_h_stamps[(blockIdx.x * 8 + 7) * 4 + 3] = __builtin_amdgcn_s_memtime();
  }
#endif
  // 2^-(ea + eb): the planes hold A 2^ea and B 2^eb
  const float alpha = TAIL ? 1.f : g.alpha * ofb_h_pow2(-(ofb_h_hdr(g.A)->e + ofb_h_hdr(g.B)->e));
  const char* Apl = ofb_h_planes(g.A);
  const char* Bpl = ofb_h_planes(g.B);
  // Output bound folded into the product (plan.stagger bit 1; launches with full rounds whose side vectors are short): every workgroup
  // derives the same number before its first stage is requested (the loads are OLDER than every LDS-DMA piece, so the hand-counted
  // vmcnt waits of the K loop still cover what they must), workgroup 0 publishes it for the consumers and the tail's fix-up
  int fold_e = 0;
  if (!TAIL && (p.stagger & 2)) {
    float* red = reinterpret_cast<float*>(lds);
    const float bound = gemm_h_bound_block<NW>(g, red);
    if (blockIdx.x == 0 && t == 0) gemm_h_bound_publish(g, bound);
    fold_e = __builtin_amdgcn_readfirstlane(ofb_h_exp(bound));
    __syncthreads();                                      // red[] is read before the first stage lands on it
  }
  // plan.stagger bits 4-6 (single-round launches, host: plan_h): the first-dispatched workgroup of a CU that has a second one (blockIdx +
  // W / 2 holds a tile too) sleeps 128 n cycles at every stage hand-over - it runs its units ~1.4x faster than its neighbour and
  // would otherwise finish early and leave the CU to the slower one: the launch ends when the LAST workgroup does
  const int ysl = (!TAIL && (int)blockIdx.x + (p.W >> 1) < p.ntiles) ? (p.stagger >> 4) & 7 : 0;
  bool prefetched = false;                                // the unit's first stages were requested by the previous unit (direct epilogue)
  while (true) {
    const int nk16 = cur.it1 - cur.it0, nst = (nk16 + KH - 1) / KH;
    const bool last_full = nk16 == nst * KH;              // KH == 2: an odd number of K16 steps ends in a half stage
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < (int)(sizeof(acc_t) / 4); ++r) acc[i][j][r] = 0.f;
    const size_t a_k16 = A_KC ? GRAN : (size_t)4 * a_ncb * GRAN, b_k16 = B_KC ? GRAN : (size_t)4 * b_ncb * GRAN;
    const size_t a_step = a_k16 * KH, b_step = b_k16 * KH;
    const char* a_base = Apl + (A_KC ? (size_t)(cur.m0 / 4) * a_ncb * GRAN : (size_t)(cur.m0 / 16) * GRAN) + cur.it0 * a_k16;
    const char* b_base = Bpl + (B_KC ? (size_t)(cur.n0 / 4) * b_ncb * GRAN : (size_t)(cur.n0 / 16) * GRAN) + cur.it0 * b_k16;
    OFB_HSTAMP(0);
    if (prefetched) {
      // the previous unit's direct epilogue requested this unit's first stages before it touched its accumulators (below); its stores
      // are YOUNGER than those pieces and of a number only the compiler knows: wait for everything
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      prefetched = false;
    } else {
      __builtin_amdgcn_s_barrier();                      // every wave is past the previous unit's LDS traffic
      issue(0, a_base, b_base);
      if (nst > 1) issue(1, a_base + a_step, b_base + b_step);
      if (NST > 2 && nst > 2) issue(2, a_base + 2 * a_step, b_base + 2 * b_step);
      if (NST > 3 && nst > 3) issue(3, a_base + 3 * a_step, b_base + 3 * b_step);
      vm_wait(((nst < NST ? nst : NST) - 1) * n_w);      // stage 0 landed; the other prologue stages may be in flight
    }
    __builtin_amdgcn_s_barrier();
    OFB_HSTAMP(1);
    if constexpr (MF == 32) {
      rdA(alo, 0, 0, 0);
      rdB(bb[0], 0, 0);
    } else {
      if (q_buf) flip16();                                    // every unit starts in buffer 0
      rdA16(fa_lo, 0, HML);
      rdB16(fb_l, 0, HNL);
      if (OFB_LAB_ABLATE == 4) { rdA16(fa_hi, HML, HMH); rdB16(fb_r, HNL, HNR); }
    }
    // the hand-over inside the LAST sub-step of stage i: this wave is done reading buf(i), its pieces of stage i+1 have landed
    // (later stages may fly), everybody agrees (barrier), stage i + NST goes into buf(i)
    auto handover = [&](int i, int buf) __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      int young = nst - i - 2;
      young = young < 0 ? 0 : (young > NST - 2 ? NST - 2 : young);
      vm_wait(young * n_w);
      __builtin_amdgcn_s_barrier();
      if (i + NST < nst) issue(buf, a_base + (size_t)(i + NST) * a_step, b_base + (size_t)(i + NST) * b_step, 1);
      if (ysl) {                                           // OFB_TUNE_GEMM_YIELD: the faster workgroup of the CU steps aside for 128 n cycles
        if (ysl == 1) __builtin_amdgcn_s_sleep(2);
        else if (ysl == 2) __builtin_amdgcn_s_sleep(4);
        else if (ysl == 3) __builtin_amdgcn_s_sleep(6);
        else if (ysl == 4) __builtin_amdgcn_s_sleep(8);
        else if (ysl == 5) __builtin_amdgcn_s_sleep(10);
        else if (ysl == 6) __builtin_amdgcn_s_sleep(12);
        else __builtin_amdgcn_s_sleep(14);
      }
    };
    auto stage = [&](int i, int buf, bool full) __attribute__((always_inline)) {
      const int nbuf = buf + 1 == NST ? 0 : buf + 1;
      if constexpr (MF == 16) {
        // one K32 MFMA step in four quarters of NMQ MFMAs: (lower, left) (upper, left) | hand-over | (lower, right) (upper, right); the
        // fragment reads of the halves that come next ride in the MFMA gaps: upper A, right B of this stage, then left B and lower A
        // of the next one (after the last stage they fetch a stale buffer that nothing consumes)
        __builtin_amdgcn_sched_barrier(0);
        if (OFB_LAB_ABLATE != 4) rdA16(fa_hi, HML, HMH);
        if (!full) { clear_hi(fa_lo); clear_hi(fb_l); }
        OFB_MMA_QUARTER(fa_lo, HML, fb_l, HNL, 0, 0)
        OFB_SPREAD(NQ1, RAH)
        __builtin_amdgcn_sched_barrier(0);
        if (OFB_LAB_ABLATE != 4) rdB16(fb_r, HNL, HNR);
        if (!full) clear_hi(fa_hi);
        OFB_MMA_QUARTER(fa_hi, HMH, fb_l, HNL, HML, 0)
        OFB_SPREAD(NQ2, RBR)
        __builtin_amdgcn_sched_barrier(0);
        if (full) handover(i, buf);
        flip16();
        __builtin_amdgcn_sched_barrier(0);
        if (OFB_LAB_ABLATE != 4) rdB16(fb_l, 0, HNL);
        if (!full) clear_hi(fb_r);
        OFB_MMA_QUARTER(fa_lo, HML, fb_r, HNR, 0, HNL)
        OFB_SPREAD(NQ3, RBL)
        __builtin_amdgcn_sched_barrier(0);
        if (OFB_LAB_ABLATE != 4) rdA16(fa_lo, 0, HML);
        OFB_MMA_QUARTER(fa_hi, HMH, fb_r, HNR, HML, HNL)
        OFB_SPREAD(NQ4, RAL)
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (KH == 1) {
        __builtin_amdgcn_sched_barrier(0);
        rdA(ahi, buf, 0, HA);
        OFB_MMA_HALF(alo, bb[0], 0)
        OFB_SPREAD(NMF, RD1)
        __builtin_amdgcn_sched_barrier(0);
        handover(i, buf);
        __builtin_amdgcn_sched_barrier(0);
        rdA(alo, nbuf, 0, 0);
        rdB(bb[1], nbuf, 0);
        OFB_MMA_HALF(ahi, bb[0], HA)
        OFB_SPREAD(NMF, RD2)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NI; ++j) { bb[0][j][0] = bb[1][j][0]; bb[0][j][1] = bb[1][j][1]; }
      } else {
        // sub-step 0, first half: lower row blocks x B(kh 0); the reads of the upper row blocks ride in the MFMA gaps
        __builtin_amdgcn_sched_barrier(0);
        rdA(ahi, buf, 0, HA);
        OFB_MMA_HALF(alo, bb[0], 0)
        OFB_SPREAD(NMF, RD1)
        __builtin_amdgcn_sched_barrier(0);
        if (full) {
          // sub-step 0, second half: upper row blocks; the reads of sub-step 1 ride in the gaps
          rdA(alo, buf, 1, 0);
          rdB(bb[1], buf, 1);
          OFB_MMA_HALF(ahi, bb[0], HA)
          OFB_SPREAD(NMF, RD2)
          __builtin_amdgcn_sched_barrier(0);
          // sub-step 1, first half
          rdA(ahi, buf, 1, HA);
          OFB_MMA_HALF(alo, bb[1], 0)
          OFB_SPREAD(NMF, RD1)
          __builtin_amdgcn_sched_barrier(0);
          handover(i, buf);
          // sub-step 1, second half: the reads of stage i+1, sub-step 0 ride in the gaps (after the last stage they fetch a stale
          // buffer that nothing consumes)
          __builtin_amdgcn_sched_barrier(0);
          rdA(alo, nbuf, 0, 0);
          rdB(bb[0], nbuf, 0);
          OFB_MMA_HALF(ahi, bb[1], HA)
          OFB_SPREAD(NMF, RD2)
          __builtin_amdgcn_sched_barrier(0);
        } else {
          // the unit's last, half stage (odd number of K16 steps): nothing follows
          OFB_MMA_HALF(ahi, bb[0], HA)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    // A wave whose whole 64 x 96 part of the tile lies outside the matrix (ragged shapes: N = 264 in 192-column tiles, ...) takes
    // part in the staging and the barriers only: no fragment reads, no MFMAs (its accumulators stay zero and are never stored)
    if (cur.m0 + wm0 < g.M && cur.n0 + wn0 < g.N) {
      int buf = 0;
      if constexpr (MF == 16) {                               // one copy of the full stage + one of the half stage
        const int nfull = last_full ? nst : nst - 1;
        for (int i = 0; i < nfull; ++i) {
          stage(i, buf, true);
          buf = buf + 1 == NST ? 0 : buf + 1;
        }
        if (!last_full) stage(nst - 1, buf, false);
      } else {
        for (int i = 0; i + 1 < nst; ++i) {
          stage(i, buf, true);
          buf = buf + 1 == NST ? 0 : buf + 1;
        }
        if (last_full) stage(nst - 1, buf, true);
        else stage(nst - 1, buf, false);
      }
    } else {
      int buf = 0;
      for (int i = 0; i < nst; ++i) {
        if (i + 1 < nst || last_full || KH == 1) {
          int young = nst - i - 2;
          young = young < 0 ? 0 : (young > NST - 2 ? NST - 2 : young);
          vm_wait(young * n_w);
          __builtin_amdgcn_s_barrier();
          if (i + NST < nst) issue(buf, a_base + (size_t)(i + NST) * a_step, b_base + (size_t)(i + NST) * b_step);
        }
        buf = buf + 1 == NST ? 0 : buf + 1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // the trailing (unused) fragment reads
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // (a half last stage leaves its own DMA unwaited)
    OFB_HSTAMP(2);
    const Seg nxt = get_seg<TAIL>(p, v, sidx + 1, (int)blockIdx.x);

    // DIRECT epilogue (the two products that write 8 bytes per output element: fc1 -> GELU planes + GELU', dH <- GELU' -> planes): a
    // lane of a 16x16x32 accumulator block holds 4 consecutive rows of ONE column - exactly one 8-byte column slot of each H-format
    // plane - and the saved derivative travels in T-layout (aux_t_index: 16 contiguous bytes per lane and block), so the whole
    // epilogue runs from the accumulator registers: no LDS park, no read-back, no barriers.  LDS is then free while the epilogue
    // runs: the NEXT unit's first stages are requested before the accumulators are touched (their DMA round trip hides under the
    // epilogue's arithmetic and stores).  Row- and column-interior tiles only (whole tile inside the matrix); the others park.
    constexpr bool DIRECT_CT = MF == 16 && !TAIL && CF::WM == 2 && WN == 2 && MB == 4 && NB == 6 && BMT == 128 && BN == 192 && NST == 2 &&
                               (EPI == (E_P | E_GELUG | E_AUXT) || EPI == (E_P | E_MULAUX | E_AUXT)) && OFB_LAB_ABLATE != 9;
    bool direct_done = false;
    if constexpr (DIRECT_CT) {
      if ((p.stagger & 256) != 0 && cur.m0 + BMT <= g.M && cur.n0 + BN <= g.N) {
        direct_done = true;
        constexpr bool GELUG = (EPI & E_GELUG) != 0;
        // the epilogue's arithmetic and store issue go AHEAD of the co-resident workgroup's K loop (s_setprio 1 here, 0 again at the
        // end): fc1 126.5 / 130.1 -> 118.6 / 118.4 us, dH 117.6 -> 115.4, the twelve products 10.77 -> 10.58 ms; priority 3 gains less
        // (profiles/r06_gemm_epilogue_priority.txt).  The sooner a workgroup's stores are issued, the sooner its next K loop starts.
        __builtin_amdgcn_s_setprio(OFB_EPI_PRIO);
        if (nxt.ok) {
          const int n_nk16 = nxt.it1 - nxt.it0, n_nst = (n_nk16 + KH - 1) / KH;
          const char* na = Apl + (A_KC ? (size_t)(nxt.m0 / 4) * a_ncb * GRAN : (size_t)(nxt.m0 / 16) * GRAN) + nxt.it0 * a_k16;
          const char* nb = Bpl + (B_KC ? (size_t)(nxt.n0 / 4) * b_ncb * GRAN : (size_t)(nxt.n0 / 16) * GRAN) + nxt.it0 * b_k16;
          __builtin_amdgcn_s_barrier();                             // every wave has read its last fragments: the stage buffers are free
          issue(0, na, nb);
          if (n_nst > 1) issue(1, na + a_step, nb + b_step);
          prefetched = true;
        }
        const float so = ofb_h_pow2((p.stagger & 2) ? fold_e : reinterpret_cast<const ofb_hhdr*>(g.Cp)->e);
        const int c16 = lane & 15, g4 = lane >> 4, nt192 = (g.N + 191) / 192;
        char* Cpl = (char*)g.Cp + OFB_HHDR;
        float* auxq = g.aux + ((size_t)(((cur.m0 >> 7) * nt192 + cur.n0 / BN) * 4 + w) * 24) * 256 + lane * 4;   // this lane's entries, 256 floats per block
        float bv[NB], gv[NB];
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
          const int col = cur.n0 + wn0 + 16 * ni + c16;
          bv[ni] = g.bias ? g.bias[col] : 0.f;
          gv[ni] = g.colscale ? g.colscale[col] : 1.f;
        }
        f32x4 sa[2][MB];                                             // MULAUX: the saved derivative of column block ni, requested one block ahead
        if constexpr (!GELUG) {
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) sa[0][mi] = OFB_NT_LOAD(reinterpret_cast<const f32x4*>(auxq + (mi * NB) * 256));
        }
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
          if constexpr (!GELUG) {
            if (ni + 1 < NB) {
#pragma unroll
              for (int mi = 0; mi < MB; ++mi) sa[(ni + 1) & 1][mi] = OFB_NT_LOAD(reinterpret_cast<const f32x4*>(auxq + (mi * NB + ni + 1) * 256));
            }
          }
          const int col = cur.n0 + wn0 + 16 * ni + c16;
          float csum = 0.f;
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) {
            float o[4];
            f32x4 ax;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float val = (acc[mi][ni][r] * alpha + bv[ni]) * gv[ni];
              if constexpr (GELUG) {
                float Phi, phi;
                ofb_gelu_parts(val, Phi, phi);
                ax[r] = Phi + val * phi;
                val *= Phi;
              } else {
                val *= sa[ni & 1][mi][r];
              }
              o[r] = val;
            }
            if constexpr (GELUG) OFB_NT_STORE(ax, reinterpret_cast<f32x4*>(auxq + (mi * NB + ni) * 256));      // read by the backward only
            unsigned h1a, h2a, h1b, h2b;
            ofb_hsplit_pair(o[0] * so, o[1] * so, h1a, h2a);
            ofb_hsplit_pair(o[2] * so, o[3] * so, h1b, h2b);
            char* slot = Cpl + ((size_t)(((cur.m0 + wm0 + 16 * mi) >> 2) + g4) * g.c_ncb + (col >> 4)) * GRAN + (col & 15) * 8;
#if OFB_DIRECT_PAIR
            // two neighbouring lanes hold the 8-byte slots of two neighbouring columns: they swap one slot (DPP quad_perm 1,0,3,2), the
            // even lane then stores BOTH h1 slots (16 contiguous bytes), the odd lane both h2 slots - one dwordx4 store per lane and
            // block instead of two dwordx2
            const bool even = (lane & 1) == 0;
            const unsigned sa_ = even ? h2a : h1a, sb_ = even ? h2b : h1b;          // what the neighbour stores for me
            const unsigned ra = (unsigned)__builtin_amdgcn_mov_dpp((int)sa_, 0xB1, 0xF, 0xF, true);
            const unsigned rb = (unsigned)__builtin_amdgcn_mov_dpp((int)sb_, 0xB1, 0xF, 0xF, true);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v4 = even ? (u32x4){h1a, h1b, ra, rb} : (u32x4){ra, rb, h2a, h2b};
            OFB_NT_STORE(v4, reinterpret_cast<u32x4*>(even ? slot : slot + 120));    // odd: plane 2 of the pair = slot - 8 + 128
#else
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            OFB_NT_STORE(((u32x2){h1a, h1b}), reinterpret_cast<u32x2*>(slot));
            OFB_NT_STORE(((u32x2){h2a, h2b}), reinterpret_cast<u32x2*>(slot + 128));
#endif
            csum += (o[0] + o[1]) + (o[2] + o[3]);
          }
          if (g.colpart) {
            // column sums of the wave's 64 rows: the four lane groups hold four row quads of the column; the tile's two wave rows
            // leave one partial row each (rows 2 tile, 2 tile + 1: ofb_gemm_h_colpart_rows counts them)
            OFB_XOR_STEP(csum, ofb_add_, 16)                  // (register-only exchanges: ofb_common.h)
            OFB_XOR_STEP(csum, ofb_add_, 32)
            if (lane < 16) g.colpart[(size_t)(2 * (cur.m0 / BMT) + (w / WN)) * g.N + col] = csum;
          }
        }
        __builtin_amdgcn_s_setprio(0);
      }
    }
    if (!direct_done)

#if OFB_LAB_ABLATE == 9                                   // lab: no epilogue at all (the accumulators are kept alive)
    if constexpr (MF == 16) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) asm volatile("" :: "v"(acc[i][j]));
    } else
#endif
    {
      // Epilogue through LDS (the stage buffers are free now).  The accumulators of HR rows of the tile are parked ROW-major as
      // T[row][BN + 4] f32 and all waves finish those rows together.  WIDE form (interior tiles of launches whose f32 side tensors are
      // 16-byte aligned with row strides that are multiples of 4): one item = (4-row group, 4 adjacent columns) per thread and trip:
      // four ds_read_b128, every f32 load / store a dwordx4 (4 instead of 16 per 16 values), the H-format store two 16-byte stores per
      // plane (4 adjacent 8-byte column slots are contiguous), bias / gate one float4.  The store tail of a 128 x 192 tile was
      // issue-bound: 7 instructions per 4 outputs before, 10 per 16 now.  Edge tiles and unaligned launches take the
      // NARROW form: one (4-row group, column) item per thread, dword accesses, per-element guards.
      constexpr int TLD = BN + 4;
      static_assert((HR + HR / 4) * TLD * 4 <= NST * STAGE, "epilogue patch + column-sum rows");
      // (the row-norm hand-over needs room for its parts behind them and four threads per parked row: the 128 x 192 tiles; the host
      //  keeps launches that ask for it off the other tiles)
      constexpr bool RN_OK = ((HR + HR / 4) * TLD + HR * (BN / 4) + 4) * 4 <= NST * STAGE && CF::NT >= 4 * HR && (BN / 4) % 16 == 0;
      constexpr bool ANY = (EPI & E_ANY) != 0;
      const bool has_c = ANY ? g.C != nullptr : (EPI & E_C) != 0, has_p = ANY ? g.Cp != nullptr : (EPI & E_P) != 0;
      const bool gelu = ANY ? g.act == OFB_ACT_GELU : (EPI & E_GELU) != 0, dg = ANY ? g.act == OFB_ACT_DGELU : (EPI & E_DGELU) != 0;
      const bool gelug = ANY ? act_is_gelug(g.act) : (EPI & E_GELUG) != 0, mula = ANY ? act_is_mulaux(g.act) : (EPI & E_MULAUX) != 0;
      const bool auxt = ANY ? act_is_t(g.act) : (EPI & E_AUXT) != 0;           // the aux tensor is in T-layout (aux_t_index)
      const int nt192 = (g.N + 191) / 192;
      const bool has_rs = ANY ? g.rowscale != nullptr : (EPI & E_RS) != 0, has_res = ANY ? g.resid != nullptr : (EPI & E_RES) != 0;
      // E_RN (rn_out): per tile, max over its rows of rn_rowfac[row] * |rn_gamma (.) output row, this tile's columns|_2 - what the
      // LayerNorm backward that consumes this gradient needs for the exponent of ITS output planes (rowops.hip: the bound pass it
      // otherwise runs over the whole gradient).  Every item parks its rows' sums of squares in Rp[row][column quad] (plain stores: the
      // 48 items of a row group adding into one word by LDS atomics serialised the CU's atomic unit, +19 us per launch), four threads
      // per row add them up after the pass; the tile's maximum gathers in Rmx[0]
      const bool rn = RN_OK && !TAIL && (ANY ? g.rn_out != nullptr : (EPI & E_RN) != 0);
      float* T = reinterpret_cast<float*>(lds);
      float* Rp = reinterpret_cast<float*>(lds) + (HR + HR / 4) * TLD;         // [HR][BN / 4]
      float* Rmx = Rp + HR * (BN / 4);
      if (rn && t == 0) Rmx[0] = 0.f;                                  // (visible behind the barriers below; read back at the tile's end)
      float* __restrict__ Cout = g.C;
      float* __restrict__ auxw = g.aux;
      const float* __restrict__ auxr = g.aux;
      const float* __restrict__ resid = g.resid;
      const float* __restrict__ rowscale = g.rowscale;
      char* __restrict__ Cpl = g.Cp ? (char*)g.Cp + OFB_HHDR : nullptr;
      const float so = (!TAIL && has_p) ? ofb_h_pow2((p.stagger & 2) ? fold_e : reinterpret_cast<const ofb_hhdr*>(g.Cp)->e) : 1.f;   // (else: written by the bound kernel)
      const int rp_out = (g.M + 15) & ~15;
      // WIDE also serves tiles that stick out of the matrix on the COLUMN side (N = 264, 480, ... on 192-wide tiles: half of all
      // tiles of the pruned / finetune shapes): N is a multiple of 4 there, so a column quad lies wholly inside or wholly outside
      const bool wide = TAIL || (cur.m0 + BMT <= g.M && (p.stagger & 1) != 0);   // p.stagger bit 0: "wide epilogue allowed" (alignment checked on the host)
      constexpr int NQ = BN / 4, NIT = ((HR / 4) * NQ + CF::NT - 1) / CF::NT;         // column quads per row; items per full pass and thread
#ifdef OFB_LAB_EPI_PRIO_ALL
      __builtin_amdgcn_s_setprio(OFB_LAB_EPI_PRIO_ALL);             // lab: the parked epilogue ahead of the co-resident K loop as well
#endif
      __builtin_amdgcn_s_barrier();                                 // every wave has finished its fragment reads / its LDS-DMA
#pragma unroll
      for (int half = 0; half < NPASS; ++half) {
        const int prow = (BMT - HR * half) < HR ? (BMT - HR * half) : HR;           // rows of this pass (the last one of a 112-row tile: 48)
        const int NITEM = (prow / 4) * NQ;
        if constexpr (MF == 32) {
          if (wm0 / HR == half) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                  T[((wm0 % HR) + 32 * mi + 8 * (r >> 2) + 4 * h + (r & 3)) * TLD + wn0 + 32 * ni + l31] = acc[mi][ni][r];
          }
        } else {                                                      // 16 x 16 blocks (column on lane & 15, rows 4 (lane >> 4) + r): those of this pass
#pragma unroll
          for (int mi = 0; mi < MB; ++mi) {
            if ((wm0 + 16 * mi) / HR != half) continue;
#pragma unroll
            for (int ni = 0; ni < NB; ++ni)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                T[((wm0 + 16 * mi) % HR + 4 * (lane >> 4) + r) * TLD + wn0 + 16 * ni + (lane & 15)] = acc[mi][ni][r];
          }
        }
        __syncthreads();
        if (wide) {
          float* S = T + HR * TLD;                                    // [HR / 4 row groups][TLD]: the items' column sums (colpart only)
          // (one item at a time: with the items unrolled hipcc keeps three items' side inputs next to the other half's accumulators
          // and spills)
#pragma unroll 1
          for (int k = 0; k < NIT; ++k) {
            const int id = t + CF::NT * k;
            const bool initem = id < NITEM;
            const int rgl = initem ? id / NQ : 0, cq = initem ? id - (id / NQ) * NQ : 0;
            const int row0 = cur.m0 + HR * half + 4 * rgl, lcol = 4 * cq, col0 = cur.n0 + lcol;
            const bool live = initem && (TAIL || col0 < g.N);                       // this quad's columns exist
            const bool zpad = !TAIL && initem && !live && has_p && col0 < g.c_ncb * 16;      // padding columns of an H-format output: zeros
            const int col = live ? col0 : 0;
            f32x4 v[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) v[tt] = *reinterpret_cast<const f32x4*>(T + (4 * rgl + tt) * TLD + lcol);
            if (TAIL) {                                               // raw partial tile -> workspace[slot][BM][BN]
              float* __restrict__ ws = g.workspace + (size_t)cur.slot * (BM * BN) + (size_t)(HR * half + 4 * rgl) * BN + lcol;
              if (live) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(ws + tt * BN) = v[tt];
              }
              continue;
            }
            // side inputs of the item first (one exposed latency), then arithmetic, then stores
            f32x4 sa[4], sr[4];
            float rsv[4];
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, c4 = {1.f, 1.f, 1.f, 1.f};
            if (g.bias) b4 = *reinterpret_cast<const f32x4*>(g.bias + col);
            if (g.colscale) c4 = *reinterpret_cast<const f32x4*>(g.colscale + col);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
              const int row = row0 + tt;
              rsv[tt] = has_rs ? rowscale[g.rs_div == 1 ? row : row / g.rs_div] : 1.f;
              if ((dg || mula) && !auxt) sa[tt] = OFB_NT_LOAD(reinterpret_cast<const f32x4*>(auxr + (size_t)row * g.ldaux + col));   // last use of the saved derivative
              if (has_res) sr[tt] = *reinterpret_cast<const f32x4*>(resid + (size_t)row * g.ldr + col);
            }
            if (mula && auxt) {
              // T-layout: the item's four columns are four adjacent lanes' 16-byte entries (each: the four rows of one column)
              const float* q = auxr + aux_t_index(row0, col, nt192);
              f32x4 ft[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) ft[e] = OFB_NT_LOAD(reinterpret_cast<const f32x4*>(q + 4 * e));
#pragma unroll
              for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int e = 0; e < 4; ++e) sa[tt][e] = ft[e][tt];
            }
            f32x4 o[4], ax[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float val = (v[tt][e] * alpha + b4[e]) * c4[e];
                if (gelu) {
                  ax[tt][e] = val;
                  val = ofb_gelu(val);
                } else if (gelug) {
                  float Phi, phi;
                  if (OFB_LAB_ABLATE == 8) { Phi = 0.5f + 0.1f * val; phi = 0.1f; } else ofb_gelu_parts(val, Phi, phi);
                  ax[tt][e] = Phi + val * phi;
                  val *= Phi;
                } else if (dg) {
                  val *= ofb_dgelu(sa[tt][e]);
                } else if (mula) {
                  val *= sa[tt][e];
                }
                val *= rsv[tt];
                if (has_res) val += sr[tt][e];
                o[tt][e] = val;
              }
            }
            if (rn && initem) {
              const f32x4 gm = *reinterpret_cast<const f32x4*>(g.rn_gamma + col);
#pragma unroll
              for (int tt = 0; tt < 4; ++tt) {
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float q = gm[e] * o[tt][e]; ss += q * q; }
                Rp[(4 * rgl + tt) * NQ + cq] = live ? ss : 0.f;
              }
            }
            if (zpad) {
              char* slot = Cpl + ((size_t)(row0 >> 2) * g.c_ncb + (col0 >> 4)) * GRAN + (col0 & 15) * 8;
              const uint4 z = make_uint4(0u, 0u, 0u, 0u);
              *reinterpret_cast<uint4*>(slot) = z; *reinterpret_cast<uint4*>(slot + 16) = z;
              *reinterpret_cast<uint4*>(slot + 128) = z; *reinterpret_cast<uint4*>(slot + 144) = z;
            }
            if (OFB_LAB_ABLATE == 5) { asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3])); if (gelu || gelug) asm volatile("" :: "v"(ax[0]), "v"(ax[1]), "v"(ax[2]), "v"(ax[3])); }
            if (live && OFB_LAB_ABLATE != 5) {
              if (gelug && auxt && OFB_LAB_ABLATE != 6) {
                float* q = auxw + aux_t_index(row0, col, nt192);
#pragma unroll
                for (int e = 0; e < 4; ++e) OFB_NT_STORE(((f32x4){ax[0][e], ax[1][e], ax[2][e], ax[3][e]}), reinterpret_cast<f32x4*>(q + 4 * e));
              } else if (((gelu && auxw) || gelug) && OFB_LAB_ABLATE != 6) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) OFB_NT_STORE(ax[tt], reinterpret_cast<f32x4*>(auxw + (size_t)(row0 + tt) * g.ldaux + col));   // read by the backward only
              }
              if (has_c) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) *reinterpret_cast<f32x4*>(Cout + (size_t)(row0 + tt) * g.ldc + col) = o[tt];
              }
              if (has_p && OFB_LAB_ABLATE != 7) {
                // column e of the quad: rows 0..3 -> one 8-byte slot per plane; the quad's four slots are 32 contiguous bytes per plane
                unsigned h1[4][2], h2[4][2];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  ofb_hsplit_pair(o[0][e] * so, o[1][e] * so, h1[e][0], h2[e][0]);
                  ofb_hsplit_pair(o[2][e] * so, o[3][e] * so, h1[e][1], h2[e][1]);
                }
                char* slot = Cpl + ((size_t)(row0 >> 2) * g.c_ncb + (col >> 4)) * GRAN + (col & 15) * 8;
                // (H-format outputs are 155-MB streams at the DeiT-S sizes: written past the caches, -0.07 ms per step in a same-box A/B)
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                OFB_NT_STORE(((u32x4){h1[0][0], h1[0][1], h1[1][0], h1[1][1]}), reinterpret_cast<u32x4*>(slot));
                OFB_NT_STORE(((u32x4){h1[2][0], h1[2][1], h1[3][0], h1[3][1]}), reinterpret_cast<u32x4*>(slot + 16));
                OFB_NT_STORE(((u32x4){h2[0][0], h2[0][1], h2[1][0], h2[1][1]}), reinterpret_cast<u32x4*>(slot + 128));
                OFB_NT_STORE(((u32x4){h2[2][0], h2[2][1], h2[3][0], h2[3][1]}), reinterpret_cast<u32x4*>(slot + 144));
              }
            }
            if (g.colpart && initem) {
              f32x4 cs4;
#pragma unroll
              for (int e = 0; e < 4; ++e) cs4[e] = live ? (o[0][e] + o[1][e]) + (o[2][e] + o[3][e]) : 0.f;
              *reinterpret_cast<f32x4*>(S + rgl * TLD + lcol) = cs4;
            }
          }
          if (rn) {
            __syncthreads();                                          // every item's row parts are in
            if (RN_OK && t < 4 * prow) {
              const int rr = t >> 2, part = t & 3, row = cur.m0 + HR * half + rr;
              float ss = 0.f;
#pragma unroll
              for (int q4 = 0; q4 < NQ / 16; ++q4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(Rp + rr * NQ + (NQ / 4) * part + 4 * q4);
                ss += (v[0] + v[1]) + (v[2] + v[3]);
              }
              OFB_XOR_STEP(ss, ofb_add_, 1) OFB_XOR_STEP(ss, ofb_add_, 2)
              const float v = row < g.M ? g.rn_rowfac[row] * sqrtf(ss) : 0.f;
              const float m = ofb_wave_max_pos(v);
              if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(Rmx), __float_as_uint(m));
            }
          }
          if (!TAIL && g.colpart) {
            // column sums of this pass: one thread per column adds the row groups' sums in order; the passes of a tile in order too
            __syncthreads();
            if (t < BN && cur.n0 + t < g.N) {
              float sum = 0.f;
#pragma unroll
              for (int rg = 0; rg < HR / 4; ++rg) sum += (4 * rg < prow) ? S[rg * TLD + t] : 0.f;
              // (a launch that may take the direct epilogue keeps TWO partial rows per tile: a parked tile fills the first, zeroes the second)
              const bool two = DIRECT_CT && (p.stagger & 256) != 0;
              float* cp = g.colpart + (size_t)((two ? 2 : 1) * (cur.m0 / BMT)) * g.N + cur.n0 + t;
              if (half == 0) { *cp = sum; if (two) cp[g.N] = 0.f; } else *cp += sum;
            }
          }
        } else {
          // NARROW form: thread = (column lane + 64 c, 4-row groups w, w + NW, ...), per-element guards
          constexpr int NC = (BN + 63) / 64, NRG = (HR / 4) / NW;
          static_assert((HR / 4) % NW == 0, "epilogue item map");
          float cacc[NC];
#pragma unroll
          for (int c = 0; c < NC; ++c) cacc[c] = 0.f;
#pragma unroll 1
          for (int k = 0; k < NRG; ++k) {
            const int rgl = w + NW * k, row0 = cur.m0 + HR * half + 4 * rgl;
            if (4 * rgl >= prow) continue;                              // (a short last pass)
            float rss[4] = {0.f, 0.f, 0.f, 0.f};                        // E_RN: this lane's share of the four rows' sums of squares
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const int lcol = lane + 64 * c, col = cur.n0 + lcol;
              if (lcol >= BN) continue;
              const bool colok = col < g.N;
              const int colc = colok ? col : g.N - 1;
              const float biasv = g.bias ? g.bias[colc] : 0.f, csv = g.colscale ? g.colscale[colc] : 1.f;
              float pv[4];
#pragma unroll
              for (int tt = 0; tt < 4; ++tt) {
                const int row = row0 + tt;
                const bool ok = colok && row < g.M;
                const int rowc = row < g.M ? row : g.M - 1;
                float val = (T[(4 * rgl + tt) * TLD + lcol] * alpha + biasv) * csv;
                const size_t ai = auxt ? aux_t_index(rowc, colc, nt192) : (size_t)rowc * g.ldaux + colc;      // (== (row, col) wherever ok)
                if (gelu) {
                  if (auxw && ok) auxw[ai] = val;
                  val = ofb_gelu(val);
                } else if (gelug) {
                  float Phi, phi;
                  ofb_gelu_parts(val, Phi, phi);
                  if (ok) auxw[ai] = Phi + val * phi;
                  val *= Phi;
                } else if (dg) {
                  val *= ofb_dgelu(auxr[ai]);
                } else if (mula) {
                  val *= auxr[ai];
                }
                if (has_rs) val *= rowscale[g.rs_div == 1 ? rowc : rowc / g.rs_div];
                if (has_res) val += resid[(size_t)rowc * g.ldr + colc];
                if (has_c && ok) Cout[(size_t)row * g.ldc + col] = val;
                pv[tt] = ok ? val : 0.f;
              }
              cacc[c] += (pv[0] + pv[1]) + (pv[2] + pv[3]);
              if (rn) {
                const float gmv = colok ? g.rn_gamma[colc] : 0.f;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) { const float q = gmv * pv[tt]; rss[tt] += q * q; }
              }
              if (has_p && row0 < rp_out && col < g.c_ncb * 16)
                store_h4(Cpl + ((size_t)(row0 >> 2) * g.c_ncb + (col >> 4)) * GRAN + (col & 15) * 8, pv[0] * so, pv[1] * so, pv[2] * so,
                         pv[3] * so);
            }
            if (rn) {                                                   // a wave holds whole tile rows here: their sums by shuffles
              float m = 0.f;
#pragma unroll
              for (int tt = 0; tt < 4; ++tt) {
                const float ss = ofb_wave_sum(rss[tt]);
                if (row0 + tt < g.M) m = fmaxf(m, g.rn_rowfac[row0 + tt] * sqrtf(ss));
              }
              if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(Rmx), __float_as_uint(m));
            }
          }
          if (g.colpart) {
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NC; ++c)
              if (lane + 64 * c < BN) T[w * TLD + lane + 64 * c] = cacc[c];
            __syncthreads();
            if (t < BN && cur.n0 + t < g.N) {
              float sum = 0.f;
#pragma unroll
              for (int ww = 0; ww < NW; ++ww) sum += T[ww * TLD + t];
              // (a launch that may take the direct epilogue keeps TWO partial rows per tile: a parked tile fills the first, zeroes the second)
              const bool two = DIRECT_CT && (p.stagger & 256) != 0;
              float* cp = g.colpart + (size_t)((two ? 2 : 1) * (cur.m0 / BMT)) * g.N + cur.n0 + t;
              if (half == 0) { *cp = sum; if (two) cp[g.N] = 0.f; } else *cp += sum;
            }
          }
        }
        if (half + 1 < NPASS) __syncthreads();                    // T is rewritten by the next pass (the next unit starts with a barrier)
      }
      if (rn) {
        __syncthreads();
        if (t == 0) g.rn_out[(cur.m0 / BMT) * p.nt + cur.n0 / BN] = Rmx[0];
      }
#ifdef OFB_LAB_EPI_PRIO_ALL
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    OFB_HSTAMP(3);
    ++sidx;
    if (!nxt.ok) break;
    cur = nxt;
  }
#ifdef OFB_H_STAMPS
  if (t == 0 && blockIdx.x < 1024) {
    ofb_h_stamps[(blockIdx.x * 8 + 7) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// Sums the partial tiles of each streamed tail tile in a fixed contributor order and applies the epilogue.
// grid (R, BM / 4), BN threads: block (r, rg) handles rows [4*rg, 4*rg+4) of tail tile r, one column per thread.
template <class CF>
__global__ __launch_bounds__(CF::BN) void gemm_h_fixup_kernel(const ofb_gemm_h_args g, const Plan p) {
  constexpr int BM = CF::BM, BN = CF::BN;
  const int r = blockIdx.x, rgl = blockIdx.y, t = threadIdx.x;
  const int tile = p.full_rounds * p.W + r;
  int m0, n0;
  tile_coord(p, tile, m0, n0);
  const int lo = r * p.I, hi = lo + p.I;
  const int v0 = p.S ? 0 : lo / p.q, v1 = p.S ? (p.I + p.qs - 1) / p.qs - 1 : (hi - 1) / p.q;
  auto slot_of = [&](int v) { return p.S ? v * p.R + r : ((v * p.q < lo) ? 2 * v + 1 : 2 * v); };
  float sum[4] = {0.f, 0.f, 0.f, 0.f};
  int v = v0;
  for (; v + 3 <= v1; v += 4) {                       // four contributors per trip in flight, added in contributor order
    float x[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* ws = g.workspace + (size_t)slot_of(v + u) * (BM * BN) + (size_t)(4 * rgl) * BN + t;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) x[u][tt] = ws[tt * BN];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) sum[tt] += x[u][tt];
  }
  for (; v <= v1; ++v) {
    const float* ws = g.workspace + (size_t)slot_of(v) * (BM * BN) + (size_t)(4 * rgl) * BN + t;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) sum[tt] += ws[tt * BN];
  }
  const float alpha = g.alpha * ofb_h_pow2(-(ofb_h_hdr(g.A)->e + ofb_h_hdr(g.B)->e));
  const float so = g.Cp ? ofb_h_pow2(reinterpret_cast<const ofb_hhdr*>(g.Cp)->e) : 1.f;
  const int row0 = m0 + 4 * rgl, col = n0 + t;
  const bool colok = col < g.N;
  const float bias = (g.bias && colok) ? g.bias[col] : 0.f, cs = (g.colscale && colok) ? g.colscale[col] : 1.f;
  float pv[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int row = row0 + tt;
    const bool ok = colok && row < g.M;
    const float val = epi_value(g, alpha, sum[tt], row, col, bias, cs, ok);
    if (g.C && ok) g.C[(size_t)row * g.ldc + col] = val;
    pv[tt] = ok ? val : 0.f;
  }
  if (g.Cp && row0 < ((g.M + 15) & ~15) && col < g.c_ncb * 16)
    store_h4((char*)g.Cp + OFB_HHDR + ((size_t)(row0 >> 2) * g.c_ncb + (col >> 4)) * GRAN + (col & 15) * 8, pv[0] * so, pv[1] * so, pv[2] * so,
             pv[3] * so);
  if (g.colpart && colok)
    g.colpart[((size_t)(p.mt - p.R / p.nt) + (size_t)(r / p.nt) * (BM / 4) + rgl) * g.N + col] = (pv[0] + pv[1]) + (pv[2] + pv[3]);
}

// Run-time switches (ofb_tune): -1 = not set yet (the environment variable of the same meaning is read once, then the default)
int h_tune[OFB_TUNE_COUNT] = {-1, -1, -1, -1, -1, -1, -1};
int h_switch(int key, const char* env, int dflt) {
  if (h_tune[key] < 0) { const char* e = getenv(env); h_tune[key] = e ? atoi(e) : dflt; }
  return h_tune[key];
}

// CUs the plans count on: the device's, or fewer when the caller says that some are held by another stream's kernels
// (OFB_TUNE_GEMM_CUS: a data-parallel exchange during backward) - the persistent grid then leaves those slots to the hardware
int h_cu_count() {
  static int n = 0;
  if (n == 0) {
    hipDeviceProp_t prop;
    int devid = 0;
    n = 256;
    if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess && prop.multiProcessorCount > 0)
      n = prop.multiProcessorCount;
  }
  const int lim = h_switch(OFB_TUNE_GEMM_CUS, "OFB_GEMM_H_CUS", 0);
  return (lim > 0 && lim < n) ? lim : n;
}

int h_tile_choice(const ofb_gemm_h_args& g) {
  static int env_forced = -1;
  if (env_forced < 0) { const char* e = getenv("OFB_GEMM_H_TILE"); env_forced = e ? atoi(e) : 0; }
  const int forced = h_tune[OFB_TUNE_GEMM_TILE] >= 0 ? h_tune[OFB_TUNE_GEMM_TILE] : env_forced;     // ofb_tune (0 included) overrides the environment
  if (forced == 128) return 128;
#ifdef OFB_GEMM_H_LAB
  if (h_tune[OFB_TUNE_GEMM_TILE] == 1281 || h_tune[OFB_TUNE_GEMM_TILE] == 1283) return h_tune[OFB_TUNE_GEMM_TILE];
  if (forced == 1281 || forced == 1283) return forced;
#endif
  // The 112-row tile (C112F): token-row products (A reduced along its columns) whose 128-row tiles fill more than half a round but
  // not more than one round of the 2 x CUs workgroups, and whose 112-row tiles still fit one round: the launch ends with its slowest
  // co-resident pair either way, and that pair has 7/8 of the rows.  OFB_GEMM_H_TILE=128 / ofb_tune keeps the 128-row tile.
  if (g.a_kc && !g.colpart && forced != 96 && forced != 97 && h_switch(OFB_TUNE_GEMM_MFMA, "OFB_GEMM_H_MFMA", 16) == 16 &&
      h_switch(OFB_TUNE_GEMM_T112, "OFB_GEMM_H_T112", 0) != 0) {
    const int W2 = h_cu_count() * C128F::WGS, nt = ofb_cdiv(g.N, C128F::BN);
    const long long n128 = (long long)ofb_cdiv(g.M, C128F::BMT) * nt, n112 = (long long)ofb_cdiv(g.M, C112F::BMT) * nt;
    if (2 * n128 > W2 && n128 <= W2 && n112 <= W2) return 112;
  }
  if (g.a_kc && !g.colpart && !g.rn_out && g.M >= 4 * C96::BM) {
    const int c192 = ofb_cdiv(g.N, 192) * 192, c96 = ofb_cdiv(g.N, 96) * 96;
    if (forced == 96) return 96;
    // Round 5: with the 128 x 192 tile on 16x16x32 (two K32 stages) the 256 x 96 tile (32x32x16, three K16 stages: a 16x16x32 form
    // does not fit two workgroups' LDS) no longer pays for the widths it was built for - forced onto 128 x 192 the pruned search
    // step runs 10.1-11.2 k images/s against 9.99 k, the finetune subnet 13.15 k against 13.05 k (same box,
    // profiles/r05_tile_choice_pruned_finetune.txt).  OFB_GEMM_H_TILE=97 restores the padded-columns model of rounds 3-4.
    if (forced == 97 && c96 < c192) {
      const double rounds = (double)ofb_cdiv(g.M, C96::BM) * (c96 / 96) / (double)(h_cu_count() * C96::WGS);
      const double frac = rounds - (double)(long long)rounds;
      if (!(frac > 0.0 && frac < 0.3 && g.K < 384)) return 96;
    }
  }
  return 128;
}

// the launch has the form of the direct epilogue (gemm_h_kernel: DIRECT_CT): planes out only, the saved derivative in T-layout, no row
// scale / residual / row-norm request, the 128 x 192 tile on 16x16x32.  OFB_TUNE_GEMM_DIRECT / OFB_GEMM_H_DIRECT=0: every tile parks.
bool h_direct_ok(const ofb_gemm_h_args& g) {
  return act_is_t(g.act) && g.Cp && !g.C && !g.rowscale && !g.resid && !g.rn_out && g.aux && g.c_ncb * 16 >= g.N &&
         h_switch(OFB_TUNE_GEMM_DIRECT, "OFB_GEMM_H_DIRECT", 1) != 0 && h_switch(OFB_TUNE_GEMM_MFMA, "OFB_GEMM_H_MFMA", 16) == 16 &&
         h_tile_choice(g) == 128;
}

template <class CF>
Plan plan_h(const ofb_gemm_h_args& g) {
  int W = h_cu_count() * CF::WGS;
  if (!g.a_kc && !g.b_kc) {
    // weight-gradient form: ONE workgroup per CU (half the partial-tile traffic of two at an equal step time)
    static int dw_wgs = -1;
    if (dw_wgs < 0) { const char* e = getenv("OFB_GEMM_H_DW_WGS"); dw_wgs = e ? atoi(e) : 1; }
    if (dw_wgs > 0 && dw_wgs < CF::WGS) W = h_cu_count() * dw_wgs;
  }
  const int tiles = ofb_cdiv(g.M, CF::BMT) * ofb_cdiv(g.N, CF::BN);
  const long long iters = (long long)tiles * ofb_cdiv(g.K, 16);
  if (iters < W) W = (int)iters;
  Plan p = make_plan(g.M, g.N, g.K, W, CF::BMT, CF::BN, 16);
  if (CF::BMT != CF::BM && p.R > 0) { p.full_rounds += 1; p.R = 0; p.q = 0; p.S = 0; p.qs = 0; }   // (no stream-K form of the 112-row tile)
  if (p.R > 0 && g.rn_out) { p.full_rounds += 1; p.R = 0; p.q = 0; p.S = 0; p.qs = 0; }   // (the fix-up kernel has no row-norm form)
  if (p.R > 0 && g.colpart && p.R % p.nt != 0) {                     // column sums of tail tiles come from the fix-up kernel, which
    p.full_rounds += 1; p.R = 0; p.q = 0; p.S = 0; p.qs = 0;         // addresses them per whole tile row
  }
  // A remainder of at most half a round behind >= 2 full rounds runs as one more, partial round on the FIRST-dispatched workgroup of
  // every CU (get_seg's spread form): those run their units ~1.4x faster than the later-dispatched ones, so full_rounds + 1 of
  // their units end about when the others' full_rounds do - the remainder costs (almost) no time, no partial tiles, no fix-up launch
  const bool spread_on = h_switch(OFB_TUNE_GEMM_SCHED, "OFB_GEMM_H_SPREAD", 2) != 0;
  if (spread_on && p.R > 0 && p.full_rounds >= 2 && 2 * p.R <= W && (g.a_kc || g.b_kc)) {
    p.full_rounds += 1; p.R = 0; p.q = 0; p.S = 0; p.qs = 0;
  }
  if (p.R > 0) {
    // costs in K16 steps of one workgroup: a streamed tail = its K steps + the partial tiles' round trip through HBM + the extra
    // launches, against one more (partly idle) data-parallel round of I steps
    const double PC = 0.03, FIX = 5.0;
    double best = 0.9 * p.I;
    int bestS = -1;
    const int smax = W / p.R;
    for (int S = 2; S <= smax; ++S) {
      const double c = (double)((p.I + S - 1) / S) + (double)p.R * S * PC + FIX;
      if (c < best) { best = c; bestS = S; }
    }
    if (smax < 2) {
      const double c = (double)(((long long)p.R * p.I + W - 1) / W) + (double)(2 * p.R < 2 * W ? 2 * p.R : 2 * W) * PC + FIX;
      if (c < best) { best = c; bestS = 0; }
    }
    if (bestS < 0) { p.full_rounds += 1; p.R = 0; p.q = 0; p.S = 0; p.qs = 0; }
    else if (bestS > 0) { p.S = bestS; p.qs = (p.I + bestS - 1) / bestS; }
    else { p.S = 0; p.qs = 0; }
  }
  // "wide epilogue allowed" rides in plan.stagger: every f32 side tensor 16-byte aligned with a row stride that is a multiple of 4,
  // N a multiple of 4 (so that tile-interior column quads stay aligned)
  auto al = [](const void* q, int ld) { return q == nullptr || (ofb_aligned16(q) && (ld & 3) == 0); };
  static int narrow = -1;
  if (narrow < 0) { const char* e = getenv("OFB_GEMM_H_NARROW"); narrow = e ? atoi(e) : 0; }
  const bool wide_ok = !narrow && (g.N & 3) == 0 && al(g.C, g.ldc) && al(g.aux, g.ldaux) && al(g.resid, g.ldr) && al(g.bias, 0) && al(g.colscale, 0) &&
                       (g.Cp == nullptr || (g.c_ncb * 16 >= g.N));
  // bit 1: the output bound is computed by the product kernel itself (gemm_h_kernel) instead of a one-block launch ahead of it
  static int fold = -1;
  if (fold < 0) { const char* e = getenv("OFB_GEMM_H_BOUND_FOLD"); fold = e ? atoi(e) : 1; }
  const long long side = (g.out_bound ? 0 : (long long)(g.bias ? g.N : 0) + (g.colscale ? g.N : 0) + (g.rowscale ? (g.M + g.rs_div - 1) / g.rs_div : 0));
  const bool fold_ok = fold && (g.Cp || g.cbound_out) && p.full_rounds > 0 && side <= 8192;
  // bit 3: the partial last round of a multi-round launch is spread over the XCDs by the raw block index (gemm_plan.h: get_seg)
  const bool spread = spread_on && p.full_rounds >= 2 && p.R == 0 && p.ntiles < p.full_rounds * p.W;
  // bit 7: a launch of ONE partial round hands every XCD an equal share of the tiles, in dispatch order (gemm_plan.h: get_seg);
  // bits 4-6: and the first-dispatched workgroup of every fully occupied CU yields at its stage hand-overs (gemm_h_kernel: ysl)
  const int yl = h_switch(OFB_TUNE_GEMM_YIELD, "OFB_GEMM_H_YIELD", 4);
  const bool balance1 = h_switch(OFB_TUNE_GEMM_SCHED, "OFB_GEMM_H_SPREAD", 2) >= 2 && p.R == 0 && p.full_rounds == 1 && p.ntiles < p.W && p.ntiles >= 8;
  const bool one_round = balance1 && 2 * p.ntiles > p.W && CF::WGS == 2 && p.W == h_cu_count() * 2;
  // bit 8: interior tiles of this launch take the direct epilogue (gemm_h_kernel: DIRECT_CT; h_direct_ok: the launch has the form)
  const bool direct = std::is_same<CF, C128F>::value && p.R == 0 && h_direct_ok(g);
  p.stagger = (wide_ok ? 1 : 0) | (fold_ok ? 2 : 0) | (spread ? 8 : 0) | ((yl > 0 && one_round) ? ((yl > 7 ? 7 : yl) << 4) : 0) | (balance1 ? 128 : 0) |
              (direct ? 256 : 0);
  return p;
}

template <class CF, bool A_KC, bool B_KC, int EPI>
void launch_full(const ofb_gemm_h_args& g, const Plan& p, hipStream_t s) {
  hipLaunchKernelGGL((gemm_h_kernel<CF, A_KC, B_KC, false, EPI>), dim3(p.W), dim3(CF::NT), 0, s, g, p);
}

template <class CF, bool A_KC, bool B_KC>
int launch_h(const ofb_gemm_h_args& g, const Plan& p, hipStream_t s) {
  if ((g.Cp || g.cbound_out) && !(p.stagger & 2)) hipLaunchKernelGGL(gemm_h_bound_kernel, dim3(1), dim3(256), 0, s, g);
  if (p.full_rounds > 0) {
    const int f = (g.C ? E_C : 0) | (g.Cp ? E_P : 0) | (g.act == OFB_ACT_GELU ? E_GELU : 0) | (g.act == OFB_ACT_DGELU ? E_DGELU : 0) |
                  (act_is_gelug(g.act) ? E_GELUG : 0) | (act_is_mulaux(g.act) ? E_MULAUX : 0) | (act_is_t(g.act) ? E_AUXT : 0) |
                  (g.rowscale ? E_RS : 0) | (g.resid ? E_RES : 0) | (g.rn_out ? E_RN : 0);
    switch (f) {     // the forms the model issues; anything else takes the generic (run-time flags) instantiation
      case E_C: launch_full<CF, A_KC, B_KC, E_C>(g, p, s); break;
      case E_C | E_RES: launch_full<CF, A_KC, B_KC, E_C | E_RES>(g, p, s); break;
      case E_C | E_RES | E_RN: launch_full<CF, A_KC, B_KC, E_C | E_RES | E_RN>(g, p, s); break;
      case E_C | E_RN: launch_full<CF, A_KC, B_KC, E_C | E_RN>(g, p, s); break;
      case E_C | E_RS | E_RES: launch_full<CF, A_KC, B_KC, E_C | E_RS | E_RES>(g, p, s); break;
      case E_P | E_GELUG: launch_full<CF, A_KC, B_KC, E_P | E_GELUG>(g, p, s); break;
      case E_P | E_MULAUX: launch_full<CF, A_KC, B_KC, E_P | E_MULAUX>(g, p, s); break;
      case E_P | E_GELUG | E_AUXT: launch_full<CF, A_KC, B_KC, E_P | E_GELUG | E_AUXT>(g, p, s); break;
      case E_P | E_MULAUX | E_AUXT: launch_full<CF, A_KC, B_KC, E_P | E_MULAUX | E_AUXT>(g, p, s); break;
      case E_P: launch_full<CF, A_KC, B_KC, E_P>(g, p, s); break;
      default: launch_full<CF, A_KC, B_KC, E_ANY>(g, p, s); break;
    }
  }
  if (p.R > 0) {
    hipLaunchKernelGGL((gemm_h_kernel<CF, A_KC, B_KC, true, 0>), dim3(p.W), dim3(CF::NT), 0, s, g, p);
    hipLaunchKernelGGL(gemm_h_fixup_kernel<CF>, dim3(p.R, CF::BM / 4), dim3(CF::BN), 0, s, g, p);
  }
  return ofb_launch_status();
}

template <class CF>
int run_h(const ofb_gemm_h_args& g, hipStream_t s) {
  const Plan p = plan_h<CF>(g);
  if ((long long)p.W * p.I > 0x7fffffffLL / 2) return OFB_ELIMIT;
  if (p.R && (!g.workspace || g.workspace_bytes < (int64_t)2 * p.W * CF::BM * CF::BN * (int64_t)sizeof(float))) return OFB_EINVAL;
  if (g.a_kc && g.b_kc) return launch_h<CF, true, true>(g, p, s);
  if (g.a_kc) return launch_h<CF, true, false>(g, p, s);
  return launch_h<CF, false, false>(g, p, s);
}

}  // namespace

extern "C" int64_t ofb_hformat_bytes(int32_t R, int32_t C) {
  if (R <= 0 || C <= 0) return 0;
  // Tile-granular reads run past the matrix: mode KC reads whole 128- / 192- / 256-row tiles and one granule column past an odd K16
  // count, so the row groups are allocated up to the next 256-row boundary plus one more 256-row tile; mode KR reads whole tiles of
  // columns and up to 8 row groups per stage.  The slack is never initialised and only ever feeds accumulators that are not stored
  // (rows / columns beyond the matrix) or MFMAs that are not issued (the second half of an odd last stage).
  const int64_t ncb = (C + 15) / 16, rgs = (int64_t)((R + 255) / 256) * 64 + 64;
  return OFB_HHDR + (rgs * ncb + 16) * GRAN;
}

namespace {
// header.amax for the conversion kernels: measured (memset + statistics pass) or the caller's bound
int h_prepare_header(const float* X, int R, int C, int ld, void* P, const float* rowscale, int rs_div, const float* bound, hipStream_t s) {
  if (bound) return 0;                                 // the conversion kernel reads the caller's scalar itself
  if (hipMemsetAsync(P, 0, 16, s) != hipSuccess) return (int)hipGetLastError();
  const int nb = ofb_cdiv(R, 4) < 512 ? ofb_cdiv(R, 4) : 512;
  hipLaunchKernelGGL(hstat_kernel, dim3(nb), dim3(256), 0, s, X, R, C, ld, (ofb_hhdr*)P, rowscale, rs_div);
  return 0;
}
}  // namespace

extern "C" int ofb_to_hformat(const float* X, int32_t R, int32_t C, int32_t ld, void* P, const float* rowscale, int32_t rs_div,
                              const float* bound, void* stream) {
  if (!X || !P || R <= 0 || C <= 0 || ld < C) return OFB_EINVAL;
  if (rowscale && rs_div <= 0) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int ncb = (C + 15) / 16, rgs = ((R + 15) / 16) * 4;
  if (int rc = h_prepare_header(X, R, C, ld, P, rowscale, rs_div, bound, s)) return rc;
  hipLaunchKernelGGL(to_hformat_kernel, dim3((ncb * 16 + 255) / 256, rgs), dim3(256), 0, s, X, R, C, ld, (char*)P, ncb, rowscale, rs_div, bound);
  return ofb_launch_status();
}

// img [B][Cin][H][W] -> planes of the patch matrix [B * (H/patch) * (W/patch)][Cin * patch * patch]
extern "C" int ofb_patchify_hformat(const float* img, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t patch, void* P, void* stream) {
  if (!img || !P || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || patch <= 0 || H % patch || W % patch) return OFB_EINVAL;
  const int64_t R = (int64_t)B * (H / patch) * (W / patch), Cc = (int64_t)Cin * patch * patch;
  if (R > 0x7fffffff / 4 || Cc > 65536) return OFB_ELIMIT;
  hipStream_t s = (hipStream_t)stream;
  const int ncb = (int)((Cc + 15) / 16), rgs = (int)(((R + 15) / 16) * 4);
  if (hipMemsetAsync(P, 0, 16, s) != hipSuccess) return (int)hipGetLastError();
  hipLaunchKernelGGL(hstat_flat_kernel, dim3(2048), dim3(256), 0, s, img, (size_t)B * Cin * H * W, (ofb_hhdr*)P);
  hipLaunchKernelGGL(patchify_hformat_kernel, dim3((ncb * 16 + 255) / 256, rgs), dim3(256), 0, s, img, B, Cin, H, W, patch, (char*)P, ncb);
  return ofb_launch_status();
}

// jobs_dev: n_jobs descriptors in device memory; max_R / max_C: the largest R and C among them (grid extent); scratch: n_jobs * 64 floats
extern "C" int ofb_to_hformat_multi(const ofb_hformat_job* jobs_dev, int32_t n_jobs, int32_t max_R, int32_t max_C, float* scratch,
                                    void* stream) {
  if (!jobs_dev || !scratch || n_jobs <= 0 || n_jobs > 65535 || max_R <= 0 || max_C <= 0) return OFB_EINVAL;
  const int ncb = (max_C + 15) / 16, rgs = ((max_R + 15) / 16) * 4;
  hipLaunchKernelGGL(hstat_multi_kernel, dim3(HM_NB, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, scratch);
  hipLaunchKernelGGL(to_hformat_multi_kernel, dim3((ncb * 16 + 255) / 256, rgs, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, scratch);
  return ofb_launch_status();
}

namespace {
int to_hformat_colsum_launch(const float* X, int32_t R, int32_t C, int32_t ld, void* P, const float* rowscale, int32_t rs_div,
                             float* partial, const float* bound, int32_t n_bound, void* stream) {
  if (!X || !P || !partial || R <= 0 || C <= 0 || ld < C) return OFB_EINVAL;
  if (rowscale && rs_div <= 0) return OFB_EINVAL;
  if (bound && n_bound <= 0) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int ncb = (C + 15) / 16, rgs = ((R + 15) / 16) * 4, slabs = (rgs + CS_SLAB_RG - 1) / CS_SLAB_RG;
  if (int rc = h_prepare_header(X, R, C, ld, P, rowscale, rs_div, bound, s)) return rc;
  hipLaunchKernelGGL(to_hformat_colsum_kernel, dim3((ncb * 16 + 63) / 64, slabs), dim3(256), 0, s, X, R, C, ld, (char*)P, ncb, rgs, rowscale,
                     rs_div, partial, bound, n_bound);
  return ofb_launch_status();
}
}  // namespace

extern "C" int ofb_to_hformat_colsum(const float* X, int32_t R, int32_t C, int32_t ld, void* P, const float* rowscale, int32_t rs_div,
                                     float* partial, const float* bound, void* stream) {
  return to_hformat_colsum_launch(X, R, C, ld, P, rowscale, rs_div, partial, bound, 1, stream);
}

// bound: n_bound device floats whose maximum is >= max |x * rowscale| (the per-workgroup maxima of ofb_attention_bwd_wgmax)
extern "C" int ofb_to_hformat_colsum_nb(const float* X, int32_t R, int32_t C, int32_t ld, void* P, const float* rowscale, int32_t rs_div,
                                        float* partial, const float* bound, int32_t n_bound, void* stream) {
  if (!bound) return OFB_EINVAL;
  return to_hformat_colsum_launch(X, R, C, ld, P, rowscale, rs_div, partial, bound, n_bound, stream);
}

extern "C" int ofb_from_hformat(const void* P, int32_t R, int32_t C, float* X, int32_t ld, void* stream) {
  if (!X || !P || R <= 0 || C <= 0 || ld < C) return OFB_EINVAL;
  const int ncb = (C + 15) / 16, rgs = (R + 3) / 4;
  hipLaunchKernelGGL(from_hformat_kernel, dim3((C + 255) / 256, rgs), dim3(256), 0, (hipStream_t)stream, (const char*)P, ncb, X, R, C, ld);
  return ofb_launch_status();
}

extern "C" int32_t ofb_colsum_h_slabs(int32_t R) { return R > 0 ? (((R + 15) / 16) * 4 + CS_SLAB_RG - 1) / CS_SLAB_RG : 0; }

extern "C" int ofb_colsum_h(const void* P, int32_t R, int32_t C, float* partial, void* stream) {
  if (!P || !partial || R <= 0 || C <= 0) return OFB_EINVAL;
  const int ncb = (C + 15) / 16, rgs = ((R + 15) / 16) * 4, slabs = ofb_colsum_h_slabs(R);
  hipLaunchKernelGGL(colsum_h_kernel, dim3((ncb + 3) / 4, slabs), dim3(256), 0, (hipStream_t)stream, (const char*)P, ncb, rgs, partial, ncb * 16);
  return ofb_launch_status();
}

extern "C" int64_t ofb_gemm_h_workspace_bytes(const ofb_gemm_h_args* args) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
  const int tile = h_tile_choice(*args);
  if (tile == 96) { const Plan p = plan_h<C96>(*args); return p.R ? (int64_t)2 * p.W * C96::BM * C96::BN * (int64_t)sizeof(float) : 0; }
  if (tile == 112) return 0;                                  // (whole rounds only)
  const Plan p = plan_h<C128>(*args);
  return p.R ? (int64_t)2 * p.W * C128::BM * C128::BN * (int64_t)sizeof(float) : 0;
}

extern "C" int32_t ofb_gemm_h_colpart_rows(const ofb_gemm_h_args* args) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
  ofb_gemm_h_args g = *args;
  if (!g.colpart) g.colpart = reinterpret_cast<float*>(16);
  const Plan p = plan_h<C128F>(g);                            // (C128 and C128F share the tile geometry; the direct form is C128F's)
  if (p.stagger & 256) return 2 * p.mt;                       // direct epilogue: one partial row per wave row of a tile
  return p.R ? (p.mt - p.R / p.nt) + (p.R / p.nt) * (C128::BM / 4) : p.mt;
}

/* floats of an aux tensor in T-layout (OFB_ACT_GELU_GRAD_T / OFB_ACT_MULAUX_T) for an [M][N] output: whole 128 x 192 tiles */
extern "C" int64_t ofb_gemm_h_aux_t_floats(int32_t M, int32_t N) {
  if (M <= 0 || N <= 0) return 0;
  return (int64_t)((M + 127) / 128) * ((N + 191) / 192) * 128 * 192;
}

extern "C" int32_t ofb_gemm_h_rn_tiles(const ofb_gemm_h_args* args, int32_t* col_tiles) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
  ofb_gemm_h_args g = *args;
  // 0 = "do not ask for rn_out here": the product would take the 256 x 96 tile (a width that pads badly on 192 columns), which has no
  // room for the row-norm parts; forcing it onto 128 x 192 tiles costs more than the LayerNorm bound pass it would save
  g.rn_out = nullptr;
  if (h_tile_choice(g) == 96) return 0;
  g.rn_out = reinterpret_cast<float*>(16);                  // the geometry of the launch that WILL carry rn_out
  const int tile = h_tile_choice(g);
  const bool c96 = tile == 96;
  const int bm = c96 ? C96::BM : tile == 112 ? C112F::BMT : C128::BM, bn = c96 ? C96::BN : C128::BN;
  const int nt = ofb_cdiv(args->N, bn);
  if (col_tiles) *col_tiles = nt;
  return ofb_cdiv(args->M, bm) * nt;
}

extern "C" int ofb_gemm_h(const ofb_gemm_h_args* args, void* stream) {
  if (!args) return OFB_EINVAL;
  const ofb_gemm_h_args& g = *args;
  if (!g.A || !g.B || (!g.C && !g.Cp) || g.M <= 0 || g.N <= 0 || g.K <= 0) return OFB_EINVAL;
  if (g.a_kc == 0 && g.b_kc == 1) return OFB_ELIMIT;           // A^T * B^T is not on the path
  if (g.colpart && C128::BM != 128) return OFB_ELIMIT;
  if (g.rowscale && g.rs_div <= 0) return OFB_EINVAL;
  if (g.act < OFB_ACT_NONE || g.act > OFB_ACT_MULAUX_T) return OFB_EINVAL;
  if ((g.act == OFB_ACT_DGELU || act_is_gelug(g.act) || act_is_mulaux(g.act)) && !g.aux) return OFB_EINVAL;
  if (act_is_t(g.act) && !ofb_aligned16(g.aux)) return OFB_EINVAL;
  if (g.C && g.ldc < g.N) return OFB_EINVAL;
  if (g.Cp && g.c_ncb < (g.N + 15) / 16) return OFB_EINVAL;
  if (g.rn_out && (!g.rn_gamma || !g.rn_rowfac || !ofb_aligned16(g.rn_gamma))) return OFB_EINVAL;
  if ((g.Cp || g.cbound_out) && g.resid && !g.out_bound) return OFB_ELIMIT;   // no bound for a residual sum without scanning it: the caller supplies one
  if (g.a_ncb < ((g.a_kc ? g.K : g.M) + 15) / 16 || g.b_ncb < ((g.b_kc ? g.K : g.N) + 15) / 16) return OFB_EINVAL;
  if (!ofb_aligned16(g.A) || !ofb_aligned16(g.B) || (g.Cp && !ofb_aligned16(g.Cp))) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(0, s, 2.0 * g.M * g.N * (double)g.K);
  const int tile = h_tile_choice(g);
  const int mf = h_switch(OFB_TUNE_GEMM_MFMA, "OFB_GEMM_H_MFMA", 16);
#ifdef OFB_GEMM_H_LAB
  const int rc = tile == 1283 ? run_h<C128S>(g, s) : tile == 1281 ? run_h<C128K1>(g, s) : tile == 112 ? run_h<C112F>(g, s) : (tile == 96 ? run_h<C96>(g, s) : (mf == 16 ? run_h<C128F>(g, s) : run_h<C128>(g, s)));
#else
  const int rc = tile == 96 ? run_h<C96>(g, s) : tile == 112 ? run_h<C112F>(g, s) : (mf == 16 ? run_h<C128F>(g, s) : run_h<C128>(g, s));
#endif
  ofb_prof_post(0, s);
  return rc;
}

extern "C" int ofb_tune(int32_t key, int32_t value) {
  if (key < 0 || key >= OFB_TUNE_COUNT || value < 0) return OFB_EINVAL;
  bool ok = true;                                           // a value outside a key's set would silently select some other path
  switch (key) {
    case OFB_TUNE_GEMM_MFMA: ok = value == 16 || value == 32; break;
    case OFB_TUNE_GEMM_SCHED: ok = value <= 2; break;
    case OFB_TUNE_GEMM_TILE:
      ok = value == 0 || value == 96 || value == 97 || value == 128;
#ifdef OFB_GEMM_H_LAB
      ok = ok || value == 1281 || value == 1283;
#endif
      break;
    case OFB_TUNE_GEMM_T112: ok = value <= 1; break;
    case OFB_TUNE_GEMM_YIELD: ok = value <= 32; break;
    case OFB_TUNE_GEMM_DIRECT: ok = value <= 1; break;
    case OFB_TUNE_GEMM_CUS: ok = value <= 1024; break;
  }
  if (!ok) return OFB_EINVAL;
  h_tune[key] = value;
  return 0;
}

#ifdef OFB_H_STAMPS
extern "C" int ofb_diag_h_stamps(unsigned long long* out_host) {      /* lab only, not part of the ABI */
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(ofb_h_stamps), sizeof(unsigned long long) * 1024 * 8 * 4);
}
#endif
