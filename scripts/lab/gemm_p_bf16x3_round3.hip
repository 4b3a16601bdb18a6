// f32-accurate GEMM on operands that are ALREADY split into three bf16 planes ("P-format"), staged by LDS-DMA.
//
// Why: the exact 3-way split x = hi + mid + lo that lets the bf16 matrix pipe deliver fp32 accuracy (six MFMA terms per product,
// gemm.hip) costs ~9 VALU per value pair.  Done inside the K loop it is repeated for every tile that touches an operand
// (activations 3-12x, weights ~200x) and it is what bounds that loop (VALU issue + VGPR->LDS stores beside the MFMAs, on a
// power-limited chip).  Here every producer on the path (LayerNorm, GEMM epilogues, attention, the optimizer for weights) emits
// the planes ONCE, and the GEMM's K loop holds nothing but LDS-DMA, fragment reads and MFMAs.
//
// P-format of a matrix X[R][C]  (ofb_hip.h: ofb_pformat_bytes / ofb_to_pformat):
//   granules of 4 rows x 16 columns, 384 B each, stored [ceil(R/16)*4][ceil(C/16)]; inside a granule
//   [plane hi | mid | lo][c % 16][r % 4] bf16 (128 B per plane).  Rows >= R and columns >= C of the last granules hold zeros.
// One layout serves both consumers of an activation / weight:
//   mode KC (reduction along C, the MFMA row/column index is R): fragment = two ds_read_b64_tr_b16 (the hardware 4x16 transpose)
//   mode KR (reduction along R, the MFMA row/column index is C): fragment = two ds_read_b64 (4 consecutive r are contiguous)
// and a 32x32 accumulator block (lane = column, 4 consecutive rows per register group) stores P-format with 8-byte stores that
// fill whole 128-B lines, so a GEMM epilogue can feed the next GEMM directly.
//
// Kernel (product configuration C128): 128 x 192 tile, 4 waves (2 x 2), wave tile 64 x 96 = 2 x 3 blocks of
// v_mfma_f32_32x32x16_bf16, K step 16 (36 MFMAs per wave), TWO workgroups per CU, two 30-KB LDS stages filled by
// global_load_lds_dwordx4 two steps ahead (inline asm: through the builtin hipcc drains every LDS-DMA with vmcnt(0) before the next
// ds_read), one barrier per K step placed BETWEEN the two halves of the step with the fragment reads of the next half-step issued
// in the MFMA gaps of the current one; epilogue through LDS with compile-time forms.  Scheduling is the hybrid stream-K of
// gemm_plan.h (data-parallel rounds + K-split tail + deterministic fix-up).
// Measured alternatives (same-box A/B, scripts/lab/ab_gemm_p.sh): 256 x 192 / 8 waves / 3 stages (C192, lab builds) is not faster
// on any product of the step; v_mfma_f32_16x16x32_bf16 fragments carrying two planes each (scripts/lab/gemm_p_mfma16_paired.hip.txt)
// hold a higher clock but need 24 KB instead of 15 KB of fragment reads per wave and step: equal step time; s_setprio around the
// MFMA bursts and a half-unit start stagger of a CU's two workgroups: no gain.
#include "ofb_common.h"
#include "gemm_plan.h"
#include <cstdlib>
#include <type_traits>

namespace {

using ofb_plan::Plan; using ofb_plan::make_plan; using ofb_plan::tile_coord; using ofb_plan::Seg; using ofb_plan::get_seg;

typedef __bf16 pbf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef short ps16x4 __attribute__((ext_vector_type(4)));
typedef short ps16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x2 __attribute__((ext_vector_type(2)));
#define OFB_LDSP(p) ((__attribute__((address_space(3))) void*)(p))

// Tile configuration: WM x WN waves, wave tile (32 MI) x (32 NI), NST LDS stages, WGS workgroups per CU.  BN = 192 fits every
// width of the DeiT family (192 | 384, 768, 1152, 1536, 2304, 3072) without padded columns.
//   C192: 256 x 192, 8 waves, one workgroup per CU, 3 stages: the leanest main loop (fewest LDS-DMA bytes and barriers per MFMA);
//         its epilogue is serial with its main loop, so it serves the products whose outputs are small next to their K loop.
//   C128: 128 x 192, 4 waves, TWO workgroups per CU, 2 stages: the co-resident workgroup's MFMAs run under this one's epilogue
//         (GELU / P-format split VALU work, output stores) - for the output-heavy products (fc1, its input gradient, qkv).
template <int WM_, int WN_, int MI_, int NI_, int NST_, int WGS_>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, MI = MI_, NI = NI_, NST = NST_, WGS = WGS_, NW = WM * WN;
  static constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN, NT = 64 * NW;
  static constexpr int A_BYTES = BM * 96, B_BYTES = BN * 96, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024;          // 1-KB LDS-DMA pieces per stage
  static constexpr int QA = A_PIECES / NW, QB = (B_PIECES + NW - 1) / NW;             // pieces per wave (the last B piece only for some waves)
  static constexpr int HA = MI / 2;                                                   // row blocks per half step
  static constexpr int HR = (NST * STAGE >= BN * 132 * 4) ? 128 : 64;                 // rows of the tile parked in LDS per epilogue pass
  static constexpr int TROW = HR + 4, ITEMS = BN * (HR / 4) / NT;
  static_assert(A_PIECES % NW == 0 && MI % 2 == 0 && BM % HR == 0 && (32 * MI) <= HR && HR % (32 * MI) == 0, "piece / epilogue schedule");
  static_assert(BN * TROW * 4 <= NST * STAGE && QA >= 3 && QA <= 6 && QB <= 5 && NST >= 2 && NST <= 3, "LDS budget / schedule");
};
using C192 = Cfg<4, 2, 2, 3, 3, 1>;
using C128 = Cfg<2, 2, 2, 3, 2, 2>;
//   C96:  256 x 96, 4 waves stacked along M (the same 64 x 96 wave tile), two workgroups per CU, 2 stages: for output widths that pad
//         badly on 192 columns (N mod 192 in (0, 96]: the pruned / finetune widths 264, 480, 672, ...).  A tile with dead wave
//         columns costs nearly a whole tile's time (the K loop is paced by its stages and barriers), so 264 columns on 192-wide
//         tiles run at 69 % column efficiency, on 96-wide tiles at 92 %.
using C96 = Cfg<4, 1, 2, 3, 2, 2>;
constexpr int GRAN = 384;
constexpr int CS_SLAB_RG = 64;                      // row groups (256 rows) per column-sum slab

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {          // v_cvt_pk_bf16_f32: a -> low half, b -> high half (RNE)
  pf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pbf16x2));
}
// x = hi + mid + lo, each a bf16 (24 significant bits in total: exact for finite f32 in the normal range); two values at a time
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = pk_bf16(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  mid = pk_bf16(ra, rb);
  lo = pk_bf16(ra - __uint_as_float(mid << 16), rb - __uint_as_float(mid & 0xffff0000u));
}
// four consecutive rows of one column -> the 8-byte column slot of each plane slab
__device__ __forceinline__ void store_p4(char* slot, float v0, float v1, float v2, float v3) {
  unsigned h0, m0, l0, h1, m1, l1;
  split_pair(v0, v1, h0, m0, l0);
  split_pair(v2, v3, h1, m1, l1);
  *reinterpret_cast<uint2*>(slot) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(slot + 128) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(slot + 256) = make_uint2(l0, l1);
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// ---- f32 <-> P-format ---------------------------------------------------------------------------------------------
// X[R][C] row-major (ld) -> P; value = X * rowscale[r / rs_div] (optional); zero padding up to (Rp, Cp)
__global__ void to_pformat_kernel(const float* __restrict__ X, int R, int C, int ld, char* __restrict__ P, int ncb,
                                  const float* __restrict__ rowscale, int rs_div) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, rg = blockIdx.y;
  if (c >= ncb * 16) return;
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = 4 * rg + t;
    float x = (r < R && c < C) ? X[(size_t)r * ld + c] : 0.f;
    if (rowscale && r < R) x *= rowscale[rs_div == 1 ? r : r / rs_div];
    v[t] = x;
  }
  store_p4(P + ((size_t)rg * ncb + (c >> 4)) * GRAN + (c & 15) * 8, v[0], v[1], v[2], v[3]);
}
// Patch matrix of a conv-as-GEMM straight from the images (models/layers.py:177: Conv2d(k = s = patch) == patchify + Linear): row
// (b, py, px) x column (c, i, j) = img[b][c][py * patch + i][px * patch + j], written as planes without the [B*L][C*patch^2] f32 copy in between
__global__ void patchify_pformat_kernel(const float* __restrict__ img, int B, int Cin, int Hh, int Ww, int patch, char* __restrict__ P,
                                        int ncb) {
  const int gw = Ww / patch, L = (Hh / patch) * gw, R = B * L, Cc = Cin * patch * patch;
  const int c = blockIdx.x * blockDim.x + threadIdx.x, rg = blockIdx.y;
  if (c >= ncb * 16) return;
  const int ch = c / (patch * patch), rem = c - ch * patch * patch, i = rem / patch, j = rem - i * patch;
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = 4 * rg + t;
    float x = 0.f;
    if (r < R && c < Cc) {
      const int b = r / L, l = r - b * L, py = l / gw, px = l - py * gw;
      x = img[(((size_t)b * Cin + ch) * Hh + py * patch + i) * Ww + px * patch + j];
    }
    v[t] = x;
  }
  store_p4(P + ((size_t)rg * ncb + (c >> 4)) * GRAN + (c & 15) * 8, v[0], v[1], v[2], v[3]);
}
// Many matrices in ONE launch (the weights of the model, once per optimizer step): blockIdx.z picks the matrix, the grid covers the
// largest one and the blocks beyond a matrix's own extent leave at once.
__global__ void to_pformat_multi_kernel(const ofb_pformat_job* __restrict__ jobs) {
  const ofb_pformat_job j = jobs[blockIdx.z];
  const int c = blockIdx.x * blockDim.x + threadIdx.x, rg = blockIdx.y, ncb = (j.C + 15) >> 4;
  if (c >= ncb * 16 || rg >= ((j.R + 15) >> 4) * 4) return;
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = 4 * rg + t;
    float x = (r < j.R && c < j.C) ? j.X[(size_t)r * j.ld + c] : 0.f;
    if (j.rowscale && r < j.R) x *= j.rowscale[r];
    v[t] = x;
  }
  store_p4((char*)j.P + ((size_t)rg * ncb + (c >> 4)) * GRAN + (c & 15) * 8, v[0], v[1], v[2], v[3]);
}
// The same conversion for a gradient whose column sums are wanted as well (bias gradients: db = colsum(dY)): one pass over dY
// writes the planes AND partial[slab][c] = sum of the slab's (scaled) rows, added in row order; the slabs are summed by ofb_colsum.
// grid (ceil(C16 / 64), slabs of 256 rows), 256 threads: wave w converts the row groups w, w + 4, ... of the slab, lane = column.
__global__ __launch_bounds__(256) void to_pformat_colsum_kernel(const float* __restrict__ X, int R, int C, int ld, char* __restrict__ P,
                                                                int ncb, int rgs, const float* __restrict__ rowscale, int rs_div,
                                                                float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = blockIdx.x * 64 + lane, slab = blockIdx.y;
  float s = 0.f;
  if (c < ncb * 16) {
    const int rg1 = min(rgs, (slab + 1) * CS_SLAB_RG);
    for (int rg = slab * CS_SLAB_RG + w; rg < rg1; rg += 4) {
      float v[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = 4 * rg + t;
        float x = (r < R && c < C) ? X[(size_t)r * ld + c] : 0.f;
        if (rowscale && r < R) x *= rowscale[rs_div == 1 ? r : r / rs_div];
        v[t] = x;
      }
      s += (v[0] + v[1]) + (v[2] + v[3]);
      store_p4(P + ((size_t)rg * ncb + (c >> 4)) * GRAN + (c & 15) * 8, v[0], v[1], v[2], v[3]);
    }
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < ncb * 16) partial[(size_t)slab * (ncb * 16) + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// P -> X[R][C] (hi + mid + lo is exact)
__global__ void from_pformat_kernel(const char* __restrict__ P, int ncb, float* __restrict__ X, int R, int C, int ld) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, rg = blockIdx.y;
  if (c >= C) return;
  const char* slot = P + ((size_t)rg * ncb + (c >> 4)) * GRAN + (c & 15) * 8;
  const uint2 h = *reinterpret_cast<const uint2*>(slot), m = *reinterpret_cast<const uint2*>(slot + 128),
              l = *reinterpret_cast<const uint2*>(slot + 256);
  const float v[4] = {bf16_lo(h.x) + (bf16_lo(m.x) + bf16_lo(l.x)), bf16_hi(h.x) + (bf16_hi(m.x) + bf16_hi(l.x)),
                      bf16_lo(h.y) + (bf16_lo(m.y) + bf16_lo(l.y)), bf16_hi(h.y) + (bf16_hi(m.y) + bf16_hi(l.y))};
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (4 * rg + t < R) X[(size_t)(4 * rg + t) * ld + c] = v[t];
}

// out partial[slab][c] = sum over the slab's rows of X[r][c] (hi + mid + lo, rows in order): column sums of a P-format matrix (bias
// gradients of tensors that exist only in P-format).  grid (ncb / 4, slabs), 256 threads = 4 waves x 64 columns; wave w takes the row
// groups w, w + 4, ... of the slab, the four waves' sums are added in wave order (deterministic).
__global__ __launch_bounds__(256) void colsum_p_kernel(const char* __restrict__ P, int ncb, int rgs, float* __restrict__ partial, int ld) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, cb = blockIdx.x * 4 + (lane >> 4), slab = blockIdx.y;
  float s = 0.f;
  if (cb < ncb) {
    const int rg1 = min(rgs, (slab + 1) * CS_SLAB_RG);
    for (int rg = slab * CS_SLAB_RG + w; rg < rg1; rg += 4) {
      const char* slot = P + ((size_t)rg * ncb + cb) * GRAN + (lane & 15) * 8;
      const uint2 h = *reinterpret_cast<const uint2*>(slot), m = *reinterpret_cast<const uint2*>(slot + 128),
                  l = *reinterpret_cast<const uint2*>(slot + 256);
      s += bf16_lo(h.x) + (bf16_lo(m.x) + bf16_lo(l.x));
      s += bf16_hi(h.x) + (bf16_hi(m.x) + bf16_hi(l.x));
      s += bf16_lo(h.y) + (bf16_lo(m.y) + bf16_lo(l.y));
      s += bf16_hi(h.y) + (bf16_hi(m.y) + bf16_hi(l.y));
    }
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && cb < ncb) partial[(size_t)slab * ld + blockIdx.x * 64 + lane] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ---- the GEMM -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz(int tg) { return ((tg >> 1) & 3) << 1; }

// v = alpha*acc (+bias)(*colscale); act; (*rowscale); (+resid)   -- shared by the fused epilogue and the fix-up kernel
__device__ __forceinline__ float epi_value(const ofb_gemm_p_args& g, float accv, int row, int col, float bias, float cs, bool ok) {
  float v = (accv * g.alpha + bias) * cs;
  if (g.act == OFB_ACT_GELU) {
    if (g.aux && ok) g.aux[(size_t)row * g.ldaux + col] = v;
    v = ofb_gelu(v);
  } else if (g.act == OFB_ACT_GELU_GRAD) {
    float Phi, phi;
    ofb_gelu_parts(v, Phi, phi);
    if (ok) g.aux[(size_t)row * g.ldaux + col] = Phi + v * phi;
    v *= Phi;
  } else if (g.act == OFB_ACT_DGELU) {
    v *= ofb_dgelu(ok ? g.aux[(size_t)row * g.ldaux + col] : 0.f);
  } else if (g.act == OFB_ACT_MULAUX) {
    v *= ok ? g.aux[(size_t)row * g.ldaux + col] : 0.f;
  }
  if (g.rowscale) v *= ok ? g.rowscale[g.rs_div == 1 ? row : row / g.rs_div] : 1.f;
  if (g.resid) v += ok ? g.resid[(size_t)row * g.ldr + col] : 0.f;
  return v;
}

#ifdef OFB_P_STAMPS
// lab only (scripts/lab/stamp_gemm_p.py): s_memtime stamps of wave 0 of every workgroup: [wg][unit][4] = unit start, K loop start,
// K loop end, epilogue end
__device__ unsigned long long ofb_p_stamps[1024 * 8 * 4];
#define OFB_PSTAMP(slot) do { if (t == 0 && sidx < 8 && blockIdx.x < 1024) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ofb_p_stamps[(blockIdx.x * 8 + sidx) * 4 + slot] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define OFB_PSTAMP(slot) do { } while (0)
#endif
#define OFB_VMW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void vm_wait(int n) {           // n is wave-uniform
  switch (n) {
    OFB_VMW(0) OFB_VMW(5) OFB_VMW(6) OFB_VMW(7) OFB_VMW(8) OFB_VMW(9) OFB_VMW(10) OFB_VMW(12) OFB_VMW(14) OFB_VMW(16) OFB_VMW(18)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// Epilogue forms are compile-time (EPI = set of E_* bits): with run-time flags hipcc has to assume that a side-input load may
// follow an aliasing store and puts s_waitcnt vmcnt(0) between the stores of every element (measured: a 128 x 192 tile took longer
// to store than to compute).  Each instantiation is straight-line: all loads of a pass, then arithmetic, then stores.
enum : int { E_C = 1, E_P = 2, E_GELU = 4, E_DGELU = 8, E_RS = 16, E_RES = 32, E_ANY = 64, E_GELUG = 128, E_MULAUX = 256 };

template <class CF, bool A_KC, bool B_KC, bool TAIL, int EPI>
__global__ __launch_bounds__(CF::NT, 2) void gemm_p_kernel(const ofb_gemm_p_args g, const Plan p) {
  constexpr int BM = CF::BM, BN = CF::BN, WN = CF::WN, MI = CF::MI, NI = CF::NI, HA = CF::HA, NST = CF::NST, NW = CF::NW;
  constexpr int STAGE = CF::STAGE, A_BYTES = CF::A_BYTES, QA = CF::QA, QB = CF::QB, HR = CF::HR, TROW = CF::TROW;
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE];
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, h = lane >> 5;
  const int wm0 = (w / WN) * (32 * MI), wn0 = (w % WN) * (32 * NI);
  const int a_ncb = g.a_ncb, b_ncb = g.b_ncb;
  const bool blast = w + NW * (QB - 1) < CF::B_PIECES;   // this wave moves a QB-th B piece per stage (wave-uniform)
  const int n_w = QA + QB - (blast ? 0 : 1);             // LDS-DMA instructions of this wave per stage

  // LDS-DMA source offsets (bytes, relative to the tile / stage base) of this wave's 1-KB pieces (ids w, w + NW, ...)
  //   KC: piece = (block j of 32 rows, plane): lane -> granule tg = l>>3 of the block, 16-B chunk cp = (l&7) ^ swz(tg); the XOR on
  //       the SOURCE chunk (LDS stays lane-linear) makes the transposed fragment reads bank-conflict free
  //   KR: the stage image is a linear copy of [4 row groups][B? / 16 granules], cut into 1-KB pieces
  unsigned a_off[QA], b_off[QB];
  {
    const int tg = lane >> 3, cp = (lane & 7) ^ swz(tg);
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int piece = w + NW * q, j = piece / 3, pl = piece % 3, bl = piece * 1024 + lane * 16;
      a_off[q] = A_KC ? (unsigned)(((8 * j + tg) * a_ncb) * GRAN + pl * 128 + cp * 16)
                      : (unsigned)(((bl / (BM / 16 * GRAN)) * a_ncb) * GRAN + bl % (BM / 16 * GRAN));
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int piece = w + NW * q, j = piece / 3, pl = piece % 3, bl = piece * 1024 + lane * 16;
      b_off[q] = B_KC ? (unsigned)(((8 * j + tg) * b_ncb) * GRAN + pl * 128 + cp * 16)
                      : (unsigned)(((bl / (BN / 16 * GRAN)) * b_ncb) * GRAN + bl % (BN / 16 * GRAN));
    }
  }
  // fragment read offsets (bytes inside an operand's stage image), two 8-byte reads per fragment
  int a_r0, a_r1, b_r0, b_r1;
  {
    const int gq = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, hh = gq >> 1, tg = 4 * (gq & 1) + pp;
    const int c0 = 8 * hh + q, c1 = c0 + 4;
    const int kc0 = (8 * tg + ((c0 >> 1) ^ swz(tg))) * 16 + (c0 & 1) * 8, kc1 = (8 * tg + ((c1 >> 1) ^ swz(tg))) * 16 + (c1 & 1) * 8;
    const int kr_a0 = ((2 * h) * (BM / 16) + (l31 >> 4)) * GRAN + (l31 & 15) * 8, kr_a1 = kr_a0 + (BM / 16) * GRAN;
    const int kr_b0 = ((2 * h) * (BN / 16) + (l31 >> 4)) * GRAN + (l31 & 15) * 8, kr_b1 = kr_b0 + (BN / 16) * GRAN;
    a_r0 = A_KC ? kc0 : kr_a0; a_r1 = A_KC ? kc1 : kr_a1;
    b_r0 = B_KC ? kc0 : kr_b0; b_r1 = B_KC ? kc1 : kr_b1;
  }

  f32x16 acc[MI][NI];                                // zeroed at the start of every unit: nothing of it lives across an epilogue

  // LDS-DMA in inline asm (through the builtin hipcc drains every LDS-DMA with vmcnt(0) before the next ds_read): invisible to
  // its wait-count bookkeeping, counted by hand (n_w per wave and stage).  The pieces of one wave sit NW KB apart in the stage
  // image, A's first, then B's (A_BYTES = QA * NW KB).  M0 is written in the statement that uses it and restored afterwards.
  const unsigned lds0 = (unsigned)(size_t)OFB_LDSP(lds) + (unsigned)w * 1024u;
  auto dma = [&](unsigned ldsaddr, unsigned voff, const char* base) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldsaddr), "v"(voff), "s"(base) : "memory");
  };
  auto issue = [&](int buf, const char* a_src, const char* b_src) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep) :: "memory");
    const unsigned l0 = lds0 + (unsigned)buf * (unsigned)STAGE;
#pragma unroll
    for (int q = 0; q < QA; ++q) dma(l0 + q * (NW * 1024), a_off[q], a_src);
#pragma unroll
    for (int q = 0; q < QB - 1; ++q) dma(l0 + (QA + q) * (NW * 1024), b_off[q], b_src);
    if (blast) dma(l0 + (QA + QB - 1) * (NW * 1024), b_off[QB - 1], b_src);
    asm volatile("s_mov_b32 m0, %0" :: "s"(keep) : "memory");
  };
  auto frag = [&](const char* base, int r0, int r1, bool kc, int blk, int plane) __attribute__((always_inline)) -> pbf16x8 {
    ps16x4 lo4, hi4;
    if (kc) {
      const char* q = base + (blk * 3 + plane) * 1024;
      lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4*)(q + r0));
      hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4*)(q + r1));
    } else {
      const char* q = base + (2 * blk) * GRAN + plane * 128;
      lo4 = *reinterpret_cast<const ps16x4*>(q + r0);
      hi4 = *reinterpret_cast<const ps16x4*>(q + r1);
    }
    ps16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(pbf16x8, v);
  };
  pbf16x8 alo[HA][3], ahi[HA][3], bb[2][NI][3];         // A fragments of the two half steps, B fragments of this / the next K step
  auto rdA = [&](pbf16x8 (&dst)[HA][3], int buf, int blk0) __attribute__((always_inline)) {
    const char* la = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < HA; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[i][pl] = frag(la, a_r0, a_r1, A_KC, (wm0 >> 5) + blk0 + i, pl);
  };
  auto rdB = [&](pbf16x8 (&dst)[NI][3], int buf) __attribute__((always_inline)) {
    const char* lb = lds + buf * STAGE + A_BYTES;
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[j][pl] = frag(lb, b_r0, b_r1, B_KC, (wn0 >> 5) + j, pl);
  };
  // six product terms, smallest first: (mid,mid) (hi,lo) (lo,hi) (hi,mid) (mid,hi) (hi,hi)
  constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
  constexpr int NMF = 6 * HA * NI, RD2 = 6 * HA + 6 * NI;       // MFMAs per half step; fragment reads riding in the second half
#define OFB_MMA_HALF(AF, BF, BLK0)                                                                                                   \
  _Pragma("unroll") for (int q = 0; q < 6; ++q) _Pragma("unroll") for (int i = 0; i < HA; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) \
      acc[BLK0 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[i][TA[q]], BF[j][TB[q]], acc[BLK0 + i][j], 0, 0, 0);
#define OFB_INTERLEAVE(NM, ND)                                                               \
  _Pragma("unroll") for (int z_ = 0; z_ < (NM); ++z_) {                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
    __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);                                      \
  }

  const int v = ofb_xcd_remap(blockIdx.x, p.W);     // consecutive v share an XCD (and thus operand panels in its L2)
  int sidx = 0;
  Seg cur = get_seg<TAIL>(p, v, 0);
  if (!cur.ok) return;
  if (CF::WGS > 1 && p.stagger > 0) {
    // de-phase the two workgroups of a CU: in lock step both sit in their epilogues together (vector pipe and stores busy, matrix
    // pipe idle) and then in their K loops together; out of phase one's epilogue runs under the other's MFMAs.  The workgroup in
    // the CU's odd threadgroup slot (HW_ID.TG_ID) starts about half a unit late.
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
#ifdef OFB_P_STAMPS
    if (t == 0 && blockIdx.x < 1024) ofb_p_stamps[(blockIdx.x * 8 + 7) * 4 + 3] = hw;
#endif
    if ((hw >> 16) & 1)
      for (int z = 0; z < p.stagger; ++z) __builtin_amdgcn_s_sleep(127);
  }
  while (true) {
    const int nk = cur.it1 - cur.it0;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const size_t a_step = A_KC ? GRAN : (size_t)4 * a_ncb * GRAN, b_step = B_KC ? GRAN : (size_t)4 * b_ncb * GRAN;
    const char* a_base = (const char*)g.A + (A_KC ? (size_t)(cur.m0 / 4) * a_ncb * GRAN : (size_t)(cur.m0 / 16) * GRAN) + cur.it0 * a_step;
    const char* b_base = (const char*)g.B + (B_KC ? (size_t)(cur.n0 / 4) * b_ncb * GRAN : (size_t)(cur.n0 / 16) * GRAN) + cur.it0 * b_step;
    OFB_PSTAMP(0);
    __builtin_amdgcn_s_barrier();                        // every wave is past the previous unit's LDS traffic
    issue(0, a_base, b_base);
    if (nk > 1) issue(1, a_base + a_step, b_base + b_step);
    if (NST > 2 && nk > 2) issue(2, a_base + 2 * a_step, b_base + 2 * b_step);
    vm_wait(((nk < NST ? nk : NST) - 1) * n_w);          // stage 0 landed; the other prologue stages may be in flight
    __builtin_amdgcn_s_barrier();
    OFB_PSTAMP(1);
    rdA(alo, 0, 0);
    rdB(bb[0], 0);
    auto step = [&](int i, int buf, auto PAR) __attribute__((always_inline)) {
      constexpr int par = decltype(PAR)::value;
      const int nbuf = buf + 1 == NST ? 0 : buf + 1;
      // first half: lower row blocks x B(i); the reads of the upper row blocks ride in the MFMA gaps
      __builtin_amdgcn_sched_barrier(0);
      rdA(ahi, buf, HA);
      OFB_MMA_HALF(alo, bb[par], 0)
      OFB_INTERLEAVE(NMF, 1)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // this wave is done reading buf(i)
      {                                                                      // own pieces of stage i+1 landed; later stages may fly
        int young = nk - i - 2;
        young = young < 0 ? 0 : (young > NST - 2 ? NST - 2 : young);
        vm_wait(young * n_w);
      }
#ifndef OFB_P_ABL_NOBAR
      __builtin_amdgcn_s_barrier();
#endif
#ifndef OFB_P_ABL_NODMA
      if (i + NST < nk) issue(buf, a_base + (size_t)(i + NST) * a_step, b_base + (size_t)(i + NST) * b_step);
#endif
      // second half: upper row blocks x B(i); the reads of step i+1 (lower row blocks and B) ride in the gaps (after the last
      // step they fetch a stale buffer that nothing consumes)
      __builtin_amdgcn_sched_barrier(0);
      rdA(alo, nbuf, 0);
      rdB(bb[par ^ 1], nbuf);
      OFB_MMA_HALF(ahi, bb[par], HA)
      OFB_INTERLEAVE(RD2 > NMF ? RD2 - NMF : 0, 2)
      OFB_INTERLEAVE(RD2 > NMF ? 2 * NMF - RD2 : NMF, 1)
      __builtin_amdgcn_sched_barrier(0);
    };
    // A wave whose whole 64 x 96 part of the tile lies outside the matrix (ragged shapes: N = 264 in 192-column tiles, ...) takes
    // part in the staging and the barriers only: no fragment reads, no MFMAs (its accumulators stay zero and are never stored); the
    // matrix pipe and the power budget go to the other waves and the co-resident workgroup
    if (cur.m0 + wm0 < g.M && cur.n0 + wn0 < g.N) {
      int buf = 0, i = 0;
      for (; i + 1 < nk; i += 2) {
        step(i, buf, std::integral_constant<int, 0>{});
        buf = buf + 1 == NST ? 0 : buf + 1;
        step(i + 1, buf, std::integral_constant<int, 1>{});
        buf = buf + 1 == NST ? 0 : buf + 1;
      }
      if (i < nk) step(i, buf, std::integral_constant<int, 0>{});
    } else {
      int buf = 0;
      for (int i = 0; i < nk; ++i) {
        int young = nk - i - 2;
        young = young < 0 ? 0 : (young > NST - 2 ? NST - 2 : young);
        vm_wait(young * n_w);
        __builtin_amdgcn_s_barrier();
        if (i + NST < nk) issue(buf, a_base + (size_t)(i + NST) * a_step, b_base + (size_t)(i + NST) * b_step);
        buf = buf + 1 == NST ? 0 : buf + 1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // the trailing (unused) fragment reads
    OFB_PSTAMP(2);

    {
      // Epilogue through LDS (the stage buffers are free now): the accumulators of HR rows of the tile are parked as
      // T[col][HR rows (+4 pad)] f32 - a lane's four consecutive rows of a column are one ds_write_b128 - and all waves finish
      // those rows together, one (4-row group, column) item per thread and trip: f32 loads / stores are coalesced along the
      // columns, the P-format store is the item's three 8-byte plane slots.  The body exists once (a direct epilogue over all
      // accumulator values of a lane does not unroll and spills); the side inputs of EPB items are requested before any of them
      // is finished, so the memory latency is paid once per batch.
      // Item map: a thread keeps its lane's columns (lane + 64 c, c < BN/64) and walks the 4-row groups rg = w, w + NW, ..., so
      // the per-column inputs (bias, gate) are loaded once per tile; the side inputs of ALL items of a pass (residual or saved
      // pre-activation, DropPath row scales) are requested before any item is finished: one exposed memory latency per pass.
      // PART: the tile width is not a multiple of 64 (C96): the last column chunk is half empty, its lanes >= BN sit out
      constexpr int NC = (BN + 63) / 64, NRG = (HR / 4) / NW, ITEMS = NC * NRG;
      constexpr bool PART = (BN % 64) != 0;
      static_assert((HR / 4) % NW == 0, "epilogue item map");
      const bool lane_in_last = !PART || lane + 64 * (NC - 1) < BN;
      constexpr bool ANY = (EPI & E_ANY) != 0;
      // which parts exist: known at compile time for the specialised forms, asked at run time by the generic form (E_ANY)
      const bool has_c = ANY ? g.C != nullptr : (EPI & E_C) != 0, has_p = ANY ? g.Cp != nullptr : (EPI & E_P) != 0;
      const bool gelu = ANY ? g.act == OFB_ACT_GELU : (EPI & E_GELU) != 0, dg = ANY ? g.act == OFB_ACT_DGELU : (EPI & E_DGELU) != 0;
      const bool gelug = ANY ? g.act == OFB_ACT_GELU_GRAD : (EPI & E_GELUG) != 0, mula = ANY ? g.act == OFB_ACT_MULAUX : (EPI & E_MULAUX) != 0;
      const bool has_rs = ANY ? g.rowscale != nullptr : (EPI & E_RS) != 0, has_res = ANY ? g.resid != nullptr : (EPI & E_RES) != 0;
      float* T = reinterpret_cast<float*>(lds);
      float* __restrict__ Cout = g.C;
      float* __restrict__ auxw = g.aux;
      const float* __restrict__ auxr = g.aux;
      const float* __restrict__ resid = g.resid;
      const float* __restrict__ rowscale = g.rowscale;
      const int rp_out = (g.M + 15) & ~15;
      float biasv[NC], csv[NC], cacc[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        cacc[c] = 0.f;
        const int col = cur.n0 + lane + 64 * c, colc = col < g.N ? col : g.N - 1;
        biasv[c] = (!TAIL && g.bias) ? g.bias[colc] : 0.f;
        csv[c] = (!TAIL && g.colscale) ? g.colscale[colc] : 1.f;
      }
      __builtin_amdgcn_s_barrier();                                 // every wave has finished its fragment reads
#pragma unroll
      for (int half = 0; half < BM / HR; ++half) {
        if (wm0 / HR == half) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
              for (int gq = 0; gq < 4; ++gq) {
                f32x4 q4 = {acc[mi][ni][4 * gq], acc[mi][ni][4 * gq + 1], acc[mi][ni][4 * gq + 2], acc[mi][ni][4 * gq + 3]};
                *reinterpret_cast<f32x4*>(T + (wn0 + 32 * ni + l31) * TROW + (wm0 % HR) + 32 * mi + 8 * gq + 4 * h) = q4;
              }
        }
        __syncthreads();
        if (TAIL) {                                                   // raw partial tile -> workspace[slot][BM][BN]
#pragma unroll
          for (int k = 0; k < NRG; ++k) {
            const int rgl = w + NW * k;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const int lcol = lane + 64 * c;
              if (PART && c == NC - 1 && !lane_in_last) continue;
              const f32x4 q4 = *reinterpret_cast<const f32x4*>(T + lcol * TROW + 4 * rgl);
              float* __restrict__ ws = g.workspace + (size_t)cur.slot * (BM * BN) + (size_t)(HR * half + 4 * rgl) * BN + lcol;
#pragma unroll
              for (int tt = 0; tt < 4; ++tt) ws[tt * BN] = q4[tt];
            }
          }
        } else {
          // all side inputs of the pass first, one exposed latency.  Interior tiles (every row and column in range: all but the last
          // row / column of tiles) take the unguarded body: with per-element guards hipcc puts each store in its own block behind
          // s_waitcnt vmcnt(0), i.e. one memory round trip per store.
          auto pass = [&](auto GUARDED) __attribute__((always_inline)) {
            constexpr bool GD = decltype(GUARDED)::value;
            // two batches of row groups per pass: all side inputs of a batch are requested before any item of it is finished
            // (one exposed latency per batch); a whole pass in one batch needs > 100 registers for side inputs and spills
            constexpr int KB = NRG > 4 ? NRG / 4 : NRG / 2, BI = KB * NC;
            static_assert(NRG % 2 == 0 && NRG % KB == 0, "epilogue batches");
#pragma unroll
            for (int k0 = 0; k0 < NRG; k0 += KB) {
              f32x4 side[BI], side2[BI], rsv[KB];
#pragma unroll
              for (int kk = 0; kk < KB; ++kk) {
                const int row0 = cur.m0 + HR * half + 4 * (w + NW * (k0 + kk));
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                  const int row = row0 + tt, rowc = (!GD || row < g.M) ? row : g.M - 1;
                  rsv[kk][tt] = has_rs ? rowscale[g.rs_div == 1 ? rowc : rowc / g.rs_div] : 1.f;
#pragma unroll
                  for (int c = 0; c < NC; ++c) {
                    // (a lane outside a PART tile's last chunk reads its neighbour tile's columns, clamped into the matrix: unused)
                    const int col = cur.n0 + lane + 64 * c, colc = ((!GD && !PART) || col < g.N) ? col : g.N - 1;
                    side[kk * NC + c][tt] = (dg || mula) ? auxr[(size_t)rowc * g.ldaux + colc] : 0.f;
                    side2[kk * NC + c][tt] = has_res ? resid[(size_t)rowc * g.ldr + colc] : 0.f;
                  }
                }
              }
#pragma unroll
              for (int kk = 0; kk < KB; ++kk) {
                const int rgl = w + NW * (k0 + kk), row0 = cur.m0 + HR * half + 4 * rgl;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                  const int lcol = lane + 64 * c, col = cur.n0 + lcol;
                  if (PART && c == NC - 1 && !lane_in_last) continue;
                  const f32x4 q4 = *reinterpret_cast<const f32x4*>(T + lcol * TROW + 4 * rgl);
                  const bool colok = !GD || col < g.N;
                  float pv[4];
#pragma unroll
                  for (int tt = 0; tt < 4; ++tt) {
                    const int row = row0 + tt;
                    const bool ok = !GD || (colok && row < g.M);
                    float val = (q4[tt] * g.alpha + biasv[c]) * csv[c];
                    if (gelu) {
                      if (auxw && ok) auxw[(size_t)row * g.ldaux + col] = val;
                      val = ofb_gelu(val);
                    } else if (gelug) {
                      float Phi, phi;
                      ofb_gelu_parts(val, Phi, phi);
                      if (ok) auxw[(size_t)row * g.ldaux + col] = Phi + val * phi;
                      val *= Phi;
                    } else if (dg) {
                      val *= ofb_dgelu(side[kk * NC + c][tt]);
                    } else if (mula) {
                      val *= side[kk * NC + c][tt];
                    }
                    val = val * rsv[kk][tt] + side2[kk * NC + c][tt];
                    if (has_c && ok) Cout[(size_t)row * g.ldc + col] = val;
                    pv[tt] = ok ? val : 0.f;
                  }
                  cacc[c] += (pv[0] + pv[1]) + (pv[2] + pv[3]);
                  if (has_p && (!GD || (row0 < rp_out && col < g.c_ncb * 16)))
                    store_p4((char*)g.Cp + ((size_t)(row0 >> 2) * g.c_ncb + (col >> 4)) * GRAN + (col & 15) * 8, pv[0], pv[1], pv[2], pv[3]);
                }
              }
            }
          };
          if (cur.m0 + BM <= g.M && cur.n0 + BN <= g.N) pass(std::false_type{});
          else pass(std::true_type{});
        }
        if (half + 1 < BM / HR) __syncthreads();                    // T is rewritten by the next pass (the next unit starts with a barrier)
      }
      if (!TAIL && g.colpart) {
        // column sums of this tile's outputs: per-thread sums over its row groups (fixed order), then the waves in order
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c) T[w * BN + lane + 64 * c] = cacc[c];
        __syncthreads();
        if (t < BN && cur.n0 + t < g.N) {
          float sum = 0.f;
#pragma unroll
          for (int ww = 0; ww < NW; ++ww) sum += T[ww * BN + t];
          g.colpart[(size_t)(cur.m0 / BM) * g.N + cur.n0 + t] = sum;
        }
      }
    }
    OFB_PSTAMP(3);
    const Seg nxt = get_seg<TAIL>(p, v, ++sidx);
    if (!nxt.ok) break;
    cur = nxt;
  }
}

// Sums the partial tiles of each streamed tail tile in a fixed contributor order and applies the epilogue.
// grid (R, BM / 4), BN threads: block (r, rg) handles rows [4*rg, 4*rg+4) of tail tile r, one column per thread.
template <class CF>
__global__ __launch_bounds__(CF::BN) void gemm_p_fixup_kernel(const ofb_gemm_p_args g, const Plan p) {
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
  const int row0 = m0 + 4 * rgl, col = n0 + t;
  const bool colok = col < g.N;
  const float bias = (g.bias && colok) ? g.bias[col] : 0.f, cs = (g.colscale && colok) ? g.colscale[col] : 1.f;
  float pv[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int row = row0 + tt;
    const bool ok = colok && row < g.M;
    const float val = epi_value(g, sum[tt], row, col, bias, cs, ok);
    if (g.C && ok) g.C[(size_t)row * g.ldc + col] = val;
    pv[tt] = ok ? val : 0.f;
  }
  if (g.Cp && row0 < ((g.M + 15) & ~15) && col < g.c_ncb * 16)
    store_p4((char*)g.Cp + ((size_t)(row0 >> 2) * g.c_ncb + (col >> 4)) * GRAN + (col & 15) * 8, pv[0], pv[1], pv[2], pv[3]);
  // column sums of the output: the tail is the last R / nt whole tile rows (plan_p); their sums go, one row per 4-row slice, behind
  // the rows that the data-parallel tiles write (one per tile row): colpart[(mt - R/nt) + (r / nt) * (BM/4) + slice][N]
  if (g.colpart && colok)
    g.colpart[((size_t)(p.mt - p.R / p.nt) + (size_t)(r / p.nt) * (BM / 4) + rgl) * g.N + col] = (pv[0] + pv[1]) + (pv[2] + pv[3]);
}

int p_cu_count() {
  static int n = 0;
  if (n == 0) {
    hipDeviceProp_t prop;
    int devid = 0;
    n = 256;
    if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess && prop.multiProcessorCount > 0)
      n = prop.multiProcessorCount;
  }
  return n;
}

// Tile configuration: C128 (two workgroups per CU) measured equal or faster than C192 on every product of the step (same-box
// scripts/gemm_step_shapes_p.py); C192 is compiled only into lab builds (-DOFB_GEMM_P_LAB, OFB_GEMM_P_TILE=192).
int p_tile_choice(const ofb_gemm_p_args& g) {
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("OFB_GEMM_P_TILE"); forced = e ? atoi(e) : 0; }
#ifdef OFB_GEMM_P_LAB
  if (forced == 192) return 192;
#endif
  if (forced == 128) return 128;
  // 96-wide tiles (C96) where they compute fewer padded columns than 192-wide ones: token-row products only (a_kc: the tall
  // operand is A; the weight-gradient form has the ragged extent on its 128-row axis) and no per-tile column sums (laid out per
  // 128-row tile row)
  if (g.a_kc && !g.colpart && g.M >= 4 * C96::BM) {
    const int c192 = ofb_cdiv(g.N, 192) * 192, c96 = ofb_cdiv(g.N, 96) * 96;
    if (forced == 96) return 96;
    if (c96 < c192) {
      // measured on the configs[4] shapes (scripts/lab/time_ragged_gemm.py, M = 50432): fewer padded columns win 7-33 % (N = 264 at
      // K >= 576, N = 480 / 672 at K = 264) EXCEPT when the 96-wide tiling ends in a small streamed tail of short K pieces
      // (N = 264 at K = 192: 591 tiles = 1.15 rounds, 68 vs 52 us; N = 224 at K = 264): a tail below 0.3 rounds needs K >= 384
      const double rounds = (double)ofb_cdiv(g.M, C96::BM) * (c96 / 96) / (double)(p_cu_count() * C96::WGS);
      const double frac = rounds - (double)(long long)rounds;
      if (!(frac > 0.0 && frac < 0.3 && g.K < 384)) return 96;
    }
  }
  return 128;
}

template <class CF>
Plan plan_p(const ofb_gemm_p_args& g) {
  int W = p_cu_count() * CF::WGS;
  if (!g.a_kc && !g.b_kc) {
    // weight-gradient form: ONE workgroup per CU.  Its 18-24 output tiles are cut along K into W pieces, each leaving a 96-KB
    // partial tile for the fix-up: half the workgroups = half that traffic (2.4 GB written + re-read per DeiT-S step) at an
    // equal step time (same-box A/B 24.9 / 25.0 ms both ways; the K loops of the co-running main-stream kernels fill the CUs).
    // OFB_GEMM_P_DW_WGS=2 restores two per CU (lab).
    static int dw_wgs = -1;
    if (dw_wgs < 0) { const char* e = getenv("OFB_GEMM_P_DW_WGS"); dw_wgs = e ? atoi(e) : 1; }
    if (dw_wgs > 0 && dw_wgs < CF::WGS) W = p_cu_count() * dw_wgs;
  }
#ifdef OFB_P_STAMPS
  { const char* e = getenv("OFB_GEMM_P_WCAP"); if (e && atoi(e) > 0 && atoi(e) < W) W = atoi(e); }   // lab: fewer resident workgroups
#endif
  const int tiles = ofb_cdiv(g.M, CF::BM) * ofb_cdiv(g.N, CF::BN);
  const long long iters = (long long)tiles * ofb_cdiv(g.K, 16);
  if (iters < W) W = (int)iters;
  Plan p = make_plan(g.M, g.N, g.K, W, CF::BM, CF::BN, 16);
  // Tail policy (costs in K steps of one workgroup, ~2.3 us): a streamed tail costs its K steps plus the partial tiles' round trip
  // through HBM (~0.019 steps per 96-KB partial at 4.5 TB/s) plus the extra launches (~3.5 steps); one more (partly idle)
  // data-parallel round costs I steps.  The split-major piece count S is chosen to minimise that sum (short tails: few large
  // pieces; weight gradients: one piece per workgroup); the tail must beat the round by 10 % to be worth its HBM traffic.
  if (p.R > 0 && g.colpart && p.R % p.nt != 0) {                     // column sums of tail tiles come from the fix-up kernel, which
    p.full_rounds += 1; p.R = 0; p.q = 0; p.S = 0; p.qs = 0;         // addresses them per whole tile row (colpart layout below)
  }
  if (p.R > 0) {
    const double PC = 0.019, FIX = 3.5;
    double best = 0.9 * p.I;
    int bestS = -1;                                                  // -1: run the remainder as one more round
    const int smax = W / p.R;
    for (int S = 2; S <= smax; ++S) {
      const double c = (double)((p.I + S - 1) / S) + (double)p.R * S * PC + FIX;
      if (c < best) { best = c; bestS = S; }
    }
    if (smax < 2) {                                                  // flattened runs: up to two partial tiles per workgroup
      const double c = (double)(((long long)p.R * p.I + W - 1) / W) + (double)(2 * p.R < 2 * W ? 2 * p.R : 2 * W) * PC + FIX;
      if (c < best) { best = c; bestS = 0; }
    }
    if (bestS < 0) { p.full_rounds += 1; p.R = 0; p.q = 0; p.S = 0; p.qs = 0; }
    else if (bestS > 0) { p.S = bestS; p.qs = (p.I + bestS - 1) / bestS; }
    else { p.S = 0; p.qs = 0; }
  }
  static int stag = -1;
  if (stag < 0) { const char* e = getenv("OFB_GEMM_P_STAGGER"); stag = e ? atoi(e) : 0; }
  p.stagger = stag;
  return p;
}

template <class CF, bool A_KC, bool B_KC, int EPI>
void launch_full(const ofb_gemm_p_args& g, const Plan& p, hipStream_t s) {
  hipLaunchKernelGGL((gemm_p_kernel<CF, A_KC, B_KC, false, EPI>), dim3(p.W), dim3(CF::NT), 0, s, g, p);
}

template <class CF, bool A_KC, bool B_KC>
int launch_p(const ofb_gemm_p_args& g, const Plan& p, hipStream_t s) {
  if (p.full_rounds > 0) {
    const int f = (g.C ? E_C : 0) | (g.Cp ? E_P : 0) | (g.act == OFB_ACT_GELU ? E_GELU : 0) | (g.act == OFB_ACT_DGELU ? E_DGELU : 0) |
                  (g.act == OFB_ACT_GELU_GRAD ? E_GELUG : 0) | (g.act == OFB_ACT_MULAUX ? E_MULAUX : 0) |
                  (g.rowscale ? E_RS : 0) | (g.resid ? E_RES : 0);
    switch (f) {     // the forms the model issues; anything else takes the generic (run-time flags) instantiation
      case E_C: launch_full<CF, A_KC, B_KC, E_C>(g, p, s); break;
      case E_C | E_RES: launch_full<CF, A_KC, B_KC, E_C | E_RES>(g, p, s); break;
      case E_C | E_RS | E_RES: launch_full<CF, A_KC, B_KC, E_C | E_RS | E_RES>(g, p, s); break;
      case E_P | E_GELUG: launch_full<CF, A_KC, B_KC, E_P | E_GELUG>(g, p, s); break;
      case E_P | E_MULAUX: launch_full<CF, A_KC, B_KC, E_P | E_MULAUX>(g, p, s); break;
      case E_P: launch_full<CF, A_KC, B_KC, E_P>(g, p, s); break;
      default: launch_full<CF, A_KC, B_KC, E_ANY>(g, p, s); break;
    }
  }
  if (p.R > 0) {
    hipLaunchKernelGGL((gemm_p_kernel<CF, A_KC, B_KC, true, 0>), dim3(p.W), dim3(CF::NT), 0, s, g, p);
    hipLaunchKernelGGL(gemm_p_fixup_kernel<CF>, dim3(p.R, CF::BM / 4), dim3(CF::BN), 0, s, g, p);
  }
  return ofb_launch_status();
}

template <class CF>
int run_p(const ofb_gemm_p_args& g, hipStream_t s) {
  const Plan p = plan_p<CF>(g);
  if ((long long)p.W * p.I > 0x7fffffffLL / 2) return OFB_ELIMIT;
  if (p.R && (!g.workspace || g.workspace_bytes < (int64_t)2 * p.W * CF::BM * CF::BN * (int64_t)sizeof(float))) return OFB_EINVAL;
  if (g.a_kc && g.b_kc) return launch_p<CF, true, true>(g, p, s);
  if (g.a_kc) return launch_p<CF, true, false>(g, p, s);
  return launch_p<CF, false, false>(g, p, s);
}

}  // namespace

extern "C" int64_t ofb_pformat_bytes(int32_t R, int32_t C) {
  if (R <= 0 || C <= 0) return 0;
  // Tile-granular reads run past the matrix: mode KC reads whole 128-row (A) or 192-row (B) tiles, so the row groups are
  // allocated up to the next 256-row boundary plus one more 256-row tile (covers any tile height <= 256 at any offset); mode KR
  // reads whole 128- / 192-column tiles, i.e. up to 12 granules behind the last row group.  The slack is never initialised and
  // only ever feeds accumulators that are not stored.
  const int64_t ncb = (C + 15) / 16, rgs = (int64_t)((R + 255) / 256) * 64 + 64;
  return (rgs * ncb + 16) * GRAN;
}

extern "C" int ofb_to_pformat(const float* X, int32_t R, int32_t C, int32_t ld, void* P, const float* rowscale, int32_t rs_div,
                              void* stream) {
  if (!X || !P || R <= 0 || C <= 0 || ld < C) return OFB_EINVAL;
  if (rowscale && rs_div <= 0) return OFB_EINVAL;
  const int ncb = (C + 15) / 16, rgs = ((R + 15) / 16) * 4;
  hipLaunchKernelGGL(to_pformat_kernel, dim3((ncb * 16 + 255) / 256, rgs), dim3(256), 0, (hipStream_t)stream, X, R, C, ld, (char*)P,
                     ncb, rowscale, rs_div);
  return ofb_launch_status();
}

// img [B][Cin][H][W] -> P-format planes of the patch matrix [B * (H/patch) * (W/patch)][Cin * patch * patch]
extern "C" int ofb_patchify_pformat(const float* img, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t patch, void* P, void* stream) {
  if (!img || !P || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || patch <= 0 || H % patch || W % patch) return OFB_EINVAL;
  const int64_t R = (int64_t)B * (H / patch) * (W / patch), Cc = (int64_t)Cin * patch * patch;
  if (R > 0x7fffffff / 4 || Cc > 65536) return OFB_ELIMIT;
  const int ncb = (int)((Cc + 15) / 16), rgs = (int)(((R + 15) / 16) * 4);
  hipLaunchKernelGGL(patchify_pformat_kernel, dim3((ncb * 16 + 255) / 256, rgs), dim3(256), 0, (hipStream_t)stream, img, B, Cin, H, W, patch,
                     (char*)P, ncb);
  return ofb_launch_status();
}

// jobs_dev: n_jobs descriptors in device memory; max_R / max_C: the largest R and C among them (grid extent)
extern "C" int ofb_to_pformat_multi(const ofb_pformat_job* jobs_dev, int32_t n_jobs, int32_t max_R, int32_t max_C, void* stream) {
  if (!jobs_dev || n_jobs <= 0 || n_jobs > 65535 || max_R <= 0 || max_C <= 0) return OFB_EINVAL;
  const int ncb = (max_C + 15) / 16, rgs = ((max_R + 15) / 16) * 4;
  hipLaunchKernelGGL(to_pformat_multi_kernel, dim3((ncb * 16 + 255) / 256, rgs, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  return ofb_launch_status();
}

extern "C" int ofb_to_pformat_colsum(const float* X, int32_t R, int32_t C, int32_t ld, void* P, const float* rowscale, int32_t rs_div,
                                     float* partial, void* stream) {
  if (!X || !P || !partial || R <= 0 || C <= 0 || ld < C) return OFB_EINVAL;
  if (rowscale && rs_div <= 0) return OFB_EINVAL;
  const int ncb = (C + 15) / 16, rgs = ((R + 15) / 16) * 4, slabs = (rgs + CS_SLAB_RG - 1) / CS_SLAB_RG;
  hipLaunchKernelGGL(to_pformat_colsum_kernel, dim3((ncb * 16 + 63) / 64, slabs), dim3(256), 0, (hipStream_t)stream, X, R, C, ld, (char*)P,
                     ncb, rgs, rowscale, rs_div, partial);
  return ofb_launch_status();
}

extern "C" int ofb_from_pformat(const void* P, int32_t R, int32_t C, float* X, int32_t ld, void* stream) {
  if (!X || !P || R <= 0 || C <= 0 || ld < C) return OFB_EINVAL;
  const int ncb = (C + 15) / 16, rgs = (R + 3) / 4;
  hipLaunchKernelGGL(from_pformat_kernel, dim3((C + 255) / 256, rgs), dim3(256), 0, (hipStream_t)stream, (const char*)P, ncb, X, R,
                     C, ld);
  return ofb_launch_status();
}

extern "C" int32_t ofb_colsum_p_slabs(int32_t R) { return R > 0 ? (((R + 15) / 16) * 4 + CS_SLAB_RG - 1) / CS_SLAB_RG : 0; }

extern "C" int ofb_colsum_p(const void* P, int32_t R, int32_t C, float* partial, void* stream) {
  if (!P || !partial || R <= 0 || C <= 0) return OFB_EINVAL;
  const int ncb = (C + 15) / 16, rgs = ((R + 15) / 16) * 4, slabs = ofb_colsum_p_slabs(R);
  hipLaunchKernelGGL(colsum_p_kernel, dim3((ncb + 3) / 4, slabs), dim3(256), 0, (hipStream_t)stream, (const char*)P, ncb, rgs, partial,
                     ncb * 16);
  return ofb_launch_status();
}

extern "C" int64_t ofb_gemm_p_workspace_bytes(const ofb_gemm_p_args* args) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
#ifdef OFB_GEMM_P_LAB
  if (p_tile_choice(*args) == 192) { const Plan p = plan_p<C192>(*args); return p.R ? (int64_t)2 * p.W * C192::BM * C192::BN * (int64_t)sizeof(float) : 0; }
#endif
  if (p_tile_choice(*args) == 96) { const Plan p = plan_p<C96>(*args); return p.R ? (int64_t)2 * p.W * C96::BM * C96::BN * (int64_t)sizeof(float) : 0; }
  const Plan p = plan_p<C128>(*args);
  return p.R ? (int64_t)2 * p.W * C128::BM * C128::BN * (int64_t)sizeof(float) : 0;
}

// rows of the colpart buffer a call with these arguments (colpart given) fills: one per 128-row tile row that runs data-parallel,
// BM/4 per tile row of the streamed tail
extern "C" int32_t ofb_gemm_p_colpart_rows(const ofb_gemm_p_args* args) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return 0;
  ofb_gemm_p_args g = *args;
  if (!g.colpart) g.colpart = reinterpret_cast<float*>(16);
  const Plan p = plan_p<C128>(g);
  return p.R ? (p.mt - p.R / p.nt) + (p.R / p.nt) * (C128::BM / 4) : p.mt;
}

extern "C" int ofb_gemm_p(const ofb_gemm_p_args* args, void* stream) {
  if (!args) return OFB_EINVAL;
  const ofb_gemm_p_args& g = *args;
  if (!g.A || !g.B || (!g.C && !g.Cp) || g.M <= 0 || g.N <= 0 || g.K <= 0) return OFB_EINVAL;
  if (g.a_kc == 0 && g.b_kc == 1) return OFB_ELIMIT;           // A^T * B^T is not on the path
  if (g.colpart && C128::BM != 128) return OFB_ELIMIT;
  if (g.rowscale && g.rs_div <= 0) return OFB_EINVAL;
  if (g.act < OFB_ACT_NONE || g.act > OFB_ACT_MULAUX) return OFB_EINVAL;
  if ((g.act == OFB_ACT_DGELU || g.act == OFB_ACT_GELU_GRAD || g.act == OFB_ACT_MULAUX) && !g.aux) return OFB_EINVAL;
  if (g.C && g.ldc < g.N) return OFB_EINVAL;
  if (g.Cp && g.c_ncb < (g.N + 15) / 16) return OFB_EINVAL;
  // granule columns of each operand's P matrix must cover its extent along that axis
  if (g.a_ncb < ((g.a_kc ? g.K : g.M) + 15) / 16 || g.b_ncb < ((g.b_kc ? g.K : g.N) + 15) / 16) return OFB_EINVAL;
  if (!ofb_aligned16(g.A) || !ofb_aligned16(g.B) || (g.Cp && !ofb_aligned16(g.Cp))) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(0, s, 2.0 * g.M * g.N * (double)g.K);
  const int tile = p_tile_choice(g);
#ifdef OFB_GEMM_P_LAB
  const int rc = tile == 192 ? run_p<C192>(g, s) : (tile == 96 ? run_p<C96>(g, s) : run_p<C128>(g, s));
#else
  const int rc = tile == 96 ? run_p<C96>(g, s) : run_p<C128>(g, s);
#endif
  ofb_prof_post(0, s);
  return rc;
}

#ifdef OFB_P_STAMPS
extern "C" int ofb_diag_p_stamps(unsigned long long* out_host) {      /* lab only, not part of the ABI */
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(ofb_p_stamps), sizeof(unsigned long long) * 1024 * 8 * 4);
}
#endif
