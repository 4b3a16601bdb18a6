// Fused gated-attention core for the OFB search step: softmax(q k^T * scale) v, forward and backward, on the
// f32-input MFMA v_mfma_f32_16x16x4_f32.  One workgroup of 13 waves per (batch, head): the sequence (N <= 208 tokens =
// 13 tiles of 16; DeiT: N = 197) is cut into 16-token tiles and wave w owns tile w, so 197 tokens pad to 208 (not 224)
// and every wave has the same amount of work.  K/V (forward) or K/Q/dO (backward) of the head live in LDS;
// probabilities never touch HBM.  Reference math: models/layers.py:510-514 (search branch) and its autograd.
//
// 16x16x4 operand conventions: lane l gives A[i = l&15][kslot = l>>4] and B[kslot = l>>4][j = l&15]; D[i][j] sits in
// reg r of lane l at i = 4*(l>>4) + r, j = l&15.  Two facts carry the design:
//  * any bijection reduction-index <-> (step, kslot) is valid as long as A and B use the same one;
//  * an accumulator tile X feeds the next MFMA directly as its A operand when that MFMA reduces over X's ROW index:
//    reg r of kslot-group g is row 4g + r, so the B operand is simply read at row 4g + r.
// Forward uses the S^T orientation (rows = keys, lane column = query): row-softmax is register-local plus two
// shuffles, and P^T feeds P.V directly.  Backward uses the S orientation (rows = queries, lane column = key): P and dS
// feed dV = P^T dO and dK = dS^T Q directly; only dS crosses LDS (transposed) for dQ = dS K, whose 13 per-wave partial
// tiles are parked in LDS and summed in a fixed order (deterministic).
#include "ofb_common.h"

#define ATT_T 16            // tokens per tile
#define ATT_NT 13           // tiles per head -> N <= 208
#define ATT_NMAX (ATT_T * ATT_NT)
#define ATT_DMAX 64
#define ATT_LD 68           // LDS row pitch (floats) of K / V / Q / dO tiles
#define ATT_DSLD 212        // pitch of the transposed dS tile [16 q][208 keys + 4]
#define ATT_THREADS (64 * ATT_NT)

namespace {

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ------------------------------------------------------------------------------------------------------------------
// forward: streaming (flash) form on the bf16 matrix pipe at fp32 accuracy
// ------------------------------------------------------------------------------------------------------------------
// Same exact three-way bf16 split and six product terms as the GEMM (csrc/gemm_p.hip), here on v_mfma_f32_16x16x32_bf16:
// lane l gives A[i = l&15][k = 8*(l>>4) + j] and B[k = 8*(l>>4) + j][col = l&15] (j = 0..7, one 16-byte fragment);
// D[i = 4*(l>>4) + r][col = l&15].  Wave w owns query tile w; keys stream through LDS in blocks of 32 (two 16-key tiles),
// double-buffered, each K / V value split ONCE per workgroup while it is staged, both as row-major planes [3][32 keys][64] bf16
// (128-B rows, 8-byte stores) with the 16-byte chunks of a row XOR-swizzled by the key so that the fragment reads are conflict free:
//   K planes: chunk ^ ((key >> 1) & 7).  A operand of S^T = K Q^T: fragment = 8 consecutive d of one key = one ds_read_b128
//   V planes: chunk ^ (((key >> 1) & 3) << 1).  A operand of O^T = V^T P^T: rows = channels, k-slot (g, j) = key 4g + j (j < 4) or
//             16 + 4g + j - 4 - the keys whose P^T values lane group g owns in its two S^T accumulators, so the probabilities feed
//             the MFMA from registers.  The fragment is two ds_read_b64_tr_b16: lane 4e + q of a 16-lane group addresses the four
//             channels 16 dt + 4q .. of key 4g + e, the hardware transpose hands lane i the keys 4g .. 4g + 3 of channel 16 dt + i.
// (Round-2a layout: padded 144-B K rows and a slot-ordered V^T image written with 2-byte scattered stores: 61 % of the kernel's LDS
// cycles were bank conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.)
// Online softmax over the key blocks (running max m, running sum l, O rescaled when m grows); lse = m + log l.
typedef __bf16 att_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 att_bf16x2 __attribute__((ext_vector_type(2)));
typedef float att_f32x2 __attribute__((ext_vector_type(2)));

#define AF_KB 32                       // keys per block
#define AF_NB ((ATT_NMAX + AF_KB - 1) / AF_KB)      // 7 blocks cover 224 >= 208 keys
#define AF_PITCH 128                   // bytes per key row of one plane: 64 bf16
#define AF_PLANE (AF_KB * AF_PITCH)
#define AF_STAGE (6 * AF_PLANE)        // K planes then V planes
typedef short att_s16x4 __attribute__((ext_vector_type(4)));
typedef short att_s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned att_pack(float a, float b) {
  att_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, att_bf16x2));
}
__device__ __forceinline__ void att_split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = att_pack(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  mid = att_pack(ra, rb);
  lo = att_pack(ra - __uint_as_float(mid << 16), rb - __uint_as_float(mid & 0xffff0000u));
}
// 8 floats -> three 8 x bf16 fragments
__device__ __forceinline__ void att_split8(const float (&x)[8], att_bf16x8 (&out)[3]) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) att_split_pair(x[2 * i], x[2 * i + 1], h[i], m[i], l[i]);
  out[0] = __builtin_bit_cast(att_bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  out[1] = __builtin_bit_cast(att_bf16x8, make_uint4(m[0], m[1], m[2], m[3]));
  out[2] = __builtin_bit_cast(att_bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}
__device__ __forceinline__ f32x4 att_mfma6(const att_bf16x8 (&a)[3], const att_bf16x8 (&b)[3], f32x4 c) {
  // (mid,mid) (hi,lo) (lo,hi) (hi,mid) (mid,hi) (hi,hi): smallest terms first
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
  return c;
}

// P-format stores (csrc/gemm_p.hip: granules of 4 rows x 16 columns, [plane][c % 16][r % 4] bf16, 384 B, stored [rows/4][ncb])
__device__ __forceinline__ char* att_p_slot(char* P, int ncb, int row, int col) {
  return P + ((size_t)(row >> 2) * ncb + (col >> 4)) * 384 + (col & 15) * 8 + (row & 3) * 2;
}
// four consecutive rows row0.. (row0 % 4 == 0: one 8-byte slot per plane) of one column; only rows [rlo, rhi) belong to the caller
__device__ __forceinline__ void att_store_p_col4(char* P, int ncb, int row0, int col, f32x4 v, int rlo, int rhi) {
  unsigned h0, m0, l0, h1, m1, l1;
  att_split_pair(v[0], v[1], h0, m0, l0);
  att_split_pair(v[2], v[3], h1, m1, l1);
  char* q = att_p_slot(P, ncb, row0, col);
  if (rlo == 0 && rhi == 4) {
    *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(q + 128) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(q + 256) = make_uint2(l0, l1);
    return;
  }
  const unsigned hs[2] = {h0, h1}, ms[2] = {m0, m1}, ls[2] = {l0, l1};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r >= rlo && r < rhi) {                             // rows of a neighbouring image in this slot are its workgroup's to write
      const int sh = (r & 1) * 16;
      *reinterpret_cast<unsigned short*>(q + 2 * r) = (unsigned short)(hs[r >> 1] >> sh);
      *reinterpret_cast<unsigned short*>(q + 2 * r + 128) = (unsigned short)(ms[r >> 1] >> sh);
      *reinterpret_cast<unsigned short*>(q + 2 * r + 256) = (unsigned short)(ls[r >> 1] >> sh);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// backward (one kernel: dq | dk | dv); P recomputed from lse
// ------------------------------------------------------------------------------------------------------------------
// PF: the output rows also leave as P-format planes oP (the operand of the projection GEMM).  The accumulators hold four CHANNELS
// of one query per lane, a plane slot is four QUERIES of one channel: every wave transposes its 16 x 64 tile through a private LDS
// patch (the stage buffers are free by then), 32 channels at a time.  As in the backward kernel the query tiles are laid from
// position -(b N % 4), so that every four tile rows are one 4-row granule of the [B N] matrix.
template <bool PF>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                               float* __restrict__ lse, char* __restrict__ oP, int p_ncb, int B, int N,
                                                               int H, int dh, float scale) {
  __shared__ __attribute__((aligned(16))) char smem[2 * AF_STAGE];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / H, head = blockIdx.x % H;
  const int ldq = 3 * H * dh, ldo = H * dh;
  const float* qbase = qkv + (size_t)b * N * ldq + head * dh;
  const float* kbase = qbase + H * dh;
  const float* vbase = qbase + 2 * H * dh;
  const int nb = (N + AF_KB - 1) / AF_KB;
  const int sft = PF ? (int)(((size_t)b * N) & 3) : 0;       // query tile position p <-> token p - sft (keys are not shifted)

  // staging items: idx < 512 -> K float4 (key = idx/16, d = 4*(idx%16)); 512 <= idx < 1024 -> V float4, same coordinates
  f32x4 sreg[2];
  auto stage_load = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = t + ATT_THREADS * i;
      sreg[i] = zero4();
      if (idx < 1024) {
        const int key = kb * AF_KB + ((idx & 511) >> 4), d4 = (idx & 15) << 2;
        if (key < N && d4 < dh) sreg[i] = *reinterpret_cast<const f32x4*>(((idx < 512) ? kbase : vbase) + (size_t)key * ldq + d4);
      }
    }
  };
  auto stage_store = [&](int buf) {
    char* st = smem + buf * AF_STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = t + ATT_THREADS * i;
      if (idx >= 1024) continue;
      const int kk = (idx & 511) >> 4, d16 = idx & 15, chunk = d16 >> 1;
      unsigned h0, m0, l0, h1, m1, l1;
      att_split_pair(sreg[i][0], sreg[i][1], h0, m0, l0);
      att_split_pair(sreg[i][2], sreg[i][3], h1, m1, l1);
      const bool isk = idx < 512;
      const int sw = isk ? ((kk >> 1) & 7) : (((kk >> 1) & 3) << 1);
      char* p = st + (isk ? 0 : 3 * AF_PLANE) + kk * AF_PITCH + ((chunk ^ sw) << 4) + ((d16 & 1) << 3);
      *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(p + AF_PLANE) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(p + 2 * AF_PLANE) = make_uint2(l0, l1);
    }
  };
  stage_load(0);
  stage_store(0);

  // this lane's query row (pre-scaled), d-slices [8g, 8g+8) and [32 + 8g, 32 + 8g + 8), as B-operand planes
  const int q = w * ATT_T + c - sft;
  const bool qvalid = q >= 0 && q < N;
  att_bf16x8 qf[2][3];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int d0 = 32 * ks + 8 * g + 4 * u;
      f32x4 v = zero4();
      if (qvalid && d0 < dh) v = *reinterpret_cast<const f32x4*>(qbase + (size_t)q * ldq + d0);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[4 * u + j] = v[j] * scale;
    }
    att_split8(x, qf[ks]);
  }
  f32x4 O[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) O[dt] = zero4();
  float m_run = -INFINITY, l_run = 0.f;
  // fragment read offsets inside a stage (bytes): K: key row 16 tk + c, chunk (4 ks + g) ^ ((c >> 1) & 7);
  // V (transposed reads): this lane addresses key 4g + (c >> 2), channels 16 dt + 4 (c & 3) ..: chunk (2 dt + ((c & 3) >> 1)) ^ swizzle
  int k_off[2], v_off[4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) k_off[ks] = c * AF_PITCH + (((4 * ks + g) ^ ((c >> 1) & 7)) << 4);
  {
    const int vkey = 4 * g + (c >> 2), q = c & 3, sw = ((vkey >> 1) & 3) << 1;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) v_off[dt] = 3 * AF_PLANE + vkey * AF_PITCH + (((2 * dt + (q >> 1)) ^ sw) << 4) + ((q & 1) << 3);
  }
  __syncthreads();

  const bool active = w * ATT_T < N + sft;
  for (int kb = 0; kb < nb; ++kb) {
    const char* st = smem + (kb & 1) * AF_STAGE;
    if (kb + 1 < nb) stage_load(kb + 1);
    if (active) {
      // S^T tiles of this key block: rows = keys, lane column = query
      f32x4 S[2];
#pragma unroll
      for (int tk = 0; tk < 2; ++tk) {
        f32x4 acc = zero4();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          att_bf16x8 kf[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            kf[pl] = *reinterpret_cast<const att_bf16x8*>(st + pl * AF_PLANE + 16 * tk * AF_PITCH + k_off[ks]);
          acc = att_mfma6(kf, qf[ks], acc);
        }
        S[tk] = acc;
      }
      // online softmax: this lane holds keys kb*32 + 16 tk + 4g + r of query c
      float mb = -INFINITY;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sv = (kb * AF_KB + 16 * tk + 4 * g + r < N) ? S[tk][r] : -INFINITY;
          S[tk][r] = sv;
          mb = fmaxf(mb, sv);
        }
      mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
      mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
      const float m_new = fmaxf(m_run, mb);                 // finite from block 0 on (key 0 is always valid)
      const float alpha = __expf(m_run - m_new);
      m_run = m_new;
      float p8[8];
      float ls = 0.f;
#pragma unroll
      for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __expf(S[tk][r] - m_new);
          p8[4 * tk + r] = pv;                               // k-slot j = 4 tk + r of group g
          ls += pv;
        }
      l_run = l_run * alpha + ls;
      att_bf16x8 pf[3];
      att_split8(p8, pf);
      // O^T[ch][query] = alpha * O^T + V^T P^T
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        att_bf16x8 vf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const char* vp = st + pl * AF_PLANE + v_off[dt];
          const att_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)(vp));
          const att_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)(vp + 16 * AF_PITCH));
          const att_s16x8 v8 = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
          vf[pl] = __builtin_bit_cast(att_bf16x8, v8);
        }
        O[dt] *= alpha;
        O[dt] = att_mfma6(vf, pf, O[dt]);
      }
    }
    if (kb + 1 < nb) stage_store((kb + 1) & 1);
    __syncthreads();
  }
  if (!active) return;
  float l = l_run;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float linv = 1.0f / l;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) O[dt] *= linv;
  if (qvalid) {
    if (g == 0) lse[((size_t)b * H + head) * N + q] = m_run + logf(l);
    // O[dt][r] = O^T[channel 16 dt + 4g + r][query c]: four consecutive channels of this lane's query
    float* op = out + ((size_t)b * N + q) * ldo + head * dh;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int ch = 16 * dt + 4 * g;
      if (ch < dh) *reinterpret_cast<f32x4*>(op + ch) = O[dt];
    }
  }
  if (PF) {
    // wave-private patch [32 ch][16 q + 4] f32 in the (now free) stage buffers: 2.5 KB per wave
    float* T = reinterpret_cast<float*>(smem) + w * (32 * 20);
    const int row0 = (int)((size_t)b * N) - sft + w * ATT_T;                            // global row of tile position 0: row0 % 4 == 0
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * dd + 4 * g + r) * 20 + c] = qvalid ? O[2 * half + dd][r] : 0.f;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int chl = lane & 31, qg = 2 * pass + (lane >> 5), ch = 32 * half + chl, p0 = w * ATT_T + 4 * qg;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&T[chl * 20 + 4 * qg]);
        const int rlo = max(0, sft - p0), rhi = min(4, N + sft - p0);
        if (ch < dh && rlo < rhi) att_store_p_col4(oP, p_ncb, row0 + 4 * qg, head * dh + ch, v, rlo, rhi);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// PF: dq | dk | dv leave the kernel as P-format planes (the operand form of the qkv gradient GEMMs, csrc/gemm_p.hip) together with
// their per-image column sums colpart[b][3 H dh] (the qkv bias gradient is their sum over b) instead of as f32 rows.  A lane's
// four accumulator rows are four consecutive tokens of one channel = one 8-byte plane slot IF they start on a 4-row granule
// boundary of the [B N] matrix: the workgroup therefore lays its 16-token tiles from position -sft (sft = b N % 4; tile position
// p holds token p - sft, the first sft positions are padding like the ones behind token N - 1).  N + 3 <= 208 is checked by the host.
// Only the image's first and last row group are shared with a neighbouring image; there the lane stores its own rows 2 bytes at a time.
#define ATT_QPITCH 20       // PF: column pitch (floats) of the parked partial dQ tiles [64 ch][16 q + 4]
template <bool PF>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                               const float* __restrict__ lse, const float* __restrict__ dout,
                                                               float* __restrict__ dqkv, char* __restrict__ dP_out, int p_ncb,
                                                               float* __restrict__ colpart, int B, int N, int H, int dh, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Kt = smem;                                   // [208][68]  unscaled K, row-major (B operand of dQ = dS K)
  float* Qs = Kt + ATT_NMAX * ATT_LD;                 // [2][16][68]
  float* dOs = Qs + 2 * ATT_T * ATT_LD;               // [2][16][68]
  float* dSs = dOs + 2 * ATT_T * ATT_LD;              // [16][212]  dS of the current query tile, [q][key]
  float* lse_s = dSs + ATT_T * ATT_DSLD;              // [208]
  float* del_s = lse_s + ATT_NMAX;                    // [208]
  float* dQp = del_s + ATT_NMAX;                      // per-wave partial dQ of the current query tile: [13][16][64], PF: [13][64][20]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / H, head = blockIdx.x % H;
  const int ldq = 3 * H * dh, ldo = H * dh;
  const size_t tok0 = (size_t)b * N;
  const int sft = PF ? (int)(tok0 & 3) : 0;           // tile position p <-> token p - sft
  const float* qbase = qkv + tok0 * ldq + head * dh;
  const float* kbase = qbase + H * dh;
  const float* vbase = qbase + 2 * H * dh;
  const float* obase = out + tok0 * ldo + head * dh;
  const float* dobase = dout + tok0 * ldo + head * dh;
  float* dqbase = dqkv + tok0 * ldq + head * dh;
  float* dkbase = dqbase + H * dh;
  float* dvbase = dqbase + 2 * H * dh;
  auto valid = [&](int pos) { return pos >= sft && pos - sft < N; };

  // ---- prologue: K -> LDS; delta[q] = rowsum(dO * O) and lse -> LDS; stage query tile 0 ----
  for (int idx = t; idx < ATT_NMAX * 16; idx += ATT_THREADS) {
    const int row = idx >> 4, c4 = (idx & 15) << 2, tok = row - sft;
    f32x4 kv = zero4(), dv = zero4(), ov = zero4();
    if (valid(row) && c4 < dh) {
      kv = *reinterpret_cast<const f32x4*>(kbase + (size_t)tok * ldq + c4);
      dv = *reinterpret_cast<const f32x4*>(dobase + (size_t)tok * ldo + c4);
      ov = *reinterpret_cast<const f32x4*>(obase + (size_t)tok * ldo + c4);
    }
    *reinterpret_cast<f32x4*>(&Kt[row * ATT_LD + c4]) = kv;
    float d = dv[0] * ov[0] + dv[1] * ov[1] + dv[2] * ov[2] + dv[3] * ov[3];
    d += __shfl_xor(d, 8, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 1, 64);
    if ((idx & 15) == 0) del_s[row] = d;
  }
  for (int i = t; i < ATT_NMAX; i += ATT_THREADS) lse_s[i] = valid(i) ? lse[((size_t)b * H + head) * N + i - sft] : 0.f;
  // staging of a 16-row query tile: threads 0..255 carry Q, 256..511 carry dO (one float4 each)
  const bool stager = t < 512;
  const int srow = (t & 255) >> 4, sc4 = (t & 15) << 2;
  f32x4 sreg = zero4();
  auto stage_load = [&](int qt) {
    sreg = zero4();
    const int qq = qt * ATT_T + srow;
    if (stager && valid(qq) && sc4 < dh)
      sreg = (t < 256) ? *reinterpret_cast<const f32x4*>(qbase + (size_t)(qq - sft) * ldq + sc4)
                       : *reinterpret_cast<const f32x4*>(dobase + (size_t)(qq - sft) * ldo + sc4);
  };
  auto stage_store = [&](int buf) {
    if (stager) *reinterpret_cast<f32x4*>(&((t < 256) ? Qs : dOs)[buf * ATT_T * ATT_LD + srow * ATT_LD + sc4]) = sreg;
  };
  stage_load(0);
  stage_store(0);

  // this wave's key rows as B operands: K (scaled) for S = Q K^T, V for dP = dO V^T
  const int key = w * ATT_T + c;
  const bool kvalid = valid(key);
  float kr[16], vr[16];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int cc = 16 * g + 4 * u;
    f32x4 kv = zero4(), vv = zero4();
    if (kvalid && cc < dh) {
      kv = *reinterpret_cast<const f32x4*>(kbase + (size_t)(key - sft) * ldq + cc);
      vv = *reinterpret_cast<const f32x4*>(vbase + (size_t)(key - sft) * ldq + cc);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { kr[4 * u + j] = kv[j] * scale; vr[4 * u + j] = vv[j]; }
  }
  f32x4 dK[4], dV[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dK[dt] = zero4(); dV[dt] = zero4(); }
  __syncthreads();

  const int nqt = (N + sft + ATT_T - 1) / ATT_T;
  float dq_sum = 0.f;                                      // PF: column sum of this thread's dQ column (4-row quad rq) over the query tiles
  for (int qt = 0; qt < nqt; ++qt) {
    const int buf = qt & 1;
    const float* Qb = Qs + buf * ATT_T * ATT_LD;
    const float* dOb = dOs + buf * ATT_T * ATT_LD;
#ifndef LAB_ATT_NOSTAGE
    if (qt + 1 < nqt) stage_load(qt + 1);                  // next tile's global loads in flight during this tile's math
#endif
    // ---- S' = scale Q K^T and dP = dO V^T: rows = queries (A from LDS), lane column = this wave's keys; 2 chains ----
    f32x4 S = zero4(), dP = zero4();
    {
      const float* qp = &Qb[c * ATT_LD + 16 * g];
      const float* dp = &dOb[c * ATT_LD + 16 * g];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + 4 * u);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dp + 4 * u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          S = mfma16(qv[j], kr[4 * u + j], S);
          dP = mfma16(dv[j], vr[4 * u + j], dP);
        }
      }
    }
    // P = exp(S' - lse[q]), dS' = P (dP - delta[q]);  query = 4g + r of the tile, key = this lane's column.
    // Padded query rows hold Q = dO = 0, lse = delta = 0 (P = 1, dS = 0, dO = 0): they add nothing.
    f32x4 P, dS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = qt * ATT_T + 4 * g + r;
      const float pv = kvalid ? __expf(S[r] - lse_s[qrow]) : 0.f;
      P[r] = pv;
      dS[r] = pv * (dP[r] - del_s[qrow]);
    }
    // dV += P^T dO, dK += dS'^T Q: the accumulators are the A operand (reduction over their row index = query)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* dop = &dOb[(4 * g + r) * ATT_LD + c];
      const float* qp2 = &Qb[(4 * g + r) * ATT_LD + c];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dV[dt] = mfma16(P[r], dop[16 * dt], dV[dt]);
        dK[dt] = mfma16(dS[r], qp2[16 * dt], dK[dt]);
      }
    }
    // dS' -> LDS [q][key]: the one transposition of the backward pass
#pragma unroll
    for (int r = 0; r < 4; ++r) dSs[(4 * g + r) * ATT_DSLD + key] = dS[r];
    __syncthreads();                                                                   // (A) dS tile complete
    // dQ partial over this wave's 16 keys: A[q][kslot g] = dS[q = c][16w + 4g + s], B = K[16w + 4g + s][16 dt + c]
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = zero4();
    {
      const f32x4 av = *reinterpret_cast<const f32x4*>(&dSs[c * ATT_DSLD + w * ATT_T + 4 * g]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* kp = &Kt[(w * ATT_T + 4 * g + s) * ATT_LD + c];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma16(av[s], kp[16 * dt], dq[dt]);
      }
    }
    // dq[dt][r] = partial dQ[query 4g + r][channel 16 dt + c]: each wave parks its partial tile in its own LDS slot and
    // 256 threads sum the 13 slots after the barrier (LDS float atomics cost ~180 cycles per wave-instruction here).
    // f32 output: slot [16 q][64 ch], a thread sums 4 channels of one query.  PF: slot [64 ch][16 q (+4)], a thread sums the 4
    // queries of one row group for one channel = one plane slot.
    if (PF) {
      float* mine = dQp + w * (ATT_DMAX * ATT_QPITCH);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(&mine[(16 * dt + c) * ATT_QPITCH + 4 * g]) = dq[dt];
    } else {
      float* mine = dQp + w * ATT_T * ATT_DMAX;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(4 * g + r) * ATT_DMAX + 16 * dt + c] = dq[dt][r];
    }
    if (qt + 1 < nqt) stage_store(buf ^ 1);
    __syncthreads();                                                                   // (B) partial dQ tiles parked, next Q/dO staged
    if (t < 256) {
      if (PF) {
        const int ch = t & 63, rq = t >> 6, p0 = qt * ATT_T + 4 * rq;                  // tile positions p0 .. p0 + 3
        f32x4 v = zero4();
#pragma unroll
        for (int ww = 0; ww < ATT_NT; ++ww) v += *reinterpret_cast<const f32x4*>(&dQp[ww * (ATT_DMAX * ATT_QPITCH) + ch * ATT_QPITCH + 4 * rq]);
        const int rlo = max(0, sft - p0), rhi = min(4, N + sft - p0);
        if (ch < dh && rlo < rhi) {
          v *= scale;
#pragma unroll
          for (int r = 0; r < 4; ++r) dq_sum += (r >= rlo && r < rhi) ? v[r] : 0.f;
          att_store_p_col4(dP_out, p_ncb, (int)tok0 - sft + p0, head * dh + ch, v, rlo, rhi);
        }
      } else {
        const int qq = qt * ATT_T + srow;
        f32x4 v = zero4();
#pragma unroll
        for (int ww = 0; ww < ATT_NT; ++ww) v += *reinterpret_cast<const f32x4*>(&dQp[ww * ATT_T * ATT_DMAX + srow * ATT_DMAX + sc4]);
        if (qq < N && sc4 < dh) {
          v *= scale;
          *reinterpret_cast<f32x4*>(dqbase + (size_t)qq * ldq + sc4) = v;
        }
      }
    }
    // no third barrier: the next partial-dQ writes come after the next (A), the next dS writes after this (B)
  }
  // ---- dK (x scale), dV: dK[dt][r] = dK[key 16w + 4g + r][channel 16 dt + c] ----
  if (!PF) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = w * ATT_T + 4 * g + r;
      if (kk < N) {
        float* dkp = dkbase + (size_t)kk * ldq + c;
        float* dvp = dvbase + (size_t)kk * ldq + c;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          if (16 * dt + c < dh) { dkp[16 * dt] = dK[dt][r] * scale; dvp[16 * dt] = dV[dt][r]; }
      }
    }
    return;
  }
  {
    const int k0 = w * ATT_T + 4 * g, row0 = (int)tok0 - sft + k0;                     // row0 % 4 == 0
    const int rlo = max(0, sft - k0), rhi = min(4, N + sft - k0);                      // this image's rows of the slot
    float* cs = dQp;                                       // [13 waves][2][64] column sums of this wave's dK / dV rows (dQp is free now)
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int ch = 16 * dt + c;
      f32x4 kq = dK[dt] * scale, vq = dV[dt];
      float sk = 0.f, sv = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r < rlo || r >= rhi) { kq[r] = 0.f; vq[r] = 0.f; }
        sk += kq[r]; sv += vq[r];
      }
      sk += __shfl_xor(sk, 16, 64); sk += __shfl_xor(sk, 32, 64);
      sv += __shfl_xor(sv, 16, 64); sv += __shfl_xor(sv, 32, 64);
      if (g == 0) { cs[(w * 2 + 0) * 64 + ch] = sk; cs[(w * 2 + 1) * 64 + ch] = sv; }
      if (ch < dh && rlo < rhi) {
        att_store_p_col4(dP_out, p_ncb, row0, H * dh + head * dh + ch, kq, rlo, rhi);
        att_store_p_col4(dP_out, p_ncb, row0, 2 * H * dh + head * dh + ch, vq, rlo, rhi);
      }
    }
    // dQ column sums: thread (channel t & 63, row quad t >> 6) holds its sum over the query tiles; the four quads are added in order
    float* qs = cs + ATT_NT * 2 * 64;
    if (t < 256) qs[(t >> 6) * 64 + (t & 63)] = dq_sum;
    __syncthreads();
    if (t < 192) {
      const int part = t >> 6, ch = t & 63;
      float sum = 0.f;
      if (part == 0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) sum += qs[rr * 64 + ch];
      } else {
#pragma unroll
        for (int ww = 0; ww < ATT_NT; ++ww) sum += cs[(ww * 2 + (part - 1)) * 64 + ch];
      }
      if (ch < dh) colpart[(size_t)b * (3 * H * dh) + part * (H * dh) + head * dh + ch] = sum;
    }
  }
}

constexpr size_t BWD_LDS_BASE = (size_t)(ATT_NMAX * ATT_LD + 4 * ATT_T * ATT_LD + ATT_T * ATT_DSLD + 2 * ATT_NMAX) * sizeof(float);
constexpr size_t BWD_LDS = BWD_LDS_BASE + (size_t)(ATT_NT * ATT_T * ATT_DMAX) * sizeof(float);
constexpr size_t BWD_LDS_P = BWD_LDS_BASE + (size_t)(ATT_NT * ATT_DMAX * ATT_QPITCH) * sizeof(float);

int check_shape(int B, int N, int H, int dh) {
  if (B <= 0 || N <= 0 || H <= 0 || dh <= 0) return OFB_EINVAL;
  if (N > ATT_NMAX || dh > ATT_DMAX || (dh & 3)) return OFB_ELIMIT;
  return OFB_OK;
}

}  // namespace

// qkv: [B*N][3*H*dh] packed as the qkv Linear writes it (q | k | v, each H*dh, head-major); out: [B*N][H*dh];
// lse: [B*H][N].  N <= 208, dh <= 64, dh % 4 == 0.
extern "C" int ofb_attention_fwd(const float* qkv, float* out, float* lse, int32_t B, int32_t N, int32_t H, int32_t dh,
                                 float scale, void* stream) {
  if (!qkv || !out || !lse) return OFB_EINVAL;
  if (int rc = check_shape(B, N, H, dh)) return rc;
  if (!ofb_aligned16(qkv)) return OFB_EINVAL;
  if (!ofb_aligned16(out) || ((H * dh) & 3)) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(1, s, 4.0 * B * H * (double)N * N * dh);
  hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3(B * H), dim3(ATT_THREADS), 0, s, qkv, out, lse, (char*)nullptr, 0, B, N, H, dh, scale);
  ofb_prof_post(1, s);
  return ofb_launch_status();
}

// out as above AND the same rows as P-format planes out_p[B*N][H*dh] (the operand of the projection GEMM); the caller zeroes
// out_p beforehand when B*N or H*dh is not a multiple of 16.  Needs N + (b N mod 4) <= 208.
extern "C" int ofb_attention_fwd_p(const float* qkv, float* out, void* out_p, float* lse, int32_t B, int32_t N, int32_t H, int32_t dh,
                                   float scale, void* stream) {
  if (!qkv || !out || !out_p || !lse) return OFB_EINVAL;
  if (int rc = check_shape(B, N, H, dh)) return rc;
  if (!ofb_aligned16(qkv) || !ofb_aligned16(out) || !ofb_aligned16(out_p) || ((H * dh) & 3)) return OFB_EINVAL;
  int smax = 0;
  for (int bb = 0; bb < B && bb < 4; ++bb) smax = ((bb * N) & 3) > smax ? ((bb * N) & 3) : smax;
  if (N + smax > ATT_NMAX) return OFB_ELIMIT;
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(1, s, 4.0 * B * H * (double)N * N * dh);
  hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3(B * H), dim3(ATT_THREADS), 0, s, qkv, out, lse, (char*)out_p, (H * dh + 15) / 16, B, N, H, dh,
                     scale);
  ofb_prof_post(1, s);
  return ofb_launch_status();
}

namespace {
int attention_bwd_launch(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv, char* dP_out, int p_ncb,
                         float* colpart, int B, int N, int H, int dh, float scale, hipStream_t s) {
  // the attribute is per device and the call is cheap: set it on every launch (a per-process flag would leave a second GPU
  // of the same process without it, and is not thread-safe)
  const void* fn = dP_out ? (const void*)attn_bwd_kernel<true> : (const void*)attn_bwd_kernel<false>;
  const size_t ldsb = dP_out ? BWD_LDS_P : BWD_LDS;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return (int)hipGetLastError();
  ofb_prof_pre(4, s, 10.0 * B * H * (double)N * N * dh);
  if (dP_out)
    hipLaunchKernelGGL(attn_bwd_kernel<true>, dim3(B * H), dim3(ATT_THREADS), ldsb, s, qkv, out, lse, dout, dqkv, dP_out, p_ncb, colpart,
                       B, N, H, dh, scale);
  else
    hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3(B * H), dim3(ATT_THREADS), ldsb, s, qkv, out, lse, dout, dqkv, dP_out, p_ncb, colpart,
                       B, N, H, dh, scale);
  ofb_prof_post(4, s);
  return ofb_launch_status();
}
}  // namespace

// dqkv: same packing as qkv (dq | dk | dv).  Needs the forward's out and lse.
extern "C" int ofb_attention_bwd(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv,
                                 int32_t B, int32_t N, int32_t H, int32_t dh, float scale, void* stream) {
  if (!qkv || !out || !lse || !dout || !dqkv) return OFB_EINVAL;
  if (int rc = check_shape(B, N, H, dh)) return rc;
  if (!ofb_aligned16(qkv) || !ofb_aligned16(out) || !ofb_aligned16(dout) || !ofb_aligned16(dqkv)) return OFB_EINVAL;
  return attention_bwd_launch(qkv, out, lse, dout, dqkv, nullptr, 0, nullptr, B, N, H, dh, scale, (hipStream_t)stream);
}

// The same gradient as P-format planes of the [B*N][3*H*dh] matrix (ofb_pformat_bytes(B*N, 3*H*dh) bytes; rows >= B*N of the
// last 16-row group must be zero beforehand when B*N % 16 != 0) plus colpart[B][3*H*dh], the column sums over each image's
// tokens (added in a fixed order).  The planes hold exactly the f32 values ofb_attention_bwd writes.
extern "C" int ofb_attention_bwd_p(const float* qkv, const float* out, const float* lse, const float* dout, void* dqkv_p,
                                   float* colpart, int32_t B, int32_t N, int32_t H, int32_t dh, float scale, void* stream) {
  if (!qkv || !out || !lse || !dout || !dqkv_p || !colpart) return OFB_EINVAL;
  if (int rc = check_shape(B, N, H, dh)) return rc;
  if (!ofb_aligned16(qkv) || !ofb_aligned16(out) || !ofb_aligned16(dout) || !ofb_aligned16(dqkv_p)) return OFB_EINVAL;
  int smax = 0;                                           // the tiles start (b N % 4) positions before the image's first token
  for (int bb = 0; bb < B && bb < 4; ++bb) smax = ((bb * N) & 3) > smax ? ((bb * N) & 3) : smax;
  if (N + smax > ATT_NMAX) return OFB_ELIMIT;
  return attention_bwd_launch(qkv, out, lse, dout, nullptr, (char*)dqkv_p, (3 * H * dh + 15) / 16, colpart, B, N, H, dh, scale,
                              (hipStream_t)stream);
}
