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

// the same six terms with the roles of the operands exchanged (a <-> b of att_mfma6): the backward forms S = Q K^T with A = Q where
// the forward forms S^T = K Q^T with A = K; issued in this order each element sees the same products in the same order and S comes
// out bit-identical to the forward's
__device__ __forceinline__ f32x4 att_mfma6_swapped(const att_bf16x8 (&a)[3], const att_bf16x8 (&b)[3], f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
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
  f32x4 sreg[3];
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
    if (g == 0) {
      // lse as TWO floats: [0][b h][q] = fl(m + log l) and, B H N floats further, its rounding residue (m - lse) + log l: the backward
      // forms P = exp((S - lse) - residue) = exp(S - m) / l without the half-ulp of |lse| a single float would put on the whole row
      const float ll = logf(l), ls = m_run + ll;
      const size_t li = ((size_t)b * H + head) * N + q;
      lse[li] = ls;
      lse[(size_t)B * H * N + li] = (m_run - ls) + ll;
    }
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

// ------------------------------------------------------------------------------------------------------------------
// backward on the split engine (round 3): one kernel, dq | dk | dv, P recomputed from lse; every product is the exact three-way
// bf16 split with six v_mfma_f32_16x16x32_bf16 terms (fp32-class accuracy, like the forward and the GEMM).
// ------------------------------------------------------------------------------------------------------------------
// One workgroup of EIGHT waves per (batch, head); wave w < 7 owns the 32 key positions 32w .. 32w+31 (224 positions cover N + 3 <= 208
// tokens plus the shift below; wave 7 owns none: it helps staging and takes a dQ tile) and keeps dK / dV of its keys in 64 accumulator registers while the workgroup sweeps the queries in
// blocks of 32.  "Key on the lane": S = Q K^T and dP = dO V^T are computed with rows = queries, lane column = key, so the P and dS
// accumulators ARE the A operands (k-slot (g, j) <-> query 16 (j >> 2) + 4 g + (j & 3)) of dV = P^T dO and dK = dS^T Q; the other
// operand of those two products comes out of the row-major Q / dO planes by ds_read_b64_tr_b16.  -lse[q] and -delta[q] are the
// initial accumulator of dP (dS = P * acc).  Q is staged pre-scaled exactly as the forward scales it (S' = (scale Q) K^T, dK =
// dS'^T (scale Q)) and the six terms of S' are issued in the forward's order, so S' is bit-identical to the forward's and
// P = exp((S' - lse) - residue) reproduces the forward's normalised probabilities (lse travels as two floats).
// Only dS crosses LDS: every wave stores its dS block as planes [key][32 q] (8-byte stores: the two halves of its A fragment),
// and after one barrier each (16-query, 16-channel) tile of dQ = dS K is reduced over ALL keys by ONE wave (A = dS by transposed
// reads, B = K by transposed reads of the K planes that sit in LDS for the whole kernel): no partial dQ tiles, no cross-wave sum,
// fixed summation order.  Eight dQ tiles per block, one per wave (two per SIMD).
// LDS (155.4 KB, one workgroup per CU): K planes [3][224][64] bf16 86 KB | dS planes [3][224 keys][32 q] 42 KB | Q and dO planes of
// the current block [3][32][64] 12 KB each | -lse, -delta 1.75 KB.  All 128-byte-row images carry the chunk swizzle
// chunk ^ (((row >> 1) & 3) << 1): conflict-free for the ds_read_b128 row reads AND for the transposed reads; the 64-byte dS rows
// swap their two 32-byte halves with bit 2 of the key.
// PF: dq | dk | dv leave the kernel as P-format planes (the operand form of the qkv gradient GEMMs, csrc/gemm_p.hip) together with
// their per-image column sums colpart[b][3 H dh] (the qkv bias gradient is their sum over b).  A lane's four accumulator rows are
// four consecutive tokens of one channel = one 8-byte plane slot IF they start on a 4-row granule boundary of the [B N] matrix: the
// workgroup lays its positions from -sft (sft = b N % 4; position p holds token p - sft).  Only the image's first and last row
// group are shared with a neighbouring image; there the lane stores its own rows 2 bytes at a time.
#define AB_NW 8                       /* waves: 0..6 own 32 key positions each, wave 7 only stages and takes a dQ tile */
#define AB_NKW 7
#define AB_THREADS (64 * AB_NW)
#define AB_QB 32
#define AB_NPOS (32 * AB_NKW)
#define AB_KPL (AB_NPOS * 128)
#define AB_DSPL (AB_NPOS * 64)
#define AB_STPL (AB_QB * 128)
#define AB_OFF_DS (3 * AB_KPL)
#define AB_OFF_STQ (AB_OFF_DS + 3 * AB_DSPL)
#define AB_OFF_STO (AB_OFF_STQ + 3 * AB_STPL)
#define AB_OFF_NLSE (AB_OFF_STO + 3 * AB_STPL)
#define AB_OFF_NDEL (AB_OFF_NLSE + AB_NPOS * 4)
#define AB_OFF_NRES (AB_OFF_NDEL + AB_NPOS * 4)
#define AB_LDS_BYTES (AB_OFF_NRES + AB_NPOS * 4)

__device__ __forceinline__ int ab_swz(int row) { return ((row >> 1) & 3) << 1; }
__device__ __forceinline__ att_bf16x8 ab_tr2(const char* p0, const char* p1) {
  const att_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)(p0));
  const att_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)(p1));
  const att_s16x8 v8 = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  return __builtin_bit_cast(att_bf16x8, v8);
}

#ifdef OFB_ATT_STAMPS
// lab only (scripts/lab/stamp_att.py): s_memtime stamps of every wave of workgroup 0: [wave][64 slots]
__device__ unsigned long long ofb_att_stamps[8 * 64];
#define AB_STAMP(slot) do { if (blockIdx.x == gridDim.x - 100 && lane == 0 && (slot) < 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ofb_att_stamps[w * 64 + (slot)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define AB_STAMP(slot) do { } while (0)
#endif

template <bool PF>
__global__ __launch_bounds__(AB_THREADS, 2) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                                 const float* __restrict__ lse, const float* __restrict__ dout,
                                                                 float* __restrict__ dqkv, char* __restrict__ dP_out, int p_ncb,
                                                                 float* __restrict__ colpart, int B, int N, int H, int dh, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kpl = smem;
  char* dSpl = smem + AB_OFF_DS;
#ifdef OFB_ATT_STAMPS
  { const int lane = threadIdx.x & 63, w = threadIdx.x >> 6; AB_STAMP(0); }
#endif
  char* stQ = smem + AB_OFF_STQ;
  char* stO = smem + AB_OFF_STO;
  float* nlse = reinterpret_cast<float*>(smem + AB_OFF_NLSE);
  float* ndel = reinterpret_cast<float*>(smem + AB_OFF_NDEL);
  float* nres = reinterpret_cast<float*>(smem + AB_OFF_NRES);
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / H, head = blockIdx.x % H;
  const int ldq = 3 * H * dh, ldo = H * dh;
  const size_t tok0 = (size_t)b * N;
  const int sft = PF ? (int)(tok0 & 3) : 0;           // position p <-> token p - sft
  const int npos = N + sft, nqb = (npos + AB_QB - 1) / AB_QB, nct = (dh + 15) >> 4, nks = (dh + 31) >> 5;
  const float* qbase = qkv + tok0 * ldq + head * dh;
  const float* kbase = qbase + H * dh;
  const float* vbase = qbase + 2 * H * dh;
  const float* obase = out + tok0 * ldo + head * dh;
  const float* dobase = dout + tok0 * ldo + head * dh;
  auto valid = [&](int pos) { return pos >= sft && pos < npos; };

  // ---- prologue: K planes (all positions; zeros outside the image) and -lse -> LDS; -delta[q] = -rowsum(dO * O) is formed block by
  // block in the staging pass (dO and O are then read once).  3584 (position, 4-channel) items, 7 per thread, all loads issued first
  // staging of a 32-position query block: thread t carries the Q, dO and O float4 of (row t / 16, d = 4 (t % 16))
  f32x4 sreg[3];
  // (scalar base pointer + one 32-bit per-lane offset: nothing 64-bit per lane that hipcc would hoist out of the block loop and spill)
  auto stage_load = [&](int qb) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {                           // i = 0: Q, 1: dO, 2: O (512 threads = 512 items each)
      const float* base = i == 0 ? qbase : (i == 1 ? dobase : obase);
      const int ld = i == 0 ? ldq : ldo;
      const int pos = qb * AB_QB + (t >> 4), d4 = (t & 15) << 2;
      sreg[i] = zero4();
      if (valid(pos) && d4 < dh) sreg[i] = *reinterpret_cast<const f32x4*>(base + (unsigned)((pos - sft) * ld + d4));
    }
  };
  auto stage_store = [&](int qb) {
    {                                                       // -delta of the row: 16 lanes hold its 64 channels
      float d = sreg[1][0] * sreg[2][0] + sreg[1][1] * sreg[2][1] + sreg[1][2] * sreg[2][2] + sreg[1][3] * sreg[2][3];
      d += __shfl_xor(d, 8, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 1, 64);
      if ((t & 15) == 0) ndel[qb * AB_QB + (t >> 4)] = -d;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = t >> 4, c4 = t & 15;
      unsigned h0, m0, l0, h1, m1, l1;
      const float qsc = (i == 0) ? scale : 1.0f;            // Q is staged pre-scaled: S' = (scale Q) K^T and dK = dS'^T (scale Q)
      att_split_pair(sreg[i][0] * qsc, sreg[i][1] * qsc, h0, m0, l0);
      att_split_pair(sreg[i][2] * qsc, sreg[i][3] * qsc, h1, m1, l1);
      char* p = ((i == 0) ? stQ : stO) + row * 128 + (((c4 >> 1) ^ ab_swz(row)) << 4) + ((c4 & 1) << 3);
      *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(p + AB_STPL) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(p + 2 * AB_STPL) = make_uint2(l0, l1);
    }
  };

  // EVERY global load of the prologue is issued before the first one is consumed (one memory round trip instead of four: the
  // workgroup is alone on its CU, nothing else hides them): K rows, lse, query block 0, this wave's V rows
  constexpr int NIT = AB_NPOS * 16 / AB_THREADS;
  static_assert(NIT * AB_THREADS == AB_NPOS * 16 && AB_NPOS <= AB_THREADS, "prologue items");
  const bool has_keys = w < AB_NKW && 32 * w < npos;
  f32x4 kv[NIT], vraw[2][2][2];
  float lse_a = 0.f, lse_b = 0.f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = t + AB_THREADS * i, pos = idx >> 4, c4 = idx & 15, tok = pos - sft;
    kv[i] = zero4();
    if (valid(pos) && 4 * c4 < dh) kv[i] = *reinterpret_cast<const f32x4*>(kbase + (unsigned)(tok * ldq + 4 * c4));
  }
  if (t < AB_NPOS && valid(t)) {
    const size_t li = ((size_t)b * H + head) * N + t - sft;
    lse_a = lse[li];
    lse_b = lse[(size_t)B * H * N + li];                    // the forward's second float of lse (rounding residue)
  }
  stage_load(0);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int pos = 32 * w + 16 * kt + c;
    const bool kv_ok = valid(pos);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int d0 = 32 * ks + 8 * g + 4 * u;
        vraw[kt][ks][u] = zero4();
        if (kv_ok && d0 < dh) vraw[kt][ks][u] = *reinterpret_cast<const f32x4*>(vbase + (unsigned)((pos - sft) * ldq + d0));
      }
  }
  AB_STAMP(58);
  // K planes
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = t + AB_THREADS * i, pos = idx >> 4, c4 = idx & 15;
    unsigned h0, m0, l0, h1, m1, l1;
    att_split_pair(kv[i][0], kv[i][1], h0, m0, l0);
    att_split_pair(kv[i][2], kv[i][3], h1, m1, l1);
    char* p = Kpl + pos * 128 + ((((c4 >> 1) ^ ab_swz(pos))) << 4) + ((c4 & 1) << 3);
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + AB_KPL) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(p + 2 * AB_KPL) = make_uint2(l0, l1);
  }
  AB_STAMP(59);
  if (t < AB_NPOS) { nlse[t] = -lse_a; nres[t] = -lse_b; }
  stage_store(0);
  AB_STAMP(61);
  // this wave's V rows as B operand of dP = dO V^T (lane: key 32 w + 16 kt + c, d = 32 ks + 8 g ..); the B operand of S' = (scale Q) K^T
  // is read from the K planes every block (row reads: the registers are needed for the accumulators)
  att_bf16x8 Vb[2][2][3];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float xv[8];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[4 * u + j] = vraw[kt][ks][u][j];
      att_split8(xv, Vb[kt][ks]);
    }
  AB_STAMP(62);
  bool kval[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) kval[kt] = valid(32 * w + 16 * kt + c);

  f32x4 dKa[2][4], dVa[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { dKa[kt][ct] = zero4(); dVa[kt][ct] = zero4(); }

  // dQ tiles of a block: tile i = (qt = i / nct, ct = i % nct), owned by wave i (eight tiles at dh = 64: one per wave, two per SIMD)
  const int ntile = 2 * nct;
  float dqs[2] = {0.f, 0.f};                               // PF: column sum of this wave's dQ tile, this lane's channel, rows 4g..

  __syncthreads();
  AB_STAMP(1);

  const bool has_tile = w < ntile;                        // this wave's dQ tile: (qt, ct) = (w / nct, w % nct)
  const int tq = has_tile ? w / nct : 0, tc = has_tile ? w - tq * nct : 0;
  for (int qb = 0; qb < nqb; ++qb) {
    AB_STAMP(2 + 8 * qb);
    // fragment addresses (bytes inside a plane), re-derived every block from "laundered" lane ids: left to itself hipcc hoists the
    // ~40 block-invariant LDS addresses of the body out of the loop and then spills them around it
    int cl = c, gl = g;
    asm volatile("" : "+v"(cl), "+v"(gl));
    const int a_row = cl * 128, a_sw = ab_swz(cl);          // row reads: row 16 qt + c (same swizzle for both qt: 16 is a multiple of 8)
    const int tr_row = 4 * gl + (cl >> 2), tr_sw = ab_swz(tr_row), tr_p = cl & 3;   // transposed reads: row (+16 for the second read)
    if (has_keys) {
      // ---- S' and dP - delta for the block's 32 queries x this wave's 32 keys ----
      f32x4 S[2][2], dPa[2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const f32x4 nd = *reinterpret_cast<const f32x4*>(&ndel[qb * AB_QB + 16 * qt + 4 * gl]);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) { S[qt][kt] = zero4(); dPa[qt][kt] = nd; }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks < nks) {
          const int off = a_row + ((((4 * ks + gl)) ^ a_sw) << 4);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            // (the fences keep hipcc from hoisting every group's fragment reads to the top of the block: 140 registers of operands)
            __builtin_amdgcn_sched_barrier(0);
            att_bf16x8 qa[3], oa[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
              qa[pl] = *reinterpret_cast<const att_bf16x8*>(stQ + pl * AB_STPL + 16 * qt * 128 + off);
              oa[pl] = *reinterpret_cast<const att_bf16x8*>(stO + pl * AB_STPL + 16 * qt * 128 + off);
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              att_bf16x8 kb[3];
#pragma unroll
              for (int pl = 0; pl < 3; ++pl) kb[pl] = *reinterpret_cast<const att_bf16x8*>(Kpl + pl * AB_KPL + (32 * w + 16 * kt) * 128 + off);
              S[qt][kt] = att_mfma6_swapped(qa, kb, S[qt][kt]);              // bit-identical to the forward's S
              dPa[qt][kt] = att_mfma6(oa, Vb[kt][ks], dPa[qt][kt]);
            }
          }
        }
      }
      AB_STAMP(3 + 8 * qb);
      // ---- P = exp((S' - lse) - residue), dS' = P (dP - delta); as A fragments (k-slot (g, 4 qt + r) = query 16 qt + 4 g + r) ----
      att_bf16x8 pf[2][3], dsf[2][3];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        float p8[8], d8[8];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const f32x4 nl = *reinterpret_cast<const f32x4*>(&nlse[qb * AB_QB + 16 * qt + 4 * gl]);
          const f32x4 nr = *reinterpret_cast<const f32x4*>(&nres[qb * AB_QB + 16 * qt + 4 * gl]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = kval[kt] ? __expf((S[qt][kt][r] + nl[r]) + nr[r]) : 0.f;
            p8[4 * qt + r] = pv;
            d8[4 * qt + r] = pv * dPa[qt][kt][r];
          }
        }
        att_split8(p8, pf[kt]);
        att_split8(d8, dsf[kt]);
        // dS planes [key][32 q]: the fragment's two halves are the 8-byte (4 q) slots of q tile 0 / 1
        const int key = 32 * w + 16 * kt + cl, hsw = (key >> 2) & 1;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const uint4 v = __builtin_bit_cast(uint4, dsf[kt][pl]);
          char* p = dSpl + pl * AB_DSPL + key * 64 + 8 * gl;
          *reinterpret_cast<uint2*>(p + 32 * (0 ^ hsw)) = make_uint2(v.x, v.y);
          *reinterpret_cast<uint2*>(p + 32 * (1 ^ hsw)) = make_uint2(v.z, v.w);
        }
      }
      AB_STAMP(4 + 8 * qb);
      // ---- dV += P^T dO, dK += dS'^T Q: B = dO / Q [query slots][channel] by transposed reads ----
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        if (ct < nct) {
          __builtin_amdgcn_sched_barrier(0);
          att_bf16x8 bo[3], bq[3];
          const int off = tr_row * 128 + (((2 * ct + (tr_p >> 1)) ^ tr_sw) << 4) + ((tr_p & 1) << 3);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            bo[pl] = ab_tr2(stO + pl * AB_STPL + off, stO + pl * AB_STPL + off + 16 * 128);
            bq[pl] = ab_tr2(stQ + pl * AB_STPL + off, stQ + pl * AB_STPL + off + 16 * 128);
          }
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            dVa[kt][ct] = att_mfma6(pf[kt], bo, dVa[kt][ct]);
            dKa[kt][ct] = att_mfma6(dsf[kt], bq, dKa[kt][ct]);
          }
        }
      }
    }
    AB_STAMP(5 + 8 * qb);
    __syncthreads();                                        // (Y) dS planes of this block complete; nobody reads the Q / dO planes any more
    AB_STAMP(6 + 8 * qb);
    // next block's Q / dO / O rows: the global loads fly under this wave's dQ tile (the registers they occupy are free in this phase only)
    if (qb + 1 < nqb) stage_load(qb + 1);
    // ---- dQ tile of this wave (16 q x 16 ch), reduced over ALL keys: A = dS, B = K, both by transposed reads.  A chain of up to
    // seven 6-MFMA steps, each behind an LDS round trip: the fragments of key block kb + 1 are requested before block kb multiplies ----
    if (has_tile) {
      const int offa_q = 8 * tr_p, offb_c = (((2 * tc + (tr_p >> 1)) ^ tr_sw) << 4) + ((tr_p & 1) << 3);
      att_bf16x8 af[2][3], bk[2][3];
      auto rd = [&](int kb, int slot) __attribute__((always_inline)) {
        const int k0 = 32 * kb + tr_row;                    // second read: + 16 keys (same half-swap bit: 16 is a multiple of 8)
        const int offa = k0 * 64 + 32 * (tq ^ ((k0 >> 2) & 1)) + offa_q, offb = k0 * 128 + offb_c;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          af[slot][pl] = ab_tr2(dSpl + pl * AB_DSPL + offa, dSpl + pl * AB_DSPL + offa + 16 * 64);
          bk[slot][pl] = ab_tr2(Kpl + pl * AB_KPL + offb, Kpl + pl * AB_KPL + offb + 16 * 128);
        }
      };
      f32x4 acc = zero4(), acc1 = zero4();                  // even / odd key blocks on two chains, added once at the end (fixed order)
      rd(0, 0);
#pragma unroll
      for (int kb = 0; kb < AB_NKW; ++kb) {
        if (kb < nqb) {
          __builtin_amdgcn_sched_barrier(0);
          if (kb + 1 < nqb) rd(kb + 1, (kb + 1) & 1);
          if (kb & 1) acc1 = att_mfma6(af[kb & 1], bk[kb & 1], acc1);
          else acc = att_mfma6(af[kb & 1], bk[kb & 1], acc);
        }
      }
      acc += acc1;
      // acc[r] = dQ'[position 32 qb + 16 qt + 4 g + r][channel 16 ct + c]
      const int p0 = qb * AB_QB + 16 * tq + 4 * gl, ch = 16 * tc + cl;
      const int rlo = max(0, sft - p0), rhi = min(4, npos - p0);
      if (ch < dh && rlo < rhi) {
        acc *= scale;
        if (PF) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dqs[0] += (r >= rlo && r < rhi) ? acc[r] : 0.f;
          att_store_p_col4(dP_out, p_ncb, (int)tok0 - sft + p0, head * dh + ch, acc, rlo, rhi);
        } else {
          float* dqp = dqkv + (tok0 + p0) * ldq + head * dh + ch;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r >= rlo && r < rhi) dqp[(size_t)r * ldq] = acc[r];
        }
      }
    }
    AB_STAMP(7 + 8 * qb);
    __builtin_amdgcn_sched_barrier(0);
    if (qb + 1 < nqb) stage_store(qb + 1);
    AB_STAMP(8 + 8 * qb);
    __syncthreads();                                        // (X) next block's Q / dO planes visible; the dS planes may be rewritten
    AB_STAMP(9 + 8 * qb);
  }

  // ---- dK, dV: acc[kt][ct][r] = d[position 32 w + 16 kt + 4 g + r][channel 16 ct + c] ----
  float* cs = reinterpret_cast<float*>(stQ);               // [7 key waves][2][64] column sums of the waves' dK / dV rows, then [8 tiles][16] of dQ
  float* qs = cs + AB_NKW * 2 * 64;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int ch = 16 * ct + c;
    float sk = 0.f, sv = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int p0 = 32 * w + 16 * kt + 4 * g;
      const int rlo = max(0, sft - p0), rhi = min(4, npos - p0);
      f32x4 kq = dKa[kt][ct], vq = dVa[kt][ct];           // dK accumulated against the pre-scaled Q
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r < rlo || r >= rhi) { kq[r] = 0.f; vq[r] = 0.f; }
        sk += kq[r]; sv += vq[r];
      }
      if (ch < dh && rlo < rhi) {
        if (PF) {
          att_store_p_col4(dP_out, p_ncb, (int)tok0 - sft + p0, H * dh + head * dh + ch, kq, rlo, rhi);
          att_store_p_col4(dP_out, p_ncb, (int)tok0 - sft + p0, 2 * H * dh + head * dh + ch, vq, rlo, rhi);
        } else {
          float* dkp = dqkv + (tok0 + p0) * ldq + H * dh + head * dh + ch;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r >= rlo && r < rhi) { dkp[(size_t)r * ldq] = kq[r]; dkp[(size_t)r * ldq + H * dh] = vq[r]; }
        }
      }
    }
    if (PF) {
      sk += __shfl_xor(sk, 16, 64); sk += __shfl_xor(sk, 32, 64);
      sv += __shfl_xor(sv, 16, 64); sv += __shfl_xor(sv, 32, 64);
      if (g == 0 && w < AB_NKW) { cs[(w * 2 + 0) * 64 + ch] = sk; cs[(w * 2 + 1) * 64 + ch] = sv; }
    }
  }
  AB_STAMP(60);
  if (!PF) return;
  if (has_tile) {
    float v = dqs[0];
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (g == 0) qs[w * 16 + c] = v;
  }
  __syncthreads();
  if (t < 192) {
    const int part = t >> 6, ch = t & 63;
    float sum = 0.f;
    if (part == 0) {
      const int ct = ch >> 4;
      if (ct < nct) sum = qs[ct * 16 + (ch & 15)] + qs[(nct + ct) * 16 + (ch & 15)];
    } else {
#pragma unroll
      for (int ww = 0; ww < AB_NKW; ++ww) sum += cs[(ww * 2 + (part - 1)) * 64 + ch];
    }
    if (ch < dh) colpart[(size_t)b * (3 * H * dh) + part * (H * dh) + head * dh + ch] = sum;
  }
}

constexpr size_t BWD_LDS = AB_LDS_BYTES;
constexpr size_t BWD_LDS_P = AB_LDS_BYTES;

int check_shape(int B, int N, int H, int dh) {
  if (B <= 0 || N <= 0 || H <= 0 || dh <= 0) return OFB_EINVAL;
  if (N > ATT_NMAX || dh > ATT_DMAX || (dh & 3)) return OFB_ELIMIT;
  return OFB_OK;
}

}  // namespace

// qkv: [B*N][3*H*dh] packed as the qkv Linear writes it (q | k | v, each H*dh, head-major); out: [B*N][H*dh];
// lse: [2][B*H][N] (row log-sum-exp, then its rounding residue: see the forward kernel).  N <= 208, dh <= 64, dh % 4 == 0.
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
    hipLaunchKernelGGL(attn_bwd_kernel<true>, dim3(B * H), dim3(AB_THREADS), ldsb, s, qkv, out, lse, dout, dqkv, dP_out, p_ncb, colpart,
                       B, N, H, dh, scale);
  else
    hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3(B * H), dim3(AB_THREADS), ldsb, s, qkv, out, lse, dout, dqkv, dP_out, p_ncb, colpart,
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

#ifdef OFB_ATT_STAMPS
extern "C" int ofb_diag_att_stamps(unsigned long long* out_host) {      /* lab only, not part of the ABI */
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(ofb_att_stamps), sizeof(unsigned long long) * 8 * 64);
}
#endif
