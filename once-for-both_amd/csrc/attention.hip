// Fused gated-attention core for the OFB search step: softmax(q k^T * scale) v, forward and backward, on the f16 matrix pipe at
// fp32-class accuracy: every operand value is split into two f16 numbers of a power-of-two scaled copy (hformat.h: x 2^e = h1 + h2)
// and every product is three v_mfma_f32_16x16x32_f16 terms (h2 h1, h1 h2, h1 h1) with f32 accumulation - the GEMM's engine
// (csrc/gemm_h.hip).  The exponents come from device-side upper bounds: of |qkv| (the qkv GEMM's Cauchy-Schwarz bound, cbound_out),
// of |dout| (the dO GEMM's), constants for the probabilities (<= 1) and row-norm bounds for dS (backward, below).
// Forward: one workgroup of 13 waves per (batch, head): the sequence (N <= 208 tokens = 13 tiles of 16; DeiT: N = 197) is cut into
// 16-token tiles and wave w owns query tile w.  Backward: one workgroup of 8 waves per (batch, head), keys on the lanes.
// Longer sequences (N <= 4096: 384-px inputs, patch 8) run the same kernels chunked: the forward spreads the query tiles over
// ceil(N / 208) workgroups per (batch, head) (blockIdx.y), the backward is launched once per 224 keys (attn_bwd_kernel<true>).
// Probabilities never touch HBM.  Reference math: models/layers.py:510-514 (search branch) and its autograd.
//
// 16x16x32 operand conventions: lane l gives A[i = l&15][k = 8*(l>>4) + j] and B[k = 8*(l>>4) + j][col = l&15] (j = 0..7, one 16-byte
// fragment); D[i = 4*(l>>4) + r][col = l&15].  Two facts carry the design:
//  * any bijection reduction-index <-> (step, kslot) is valid as long as A and B use the same one;
//  * an accumulator tile X feeds the next MFMA directly as its A operand when that MFMA reduces over X's ROW index.
// Forward uses the S^T orientation (rows = keys, lane column = query): row-softmax is register-local plus two shuffles, and P^T feeds
// P.V directly.  Backward uses the S orientation (rows = queries, lane column = key): P and dS feed dV = P^T dO and dK = dS^T Q
// directly; only dS crosses LDS for dQ = dS K.
#include "hformat.h"

#define ATT_T 16            // tokens per tile
#define ATT_NT 13           // tiles per head -> N <= 208
#define ATT_NMAX (ATT_T * ATT_NT)
#define ATT_DMAX 64
#define ATT_LD 68           // LDS row pitch (floats) of K / V / Q / dO tiles
#define ATT_DSLD 212        // pitch of the transposed dS tile [16 q][208 keys + 4]
#define ATT_THREADS (64 * ATT_NT)

namespace {

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// ------------------------------------------------------------------------------------------------------------------
// forward: streaming (flash) form
// ------------------------------------------------------------------------------------------------------------------
// Wave w owns query tile w; keys stream through LDS in blocks of 32 (two 16-key tiles), double-buffered, each K / V value split ONCE
// per workgroup while it is staged, both as row-major planes [2][32 keys][64] f16 (128-B rows, 8-byte stores) with the 16-byte chunks
// of a row XOR-swizzled by the key so that the fragment reads are conflict free:
//   K planes: chunk ^ ((key >> 1) & 7).  A operand of S^T = K Q^T: fragment = 8 consecutive d of one key = one ds_read_b128
//   V planes: chunk ^ (((key >> 1) & 3) << 1).  A operand of O^T = V^T P^T: rows = channels, k-slot (g, j) = key 4g + j (j < 4) or
//             16 + 4g + j - 4 - the keys whose P^T values lane group g owns in its two S^T accumulators, so the probabilities feed
//             the MFMA from registers.  The fragment is two ds_read_b64_tr_b16: lane 4e + q of a 16-lane group addresses the four
//             channels 16 dt + 4q .. of key 4g + e, the hardware transpose hands lane i the keys 4g .. 4g + 3 of channel 16 dt + i.
// Scaling: q (pre-multiplied by `scale`), k, v are split as x 2^e with ONE e from the bound of |qkv|; S^T leaves its MFMA chain
// times 2^(2e) and is multiplied back (exact); the probabilities are split as p 2^14; O accumulates in units 2^(e + 14).
// Online softmax over the key blocks (running max m, running sum l, O rescaled when m grows); lse = m + log l.
typedef short att_s16x4 __attribute__((ext_vector_type(4)));
typedef short att_s16x8 __attribute__((ext_vector_type(8)));
typedef ofb_f16x8 att_hx8;

#ifndef OFB_ATT_STAGE_OFF_SIMD0
#define OFB_ATT_STAGE_OFF_SIMD0 1     /* forward: staging by the nine waves that do not sit on SIMD 0 (0: lab, all thirteen stage) */
#endif
#define AF_KB 32                       // keys per block
#define AF_NB ((ATT_NMAX + AF_KB - 1) / AF_KB)      // 7 blocks cover 224 >= 208 keys
#define AF_PITCH 128                   // bytes per key row of one plane: 64 f16
#define AF_PLANE (AF_KB * AF_PITCH)
#define AF_STAGE (4 * AF_PLANE)        // K planes then V planes
#define ATT_PE 14                      // probabilities are split as p 2^14
#define AF_OPATCH (16 * 68 * 4)         // a wave's output patch in LDS: [16 queries][64 channels + 4] f32 (the plane patch [32 ch][20] fits inside)

// 8 ALREADY SCALED floats -> two 8 x f16 fragments
__device__ __forceinline__ void att_split8(const float (&x)[8], att_hx8 (&out)[2]) {
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ofb_hsplit_pair(x[2 * i], x[2 * i + 1], h[i], l[i]);
  out[0] = __builtin_bit_cast(att_hx8, make_uint4(h[0], h[1], h[2], h[3]));
  out[1] = __builtin_bit_cast(att_hx8, make_uint4(l[0], l[1], l[2], l[3]));
}
__device__ __forceinline__ f32x4 att_mfma3(const att_hx8 (&a)[2], const att_hx8 (&b)[2], f32x4 c) {
  // (h2,h1) (h1,h2) (h1,h1): smallest terms first
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], c, 0, 0, 0);
  return c;
}
// the same three terms with the roles of the operands exchanged (a <-> b of att_mfma3): the backward forms S = Q K^T with A = Q where
// the forward forms S^T = K Q^T with A = K; issued in this order each element sees the same products in the same order and S comes
// out bit-identical to the forward's
__device__ __forceinline__ f32x4 att_mfma3_swapped(const att_hx8 (&a)[2], const att_hx8 (&b)[2], f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], c, 0, 0, 0);
  return c;
}

// H-format stores (hformat.h: granules of 4 rows x 16 columns, [plane][c % 16][r % 4] f16, 256 B, stored [rows/4][ncb]); P = planes base
__device__ __forceinline__ char* att_h_slot(char* P, int ncb, int row, int col) {
  return P + ((size_t)(row >> 2) * ncb + (col >> 4)) * OFB_HGRAN + (col & 15) * 8 + (row & 3) * 2;
}
// four consecutive rows row0.. (row0 % 4 == 0: one 8-byte slot per plane) of one column, ALREADY SCALED; only rows [rlo, rhi) belong
// to the caller
__device__ __forceinline__ void att_store_h_col4(char* P, int ncb, int row0, int col, f32x4 v, int rlo, int rhi) {
  unsigned h0, l0, h1, l1;
  ofb_hsplit_pair(v[0], v[1], h0, l0);
  ofb_hsplit_pair(v[2], v[3], h1, l1);
  char* q = att_h_slot(P, ncb, row0, col);
  if (rlo == 0 && rhi == 4) {
    *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(q + 128) = make_uint2(l0, l1);
    return;
  }
  const unsigned hs[2] = {h0, h1}, ls[2] = {l0, l1};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r >= rlo && r < rhi) {                             // rows of a neighbouring image in this slot are its workgroup's to write
      const int sh = (r & 1) * 16;
      *reinterpret_cast<unsigned short*>(q + 2 * r) = (unsigned short)(hs[r >> 1] >> sh);
      *reinterpret_cast<unsigned short*>(q + 2 * r + 128) = (unsigned short)(ls[r >> 1] >> sh);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// forward kernel
// ------------------------------------------------------------------------------------------------------------------
// PF: the output rows also leave as H-format planes oP (the operand of the projection GEMM; |out| <= max|v|, so the planes share the
// qkv exponent).  The accumulators hold four CHANNELS of one query per lane, a plane slot is four QUERIES of one channel: every wave
// transposes its 16 x 64 tile through a private LDS patch (the stage buffers are free by then), 32 channels at a time.  The query
// tiles are laid from position -(b N % 4), so that every four tile rows are one 4-row granule of the [B N] matrix.
template <bool PF>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                               float* __restrict__ lse, char* __restrict__ oP, int p_ncb, int B, int N,
                                                               int H, int dh, float scale, const float* __restrict__ qkv_bound) {
  __shared__ __attribute__((aligned(16))) char smem[(2 * AF_STAGE > ATT_NT * AF_OPATCH) ? 2 * AF_STAGE : ATT_NT * AF_OPATCH];   // stages; later the waves' output patches
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / H, head = blockIdx.x % H;
  const int wq = blockIdx.y * ATT_NT + w;                    // this wave's query tile (blockIdx.y > 0 only when N > 208)
  const int ldq = 3 * H * dh, ldo = H * dh;
  const float* qbase = qkv + (size_t)b * N * ldq + head * dh;
  const float* kbase = qbase + H * dh;
  const float* vbase = qbase + 2 * H * dh;
  const int nb = (N + AF_KB - 1) / AF_KB;
  const int sft = PF ? (int)(((size_t)b * N) & 3) : 0;       // query tile position p <-> token p - sft (keys are not shifted)
  const float qb = qkv_bound[0];
  const int he = ofb_h_exp(qb);                              // q, k, v are split as x 2^he
  const float hs = ofb_h_pow2(he), s_inv = ofb_h_pow2(-2 * he);
  if (PF && blockIdx.x == 0 && blockIdx.y == 0 && t == 0) { ofb_hhdr* h = reinterpret_cast<ofb_hhdr*>(oP); h->e = he; h->amax = qb; h->rn2sq = 0.f; h->cn2sq = 0.f; }

  // staging items: idx < 512 -> K float4 (key = idx/16, d = 4*(idx%16)); 512 <= idx < 1024 -> V float4, same coordinates
  // Who stages: the thirteen waves sit round-robin on the four SIMDs, so SIMD 0 hosts FOUR of them (0, 4, 8, 12) and paces every key block
  // (profiles/r06_attention_forward_stamps.txt: everybody waits at the block barrier for wave 12).  The staging - global loads, f16
  // split, LDS stores - is therefore done by the NINE waves of the other SIMDs, two items each; SIMD 0's waves only multiply.
#if OFB_ATT_STAGE_OFF_SIMD0
  const bool stager = (w & 3) != 0;
  const int sid = stager ? ((w >> 2) * 3 + (w & 3) - 1) * 64 + lane : 1 << 20;      // 0 .. 575 (else: no items)
  constexpr int SSTRIDE = 576;
#else
  const int sid = t;
  constexpr int SSTRIDE = ATT_THREADS;
#endif
  f32x4 sreg[3];
  auto stage_load = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = sid + SSTRIDE * i;
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
      const int idx = sid + SSTRIDE * i;
      if (idx >= 1024) continue;
      const int kk = (idx & 511) >> 4, d16 = idx & 15, chunk = d16 >> 1;
      unsigned h0, l0, h1, l1;
      ofb_hsplit_pair(sreg[i][0] * hs, sreg[i][1] * hs, h0, l0);
      ofb_hsplit_pair(sreg[i][2] * hs, sreg[i][3] * hs, h1, l1);
      const bool isk = idx < 512;
      const int sw = isk ? ((kk >> 1) & 7) : (((kk >> 1) & 3) << 1);
      char* p = st + (isk ? 0 : 2 * AF_PLANE) + kk * AF_PITCH + ((chunk ^ sw) << 4) + ((d16 & 1) << 3);
      *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(p + AF_PLANE) = make_uint2(l0, l1);
    }
  };
  stage_load(0);
  stage_store(0);

  // this lane's query row (pre-scaled), d-slices [8g, 8g+8) and [32 + 8g, 32 + 8g + 8), as B-operand planes
  const int q = wq * ATT_T + c - sft;
  const bool qvalid = q >= 0 && q < N;
  att_hx8 qf[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int d0 = 32 * ks + 8 * g + 4 * u;
      f32x4 v = zero4();
      if (qvalid && d0 < dh) v = *reinterpret_cast<const f32x4*>(qbase + (size_t)q * ldq + d0);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[4 * u + j] = (v[j] * scale) * hs;
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
    for (int dt = 0; dt < 4; ++dt) v_off[dt] = 2 * AF_PLANE + vkey * AF_PITCH + (((2 * dt + (q >> 1)) ^ sw) << 4) + ((q & 1) << 3);
  }
  __syncthreads();

  const bool active = wq * ATT_T < N + sft;
  const float p_sc = ofb_h_pow2(ATT_PE);
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
          att_hx8 kf[2];
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
            kf[pl] = *reinterpret_cast<const att_hx8*>(st + pl * AF_PLANE + 16 * tk * AF_PITCH + k_off[ks]);
          acc = att_mfma3(kf, qf[ks], acc);
        }
        S[tk] = acc * s_inv;                                 // exact (power of two)
      }
      // online softmax: this lane holds keys kb*32 + 16 tk + 4g + r of query c
      float mb = -INFINITY;
      if ((kb + 1) * AF_KB <= N) {                            // (only the last key block has positions past the end: no masks elsewhere)
#pragma unroll
        for (int tk = 0; tk < 2; ++tk)
#pragma unroll
          for (int r = 0; r < 4; ++r) mb = fmaxf(mb, S[tk][r]);
      } else {
#pragma unroll
        for (int tk = 0; tk < 2; ++tk)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sv = (kb * AF_KB + 16 * tk + 4 * g + r < N) ? S[tk][r] : -INFINITY;
            S[tk][r] = sv;
            mb = fmaxf(mb, sv);
          }
      }
      OFB_XOR_STEP(mb, fmaxf, 16)                         // (register-only exchanges: ofb_common.h)
      OFB_XOR_STEP(mb, fmaxf, 32)
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
          p8[4 * tk + r] = pv * p_sc;                        // k-slot j = 4 tk + r of group g
          ls += pv;
        }
      l_run = l_run * alpha + ls;
      att_hx8 pf[2];
      att_split8(p8, pf);
      // O^T[ch][query] = alpha * O^T + V^T P^T   (units 2^(he + ATT_PE))
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        att_hx8 vf[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const char* vp = st + pl * AF_PLANE + v_off[dt];
          const att_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)(vp));
          const att_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)(vp + 16 * AF_PITCH));
          const att_s16x8 v8 = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
          vf[pl] = __builtin_bit_cast(att_hx8, v8);
        }
        O[dt] *= alpha;
        O[dt] = att_mfma3(vf, pf, O[dt]);
      }
    }
    if (kb + 1 < nb) stage_store((kb + 1) & 1);
    __syncthreads();
  }
  if (!active) return;
  float l = l_run;
  OFB_XOR_STEP(l, ofb_add_, 16)
  OFB_XOR_STEP(l, ofb_add_, 32)
  const float linv = (1.0f / l) * ofb_h_pow2(-(he + ATT_PE));
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) O[dt] *= linv;
  if (qvalid) {
    if (g == 0) {
      // lse as TWO floats: [0][b h][q] = fl(m + log l) and, B H N floats further, its rounding residue (m - lse) + log l: the backward
      // forms P = exp((S - lse) - residue) = exp(S - m) / l without the half-ulp of |lse| a single float would put on the whole row
      const float ll = logf(l), ls = m_run + ll;
      const size_t li = ((size_t)b * H + head) * N + q;
      OFB_NT_STORE(ls, lse + li);                                                  // (lse and the f32 rows are read by the backward only)
      OFB_NT_STORE((m_run - ls) + ll, lse + (size_t)B * H * N + li);
    }
  }
  {
    // The f32 rows.  O[dt][r] = O^T[channel 16 dt + 4g + r][query c]: stored from these registers one instruction writes 16 B for each of
    // the four lanes that share a query - sixteen 64-byte segments in sixteen rows, and the stamps showed the waves' store issue taking up to
    // 8 k cycles (13 waves x 64 segments on one CU).  Through a wave-private LDS patch [16 q][64 ch + 4] (the stage buffers are free) sixteen
    // consecutive lanes hold one query's 256 contiguous bytes: four rows of 256 B per instruction.
    float* Tq = reinterpret_cast<float*>(smem + w * AF_OPATCH);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(&Tq[c * 68 + 16 * dt + 4 * g]) = O[dt];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int ql = 4 * pass + (lane >> 4), ck = lane & 15, qq = wq * ATT_T + ql - sft;
      const f32x4 v = *reinterpret_cast<const f32x4*>(&Tq[ql * 68 + 4 * ck]);
      if (qq >= 0 && qq < N && 4 * ck < dh) OFB_NT_STORE(v, reinterpret_cast<f32x4*>(out + ((size_t)b * N + qq) * ldo + head * dh + 4 * ck));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (PF) {
    // wave-private patch [32 ch][16 q + 4] f32 (the same LDS as the row patch above): 2.5 KB per wave
    float* T = reinterpret_cast<float*>(smem + w * AF_OPATCH);
    const int row0 = (int)((size_t)b * N) - sft + wq * ATT_T;                            // global row of tile position 0: row0 % 4 == 0
    char* oPl = oP + OFB_HHDR;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * dd + 4 * g + r) * 20 + c] = qvalid ? O[2 * half + dd][r] * hs : 0.f;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int chl = lane & 31, qg = 2 * pass + (lane >> 5), ch = 32 * half + chl, p0 = wq * ATT_T + 4 * qg;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&T[chl * 20 + 4 * qg]);
        const int rlo = max(0, sft - p0), rhi = min(4, N + sft - p0);
        if (ch < dh && rlo < rhi) att_store_h_col4(oPl, p_ncb, row0 + 4 * qg, head * dh + ch, v, rlo, rhi);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// backward: one kernel, dq | dk | dv, P recomputed from lse; every product is the two-plane f16 split with three
// v_mfma_f32_16x16x32_f16 terms (fp32-class accuracy, like the forward and the GEMM).
// ------------------------------------------------------------------------------------------------------------------
// One workgroup of EIGHT waves per (batch, head); wave w < 7 owns the 32 key positions 32w .. 32w+31 (224 positions cover N <= 208
// tokens; wave 7 owns none: it helps staging and takes a dQ tile) and keeps dK / dV of its keys in 64 accumulator registers while
// the workgroup sweeps the queries in blocks of 32.  "Key on the lane": S = Q K^T and dP = dO V^T are computed with rows = queries,
// lane column = key, so the P and dS accumulators ARE the A operands (k-slot (g, j) <-> query 16 (j >> 2) + 4 g + (j & 3)) of
// dV = P^T dO and dK = dS^T Q; the other operand of those two products comes out of the row-major Q / dO planes by
// ds_read_b64_tr_b16.  -lse[q] and -delta[q] are the initial accumulator of dP (dS = P * acc).  Q is staged pre-scaled exactly as the
// forward scales it (S' = (scale Q) K^T, dK = dS'^T (scale Q)) and the three terms of S' are issued in the forward's order, so S' is
// bit-identical to the forward's and P = exp((S' - lse) - residue) reproduces the forward's normalised probabilities.
// Only dS crosses LDS: every wave stores its dS block as planes [key][32 q] (8-byte stores: the two halves of its A fragment), and
// after one barrier each (16-query, 16-channel) tile of dQ = dS K is reduced over ALL keys by ONE wave (A = dS by transposed reads,
// B = K by transposed reads of the K planes that sit in LDS for the whole kernel): no partial dQ tiles, fixed summation order.
// Exponents: q, k, v share e(qkv bound); dO has e(dout bound); P is split as p 2^14; dS = P (dP - delta) gets ONE exponent per query
// block from |dP - delta| <= 2 max_q |dO_q|_2 max_key |V_key|_2 (Cauchy-Schwarz; the row norms are formed while the rows are staged,
// every wave derives the same number from the same LDS words), and dK, which accumulates across the blocks, is re-scaled by the
// exact power of two whenever that exponent moves.
// LDS (105 KB): K planes [2][224][64] f16 56 KB | dS planes [2][224 keys][32 q] 28 KB | Q and dO planes of the current block
// [2][32][64] 8 KB each | -lse, -delta, residue, |dO_q|_2 per position, per-wave max |V_key|_2^2.  All 128-byte-row images carry the
// chunk swizzle chunk ^ (((row >> 1) & 3) << 1): conflict-free for the ds_read_b128 row reads AND for the transposed reads; the
// 64-byte dS rows swap their two 32-byte halves with bit 2 of the key.
// Output: f32 dq | dk | dv in the qkv packing; dqkv_amax (optional) receives max |output| by atomic max (the exponent of the H-format
// copy ofb_to_hformat_colsum makes of it).
#ifndef OFB_ATT_KB_HOLD
#define OFB_ATT_KB_HOLD 0             /* 1: K fragments held across the two query tiles - built, bit-identical, NEUTRAL (round 6: profiles/r06_attention_kb_hold_neutral.txt) */
#endif
#define AB_NW 8                       /* waves: 0..6 own 32 key positions each, wave 7 only stages and takes a dQ tile */
#define AB_NKW 7
#define AB_THREADS (64 * AB_NW)
#define AB_QB 32
#define AB_NPOS (32 * AB_NKW)
#define AB_KPL (AB_NPOS * 128)
#define AB_DSPL (AB_NPOS * 64)
#define AB_STPL (AB_QB * 128)
#define AB_OFF_DS (2 * AB_KPL)
#define AB_OFF_STQ (AB_OFF_DS + 2 * AB_DSPL)
#define AB_OFF_STO (AB_OFF_STQ + 2 * AB_STPL)
#define AB_OFF_NLSE (AB_OFF_STO + 2 * AB_STPL)
#define AB_OFF_NDEL (AB_OFF_NLSE + AB_NPOS * 4)
#define AB_OFF_NRES (AB_OFF_NDEL + AB_NPOS * 4)
#define AB_OFF_NDO (AB_OFF_NRES + AB_NPOS * 4)
#define AB_OFF_VMAX (AB_OFF_NDO + AB_NPOS * 4)
#define AB_LDS_BYTES (AB_OFF_VMAX + 64)

__device__ __forceinline__ int ab_swz(int row) { return ((row >> 1) & 3) << 1; }
__device__ __forceinline__ att_hx8 ab_tr2(const char* p0, const char* p1) {
  const att_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)(p0));
  const att_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)(p1));
  const att_s16x8 v8 = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  return __builtin_bit_cast(att_hx8, v8);
}

#ifdef OFB_ATT_STAMPS
// lab only (scripts/lab/stamp_att.py): s_memtime stamps of every wave of workgroup 0: [wave][64 slots]
__device__ unsigned long long ofb_att_stamps[8 * 64];
#define AB_STAMP(slot) do { if (blockIdx.x == gridDim.x - 100 && lane == 0 && (slot) < 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ofb_att_stamps[w * 64 + (slot)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define AB_STAMP(slot) do { } while (0)
#endif

// LONG (N > 208): one launch per chunk of 224 keys (kc = chunk index, stream-ordered): the launch owns dK / dV of its keys, sweeps
// ALL queries and adds its share of dQ to what the earlier launches left (plain read-modify-write: a fixed summation order, no
// atomics); the per-position LDS arrays become a ring of seven query blocks that is filled block by block.
template <bool LONG>
__global__ __launch_bounds__(AB_THREADS, 2) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                                 const float* __restrict__ lse, const float* __restrict__ dout,
                                                                 float* __restrict__ dqkv, int B, int N, int H, int dh, float scale,
                                                                 const float* __restrict__ qkv_bound, const float* __restrict__ dout_bound,
                                                                 float* __restrict__ dqkv_amax, int kc, int amax_per_wg) {
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
  float* ndo = reinterpret_cast<float*>(smem + AB_OFF_NDO);
  float* vmax = reinterpret_cast<float*>(smem + AB_OFF_VMAX);
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / H, head = blockIdx.x % H;
  const int ldq = 3 * H * dh, ldo = H * dh;
  const size_t tok0 = (size_t)b * N;
  const int npos = N, nqb = (npos + AB_QB - 1) / AB_QB, nct = (dh + 15) >> 4, nks = (dh + 31) >> 5;
  const float* qbase = qkv + tok0 * ldq + head * dh;
  const float* kbase = qbase + H * dh;
  const float* vbase = qbase + 2 * H * dh;
  const float* obase = out + tok0 * ldo + head * dh;
  const float* dobase = dout + tok0 * ldo + head * dh;
  const int key0 = LONG ? kc * AB_NPOS : 0;                 // first key position of this launch
  const int nkb = LONG ? min(AB_NKW, (npos - key0 + 31) >> 5) : nqb;      // 32-key blocks that hold keys
  auto valid = [&](int pos) { return pos < npos; };
  auto ring = [&](int qb) { return LONG ? (qb % AB_NKW) * AB_QB : qb * AB_QB; };   // a query block's slot in the per-position arrays
  const int he = ofb_h_exp(qkv_bound[0]), hdo = ofb_h_exp(dout_bound[0]);      // q k v split as x 2^he, dO as x 2^hdo
  const float hs = ofb_h_pow2(he), hos = ofb_h_pow2(hdo), s_inv = ofb_h_pow2(-2 * he), dp_sc = ofb_h_pow2(hdo + he);

  // staging of a 32-position query block: thread t carries the Q, dO and O float4 of (row t / 16, d = 4 (t % 16))
  f32x4 sreg[3];
  float lq_a = 0.f, lq_b = 0.f;                             // LONG: lse and its residue of the row this thread's 16-lane group stages
  auto stage_load = [&](int qb) {
    if (LONG) {
      const int pos = qb * AB_QB + (t >> 4);
      lq_a = lq_b = 0.f;
      if ((t & 15) == 0 && valid(pos)) {
        const size_t li = ((size_t)b * H + head) * N + pos;
        lq_a = lse[li];
        lq_b = lse[(size_t)B * H * N + li];
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {                           // i = 0: Q, 1: dO, 2: O (512 threads = 512 items each)
      const float* base = i == 0 ? qbase : (i == 1 ? dobase : obase);
      const int ld = i == 0 ? ldq : ldo;
      const int pos = qb * AB_QB + (t >> 4), d4 = (t & 15) << 2;
      // every row is read once.  An in-bounds address for every lane and no branch around the load (positions / channels past the end
      // re-read the head's first row; stage_store drops them): behind `if (valid) load` hipcc waited for the first block's request
      // right where it was issued - a memory round trip per problem that the dQ phase was meant to hide (round 6)
      const bool in = valid(pos) && d4 < dh;
      sreg[i] = OFB_NT_LOAD(reinterpret_cast<const f32x4*>(base + (in ? (unsigned)(pos * ld + d4) : 0u)));
    }
  };
  auto stage_store = [&](int qb) {
    if (!(valid(qb * AB_QB + (t >> 4)) && ((t & 15) << 2) < dh)) {
#pragma unroll
      for (int i = 0; i < 3; ++i) sreg[i] = zero4();
    }
    {                                                       // -delta of the row and |dO_row|_2: 16 lanes hold its 64 channels
      float d = sreg[1][0] * sreg[2][0] + sreg[1][1] * sreg[2][1] + sreg[1][2] * sreg[2][2] + sreg[1][3] * sreg[2][3];
      float n2 = sreg[1][0] * sreg[1][0] + sreg[1][1] * sreg[1][1] + sreg[1][2] * sreg[1][2] + sreg[1][3] * sreg[1][3];
      OFB_XOR_STEP(d, ofb_add_, 8) OFB_XOR_STEP(d, ofb_add_, 4) OFB_XOR_STEP(d, ofb_add_, 2) OFB_XOR_STEP(d, ofb_add_, 1)
      OFB_XOR_STEP(n2, ofb_add_, 8) OFB_XOR_STEP(n2, ofb_add_, 4) OFB_XOR_STEP(n2, ofb_add_, 2) OFB_XOR_STEP(n2, ofb_add_, 1)
      if ((t & 15) == 0) {
        const int ri = ring(qb) + (t >> 4);
        ndel[ri] = -d; ndo[ri] = sqrtf(n2);
        if (LONG) { nlse[ri] = -lq_a; nres[ri] = -lq_b; }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = t >> 4, c4 = t & 15;
      unsigned h0, l0, h1, l1;
      // Q is staged pre-scaled exactly as the forward scales it: S' = (scale Q) K^T and dK = dS'^T (scale Q)
      if (i == 0) {
        ofb_hsplit_pair((sreg[0][0] * scale) * hs, (sreg[0][1] * scale) * hs, h0, l0);
        ofb_hsplit_pair((sreg[0][2] * scale) * hs, (sreg[0][3] * scale) * hs, h1, l1);
      } else {
        ofb_hsplit_pair(sreg[1][0] * hos, sreg[1][1] * hos, h0, l0);
        ofb_hsplit_pair(sreg[1][2] * hos, sreg[1][3] * hos, h1, l1);
      }
      char* p = ((i == 0) ? stQ : stO) + row * 128 + (((c4 >> 1) ^ ab_swz(row)) << 4) + ((c4 & 1) << 3);
      *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(p + AB_STPL) = make_uint2(l0, l1);
    }
  };

  // EVERY global load of the prologue is issued before the first one is consumed (one memory round trip instead of four: the
  // workgroup is alone on its CU, nothing else hides them): K rows, lse, query block 0, this wave's V rows
  constexpr int NIT = AB_NPOS * 16 / AB_THREADS;
  static_assert(NIT * AB_THREADS == AB_NPOS * 16 && AB_NPOS <= AB_THREADS, "prologue items");
  const bool has_keys = w < AB_NKW && key0 + 32 * w < npos;
  f32x4 kv[NIT], vraw[2][2][2];
  float lse_a = 0.f, lse_b = 0.f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = t + AB_THREADS * i, pos = idx >> 4, c4 = idx & 15;
    // (in-bounds addresses and no branches around the prologue's loads either; what lies past the end is dropped where it is used)
    const bool in = valid(key0 + pos) && 4 * c4 < dh;
    kv[i] = OFB_NT_LOAD(reinterpret_cast<const f32x4*>(kbase + (in ? (unsigned)((key0 + pos) * ldq + 4 * c4) : 0u)));
  }
  if (!LONG && t < AB_NPOS && valid(t)) {
    const size_t li = ((size_t)b * H + head) * N + t;
    lse_a = lse[li];
    lse_b = lse[(size_t)B * H * N + li];                    // the forward's second float of lse (rounding residue)
  }
  stage_load(0);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int pos = key0 + 32 * w + 16 * kt + c;
    const bool kv_ok = valid(pos);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int d0 = 32 * ks + 8 * g + 4 * u;
        const bool in = kv_ok && d0 < dh;
        vraw[kt][ks][u] = OFB_NT_LOAD(reinterpret_cast<const f32x4*>(vbase + (in ? (unsigned)(pos * ldq + d0) : 0u)));
      }
  }
  AB_STAMP(58);
  // K planes
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = t + AB_THREADS * i, pos = idx >> 4, c4 = idx & 15;
    unsigned h0, l0, h1, l1;
    const f32x4 kk = (valid(key0 + pos) && 4 * c4 < dh) ? kv[i] : zero4();
    ofb_hsplit_pair(kk[0] * hs, kk[1] * hs, h0, l0);
    ofb_hsplit_pair(kk[2] * hs, kk[3] * hs, h1, l1);
    char* p = Kpl + pos * 128 + ((((c4 >> 1) ^ ab_swz(pos))) << 4) + ((c4 & 1) << 3);
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + AB_KPL) = make_uint2(l0, l1);
  }
  AB_STAMP(59);
  if (!LONG && t < AB_NPOS) { nlse[t] = -lse_a; nres[t] = -lse_b; }
  stage_store(0);
  AB_STAMP(61);
  // this wave's V rows as B operand of dP = dO V^T (lane: key 32 w + 16 kt + c, d = 32 ks + 8 g ..); the B operand of S' = (scale Q) K^T
  // is read from the K planes every block (row reads: the registers are needed for the accumulators).  max_key |V_key|_2^2 of the wave's
  // keys goes to LDS (one word per wave): the dS bound of every block needs the head's maximum
  att_hx8 Vb[2][2][2];
  {
    float vn = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      float n2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float xv[8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x = (valid(key0 + 32 * w + 16 * kt + c) && 32 * ks + 8 * g + 4 * u < dh) ? vraw[kt][ks][u][j] : 0.f;
            n2 += x * x; xv[4 * u + j] = x * hs;
          }
        att_split8(xv, Vb[kt][ks]);
      }
      OFB_XOR_STEP(n2, ofb_add_, 16) OFB_XOR_STEP(n2, ofb_add_, 32)     // the key's 64 channels sit in the four lane groups
      vn = fmaxf(vn, n2);
    }
    OFB_XOR_STEP(vn, fmaxf, 1) OFB_XOR_STEP(vn, fmaxf, 2)
    OFB_XOR_STEP(vn, fmaxf, 4) OFB_XOR_STEP(vn, fmaxf, 8)
    if (lane == 0) vmax[w] = vn;
  }
  AB_STAMP(62);
  bool kval[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) kval[kt] = valid(key0 + 32 * w + 16 * kt + c);

  f32x4 dKa[2][4], dVa[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { dKa[kt][ct] = zero4(); dVa[kt][ct] = zero4(); }

  // dQ tiles of a block: tile i = (qt = i / nct, ct = i % nct), owned by wave i (eight tiles at dh = 64: one per wave, two per SIMD)
  const int ntile = 2 * nct;
  float omax = 0.f;                                        // max |output| this thread stores

  __syncthreads();
  AB_STAMP(1);
  float vnorm;
  {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < AB_NW; ++i) m = fmaxf(m, vmax[i]);
    // the dS bound needs max |V_key|_2 over ALL keys of the head (|delta| = |dO . O|, and O mixes every key): a chunk sees 224 of
    // them, so LONG takes the bound sqrt(dh) max|qkv| instead
    vnorm = LONG ? sqrtf((float)dh) * qkv_bound[0] : sqrtf(m);
  }

  const bool has_tile = w < ntile;                        // this wave's dQ tile: (qt, ct) = (w / nct, w % nct)
  const int tq = has_tile ? w / nct : 0, tc = has_tile ? w - tq * nct : 0;
  int eds_prev = 0;
  for (int qb = 0; qb < nqb; ++qb) {
    AB_STAMP(2 + 8 * qb);
    // fragment addresses (bytes inside a plane), re-derived every block from "laundered" lane ids: left to itself hipcc hoists the
    // ~40 block-invariant LDS addresses of the body out of the loop and then spills them around it
    int cl = c, gl = g;
    asm volatile("" : "+v"(cl), "+v"(gl));
    const int a_row = cl * 128, a_sw = ab_swz(cl);          // row reads: row 16 qt + c (same swizzle for both qt: 16 is a multiple of 8)
    const int tr_row = 4 * gl + (cl >> 2), tr_sw = ab_swz(tr_row), tr_p = cl & 3;   // transposed reads: row (+16 for the second read)
    // exponent of this block's dS: |P (dP - delta)| <= 2 max_q |dO_q|_2 max_key |V_key|_2 (every wave reads the same LDS words)
    int eds;
    {
      float dm = ndo[ring(qb) + (lane & 31)];
      OFB_XOR_STEP(dm, fmaxf, 1) OFB_XOR_STEP(dm, fmaxf, 2) OFB_XOR_STEP(dm, fmaxf, 4)
      OFB_XOR_STEP(dm, fmaxf, 8) OFB_XOR_STEP(dm, fmaxf, 16)
      eds = __builtin_amdgcn_readfirstlane(ofb_h_exp(2.002f * dm * vnorm));
    }
    if (has_keys) {
      // ---- S' and dP - delta for the block's 32 queries x this wave's 32 keys ----
      f32x4 S[2][2], dPa[2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const f32x4 nd = *reinterpret_cast<const f32x4*>(&ndel[ring(qb) + 16 * qt + 4 * gl]) * dp_sc;   // dP accumulates in units 2^(hdo + he)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) { S[qt][kt] = zero4(); dPa[qt][kt] = nd; }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks < nks) {
          const int off = a_row + ((((4 * ks + gl)) ^ a_sw) << 4);
#if OFB_ATT_KB_HOLD
          // lab form: the K fragments of this wave's two key tiles are read ONCE per k-step and held across both query
          // tiles - half the K-fragment LDS reads of this phase, +10 registers (231), same MFMAs in the same order (bit-identical);
          // measured 130.2 / 129.8 / 131.7 us against 130.7 / 130.4 / 130.7: the phase is not bound by those reads
          __builtin_amdgcn_sched_barrier(0);
          att_hx8 kbh[2][2];
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) kbh[kt][pl] = *reinterpret_cast<const att_hx8*>(Kpl + pl * AB_KPL + (32 * w + 16 * kt) * 128 + off);
#endif
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            // (the fences keep hipcc from hoisting every group's fragment reads to the top of the block)
            __builtin_amdgcn_sched_barrier(0);
            att_hx8 qa[2], oa[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
              qa[pl] = *reinterpret_cast<const att_hx8*>(stQ + pl * AB_STPL + 16 * qt * 128 + off);
              oa[pl] = *reinterpret_cast<const att_hx8*>(stO + pl * AB_STPL + 16 * qt * 128 + off);
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#if OFB_ATT_KB_HOLD
              S[qt][kt] = att_mfma3_swapped(qa, kbh[kt], S[qt][kt]);         // bit-identical to the forward's S (times 2^(2 he))
#else
              att_hx8 kb[2];
#pragma unroll
              for (int pl = 0; pl < 2; ++pl) kb[pl] = *reinterpret_cast<const att_hx8*>(Kpl + pl * AB_KPL + (32 * w + 16 * kt) * 128 + off);
              S[qt][kt] = att_mfma3_swapped(qa, kb, S[qt][kt]);              // bit-identical to the forward's S (times 2^(2 he))
#endif
              dPa[qt][kt] = att_mfma3(oa, Vb[kt][ks], dPa[qt][kt]);
            }
          }
        }
      }
      AB_STAMP(3 + 8 * qb);
      // ---- P = exp((S' - lse) - residue), dS' = P (dP - delta); as A fragments (k-slot (g, 4 qt + r) = query 16 qt + 4 g + r) ----
      att_hx8 pf[2][2], dsf[2][2];
      const float p_sc = ofb_h_pow2(ATT_PE), ds_sc = ofb_h_pow2(eds - hdo - he);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        float p8[8], d8[8];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const f32x4 nl = *reinterpret_cast<const f32x4*>(&nlse[ring(qb) + 16 * qt + 4 * gl]);
          const f32x4 nr = *reinterpret_cast<const f32x4*>(&nres[ring(qb) + 16 * qt + 4 * gl]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = kval[kt] ? __expf((S[qt][kt][r] * s_inv + nl[r]) + nr[r]) : 0.f;
            p8[4 * qt + r] = pv * p_sc;
            d8[4 * qt + r] = (pv * dPa[qt][kt][r]) * ds_sc;
          }
        }
        att_split8(p8, pf[kt]);
        att_split8(d8, dsf[kt]);
        // dS planes [key][32 q]: the fragment's two halves are the 8-byte (4 q) slots of q tile 0 / 1
        const int key = 32 * w + 16 * kt + cl, hsw = (key >> 2) & 1;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const uint4 v = __builtin_bit_cast(uint4, dsf[kt][pl]);
          char* p = dSpl + pl * AB_DSPL + key * 64 + 8 * gl;
          *reinterpret_cast<uint2*>(p + 32 * (0 ^ hsw)) = make_uint2(v.x, v.y);
          *reinterpret_cast<uint2*>(p + 32 * (1 ^ hsw)) = make_uint2(v.z, v.w);
        }
      }
      AB_STAMP(4 + 8 * qb);
      // dK accumulates dS'^T Q across the blocks in units 2^(eds + he): move it to this block's exponent (exact)
      if (qb > 0 && eds != eds_prev) {
        const float f = ofb_h_pow2(eds - eds_prev);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) dKa[kt][ct] *= f;
      }
      // ---- dV += P^T dO, dK += dS'^T Q: B = dO / Q [query slots][channel] by transposed reads ----
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        if (ct < nct) {
          __builtin_amdgcn_sched_barrier(0);
          att_hx8 bo[2], bq[2];
          const int off = tr_row * 128 + (((2 * ct + (tr_p >> 1)) ^ tr_sw) << 4) + ((tr_p & 1) << 3);
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            bo[pl] = ab_tr2(stO + pl * AB_STPL + off, stO + pl * AB_STPL + off + 16 * 128);
            bq[pl] = ab_tr2(stQ + pl * AB_STPL + off, stQ + pl * AB_STPL + off + 16 * 128);
          }
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            dVa[kt][ct] = att_mfma3(pf[kt], bo, dVa[kt][ct]);
            dKa[kt][ct] = att_mfma3(dsf[kt], bq, dKa[kt][ct]);
          }
        }
      }
    }
    eds_prev = eds;
    AB_STAMP(5 + 8 * qb);
    __syncthreads();                                        // (Y) dS planes of this block complete; nobody reads the Q / dO planes any more
    AB_STAMP(6 + 8 * qb);
    // next block's Q / dO / O rows: the global loads fly under this wave's dQ tile (the registers they occupy are free in this phase only)
    if (qb + 1 < nqb) stage_load(qb + 1);
    // ---- dQ tile of this wave (16 q x 16 ch), reduced over ALL keys: A = dS, B = K, both by transposed reads.  A chain of up to
    // seven 3-MFMA steps, each behind an LDS round trip: the fragments of key block kb + 1 are requested before block kb multiplies ----
    if (has_tile) {
      const int offa_q = 8 * tr_p, offb_c = (((2 * tc + (tr_p >> 1)) ^ tr_sw) << 4) + ((tr_p & 1) << 3);
      att_hx8 af[2][2], bk[2][2];
      auto rd = [&](int kb, int slot) __attribute__((always_inline)) {
        const int k0 = 32 * kb + tr_row;                    // second read: + 16 keys (same half-swap bit: 16 is a multiple of 8)
        const int offa = k0 * 64 + 32 * (tq ^ ((k0 >> 2) & 1)) + offa_q, offb = k0 * 128 + offb_c;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          af[slot][pl] = ab_tr2(dSpl + pl * AB_DSPL + offa, dSpl + pl * AB_DSPL + offa + 16 * 64);
          bk[slot][pl] = ab_tr2(Kpl + pl * AB_KPL + offb, Kpl + pl * AB_KPL + offb + 16 * 128);
        }
      };
      f32x4 acc = zero4(), acc1 = zero4();                  // even / odd key blocks on two chains, added once at the end (fixed order)
      rd(0, 0);
#pragma unroll
      for (int kb = 0; kb < AB_NKW; ++kb) {
        if (kb < nkb) {
          __builtin_amdgcn_sched_barrier(0);
          if (kb + 1 < nkb) rd(kb + 1, (kb + 1) & 1);
          if (kb & 1) acc1 = att_mfma3(af[kb & 1], bk[kb & 1], acc1);
          else acc = att_mfma3(af[kb & 1], bk[kb & 1], acc);
        }
      }
      acc += acc1;
      // acc[r] = dQ'[position 32 qb + 16 qt + 4 g + r][channel 16 ct + c] in units 2^(eds + he)
      const int p0 = qb * AB_QB + 16 * tq + 4 * gl, ch = 16 * tc + cl;
      const int rhi = min(4, npos - p0);
      if (ch < dh && rhi > 0) {
        acc *= scale * ofb_h_pow2(-(eds + he));
        float* dqp = dqkv + (tok0 + p0) * ldq + head * dh + ch;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < rhi) {
            if (LONG && kc > 0) acc[r] += dqp[(size_t)r * ldq];          // the earlier key chunks' share (stream-ordered launches)
            dqp[(size_t)r * ldq] = acc[r]; omax = fmaxf(omax, fabsf(acc[r]));
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
  const float k_inv = ofb_h_pow2(-(eds_prev + he)), v_inv = ofb_h_pow2(-(ATT_PE + hdo));
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int ch = 16 * ct + c;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int p0 = key0 + 32 * w + 16 * kt + 4 * g;
      const int rhi = min(4, npos - p0);
      const f32x4 kq = dKa[kt][ct] * k_inv, vq = dVa[kt][ct] * v_inv;           // dK accumulated against the pre-scaled Q
      if (ch < dh && rhi > 0 && w < AB_NKW) {
        float* dkp = dqkv + (tok0 + p0) * ldq + H * dh + head * dh + ch;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < rhi) {
            dkp[(size_t)r * ldq] = kq[r]; dkp[(size_t)r * ldq + H * dh] = vq[r];
            omax = fmaxf(omax, fmaxf(fabsf(kq[r]), fabsf(vq[r])));
          }
      }
    }
  }
  AB_STAMP(60);
  if (dqkv_amax) {
    omax = ofb_wave_max_pos(omax);
    if (lane == 0) vmax[8 + w] = omax;
    __syncthreads();
    if (t == 0) {
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < AB_NW; ++i) m = fmaxf(m, vmax[8 + i]);
      // amax_per_wg: one word per workgroup, plainly stored (no atomics, nothing to zero beforehand; LONG: the later key chunks of a
      // (batch, head) raise what the earlier launches left); the consumer reduces the B H words itself
      if (amax_per_wg) dqkv_amax[blockIdx.x] = (LONG && kc > 0) ? fmaxf(dqkv_amax[blockIdx.x], m) : m;
      else ofb_atomic_max_pos(dqkv_amax, m);
    }
  }
}

constexpr size_t BWD_LDS = AB_LDS_BYTES;

#define ATT_NLIMIT 4096     /* sequence lengths above one workgroup's 208 run chunked (forward: query chunks on blockIdx.y, backward: one launch per 224 keys) */
int check_shape(int B, int N, int H, int dh) {
  if (B <= 0 || N <= 0 || H <= 0 || dh <= 0) return OFB_EINVAL;
  if (N > ATT_NLIMIT || dh > ATT_DMAX || (dh & 3)) return OFB_ELIMIT;
  return OFB_OK;
}

// out[0] = max |x[i]| (out zeroed by a memset node ahead of the launch)
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float am = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) am = fmaxf(am, fabsf(x[i]));
  am = ofb_wave_max_pos(am);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
  __syncthreads();
  if (threadIdx.x == 0) ofb_atomic_max_pos(out, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

}  // namespace

// out[0] = max |x[i]|, i < n: an upper bound for callers that hold a tensor whose producer left none (tests, stand-alone attention calls)
extern "C" int ofb_amax(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out || n <= 0) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, 4, s) != hipSuccess) return (int)hipGetLastError();
  const int64_t nb = (n + 255) / 256;
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(256), 0, s, x, (size_t)n, out);
  return ofb_launch_status();
}

// qkv: [B*N][3*H*dh] packed as the qkv Linear writes it (q | k | v, each H*dh, head-major); out: [B*N][H*dh];
// lse: [2][B*H][N] (row log-sum-exp, then its rounding residue: see the forward kernel).  N <= 208, dh <= 64, dh % 4 == 0.
// qkv_bound: device scalar >= max |qkv| (the qkv GEMM's cbound_out, or ofb_amax).
extern "C" int ofb_attention_fwd(const float* qkv, float* out, float* lse, int32_t B, int32_t N, int32_t H, int32_t dh,
                                 float scale, const float* qkv_bound, void* stream) {
  if (!qkv || !out || !lse || !qkv_bound) return OFB_EINVAL;
  if (int rc = check_shape(B, N, H, dh)) return rc;
  if (!ofb_aligned16(qkv)) return OFB_EINVAL;
  if (!ofb_aligned16(out) || ((H * dh) & 3)) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(1, s, 4.0 * B * H * (double)N * N * dh);
  hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3(B * H, (N + ATT_NMAX - 1) / ATT_NMAX), dim3(ATT_THREADS), 0, s, qkv, out, lse, (char*)nullptr, 0,
                     B, N, H, dh, scale, qkv_bound);
  ofb_prof_post(1, s);
  return ofb_launch_status();
}

// out as above AND the same rows as H-format planes out_h[B*N][H*dh] (the operand of the projection GEMM; header included); the
// caller zeroes out_h beforehand when B*N or H*dh is not a multiple of 16.  Needs N + (b N mod 4) <= 208.
extern "C" int ofb_attention_fwd_h(const float* qkv, float* out, void* out_h, float* lse, int32_t B, int32_t N, int32_t H, int32_t dh,
                                   float scale, const float* qkv_bound, void* stream) {
  if (!qkv || !out || !out_h || !lse || !qkv_bound) return OFB_EINVAL;
  if (int rc = check_shape(B, N, H, dh)) return rc;
  if (!ofb_aligned16(qkv) || !ofb_aligned16(out) || !ofb_aligned16(out_h) || ((H * dh) & 3)) return OFB_EINVAL;
  int smax = 0;
  for (int bb = 0; bb < B && bb < 4; ++bb) smax = ((bb * N) & 3) > smax ? ((bb * N) & 3) : smax;
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(1, s, 4.0 * B * H * (double)N * N * dh);
  hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3(B * H, (N + smax + ATT_NMAX - 1) / ATT_NMAX), dim3(ATT_THREADS), 0, s, qkv, out, lse, (char*)out_h, (H * dh + 15) / 16, B, N, H, dh,
                     scale, qkv_bound);
  ofb_prof_post(1, s);
  return ofb_launch_status();
}

// dqkv: same packing as qkv (dq | dk | dv).  Needs the forward's out and lse, the bound the forward used for qkv and a bound of
// |dout| (device scalars).  dqkv_amax (optional device scalar): receives max |dqkv| - hand it to ofb_to_hformat_colsum as the bound.
namespace {
int attention_bwd_launch(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv, int32_t B, int32_t N,
                         int32_t H, int32_t dh, float scale, const float* qkv_bound, const float* dout_bound, float* dqkv_amax,
                         int amax_per_wg, void* stream) {
  if (!qkv || !out || !lse || !dout || !dqkv || !qkv_bound || !dout_bound) return OFB_EINVAL;
  if (int rc = check_shape(B, N, H, dh)) return rc;
  if (!ofb_aligned16(qkv) || !ofb_aligned16(out) || !ofb_aligned16(dout) || !ofb_aligned16(dqkv)) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  // the attribute is per device and the call is cheap: set it on every launch (a per-process flag would leave a second GPU
  // of the same process without it, and is not thread-safe)
  const bool lng = N > ATT_NMAX;
  if (hipFuncSetAttribute(lng ? (const void*)attn_bwd_kernel<true> : (const void*)attn_bwd_kernel<false>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_LDS) != hipSuccess)
    return (int)hipGetLastError();
  if (dqkv_amax && !amax_per_wg && hipMemsetAsync(dqkv_amax, 0, 4, s) != hipSuccess) return (int)hipGetLastError();
  ofb_prof_pre(4, s, 10.0 * B * H * (double)N * N * dh);
  if (!lng)
    hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3(B * H), dim3(AB_THREADS), BWD_LDS, s, qkv, out, lse, dout, dqkv, B, N, H, dh, scale,
                       qkv_bound, dout_bound, dqkv_amax, 0, amax_per_wg);
  else
    for (int kc = 0; kc * AB_NPOS < N; ++kc)               // dqkv_amax: the running dq sums are included, an upper bound of the final values
      hipLaunchKernelGGL(attn_bwd_kernel<true>, dim3(B * H), dim3(AB_THREADS), BWD_LDS, s, qkv, out, lse, dout, dqkv, B, N, H, dh, scale,
                         qkv_bound, dout_bound, dqkv_amax, kc, amax_per_wg);
  ofb_prof_post(4, s);
  return ofb_launch_status();
}
}  // namespace

extern "C" int ofb_attention_bwd(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv,
                                 int32_t B, int32_t N, int32_t H, int32_t dh, float scale, const float* qkv_bound,
                                 const float* dout_bound, float* dqkv_amax, void* stream) {
  return attention_bwd_launch(qkv, out, lse, dout, dqkv, B, N, H, dh, scale, qkv_bound, dout_bound, dqkv_amax, 0, stream);
}

// The same with the maximum left as ONE WORD PER WORKGROUP: wg_amax[B * H] (required) is plainly written, so nothing has to be zeroed
// ahead of the launch (no memset node) and no atomic is issued; ofb_to_hformat_colsum_nb takes the vector as its bound.
extern "C" int ofb_attention_bwd_wgmax(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv,
                                       int32_t B, int32_t N, int32_t H, int32_t dh, float scale, const float* qkv_bound,
                                       const float* dout_bound, float* wg_amax, void* stream) {
  if (!wg_amax) return OFB_EINVAL;
  return attention_bwd_launch(qkv, out, lse, dout, dqkv, B, N, H, dh, scale, qkv_bound, dout_bound, wg_amax, 1, stream);
}

#ifdef OFB_ATT_STAMPS
extern "C" int ofb_diag_att_stamps(unsigned long long* out_host) {      /* lab only, not part of the ABI */
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(ofb_att_stamps), sizeof(unsigned long long) * 8 * 64);
}
#endif
