// Fused gated-attention core for the OFB search step: softmax(q k^T * scale) v, forward and backward,
// on f32-input MFMA (v_mfma_f32_32x32x2_f32).  One workgroup per (batch, head); the whole K/V of the
// head (N <= 224 tokens, d <= 64) lives in LDS, probabilities never touch HBM.
// Reference math: models/layers.py:510-514 (search branch) and its autograd.
//
// Operand conventions (32x32x2: lane l gives A[i=l&31][kslot=l>>5], B[kslot=l>>5][j=l&31];
// D[i][j] in reg r at i = (r&3) + 8*(r>>2) + 4*(l>>5), j = l&31):
//  * any bijection reduction-index <-> (step, kslot) is valid as long as A and B use the same one;
//  * an accumulator tile X can feed the next MFMA directly when that MFMA reduces over X's ROW index:
//    reg s of lane-half h is row rowmap(s,h), so the other operand is read at row rowmap(s,h).
#include "ofb_common.h"

#define ATT_NMAX 224      // 7 key tiles of 32
#define ATT_KT 7
#define ATT_DMAX 64
#define KS_LD 68          // K tile pitch (b128 row reads, conflict-free)
#define QS_LD 68

namespace {

__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// ------------------------------------------------------------------------------------------------
// forward: 256 threads; wave w handles query tiles w and w+4
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                       float* __restrict__ lse, int B, int N, int H, int dh, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;                          // [224][68]
  float* Vs = smem + ATT_NMAX * KS_LD;       // [224][64]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, head = blockIdx.x % H;
  const int ldq = 3 * H * dh, ldo = H * dh;
  const float* qbase = qkv + (size_t)b * N * ldq + head * dh;
  const float* kbase = qbase + H * dh;
  const float* vbase = qbase + 2 * H * dh;

  for (int idx = t; idx < ATT_NMAX * 16; idx += 256) {
    const int row = idx >> 4, c = (idx & 15) << 2;
    f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (row < N && c < dh) {
      kv = *reinterpret_cast<const f32x4*>(kbase + (size_t)row * ldq + c);
      vv = *reinterpret_cast<const f32x4*>(vbase + (size_t)row * ldq + c);
    }
    *reinterpret_cast<f32x4*>(&Ks[row * KS_LD + c]) = kv;
    *reinterpret_cast<f32x4*>(&Vs[row * ATT_DMAX + c]) = vv;
  }
  __syncthreads();

  const int nqt = (N + 31) >> 5;
  for (int qt = w; qt < nqt; qt += 4) {
    const int q = qt * 32 + l31;
    // this lane's query row segment [32h, 32h+32), pre-scaled
    float qr[32];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = 32 * h + 4 * u;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < N && c < dh) v = *reinterpret_cast<const f32x4*>(qbase + (size_t)q * ldq + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) qr[4 * u + j] = v[j] * scale;
    }
    // S^T tiles: rows = keys (A = K from LDS), cols = queries (B = q registers)
    f32x16 S[ATT_KT];
#pragma unroll
    for (int kt = 0; kt < ATT_KT; ++kt) {
      f32x16 acc = zero16();
      const float* kp = &Ks[(kt * 32 + l31) * KS_LD + 32 * h];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + 4 * u);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[j], qr[4 * u + j], acc, 0, 0, 0);
      }
      S[kt] = acc;
      __builtin_amdgcn_sched_barrier(0);   // keep the next tile's LDS reads from being hoisted (register pressure)
    }
    // row softmax over keys: 112 values in this lane + the partner half (lane ^ 32)
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < ATT_KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + rowmap(r, h);
        const float s = (key < N) ? S[kt][r] : -INFINITY;
        S[kt][r] = s;
        m = fmaxf(m, s);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < ATT_KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = expf(S[kt][r] - m);
        S[kt][r] = p;
        l += p;
      }
    l += __shfl_xor(l, 32, 64);
    // O = P V: P^T accumulators feed the A operand directly (reduction over their row index = key)
    f32x16 O[2];
    O[0] = zero16();
    O[1] = zero16();
#pragma unroll
    for (int kt = 0; kt < ATT_KT; ++kt) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float* vp = &Vs[(kt * 32 + rowmap(s, h)) * ATT_DMAX + l31];
        O[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(S[kt][s], vp[0], O[0], 0, 0, 0);
        O[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(S[kt][s], vp[32], O[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const float linv = 1.0f / l;
    if (h == 0 && q < N) lse[((size_t)b * H + head) * N + q] = m + logf(l);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qrow = rowmap(r, h);
      const float li = __shfl(linv, qrow, 64);
      const int qq = qt * 32 + qrow;
      if (qq < N) {
        float* op = out + ((size_t)b * N + qq) * ldo + head * dh;
        if (l31 < dh) op[l31] = O[0][r] * li;
        if (32 + l31 < dh) op[32 + l31] = O[1][r] * li;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, split in two kernels that mirror the forward structure (P is recomputed from lse):
//  * attn_bwd_dq:   wave owns a QUERY tile, K/V of the head in LDS, lane = query (S^T orientation):
//                   dQ = sum_key dS^T[key][q] K[key][:] takes the dS^T accumulators directly as A operands.
//  * attn_bwd_dkv:  wave owns a KEY tile, Q/dO of the head in LDS, lane = key (S orientation):
//                   dV = P^T dO and dK = dS^T Q take the P / dS accumulators directly as A operands.
// No tile ever needs a transpose through LDS and nothing is summed across waves.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                          const float* __restrict__ lse, const float* __restrict__ dout,
                                                          float* __restrict__ dqkv, int B, int N, int H, int dh, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;                          // [224][68]
  float* Vs = smem + ATT_NMAX * KS_LD;       // [224][68]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, head = blockIdx.x % H;
  const int ldq = 3 * H * dh, ldo = H * dh;
  const size_t tok0 = (size_t)b * N;
  const float* qbase = qkv + tok0 * ldq + head * dh;
  const float* kbase = qbase + H * dh;
  const float* vbase = qbase + 2 * H * dh;
  const float* obase = out + tok0 * ldo + head * dh;
  const float* dobase = dout + tok0 * ldo + head * dh;
  float* dqbase = dqkv + tok0 * ldq + head * dh;

  for (int idx = t; idx < ATT_NMAX * 16; idx += 256) {
    const int row = idx >> 4, c = (idx & 15) << 2;
    f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (row < N && c < dh) {
      kv = *reinterpret_cast<const f32x4*>(kbase + (size_t)row * ldq + c);
      vv = *reinterpret_cast<const f32x4*>(vbase + (size_t)row * ldq + c);
    }
    *reinterpret_cast<f32x4*>(&Ks[row * KS_LD + c]) = kv;
    *reinterpret_cast<f32x4*>(&Vs[row * KS_LD + c]) = vv;
  }
  __syncthreads();

  const int nqt = (N + 31) >> 5;
  for (int qt = w; qt < nqt; qt += 4) {
    const int q = qt * 32 + l31;
    float qr[32], dor[32];
    float dl = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = 32 * h + 4 * u;
      f32x4 qv = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f}, ov = {0.f, 0.f, 0.f, 0.f};
      if (q < N && c < dh) {
        qv = *reinterpret_cast<const f32x4*>(qbase + (size_t)q * ldq + c);
        dv = *reinterpret_cast<const f32x4*>(dobase + (size_t)q * ldo + c);
        ov = *reinterpret_cast<const f32x4*>(obase + (size_t)q * ldo + c);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { qr[4 * u + j] = qv[j] * scale; dor[4 * u + j] = dv[j]; dl += dv[j] * ov[j]; }
    }
    dl += __shfl_xor(dl, 32, 64);                                   // delta[q] = rowsum(dO * O)
    const float ls = (q < N) ? lse[((size_t)b * H + head) * N + q] : 0.f;
    f32x16 dQ0 = zero16(), dQ1 = zero16();
#pragma unroll 1
    for (int kt = 0; kt < ATT_KT; ++kt) {
      f32x16 S = zero16(), dP = zero16();
      const float* kp = &Ks[(kt * 32 + l31) * KS_LD + 32 * h];
      const float* vp = &Vs[(kt * 32 + l31) * KS_LD + 32 * h];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + 4 * u);
        const f32x4 vv = *reinterpret_cast<const f32x4*>(vp + 4 * u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[j], qr[4 * u + j], S, 0, 0, 0);      // S^T[key][q]
          dP = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[j], dor[4 * u + j], dP, 0, 0, 0);   // dP^T[key][q]
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + rowmap(r, h);
        const float p = (key < N) ? expf(S[r] - ls) : 0.f;
        dP[r] = p * (dP[r] - dl);                                   // dS'^T
      }
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const float* kr = &Ks[(kt * 32 + rowmap(s2, h)) * KS_LD + l31];
        dQ0 = __builtin_amdgcn_mfma_f32_32x32x2f32(dP[s2], kr[0], dQ0, 0, 0, 0);
        dQ1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dP[s2], kr[32], dQ1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = qt * 32 + rowmap(r, h);
      if (qq < N) {
        float* dp = dqbase + (size_t)qq * ldq;
        if (l31 < dh) dp[l31] = dQ0[r] * scale;
        if (32 + l31 < dh) dp[32 + l31] = dQ1[r] * scale;
      }
    }
  }
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                           const float* __restrict__ lse, const float* __restrict__ dout,
                                                           float* __restrict__ dqkv, int B, int N, int H, int dh, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem;                          // [224][68]
  float* dOs = smem + ATT_NMAX * QS_LD;      // [224][68]
  float* lse_s = dOs + ATT_NMAX * QS_LD;     // [224]
  float* del_s = lse_s + ATT_NMAX;           // [224]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / H, head = blockIdx.x % H;
  const int ldq = 3 * H * dh, ldo = H * dh;
  const size_t tok0 = (size_t)b * N;
  const float* qbase = qkv + tok0 * ldq + head * dh;
  const float* kbase = qbase + H * dh;
  const float* vbase = qbase + 2 * H * dh;
  const float* obase = out + tok0 * ldo + head * dh;
  const float* dobase = dout + tok0 * ldo + head * dh;
  float* dkbase = dqkv + tok0 * ldq + H * dh + head * dh;
  float* dvbase = dkbase + H * dh;

  // stage Q and dO of the whole head; delta[q] = rowsum(dO * O) from the 16 lanes that share a row
  for (int idx = t; idx < ATT_NMAX * 16; idx += 256) {
    const int row = idx >> 4, c = (idx & 15) << 2;
    f32x4 qv = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f}, ov = {0.f, 0.f, 0.f, 0.f};
    if (row < N && c < dh) {
      qv = *reinterpret_cast<const f32x4*>(qbase + (size_t)row * ldq + c);
      dv = *reinterpret_cast<const f32x4*>(dobase + (size_t)row * ldo + c);
      ov = *reinterpret_cast<const f32x4*>(obase + (size_t)row * ldo + c);
    }
    float d = dv[0] * ov[0] + dv[1] * ov[1] + dv[2] * ov[2] + dv[3] * ov[3];
    d += __shfl_xor(d, 8, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 1, 64);
    *reinterpret_cast<f32x4*>(&Qs[row * QS_LD + c]) = qv;
    *reinterpret_cast<f32x4*>(&dOs[row * QS_LD + c]) = dv;
    if ((idx & 15) == 0) del_s[row] = d;
  }
  for (int i = t; i < ATT_NMAX; i += 256) lse_s[i] = (i < N) ? lse[((size_t)b * H + head) * N + i] : 0.f;
  __syncthreads();

  const int nkt = (N + 31) >> 5;
  for (int kt = w; kt < nkt; kt += 4) {
    const int key = kt * 32 + l31;
    float kr[32], vr[32];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = 32 * h + 4 * u;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (key < N && c < dh) {
        kv = *reinterpret_cast<const f32x4*>(kbase + (size_t)key * ldq + c);
        vv = *reinterpret_cast<const f32x4*>(vbase + (size_t)key * ldq + c);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { kr[4 * u + j] = kv[j] * scale; vr[4 * u + j] = vv[j]; }
    }
    f32x16 dK0 = zero16(), dK1 = zero16(), dV0 = zero16(), dV1 = zero16();
#pragma unroll 1
    for (int qt = 0; qt < ATT_KT; ++qt) {
      f32x16 S = zero16(), dP = zero16();
      const float* qp = &Qs[(qt * 32 + l31) * QS_LD + 32 * h];
      const float* dp = &dOs[(qt * 32 + l31) * QS_LD + 32 * h];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + 4 * u);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dp + 4 * u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          S = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[j], kr[4 * u + j], S, 0, 0, 0);     // S'[q][key]
          dP = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[j], vr[4 * u + j], dP, 0, 0, 0);   // dP[q][key]
        }
      }
      // padded query rows hold Q = dO = 0, lse = delta = 0: P = 1, dS = 0, and dO rows are 0, so they add nothing
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qrow = qt * 32 + rowmap(r, h);
        const float p = expf(S[r] - lse_s[qrow]);
        S[r] = p;
        dP[r] = p * (dP[r] - del_s[qrow]);
      }
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const int qrow = qt * 32 + rowmap(s2, h);
        const float* dop = &dOs[qrow * QS_LD + l31];
        const float* qp2 = &Qs[qrow * QS_LD + l31];
        dV0 = __builtin_amdgcn_mfma_f32_32x32x2f32(S[s2], dop[0], dV0, 0, 0, 0);
        dV1 = __builtin_amdgcn_mfma_f32_32x32x2f32(S[s2], dop[32], dV1, 0, 0, 0);
        dK0 = __builtin_amdgcn_mfma_f32_32x32x2f32(dP[s2], qp2[0], dK0, 0, 0, 0);
        dK1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dP[s2], qp2[32], dK1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = kt * 32 + rowmap(r, h);
      if (kk < N) {
        float* dkp = dkbase + (size_t)kk * ldq;
        float* dvp = dvbase + (size_t)kk * ldq;
        if (l31 < dh) { dkp[l31] = dK0[r] * scale; dvp[l31] = dV0[r]; }
        if (32 + l31 < dh) { dkp[32 + l31] = dK1[r] * scale; dvp[32 + l31] = dV1[r]; }
      }
    }
  }
}

constexpr size_t FWD_LDS = (size_t)(ATT_NMAX * KS_LD + ATT_NMAX * ATT_DMAX) * sizeof(float);
constexpr size_t BWD_LDS = (size_t)(2 * ATT_NMAX * KS_LD + 2 * ATT_NMAX) * sizeof(float);

int check_shape(int B, int N, int H, int dh) {
  if (B <= 0 || N <= 0 || H <= 0 || dh <= 0) return OFB_EINVAL;
  if (N > ATT_NMAX || dh > ATT_DMAX || (dh & 3)) return OFB_ELIMIT;
  return OFB_OK;
}

}  // namespace

// qkv: [B*N][3*H*dh] packed as the qkv Linear writes it (q | k | v, each H*dh, head-major); out: [B*N][H*dh];
// lse: [B*H][N].  N <= 224, dh <= 64, dh % 4 == 0.
extern "C" int ofb_attention_fwd(const float* qkv, float* out, float* lse, int32_t B, int32_t N, int32_t H, int32_t dh,
                                 float scale, void* stream) {
  if (!qkv || !out || !lse) return OFB_EINVAL;
  if (int rc = check_shape(B, N, H, dh)) return rc;
  if (!ofb_aligned16(qkv)) return OFB_EINVAL;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FWD_LDS) != hipSuccess)
      return (int)hipGetLastError();
    attr_set = true;
  }
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(1, s, 4.0 * B * H * (double)N * N * dh);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * H), dim3(256), FWD_LDS, s, qkv, out, lse, B, N, H, dh, scale);
  ofb_prof_post(1, s);
  return ofb_launch_status();
}

// dqkv: same packing as qkv (dq | dk | dv).  Needs the forward's out and lse.
extern "C" int ofb_attention_bwd(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv,
                                 int32_t B, int32_t N, int32_t H, int32_t dh, float scale, void* stream) {
  if (!qkv || !out || !lse || !dout || !dqkv) return OFB_EINVAL;
  if (int rc = check_shape(B, N, H, dh)) return rc;
  if (!ofb_aligned16(qkv) || !ofb_aligned16(out) || !ofb_aligned16(dout) || !ofb_aligned16(dqkv)) return OFB_EINVAL;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_LDS) != hipSuccess)
      return (int)hipGetLastError();
    attr_set = true;
  }
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(4, s, 14.0 * B * H * (double)N * N * dh);   // 7 products incl. the recomputed S and dP
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(B * H), dim3(256), BWD_LDS, s, qkv, out, lse, dout, dqkv, B, N, H, dh, scale);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(B * H), dim3(256), BWD_LDS, s, qkv, out, lse, dout, dqkv, B, N, H, dh, scale);
  ofb_prof_post(4, s);
  return ofb_launch_status();
}
