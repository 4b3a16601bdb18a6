// f32 GEMM on v_mfma_f32_32x32x2_f32 (exact f32 fma chains) with fused epilogues.
// Block tile 128x128x16, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs).
// Both operands are staged k-major in LDS ([k][mn], row pitch 132 floats) through registers with a
// one-tile global prefetch, so the three storage combinations the Linear layers need (x@W^T, dY@W,
// dY^T@X) share one inner loop of ds_read_b32 + MFMA.
#include "ofb_common.h"

#define BM 128
#define BN 128
#define BK 16
#define LDP 132   // LDS row pitch (floats): 16-B aligned rows, transposed b32 writes at most 2-way conflicted

namespace {

struct TileRegs { f32x4 v[2]; };

// K-contiguous storage: X[o*ld + k].  512 float4 per tile, 2 per thread: o = idx>>2, kq = idx&3.
template <bool VEC>
__device__ __forceinline__ void load_kc(TileRegs& r, const float* __restrict__ X, int ld, int o0, int O, int k0, int kend,
                                        int t) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = t + 256 * i, o = o0 + (idx >> 2), k = k0 + ((idx & 3) << 2);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (o < O) {
      const float* p = X + (size_t)o * ld + k;
      if (VEC) {
        if (k < kend) v = *reinterpret_cast<const f32x4*>(p);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k + j < kend) v[j] = p[j];
      }
    }
    r.v[i] = v;
  }
}
__device__ __forceinline__ void store_kc(const TileRegs& r, float* __restrict__ S, int t) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = t + 256 * i, o = idx >> 2, kk = (idx & 3) << 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) S[(kk + j) * LDP + o] = r.v[i][j];
  }
}
// MN-contiguous storage: X[k*ld + o].  kk = idx>>5, oq = idx&31.
template <bool VEC>
__device__ __forceinline__ void load_mc(TileRegs& r, const float* __restrict__ X, int ld, int o0, int O, int k0, int kend,
                                        int t, const float* __restrict__ kscale, int ks_div) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = t + 256 * i, k = k0 + (idx >> 5), o = o0 + ((idx & 31) << 2);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k < kend) {
      const float* p = X + (size_t)k * ld + o;
      if (VEC) {
        if (o < O) v = *reinterpret_cast<const f32x4*>(p);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (o + j < O) v[j] = p[j];
      }
      if (kscale) v *= kscale[k / ks_div];
    }
    r.v[i] = v;
  }
}
__device__ __forceinline__ void store_mc(const TileRegs& r, float* __restrict__ S, int t) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = t + 256 * i;
    *reinterpret_cast<f32x4*>(&S[(idx >> 5) * LDP + ((idx & 31) << 2)]) = r.v[i];
  }
}

template <bool A_KC, bool B_KC, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const ofb_gemm_args g) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BK * LDP];
  float* As = lds;                    // [2][BK][LDP]
  float* Bs = lds + 2 * BK * LDP;     // [2][BK][LDP]

  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, h = lane >> 5;
  const int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
  const int tile = ofb_xcd_remap(blockIdx.x, mt * nt);
  const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
  const int wm0 = (w >> 1) * 64, wn0 = (w & 1) * 64;

  // K range of this split
  int kbeg = 0, kend = g.K;
  if (g.split_k > 1) {
    const int chunk = ((g.K + BK - 1) / BK + g.split_k - 1) / g.split_k * BK;
    kbeg = blockIdx.y * chunk;
    kend = min(g.K, kbeg + chunk);
  }
  const int ntile = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileRegs ra, rb;
  auto gload = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    if (A_KC) load_kc<VEC>(ra, g.A, g.lda, m0, g.M, k0, kend, t);
    else load_mc<VEC>(ra, g.A, g.lda, m0, g.M, k0, kend, t, g.kscale, g.ks_div);
    if (B_KC) load_kc<VEC>(rb, g.B, g.ldb, n0, g.N, k0, kend, t);
    else load_mc<VEC>(rb, g.B, g.ldb, n0, g.N, k0, kend, t, nullptr, 1);
  };
  auto lstore = [&](int buf) {
    if (A_KC) store_kc(ra, As + buf * BK * LDP, t); else store_mc(ra, As + buf * BK * LDP, t);
    if (B_KC) store_kc(rb, Bs + buf * BK * LDP, t); else store_mc(rb, Bs + buf * BK * LDP, t);
  };

  if (ntile > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int kt = 0; kt < ntile; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < ntile) gload(kt + 1);
    const float* a_s = As + buf * BK * LDP + (h * (BK / 2)) * LDP + wm0 + l31;
    const float* b_s = Bs + buf * BK * LDP + (h * (BK / 2)) * LDP + wn0 + l31;
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      // lane half h supplies k = h*8 + s for both operands: any k<->(step,half) bijection is a valid MFMA feed
      const float a0 = a_s[s * LDP], a1 = a_s[s * LDP + 32];
      const float b0 = b_s[s * LDP], b1 = b_s[s * LDP + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < ntile) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  const bool partial = g.split_k > 1;
  float* Cout = partial ? g.workspace + (size_t)blockIdx.y * g.M * g.N : g.C;
  const int ldc = partial ? g.N : g.ldc;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = n0 + wn0 + 32 * ni + l31;
    if (col >= g.N) continue;
    float bias = 0.f, cs = 1.f;
    if (!partial) {
      if (g.bias) bias = g.bias[col];
      if (g.colscale) cs = g.colscale[col];
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= g.M) continue;
        float v = acc[mi][ni][r];
        if (!partial) {
          v = (v * g.alpha + bias) * cs;
          if (g.act == OFB_ACT_GELU) {
            if (g.aux) g.aux[(size_t)row * g.ldaux + col] = v;
            v = ofb_gelu(v);
          } else if (g.act == OFB_ACT_DGELU) {
            v *= ofb_dgelu(g.aux[(size_t)row * g.ldaux + col]);
          }
          if (g.rowscale) v *= g.rowscale[row / g.rs_div];
          if (g.resid) v += g.resid[(size_t)row * g.ldr + col];
        }
        Cout[(size_t)row * ldc + col] = v;
      }
    }
  }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int64_t count, float* __restrict__ out,
                                     int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    float s = accumulate ? out[i] : 0.f;
    for (int z = 0; z < splits; ++z) s += ws[(size_t)z * count + i];
    out[i] = s;
  }
}

template <bool A_KC, bool B_KC>
int launch(const ofb_gemm_args& g, bool vec, dim3 grid, hipStream_t s) {
  if (vec) hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, true>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, false>), grid, dim3(256), 0, s, g);
  return ofb_launch_status();
}

}  // namespace

extern "C" int ofb_gemm_f32(const ofb_gemm_args* args, void* stream) {
  if (!args) return OFB_EINVAL;
  const ofb_gemm_args& g = *args;
  if (!g.A || !g.B || g.M <= 0 || g.N <= 0 || g.K <= 0) return OFB_EINVAL;
  if (g.split_k > 1 ? !g.workspace : !g.C) return OFB_EINVAL;
  if (g.a_kc == 0 && g.b_kc == 1) return OFB_ELIMIT;           // A^T * B^T is not on the path
  if (g.kscale && (g.a_kc != 0 || g.ks_div <= 0)) return OFB_EINVAL;
  if (g.rowscale && g.rs_div <= 0) return OFB_EINVAL;
  if (g.act == OFB_ACT_DGELU && !g.aux) return OFB_EINVAL;
  // minimum leading dimensions for the declared storage
  if (g.lda < (g.a_kc ? g.K : g.M) || g.ldb < (g.b_kc ? g.K : g.N)) return OFB_EINVAL;
  if (g.split_k <= 1 && g.ldc < g.N) return OFB_EINVAL;
  // vector (16-B) staging needs aligned bases, ld % 4 == 0 and a contiguous extent that is a multiple of 4
  bool vec = ofb_aligned16(g.A) && ofb_aligned16(g.B) && (g.lda % 4 == 0) && (g.ldb % 4 == 0);
  vec = vec && ((g.a_kc ? g.K : g.M) % 4 == 0) && ((g.b_kc ? g.K : g.N) % 4 == 0);
  const int mt = ofb_cdiv(g.M, BM), nt = ofb_cdiv(g.N, BN);
  dim3 grid(mt * nt, g.split_k > 1 ? g.split_k : 1);
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(0, s, 2.0 * g.M * g.N * (double)g.K);
  int rc;
  if (g.a_kc && g.b_kc) rc = launch<true, true>(g, vec, grid, s);
  else if (g.a_kc) rc = launch<true, false>(g, vec, grid, s);
  else rc = launch<false, false>(g, vec, grid, s);
  ofb_prof_post(0, s);
  return rc;
}

extern "C" int ofb_splitk_reduce(const float* workspace, int32_t splits, int64_t count, float* out, int32_t accumulate,
                                 void* stream) {
  if (!workspace || !out || splits <= 0 || count <= 0) return OFB_EINVAL;
  const int blocks = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, workspace, splits, count, out,
                     accumulate);
  return ofb_launch_status();
}
