// HBM-bound row kernels: LayerNorm forward/backward (wave per token row, wavefront reductions),
// column sums (bias gradients, partial reductions) and the small weight-sized gate-folding kernels.
#include "ofb_common.h"

namespace {

constexpr int LN_MAXE = 16;   // elements per lane -> D <= 1024

// element index owned by (lane, slot j, e) for vector width V
template <int V> __device__ __forceinline__ int ln_col(int lane, int j, int e) { return (j * 64 + lane) * V + e; }

template <int V>
__device__ __forceinline__ void ln_load(float (&v)[LN_MAXE], const float* __restrict__ p, int D, int lane) {
#pragma unroll
  for (int j = 0; j < LN_MAXE / V; ++j) {
    const int c = ln_col<V>(lane, j, 0);
    if (V == 2) {
      float2 t = make_float2(0.f, 0.f);
      if (c < D) t = *reinterpret_cast<const float2*>(p + c);
      v[2 * j] = t.x; v[2 * j + 1] = t.y;
    } else {
      v[j] = (c < D) ? p[c] : 0.f;
    }
  }
}
template <int V>
__device__ __forceinline__ void ln_store(const float (&v)[LN_MAXE], float* __restrict__ p, int D, int lane) {
#pragma unroll
  for (int j = 0; j < LN_MAXE / V; ++j) {
    const int c = ln_col<V>(lane, j, 0);
    if (c < D) {
      if (V == 2) *reinterpret_cast<float2*>(p + c) = make_float2(v[2 * j], v[2 * j + 1]);
      else p[c] = v[j];
    }
  }
}

template <int V>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int D,
                                                     float eps) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[LN_MAXE], g[LN_MAXE], b[LN_MAXE];
  ln_load<V>(v, x + (size_t)row * D, D, lane);
  ln_load<V>(g, gamma, D, lane);
  ln_load<V>(b, beta, D, lane);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) s += v[i];
  const float mu = ofb_wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) {
    const int c = ln_col<V>(lane, i / V, i % V);
    const float d = (c < D) ? v[i] - mu : 0.f;
    q += d * d;
  }
  const float rs = 1.0f / sqrtf(ofb_wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) v[i] = (v[i] - mu) * rs * g[i] + b[i];
  ln_store<V>(v, y + (size_t)row * D, D, lane);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx = rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat)) (+ dres); per-block partial dgamma/dbeta
template <int V>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     float* __restrict__ dx, float* __restrict__ part, int rows, int D) {
  __shared__ float red[4 * 2 * 1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float g[LN_MAXE], ag[LN_MAXE], ab[LN_MAXE];
  ln_load<V>(g, gamma, D, lane);
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) ag[i] = ab[i] = 0.f;
  for (int row = blockIdx.x * 4 + w; row < rows; row += gridDim.x * 4) {
    float v[LN_MAXE], d[LN_MAXE];
    ln_load<V>(v, x + (size_t)row * D, D, lane);
    ln_load<V>(d, dy + (size_t)row * D, D, lane);
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) {
      const int c = ln_col<V>(lane, i / V, i % V);
      const float xh = (c < D) ? (v[i] - mu) * rs : 0.f;
      const float dg = d[i] * g[i];
      ag[i] += d[i] * xh;
      ab[i] += d[i];
      s1 += dg;
      s2 += dg * xh;
      v[i] = xh;
      d[i] = dg;
    }
    const float c1 = ofb_wave_sum(s1) / (float)D, c2 = ofb_wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) v[i] = rs * (d[i] - c1 - v[i] * c2);
    if (dres) {
      float r[LN_MAXE];
      ln_load<V>(r, dres + (size_t)row * D, D, lane);
#pragma unroll
      for (int i = 0; i < LN_MAXE; ++i) v[i] += r[i];
    }
    ln_store<V>(v, dx + (size_t)row * D, D, lane);
  }
  // cross-wave reduction of the per-lane column partials
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) {
    const int c = ln_col<V>(lane, i / V, i % V);
    if (c < D) { red[w * 2048 + c] = ag[i]; red[w * 2048 + 1024 + c] = ab[i]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { sg += red[k * 2048 + c]; sb += red[k * 2048 + 1024 + c]; }
    part[(size_t)blockIdx.x * 2 * D + c] = sg;
    part[(size_t)blockIdx.x * 2 * D + D + c] = sb;
  }
}

// partial[slab][col] = sum over the slab's rows of x[row][col] * (rowscale ? rowscale[row / rs_div] : 1)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ld, int M, int N,
                                                     const float* __restrict__ rowscale, int rs_div,
                                                     float* __restrict__ out, int rows_per_slab) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
  float s = 0.f;
  if (col < N)
    for (int r = r0 + w; r < r1; r += 4) {
      float v = x[(size_t)r * ld + col];
      if (rowscale) v *= rowscale[r / rs_div];
      s += v;
    }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && col < N) out[(size_t)blockIdx.y * N + col] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

__global__ void scale_rows_kernel(const float* __restrict__ W, const float* __restrict__ g, float* __restrict__ out, int N,
                                  int K) {
  // one float4 per thread when rows are 16-B multiples (token matrices), scalar otherwise
  const int64_t total = (int64_t)N * K;
  if ((K & 3) == 0) {
    const int64_t n4 = total >> 2;
    const int k4 = K >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
      f32x4 v = reinterpret_cast<const f32x4*>(W)[i];
      v *= g[i / k4];
      reinterpret_cast<f32x4*>(out)[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
      out[i] = W[i] * g[i / K];
  }
}

// Gate folded into a Linear layer, y = g[n] * (x W^T + b)[n]:  given the raw gradients of the UNGATED product
// (dWraw = dY^T x, dbraw = colsum dY) produce dW = g*dWraw, db = g*dbraw and dg[n] = <dWraw[n], W[n]> + dbraw[n] b[n].
__global__ __launch_bounds__(256) void gate_fold_bwd_kernel(const float* __restrict__ dWraw, const float* __restrict__ W,
                                                            const float* __restrict__ g, const float* __restrict__ dbraw,
                                                            const float* __restrict__ b, float* __restrict__ dW,
                                                            float* __restrict__ db, float* __restrict__ dg, int N, int K) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float gn = g[n];
  float s = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float d = dWraw[(size_t)n * K + k];
    s += d * W[(size_t)n * K + k];
    dW[(size_t)n * K + k] = d * gn;
  }
  s = ofb_wave_sum(s);
  if (lane == 0) {
    const float dbr = dbraw ? dbraw[n] : 0.f;
    dg[n] = s + (b ? dbr * b[n] : 0.f);
    if (db) db[n] = dbr * gn;
  }
}

}  // namespace

extern "C" int ofb_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                 int32_t rows, int32_t D, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || D <= 0) return OFB_EINVAL;
  if (D > 64 * LN_MAXE) return OFB_ELIMIT;
  const dim3 grid(ofb_cdiv(rows, 4));
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(2, s, 8.0 * rows * (double)D);
  if (D % 2 == 0) hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, dim3(256), 0, s, x, gamma, beta, y, mean, rstd, rows, D, eps);
  else hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, dim3(256), 0, s, x, gamma, beta, y, mean, rstd, rows, D, eps);
  ofb_prof_post(2, s);
  return ofb_launch_status();
}

extern "C" int32_t ofb_layernorm_bwd_blocks(int32_t rows) { return rows >= 4096 ? 1024 : ofb_cdiv(rows, 4); }

extern "C" int ofb_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                 const float* dres, float* dx, float* partials, int32_t rows, int32_t D, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !partials || rows <= 0 || D <= 0) return OFB_EINVAL;
  if (D > 64 * LN_MAXE) return OFB_ELIMIT;
  const dim3 grid(ofb_layernorm_bwd_blocks(rows));
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(3, s, 16.0 * rows * (double)D);
  if (D % 2 == 0) hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, partials, rows, D);
  else hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, partials, rows, D);
  ofb_prof_post(3, s);
  return ofb_launch_status();
}

extern "C" int32_t ofb_colsum_slabs(int32_t M, int32_t N) {
  if (M <= 256) return 1;               // short inputs (second stages, per-block partials): one launch, no scratch
  const int colblocks = ofb_cdiv(N, 64);
  int slabs = ofb_cdiv(2048, colblocks);
  if (slabs > ofb_cdiv(M, 16)) slabs = ofb_cdiv(M, 16);
  return slabs < 1 ? 1 : slabs;
}

// out[N] = column sums of x[M][ld]; scratch must hold ofb_colsum_slabs(M,N)*N floats.
extern "C" int ofb_colsum(const float* x, int32_t ld, int32_t M, int32_t N, const float* rowscale, int32_t rs_div,
                          float* out, float* scratch, void* stream) {
  if (!x || !out || M <= 0 || N <= 0 || ld < N || (rowscale && rs_div <= 0)) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int slabs = ofb_colsum_slabs(M, N);
  const int colblocks = ofb_cdiv(N, 64);
  if (slabs == 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3(colblocks, 1), dim3(256), 0, s, x, ld, M, N, rowscale, rs_div, out, M);
    return ofb_launch_status();
  }
  if (!scratch) return OFB_EINVAL;
  const int rps = ofb_cdiv(M, slabs);
  hipLaunchKernelGGL(colsum_kernel, dim3(colblocks, slabs), dim3(256), 0, s, x, ld, M, N, rowscale, rs_div, scratch, rps);
  hipLaunchKernelGGL(colsum_kernel, dim3(colblocks, 1), dim3(256), 0, s, (const float*)scratch, N, slabs, N,
                     (const float*)nullptr, 1, out, slabs);
  return ofb_launch_status();
}

extern "C" int ofb_scale_rows(const float* W, const float* g, float* out, int32_t N, int32_t K, void* stream) {
  if (!W || !g || !out || N <= 0 || K <= 0) return OFB_EINVAL;
  if ((K & 3) == 0 && (!ofb_aligned16(W) || !ofb_aligned16(out))) return OFB_EINVAL;
  const int64_t total = (K & 3) == 0 ? (int64_t)N * K / 4 : (int64_t)N * K;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, g, out, N, K);
  return ofb_launch_status();
}

extern "C" int ofb_gate_fold_bwd(const float* dWraw, const float* W, const float* g, const float* dbraw, const float* b,
                                 float* dW, float* db, float* dg, int32_t N, int32_t K, void* stream) {
  if (!dWraw || !W || !g || !dW || !dg || N <= 0 || K <= 0) return OFB_EINVAL;
  hipLaunchKernelGGL(gate_fold_bwd_kernel, dim3(ofb_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, dWraw, W, g, dbraw, b,
                     dW, db, dg, N, K);
  return ofb_launch_status();
}
