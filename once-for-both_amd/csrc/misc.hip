// Token assembly around the patch embedding, PMIM targets/loss, label-smoothing CE and multi-tensor AdamW.
#include "ofb_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// tokens[b][0][c]   = g[c] * (cls[c] + pos[0][c])
// tokens[b][1+l][c] = g[c] * ((conv[b][l][c] + pos[1+l][c]) * (1 - m[b][l]) + m[b][l] * mask_token[c])
// (models/vision_transformer.py:615-651 with the embed gate of layers.py:191 factored out of every term)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_assemble_fwd_kernel(const float* __restrict__ conv, const float* __restrict__ g,
                                                                 const float* __restrict__ pos, const float* __restrict__ cls,
                                                                 const float* __restrict__ mtok, const float* __restrict__ mask,
                                                                 float* __restrict__ tok, int B, int L, int D) {
  const int row = blockIdx.x;               // b*(L+1) + t
  const int b = row / (L + 1), t = row % (L + 1);
  const float m = (t > 0 && mask) ? mask[b * L + t - 1] : 0.f;
  for (int c = threadIdx.x; c < D; c += 256) {
    float v;
    if (t == 0) v = cls[c] + pos[c];
    else v = (conv[((size_t)b * L + t - 1) * D + c] + pos[(size_t)t * D + c]) * (1.0f - m) + m * (mtok ? mtok[c] : 0.f);
    tok[(size_t)row * D + c] = (g ? g[c] : 1.0f) * v;
  }
}

// the same for D % 4 == 0: a block takes FOUR token rows, a thread float4 columns, both trips' loads in flight together (one 1.5-KB
// row per 256-thread block was 25 k tiny blocks: 27 us for 77 MB)
__global__ __launch_bounds__(256) void embed_assemble_fwd4_kernel(const float* __restrict__ conv, const float* __restrict__ g,
                                                                  const float* __restrict__ pos, const float* __restrict__ cls,
                                                                  const float* __restrict__ mtok, const float* __restrict__ mask,
                                                                  float* __restrict__ tok, int B, int L, int D) {
  const int D4 = D >> 2, rows = B * (L + 1), items = 4 * D4;
  f32x4 cv[2], pv[2], gv[2], mv[2];
  float m[2];
  int rowi[2], c4i[2], ti[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int it = threadIdx.x + 256 * u;
    const int r = it / D4, c4 = it - r * D4, row = min(blockIdx.x * 4 + r, rows - 1);
    const int b = row / (L + 1), t = row - b * (L + 1);
    rowi[u] = (it < items && blockIdx.x * 4 + r < rows) ? row : -1; c4i[u] = c4; ti[u] = t;
    m[u] = (t > 0 && mask) ? mask[b * L + t - 1] : 0.f;
    cv[u] = (t > 0) ? reinterpret_cast<const f32x4*>(conv + ((size_t)b * L + t - 1) * D)[c4] : reinterpret_cast<const f32x4*>(cls)[c4];
    pv[u] = reinterpret_cast<const f32x4*>(pos + (size_t)t * D)[c4];
    gv[u] = g ? reinterpret_cast<const f32x4*>(g)[c4] : f32x4{1.f, 1.f, 1.f, 1.f};
    mv[u] = mtok ? reinterpret_cast<const f32x4*>(mtok)[c4] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (rowi[u] < 0) continue;
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // (element for element the expressions of embed_assemble_fwd_kernel)
      const float e = (ti[u] == 0) ? cv[u][k] + pv[u][k] : (cv[u][k] + pv[u][k]) * (1.0f - m[u]) + m[u] * mv[u][k];
      v[k] = gv[u][k] * e;
    }
    reinterpret_cast<f32x4*>(tok + (size_t)rowi[u] * D)[c4i[u]] = v;
  }
}

// grid (L+1, chunks): block (t, z) sweeps its batch chunk.  Writes dconv rows and per-(chunk) partials:
//   ppos[z][t][c] (-> dpos, dcls), pg[z*(L+1)+t][c] (-> dg), pmt[z*(L+1)+t][c] (-> dmask_token)
__global__ __launch_bounds__(256) void embed_assemble_bwd_kernel(const float* __restrict__ dtok, const float* __restrict__ conv,
                                                                 const float* __restrict__ g, const float* __restrict__ pos,
                                                                 const float* __restrict__ cls, const float* __restrict__ mtok,
                                                                 const float* __restrict__ mask, float* __restrict__ dconv,
                                                                 float* __restrict__ ppos, float* __restrict__ pg,
                                                                 float* __restrict__ pmt, int B, int L, int D, int bchunk) {
  const int t = blockIdx.x, z = blockIdx.y;
  const int b0 = z * bchunk, b1 = min(B, b0 + bchunk);
  for (int c = threadIdx.x; c < D; c += 256) {
    const float gc = g ? g[c] : 1.0f, pc = pos[(size_t)t * D + c];
    const float mt = mtok ? mtok[c] : 0.f, cl = (t == 0) ? cls[c] : 0.f;
    float apos = 0.f, ag = 0.f, amt = 0.f;
    // four images' loads in flight together (one image at a time the sweep was a chain of dependent round trips: 45 us for 116 MB);
    // the sums keep their order: image b, then b + 1, ...
    for (int bb = b0; bb < b1; bb += 4) {
      float d4[4], m4[4], cv4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = min(bb + u, b1 - 1);                   // (past the chunk's end: re-read its last image, dropped below)
        d4[u] = dtok[((size_t)b * (L + 1) + t) * D + c];
        m4[u] = (t > 0 && mask) ? mask[b * L + t - 1] : 0.f;
        cv4[u] = (t > 0) ? conv[((size_t)b * L + t - 1) * D + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = bb + u;
        if (b >= b1) break;
        const float d = d4[u];
        if (t == 0) {
          apos += d * gc;
          ag += d * (cl + pc);
        } else {
          const float m = m4[u];
          const size_t ci = ((size_t)b * L + t - 1) * D + c;
          const float cv = cv4[u];
          const float dk = d * gc * (1.0f - m);
          dconv[ci] = dk;
          apos += dk;
          amt += d * gc * m;
          ag += d * ((cv + pc) * (1.0f - m) + m * mt);
        }
      }
    }
    const size_t o = ((size_t)z * (L + 1) + t) * D + c;
    ppos[o] = apos; pg[o] = ag; pmt[o] = amt;
  }
}

// ---------------------------------------------------------------------------------------------------
// norm_targets (models/vision_transformer.py:121-141): 47x47 box statistics, count_include_pad=False.
// pass 1: horizontal zero-padded window sums of v and v^2; pass 2: vertical sums + normalisation.
// ---------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void box_h_kernel(const float* __restrict__ img, float* __restrict__ s1, float* __restrict__ s2,
                                                    int planes, int Hh, int Ww) {
  constexpr int R = K / 2, SPAN = 8 + K - 1;
  const int segs = (Ww + 7) / 8;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)planes * Hh * segs) return;
  const int seg = (int)(id % segs);
  const int64_t rowid = id / segs;
  const float* row = img + rowid * Ww;
  const int x0 = seg * 8;
  float v[SPAN];
#pragma unroll
  for (int k = 0; k < SPAN; ++k) {
    const int x = x0 - R + k;
    v[k] = (x >= 0 && x < Ww) ? row[x] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { a += v[i + k]; q += v[i + k] * v[i + k]; }
    if (x0 + i < Ww) { s1[rowid * Ww + x0 + i] = a; s2[rowid * Ww + x0 + i] = q; }
  }
}

template <int K>
__global__ __launch_bounds__(256) void box_v_norm_kernel(const float* __restrict__ img, const float* __restrict__ s1,
                                                         const float* __restrict__ s2, float* __restrict__ out, int planes,
                                                         int Hh, int Ww) {
  constexpr int R = K / 2, SPAN = 8 + K - 1;
  const int segs = (Hh + 7) / 8;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)planes * segs * Ww) return;
  const int x = (int)(id % Ww);
  const int seg = (int)((id / Ww) % segs);
  const int64_t plane = id / ((int64_t)Ww * segs);
  const float* p1 = s1 + plane * Hh * Ww;
  const float* p2 = s2 + plane * Hh * Ww;
  const int y0 = seg * 8;
  float a[SPAN], q[SPAN];
#pragma unroll
  for (int k = 0; k < SPAN; ++k) {
    const int y = y0 - R + k;
    const bool in = (y >= 0 && y < Hh);
    a[k] = in ? p1[(size_t)y * Ww + x] : 0.f;
    q[k] = in ? p2[(size_t)y * Ww + x] : 0.f;
  }
  const float cx = (float)(min(x + R, Ww - 1) - max(x - R, 0) + 1);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int y = y0 + i;
    if (y >= Hh) break;
    float sa = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { sa += a[i + k]; sq += q[i + k]; }
    const float cnt = cx * (float)(min(y + R, Hh - 1) - max(y - R, 0) + 1);
    const float mean = sa / cnt, sqm = sq / cnt;
    float var = (sqm - mean * mean) * (cnt / (cnt - 1.0f));
    var = fmaxf(var, 0.f);
    const size_t o = (size_t)plane * Hh * Ww + (size_t)y * Ww + x;
    out[o] = (img[o] - mean) / sqrtf(var + 1.e-6f);
  }
}

// The same statistics for the pixels the loss reads only (the masked patches: M = 0 elsewhere, vision_transformer.py:724-729), in
// one kernel: block = (masked patch, image plane); the patch's (P + K - 1)^2 window (P x P pixels + the K/2 halo) is loaded into
// LDS once, the horizontal and the vertical K-sums run out of LDS in the same order as box_h_kernel / box_v_norm_kernel, and the
// P x P normalised pixels are written to their place in the [planes][H][W] target buffer (the rest of it is never read).
template <int K, int PMAX>
__global__ __launch_bounds__(256) void norm_targets_masked_kernel(const float* __restrict__ img, const int32_t* __restrict__ ids,
                                                                  float* __restrict__ out, int C, int L, int gw, int P, int Hh, int Ww) {
  constexpr int R = K / 2, WIN = PMAX + K - 1, PITCH = WIN + 1;
  __shared__ float win[WIN * PITCH], h1[WIN * PMAX], h2[WIN * PMAX];
  const int patch = ids[blockIdx.x], c = blockIdx.y, b = patch / L, l = patch % L, py = l / gw, px = l % gw;
  const float* plane = img + ((size_t)b * C + c) * Hh * Ww;
  const int y0 = py * P - R, x0 = px * P - R, wn = P + K - 1;
  // the window's pixels: every thread's (up to 16) loads leave together, branch-free, and reach LDS afterwards (one element at a time
  // - load, wait, store - the window cost sixteen dependent round trips per block: most of this launch's 60 us).  Row / column of
  // element i by a float reciprocal: exact for i < 4096, wn <= 62 (the quotient is never within 0.008 of an integer).
  {
    constexpr int NLD = (WIN * WIN + 255) / 256;
    const float rwn = 1.0f / (float)wn;
    float tmp[NLD];
    int at[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int i = threadIdx.x + 256 * u;
      const int yy = (int)(((float)i + 0.5f) * rwn), xx = i - yy * wn, y = y0 + yy, x = x0 + xx;
      const bool ok = i < wn * wn && y >= 0 && y < Hh && x >= 0 && x < Ww;
      const float v = plane[ok ? (size_t)y * Ww + x : (size_t)0];
      tmp[u] = ok ? v : 0.f;
      at[u] = i < wn * wn ? yy * PITCH + xx : -1;
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u)
      if (at[u] >= 0) win[at[u]] = tmp[u];
  }
  __syncthreads();
  // horizontal zero-padded window sums of v and v^2 for the P columns of the patch, every row of the window: a thread keeps K + 3
  // values of one row in registers and forms four neighbouring sums from them (each still added left to right); consecutive
  // threads take consecutive rows (odd LDS pitch: conflict free)
  for (int i = threadIdx.x; i < wn * ((P + 3) / 4); i += 256) {
    const int yy = i % wn, j0 = 4 * (i / wn);
    const float* rowp = &win[yy * PITCH + j0];
    float v[K + 3];
#pragma unroll
    for (int k = 0; k < K + 3; ++k) v[k] = (j0 + k < wn) ? rowp[k] : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) { a += v[t + k]; q += v[t + k] * v[t + k]; }
      if (j0 + t < P) { h1[yy * PMAX + j0 + t] = a; h2[yy * PMAX + j0 + t] = q; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < P * P; i += 256) {
    const int ii = i / P, j = i % P, y = py * P + ii, x = px * P + j;
    float sa = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { sa += h1[(ii + k) * PMAX + j]; sq += h2[(ii + k) * PMAX + j]; }
    const float cx = (float)(min(x + R, Ww - 1) - max(x - R, 0) + 1);
    const float cnt = cx * (float)(min(y + R, Hh - 1) - max(y - R, 0) + 1);
    const float mean = sa / cnt, sqm = sq / cnt;
    float var = (sqm - mean * mean) * (cnt / (cnt - 1.0f));
    var = fmaxf(var, 0.f);
    out[((size_t)b * C + c) * Hh * Ww + (size_t)y * Ww + x] = (win[(ii + R) * PITCH + j + R] - mean) / sqrtf(var + 1.e-6f);
  }
}

// ---------------------------------------------------------------------------------------------------
// PMIM masked L1 (vision_transformer.py:724-729) in PATCH layout: rec[b*L+l][c*P*P + i*P + j] is pixel
// (c, P*py+i, P*px+j) after PixelShuffle.  One block per patch; unmasked patches contribute exactly 0.
// ---------------------------------------------------------------------------------------------------
// `ids` (optional): rec row i holds global patch ids[i] (only masked patches were decoded); otherwise row i is patch i.
__global__ __launch_bounds__(256) void pmim_loss_fwd_kernel(const float* __restrict__ rec, const float* __restrict__ tgt,
                                                            const float* __restrict__ mask, const int32_t* __restrict__ ids,
                                                            float* __restrict__ partial, int L, int gw, int P, int C, int img) {
  __shared__ float red[4];
  const int row = blockIdx.x, patch = ids ? ids[row] : row, b = patch / L, l = patch % L, py = l / gw, px = l % gw;
  const float m = mask[patch];
  float s = 0.f;
  if (m != 0.f) {
    const int PP = P * P;
    for (int o = threadIdx.x; o < C * PP; o += 256) {
      const int c = o / PP, i = (o % PP) / P, j = o % P;
      const float t = tgt[(((size_t)b * C + c) * img + (py * P + i)) * img + px * P + j];
      s += fabsf(t - rec[(size_t)row * C * PP + o]) * m;
    }
  }
  s = ofb_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[row] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = sum(partial) / (sum(mask) + 1e-5) / C ; out[1] = 1 / ((sum(mask)*P*P + 1e-5) * C)  (gradient scale)
__global__ __launch_bounds__(1024) void pmim_loss_final_kernel(const float* __restrict__ partial, const float* __restrict__ mask,
                                                               int nrows, int npatch, int PP, int C, float* __restrict__ out) {
  __shared__ float r1[16], r2[16];
  float s = 0.f, ms = 0.f;
  for (int i = threadIdx.x; i < nrows; i += 1024) s += partial[i];
  for (int i = threadIdx.x; i < npatch; i += 1024) ms += mask[i];
  s = ofb_wave_sum(s); ms = ofb_wave_sum(ms);
  if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s; r2[threadIdx.x >> 6] = ms; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float S = 0.f, M = 0.f;
    for (int k = 0; k < 16; ++k) { S += r1[k]; M += r2[k]; }
    const float denom = (M * (float)PP + 1e-5f) * (float)C;      // mask is upsampled PxP before the sum (:724-725,729)
    out[0] = S / denom;
    out[1] = 1.0f / denom;
  }
}

// drec = upstream * scale * sign(rec - tgt) * m
__global__ __launch_bounds__(256) void pmim_loss_bwd_kernel(const float* __restrict__ rec, const float* __restrict__ tgt,
                                                            const float* __restrict__ mask, const int32_t* __restrict__ ids,
                                                            const float* __restrict__ scale2, const float* __restrict__ upstream,
                                                            float* __restrict__ drec, int L, int gw, int P, int C, int img) {
  const int row = blockIdx.x, patch = ids ? ids[row] : row, b = patch / L, l = patch % L, py = l / gw, px = l % gw;
  const float m = mask[patch];
  const int PP = P * P;
  const float k = scale2[1] * upstream[0] * m;
  for (int o = threadIdx.x; o < C * PP; o += 256) {
    float d = 0.f;
    if (m != 0.f) {
      const int c = o / PP, i = (o % PP) / P, j = o % P;
      const float t = tgt[(((size_t)b * C + c) * img + (py * P + i)) * img + px * P + j];
      const float r = rec[(size_t)row * C * PP + o];
      d = (r > t) ? k : ((r < t) ? -k : 0.f);
    }
    drec[(size_t)row * C * PP + o] = d;
  }
}

// ---------------------------------------------------------------------------------------------------
// label-smoothing cross entropy (timm LabelSmoothingCrossEntropy; search.py:584 via losses.py:38):
// loss = mean_b[(1-s)*(lse - x_y) + s*(lse - mean_c x)]; grad (unscaled by upstream) stored alongside.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ls_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                    float* __restrict__ row_loss, float* __restrict__ grad, int Bn, int Cn,
                                                    float smoothing) {
  __shared__ float red[4];
  __shared__ float bc[2];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* x = logits + (size_t)b * Cn;
  float m = -INFINITY, sx = 0.f;
  for (int c = t; c < Cn; c += 256) { m = fmaxf(m, x[c]); sx += x[c]; }
  m = ofb_wave_max(m);
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f;
  for (int c = t; c < Cn; c += 256) se += expf(x[c] - m);
  se = ofb_wave_sum(se); sx = ofb_wave_sum(sx);
  if ((t & 63) == 0) red[t >> 6] = se;
  __syncthreads();
  if (t == 0) bc[0] = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if ((t & 63) == 0) red[t >> 6] = sx;
  __syncthreads();
  if (t == 0) bc[1] = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  const float lse = m + logf(bc[0]);
  const int y = (int)labels[b];
  if (t == 0) row_loss[b] = (1.0f - smoothing) * (lse - x[y]) + smoothing * (lse - bc[1] / (float)Cn);
  const float invB = 1.0f / (float)Bn;
  for (int c = t; c < Cn; c += 256) {
    const float p = expf(x[c] - lse);
    grad[(size_t)b * Cn + c] = (p - (c == y ? 1.0f - smoothing : 0.f) - smoothing / (float)Cn) * invB;
  }
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  s = ofb_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

// mask[b][l] = 1 if patch l is NOT among the len_keep smallest-noise patches of sample b (0 keep / 1 remove):
// rank by counting == argsort(argsort(noise)) of vision_transformer.py:597-607.
__global__ __launch_bounds__(256) void patch_mask_kernel(const float* __restrict__ noise, float* __restrict__ mask,
                                                         int32_t* __restrict__ masked_ids, int L, int len_keep) {
  extern __shared__ float nz[];
  const int b = blockIdx.x;
  for (int l = threadIdx.x; l < L; l += 256) nz[l] = noise[(size_t)b * L + l];
  __syncthreads();
  for (int l = threadIdx.x; l < L; l += 256) {
    const float v = nz[l];
    int r = 0;
    for (int k = 0; k < L; ++k) r += (nz[k] < v) || (nz[k] == v && k < l);
    mask[(size_t)b * L + l] = (r >= len_keep) ? 1.0f : 0.0f;
    if (masked_ids && r >= len_keep) masked_ids[(size_t)b * (L - len_keep) + (r - len_keep)] = b * L + l;   // global patch id
  }
}

__global__ void scale_by_scalar_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ out,
                                       int64_t n) {
  const float k = s[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = x[i] * k;
}

// index_select along the middle axis of an [outer][n_src][inner] view (compress(): weight / moment slicing)
__global__ __launch_bounds__(256) void index_select_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                           float* __restrict__ dst, int64_t n_src, int64_t n_idx, int64_t inner,
                                                           int64_t total, int32_t* __restrict__ bad) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t k = e % inner, oi = e / inner, i = oi % n_idx, o = oi / n_idx;
    const int32_t j = idx[i];
    if (j < 0 || j >= n_src) {
      if (bad) *bad = 1;
      dst[e] = 0.f;
    } else {
      dst[e] = src[(o * n_src + j) * inner + k];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// multi-tensor AdamW (optim.py:56-120): p *= 1 - lr*wd; m,v EMA; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// ---------------------------------------------------------------------------------------------------
// hyper (optional): device copy of {lr, bc1, 1 / sqrt(bc2)} read instead of the by-value arguments: the form a captured hipGraph
// replays (the host refreshes the three floats before every replay; same values, same arithmetic as the by-value form)
__global__ __launch_bounds__(256) void adamw_kernel(const ofb_adamw_tensor* __restrict__ tab, float lr, float beta1, float beta2,
                                                    float eps, float wd, float bc1, float rsqrt_bc2, const float* __restrict__ hyper,
                                                    const int32_t* __restrict__ skip) {
  if (skip && *skip) return;                            // a non-finite loss was seen (ofb_nonfinite_watch): change nothing
  const ofb_adamw_tensor tt = tab[blockIdx.y];
  if (hyper) { lr = hyper[0]; bc1 = hyper[1]; rsqrt_bc2 = hyper[2]; }
  const float step = lr / bc1, decay = 1.0f - lr * wd;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tt.n; i += (int64_t)gridDim.x * 256) {
    // gradient and moments stream past the caches (next use: a whole step away); the parameter stays (the next forward converts it)
    const float g = OFB_NT_LOAD(tt.g + i);
    const float p = tt.p[i] * decay;
    const float m = OFB_NT_LOAD(tt.m + i) * beta1 + g * (1.0f - beta1);
    const float v = OFB_NT_LOAD(tt.v + i) * beta2 + g * g * (1.0f - beta2);
    OFB_NT_STORE(m, tt.m + i);
    OFB_NT_STORE(v, tt.v + i);
    tt.p[i] = p - step * m / (sqrtf(v) * rsqrt_bc2 + eps);
  }
}

}  // namespace

extern "C" int ofb_embed_assemble_fwd(const float* conv, const float* g, const float* pos, const float* cls, const float* mask_token,
                                      const float* mask, float* tokens, int32_t B, int32_t L, int32_t D, void* stream) {
  if (!conv || !pos || !cls || !tokens || B <= 0 || L <= 0 || D <= 0) return OFB_EINVAL;
  const bool al16 = ofb_aligned16(conv) && ofb_aligned16(pos) && ofb_aligned16(cls) && ofb_aligned16(tokens) && (!g || ofb_aligned16(g)) &&
                    (!mask_token || ofb_aligned16(mask_token));
  if (D % 4 == 0 && D <= 512 && al16)                       // (4 rows x D / 4 float4 items <= 2 x 256 threads)
    hipLaunchKernelGGL(embed_assemble_fwd4_kernel, dim3(ofb_cdiv(B * (L + 1), 4)), dim3(256), 0, (hipStream_t)stream, conv, g, pos, cls,
                       mask_token, mask, tokens, B, L, D);
  else
    hipLaunchKernelGGL(embed_assemble_fwd_kernel, dim3(B * (L + 1)), dim3(256), 0, (hipStream_t)stream, conv, g, pos, cls,
                       mask_token, mask, tokens, B, L, D);
  return ofb_launch_status();
}

extern "C" int32_t ofb_embed_assemble_chunks(int32_t B) { return B >= 16 ? 8 : 1; }

// partial buffers: ppos / pg / pmt each [chunks][(L+1)][D]
extern "C" int ofb_embed_assemble_bwd(const float* dtokens, const float* conv, const float* g, const float* pos, const float* cls,
                                      const float* mask_token, const float* mask, float* dconv, float* ppos, float* pg,
                                      float* pmt, int32_t B, int32_t L, int32_t D, void* stream) {
  if (!dtokens || !conv || !pos || !cls || !dconv || !ppos || !pg || !pmt || B <= 0 || L <= 0 || D <= 0) return OFB_EINVAL;
  const int chunks = ofb_embed_assemble_chunks(B);
  hipLaunchKernelGGL(embed_assemble_bwd_kernel, dim3(L + 1, chunks), dim3(256), 0, (hipStream_t)stream, dtokens, conv, g, pos,
                     cls, mask_token, mask, dconv, ppos, pg, pmt, B, L, D, ofb_cdiv(B, chunks));
  return ofb_launch_status();
}

extern "C" int ofb_norm_targets(const float* imgs, float* out, float* scratch1, float* scratch2, int32_t planes, int32_t Hh,
                                int32_t Ww, int32_t ksize, void* stream) {
  if (!imgs || !out || !scratch1 || !scratch2 || planes <= 0 || Hh <= 0 || Ww <= 0) return OFB_EINVAL;
  if (ksize != 47) return OFB_ELIMIT;                   // the reference hard-codes 47 (vision_transformer.py:727)
  hipStream_t s = (hipStream_t)stream;
  const int64_t n1 = (int64_t)planes * Hh * ((Ww + 7) / 8), n2 = (int64_t)planes * ((Hh + 7) / 8) * Ww;
  ofb_prof_pre(5, s, 16.0 * planes * (double)Hh * Ww);
  hipLaunchKernelGGL(box_h_kernel<47>, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, imgs, scratch1, scratch2, planes, Hh, Ww);
  hipLaunchKernelGGL(box_v_norm_kernel<47>, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, imgs, (const float*)scratch1,
                     (const float*)scratch2, out, planes, Hh, Ww);
  ofb_prof_post(5, s);
  return ofb_launch_status();
}

// rec [B*L][C*P*P] patch layout, targets [B][C][img][img], mask [B*L] in {0,1}; partial [B*L]; out2 = {loss, grad scale}
// norm_targets restricted to the pixels of the listed patches (ids: global patch index b * L + l, L = gw * gw patches of P x P
// pixels per plane): writes only those pixels of out [B*C][H][W]; every other element of out is left untouched.
extern "C" int ofb_norm_targets_masked(const float* imgs, const int32_t* patch_ids, int32_t n_ids, float* out, int32_t B, int32_t C,
                                       int32_t L, int32_t P, int32_t Hh, int32_t Ww, int32_t ksize, void* stream) {
  if (!imgs || !patch_ids || !out || n_ids <= 0 || B <= 0 || C <= 0 || L <= 0 || P <= 0 || Hh <= 0 || Ww <= 0) return OFB_EINVAL;
  if (ksize != 47 || P > 16) return OFB_ELIMIT;         // the reference hard-codes 47 (vision_transformer.py:727); DeiT patches are 16 x 16
  int gw = 1;
  while (gw * gw < L) ++gw;
  if (gw * gw != L || gw * P > Hh || gw * P > Ww) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ofb_prof_pre(5, s, 16.0 * n_ids * C * (double)P * P);
  hipLaunchKernelGGL((norm_targets_masked_kernel<47, 16>), dim3(n_ids, C), dim3(256), 0, s, imgs, patch_ids, out, C, L, gw, P, Hh, Ww);
  ofb_prof_post(5, s);
  return ofb_launch_status();
}

extern "C" int ofb_pmim_loss_fwd(const float* rec, const float* targets, const float* mask, const int32_t* patch_ids,
                                 int32_t n_rows, float* partial, float* out2, int32_t B, int32_t L, int32_t P, int32_t C,
                                 void* stream) {
  if (!rec || !targets || !mask || !partial || !out2 || B <= 0 || L <= 0 || P <= 0 || C <= 0) return OFB_EINVAL;
  if (patch_ids ? n_rows <= 0 : n_rows != B * L) return OFB_EINVAL;
  int gw = 1;
  while (gw * gw < L) ++gw;
  if (gw * gw != L) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(pmim_loss_fwd_kernel, dim3(n_rows), dim3(256), 0, s, rec, targets, mask, patch_ids, partial, L, gw, P, C, gw * P);
  hipLaunchKernelGGL(pmim_loss_final_kernel, dim3(1), dim3(1024), 0, s, (const float*)partial, mask, n_rows, B * L, P * P, C, out2);
  return ofb_launch_status();
}

extern "C" int ofb_pmim_loss_bwd(const float* rec, const float* targets, const float* mask, const int32_t* patch_ids,
                                 int32_t n_rows, const float* out2, const float* upstream, float* drec, int32_t B, int32_t L,
                                 int32_t P, int32_t C, void* stream) {
  if (!rec || !targets || !mask || !out2 || !upstream || !drec || B <= 0 || L <= 0) return OFB_EINVAL;
  if (patch_ids ? n_rows <= 0 : n_rows != B * L) return OFB_EINVAL;
  int gw = 1;
  while (gw * gw < L) ++gw;
  if (gw * gw != L) return OFB_EINVAL;
  hipLaunchKernelGGL(pmim_loss_bwd_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, rec, targets, mask, patch_ids, out2,
                     upstream, drec, L, gw, P, C, gw * P);
  return ofb_launch_status();
}

extern "C" int ofb_ls_cross_entropy(const float* logits, const int64_t* labels, float* row_loss, float* loss, float* grad,
                                    int32_t B, int32_t C, float smoothing, void* stream) {
  if (!logits || !labels || !row_loss || !loss || !grad || B <= 0 || C <= 0) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ls_ce_kernel, dim3(B), dim3(256), 0, s, logits, labels, row_loss, grad, B, C, smoothing);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, (const float*)row_loss, B, loss);
  return ofb_launch_status();
}

// The two readers of the final token stream [B][T = L + 1][D] (vision_transformer.py:735-744): block r < B copies image r's cls row,
// block B + i the token row of masked patch ids[i] (global patch id p = b L + l -> token row p + p / L + 1).  One launch for what was
// three index launches, a strided copy and a gather; the backward zero-fills the stream gradient (a memset node) and scatters both
// kinds of rows (disjoint, unique: plain stores) in one launch.
__global__ __launch_bounds__(128) void token_taps_kernel(const float* __restrict__ src_stream, const float* __restrict__ rows_cls,
                                                         const float* __restrict__ rows_z, const int32_t* __restrict__ ids, int B, int T,
                                                         int D, float* __restrict__ dst_cls, float* __restrict__ dst_z,
                                                         float* __restrict__ dst_stream) {
  const int r = blockIdx.x, L = T - 1;
  size_t srow;                                               // the block's row of the stream
  if (r < B) srow = (size_t)r * T;
  else { const int p = ids[r - B]; srow = (size_t)p + p / L + 1; }
  // gather (dst_stream null): stream row -> compact row; scatter: compact row -> stream row
  if (dst_stream && (r < B ? rows_cls : rows_z) == nullptr) return;      // (scatter without that kind of row: it stays zero)
  const float* s = dst_stream ? (r < B ? rows_cls + (size_t)r * D : rows_z + (size_t)(r - B) * D) : src_stream + srow * D;
  float* d = dst_stream ? dst_stream + srow * D : (r < B ? dst_cls + (size_t)r * D : dst_z + (size_t)(r - B) * D);
  if ((D & 3) == 0 && ofb_aligned16_dev(s) && ofb_aligned16_dev(d)) {
    for (int c = threadIdx.x; c < (D >> 2); c += 128) reinterpret_cast<f32x4*>(d)[c] = reinterpret_cast<const f32x4*>(s)[c];
  } else {
    for (int c = threadIdx.x; c < D; c += 128) d[c] = s[c];
  }
}

extern "C" int ofb_token_taps_fwd(const float* stream_rows, const int32_t* patch_ids, int32_t n_ids, int32_t B, int32_t T, int32_t D,
                                  float* cls_out, float* z_out, void* stream) {
  if (!stream_rows || !cls_out || B <= 0 || T < 2 || D <= 0 || n_ids < 0 || (n_ids > 0 && (!patch_ids || !z_out))) return OFB_EINVAL;
  hipLaunchKernelGGL(token_taps_kernel, dim3(B + n_ids), dim3(128), 0, (hipStream_t)stream, stream_rows, (const float*)nullptr,
                     (const float*)nullptr, patch_ids, B, T, D, cls_out, z_out, (float*)nullptr);
  return ofb_launch_status();
}

extern "C" int ofb_token_taps_bwd(const float* dcls, const float* dz, const int32_t* patch_ids, int32_t n_ids, int32_t B, int32_t T,
                                  int32_t D, float* dstream, void* stream) {
  if (!dstream || B <= 0 || T < 2 || D <= 0 || n_ids < 0 || (dz && n_ids > 0 && !patch_ids)) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(dstream, 0, (size_t)B * T * D * sizeof(float), s) != hipSuccess) return OFB_EINVAL;
  const int n = dz ? n_ids : 0;
  if (!dcls && n == 0) return ofb_launch_status();
  hipLaunchKernelGGL(token_taps_kernel, dim3(B + n), dim3(128), 0, s, (const float*)nullptr, dcls, dz, patch_ids, B, T, D, (float*)nullptr,
                     (float*)nullptr, dstream);
  return ofb_launch_status();
}

// timm DropPath factors of all residual branches in one launch: out[r][b] = floor(keep[r] + u[r][b]) / keep[r]
__global__ void droppath_scales_kernel(const float* __restrict__ u, const float* __restrict__ keep, float* __restrict__ out, int R, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * B) return;
  const float k = keep[i / B];
  out[i] = floorf(k + u[i]) / k;
}

extern "C" int ofb_droppath_scales(const float* u, const float* keep, float* out, int32_t R, int32_t B, void* stream) {
  if (!u || !keep || !out || R <= 0 || B <= 0) return OFB_EINVAL;
  hipLaunchKernelGGL(droppath_scales_kernel, dim3(ofb_cdiv(R * B, 256)), dim3(256), 0, (hipStream_t)stream, u, keep, out, R, B);
  return ofb_launch_status();
}

extern "C" int ofb_patch_mask(const float* noise, float* mask, int32_t* masked_ids, int32_t B, int32_t L, int32_t len_keep,
                              void* stream) {
  if (!noise || !mask || B <= 0 || L <= 0 || len_keep < 0 || len_keep > L) return OFB_EINVAL;
  if (L > 8192) return OFB_ELIMIT;
  hipLaunchKernelGGL(patch_mask_kernel, dim3(B), dim3(256), L * sizeof(float), (hipStream_t)stream, noise, mask, masked_ids, L,
                     len_keep);
  return ofb_launch_status();
}

// the scalar mixing of a search micro-step's losses in one launch (it was ~14 one-element ATen launches between forward and backward)
__global__ void loss_mix_kernel(const float* __restrict__ base, const float* __restrict__ spars3, const float* __restrict__ flops,
                                const float* __restrict__ dec, float w0, float w1, float w2, float w3, float* __restrict__ out3) {
  if (threadIdx.x != 0) return;
  float arch = 0.f;
  if (spars3) arch = (spars3[0] * w0 + spars3[1] * w1) + spars3[2] * w2;
  if (flops) arch += w3 * flops[0];
  const float b = base ? base[0] : 0.f;
  const float coef = (base && dec) ? b / dec[0] : 0.f;
  out3[0] = arch;
  out3[1] = coef;
  out3[2] = dec ? (b + arch) + coef * dec[0] : b + arch;
}

extern "C" int ofb_loss_mix(const float* base, const float* spars3, const float* flops, const float* dec, float w0, float w1, float w2,
                            float w3, float* out3, void* stream) {
  if (!out3) return OFB_EINVAL;
  hipLaunchKernelGGL(loss_mix_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, base, spars3, flops, dec, w0, w1, w2, w3, out3);
  return ofb_launch_status();
}

extern "C" int ofb_scale_by_scalar(const float* x, const float* scalar_dev, float* out, int64_t n, void* stream) {
  if (!x || !scalar_dev || !out || n <= 0) return OFB_EINVAL;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, scalar_dev, out, n);
  return ofb_launch_status();
}

__global__ __launch_bounds__(256) void ema_kernel(const ofb_ema_tensor* __restrict__ tab, float decay, float omd,
                                                  const int32_t* __restrict__ skip) {
  if (skip && *skip) return;
  const ofb_ema_tensor tt = tab[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tt.n; i += (int64_t)gridDim.x * 256)
    tt.ema[i] = __fadd_rn(__fmul_rn(tt.ema[i], decay), __fmul_rn(omd, tt.src[i]));
}

extern "C" int ofb_ema_update(const ofb_ema_tensor* table_dev, int32_t n_tensors, int64_t max_numel, float decay,
                              float one_minus_decay, const int32_t* skip, void* stream) {
  if (!table_dev || n_tensors <= 0 || max_numel <= 0) return OFB_EINVAL;
  int bx = (int)((max_numel + 256 * 8 - 1) / (256 * 8));
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(ema_kernel, dim3(bx, n_tensors), dim3(256), 0, (hipStream_t)stream, table_dev, decay, one_minus_decay, skip);
  return ofb_launch_status();
}

extern "C" int ofb_index_select(const float* src, const int32_t* idx, float* dst, int64_t outer, int64_t n_src, int64_t n_idx,
                                int64_t inner, int32_t* bad, void* stream) {
  if (!src || !idx || !dst || outer <= 0 || n_src <= 0 || n_idx <= 0 || inner <= 0) return OFB_EINVAL;
  const int64_t total = outer * n_idx * inner;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(index_select_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, idx, dst, n_src, n_idx, inner,
                     total, bad);
  return ofb_launch_status();
}

// Host-to-device copy of a small table from PINNED host memory on `stream` (plain hipMemcpyAsync: legal inside a stream capture,
// where it becomes a memcpy node that every replay re-reads from the same host address; torch's own non_blocking copy also
// records allocator events, which a capture does not allow from the autograd thread).
extern "C" int ofb_upload(void* dst_dev, const void* src_pinned, int64_t nbytes, void* stream) {
  if (!dst_dev || !src_pinned || nbytes <= 0) return OFB_EINVAL;
  return (int)hipMemcpyAsync(dst_dev, src_pinned, (size_t)nbytes, hipMemcpyHostToDevice, (hipStream_t)stream);
}

namespace {
int adamw_launch(const ofb_adamw_tensor* table_dev, int n_tensors, int64_t max_numel, float lr, float beta1, float beta2, float eps,
                 float weight_decay, float bc1, float rsqrt_bc2, const float* hyper, const int32_t* skip, hipStream_t s) {
  int bx = (int)((max_numel + 256 * 8 - 1) / (256 * 8));
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  ofb_prof_pre(6, s, 0.0);
  hipLaunchKernelGGL(adamw_kernel, dim3(bx, n_tensors), dim3(256), 0, s, table_dev, lr, beta1, beta2, eps, weight_decay, bc1, rsqrt_bc2,
                     hyper, skip);
  ofb_prof_post(6, s);
  return ofb_launch_status();
}
}  // namespace

extern "C" int ofb_adamw_step(const ofb_adamw_tensor* table_dev, int32_t n_tensors, int64_t max_numel, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int32_t step, const int32_t* skip, void* stream) {
  if (!table_dev || n_tensors <= 0 || max_numel <= 0 || step <= 0) return OFB_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  return adamw_launch(table_dev, n_tensors, max_numel, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), nullptr,
                      skip, (hipStream_t)stream);
}

// The same update with {lr, 1 - beta1^step, 1 / sqrt(1 - beta2^step)} read from device memory (hyper_dev[3]): for a step captured
// in a hipGraph, whose replays must not bake the step count or the learning rate into the launch arguments.
extern "C" int ofb_adamw_step_dev(const ofb_adamw_tensor* table_dev, int32_t n_tensors, int64_t max_numel, const float* hyper_dev,
                                  float beta1, float beta2, float eps, float weight_decay, const int32_t* skip, void* stream) {
  if (!table_dev || !hyper_dev || n_tensors <= 0 || max_numel <= 0) return OFB_EINVAL;
  return adamw_launch(table_dev, n_tensors, max_numel, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, hyper_dev, skip, (hipStream_t)stream);
}

// out[i] = sum_s workspace[s * count + i] (+ out[i]): per-chunk partial buffers of the embed assembly, added in chunk order
namespace {
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int64_t count, float* __restrict__ out,
                                     int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    float s = accumulate ? out[i] : 0.f;
    for (int z = 0; z < splits; ++z) s += ws[(size_t)z * count + i];
    out[i] = s;
  }
}
}  // namespace

extern "C" int ofb_splitk_reduce(const float* workspace, int32_t splits, int64_t count, float* out, int32_t accumulate,
                                 void* stream) {
  if (!workspace || !out || splits <= 0 || count <= 0) return OFB_EINVAL;
  const int blocks = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, workspace, splits, count, out,
                     accumulate);
  return ofb_launch_status();
}

namespace {
__global__ void nonfinite_watch_kernel(const float* __restrict__ v, int n, int32_t* __restrict__ flag) {
  bool bad = false;
  for (int i = threadIdx.x; i < n; i += blockDim.x) bad |= !isfinite(v[i]);
  if (__syncthreads_or(bad) && threadIdx.x == 0) flag[0] += 1;
}
__global__ __launch_bounds__(256) void multi_copy_kernel(const ofb_copy_job* __restrict__ jobs) {
  const ofb_copy_job j = jobs[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < j.n; i += (int64_t)gridDim.x * 256) j.dst[i] = j.src ? j.src[i] : 0.f;
}
}  // namespace

extern "C" int ofb_nonfinite_watch(const float* values, int32_t n, int32_t* flag, void* stream) {
  if (!values || !flag || n <= 0) return OFB_EINVAL;
  hipLaunchKernelGGL(nonfinite_watch_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, values, n, flag);
  return ofb_launch_status();
}

extern "C" int ofb_multi_copy(const ofb_copy_job* jobs_dev, int32_t n_jobs, int64_t max_n, void* stream) {
  if (!jobs_dev || n_jobs <= 0 || n_jobs > 65535 || max_n <= 0) return OFB_EINVAL;
  int bx = (int)((max_n + 256 * 8 - 1) / (256 * 8));
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(multi_copy_kernel, dim3(bx, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  return ofb_launch_status();
}
