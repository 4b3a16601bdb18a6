// Bi-mask gate builder for ALL searchable modules in one launch, its backward, the adaptive one-hot
// (sparsity) loss and the differentiable FLOPs loss.
//   g[h][c] = w_p * sigmoid(score[h][c]) + (1 - w_p) * wm[rank_h[h]][rank_c[h][c]]
//   wm[h][c] = sum_{cells (i,j) on} softmax(alpha)[i][j] * [h < head_thr[i]] * [c < chan_thr[j]]
// Reference: models/layers.py:179-191 (embed), :494-509 (attention), :847-858 (mlp); get_weight :211-216,
// :548-557, :876-881; models/base_model.py:31-86 (losses); vision_transformer.py:759-783 (FLOPs model).
// Ranks are positions in a stable descending sort, computed by counting (tiny rows: <= 3072 entries).
#include "ofb_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// softmax over the "on" cells into LDS p[]; returns via p (0 for off cells). One wave does the work.
__device__ void cell_softmax(const ofb_gate_desc& d, float* p, int t) {
  const int cells = d.A0 * d.A1;
  if (t < 64) {
    const bool on = t < cells && d.on[t];
    const float a = on ? d.alpha[t] : -INFINITY;
    const float m = ofb_wave_max(a);
    const float e = on ? expf(a - m) : 0.f;
    const float s = ofb_wave_sum(e);
    p[t] = e / s;
  }
}

__global__ __launch_bounds__(256) void gates_fwd_kernel(const ofb_gate_desc* __restrict__ descs, int entropy, int var) {
  __shared__ float p[64];
  __shared__ float hs[16];
  __shared__ int rank_h[16];
  __shared__ float red[4];
  // the module's descriptor is copied into LDS first (94 words, one coalesced read): its fields - thresholds inside the cell loops,
  // pointers, shapes - were otherwise re-fetched from global memory wherever they are used, a chain of small dependent round trips
  __shared__ ofb_gate_desc dsh;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  static_assert(sizeof(ofb_gate_desc) % 4 == 0 && sizeof(ofb_gate_desc) / 4 <= 256, "descriptor copy");
  if (t < (int)(sizeof(ofb_gate_desc) / 4)) reinterpret_cast<unsigned*>(&dsh)[t] = reinterpret_cast<const unsigned*>(&descs[blockIdx.y])[t];
  __syncthreads();
  const ofb_gate_desc& d = dsh;
  const int HC = d.H * d.C, cells = d.A0 * d.A1;
  if ((int)blockIdx.x * 256 >= HC) return;
  cell_softmax(d, p, t);
  // head order: descending sum of sigmoid(score) per head (one wave per head row)
  for (int h = w; h < d.H; h += 4) {
    float s = 0.f;
    if (d.H > 1)
      for (int c = lane; c < d.C; c += 64) s += sigmoidf_(d.score[h * d.C + c]);
    s = ofb_wave_sum(s);
    if (lane == 0) hs[h] = s;
  }
  __syncthreads();
  if (t < d.H) {
    int r = 0;
    for (int k = 0; k < d.H; ++k) r += (hs[k] > hs[t]) || (hs[k] == hs[t] && k < t);
    rank_h[t] = r;
  }
  __syncthreads();

  const int e = blockIdx.x * 256 + t;
  // the score rows this block's 256 elements belong to, staged in LDS when they fit (MLP: one row of 1536; attention: four rows of
  // 64): the rank of an element is a count over its whole row, and 1536 dependent global loads per thread made this launch 94 us
  __shared__ float srow[2048];
  const int e0 = blockIdx.x * 256, e1 = min(e0 + 256, HC) - 1;
  const int h0 = e0 / d.C, nrows = e1 / d.C - h0 + 1;
  const bool staged = nrows * d.C <= 2048;
  if (staged)
    for (int i = t; i < nrows * d.C; i += 256) srow[i] = d.score[h0 * d.C + i];
  __syncthreads();
  float sg = 0.f;
  if (e < HC) {
    const int h = e / d.C, c = e % d.C;
    const float sc = d.score[e];
    int rc = 0;
    if (staged && d.C % 64 == 0) {
      // a wave's 64 elements are 64 consecutive columns cw .. cw + 63 of ONE row: left of them every lane counts o >= sc, right of
      // them o > sc - one compare and one add per element; the tie rule with its column test (seven instructions per element, ~11 k
      // per thread on a 1536-wide MLP row: most of this launch's 46 us) is only needed inside the wave's own 64 columns
      const float* row = srow + (h - h0) * d.C;
      const int cw = __builtin_amdgcn_readfirstlane(c), C = d.C;
      int k = 0;
      for (; k < cw; k += 4) {
        const float o0 = row[k], o1 = row[k + 1], o2 = row[k + 2], o3 = row[k + 3];
        rc += (o0 >= sc) + (o1 >= sc) + (o2 >= sc) + (o3 >= sc);
      }
      for (; k < cw + 64; k += 4) {
        const float o0 = row[k], o1 = row[k + 1], o2 = row[k + 2], o3 = row[k + 3];
        rc += ((o0 > sc) || (o0 == sc && k < c)) + ((o1 > sc) || (o1 == sc && k + 1 < c)) + ((o2 > sc) || (o2 == sc && k + 2 < c)) +
              ((o3 > sc) || (o3 == sc && k + 3 < c));
      }
      for (; k < C; k += 4) {
        const float o0 = row[k], o1 = row[k + 1], o2 = row[k + 2], o3 = row[k + 3];
        rc += (o0 > sc) + (o1 > sc) + (o2 > sc) + (o3 > sc);
      }
    } else if (staged) {
      const float* row = srow + (h - h0) * d.C;
      int k = 0;
      for (; k + 4 <= d.C; k += 4) {
        const float o0 = row[k], o1 = row[k + 1], o2 = row[k + 2], o3 = row[k + 3];
        rc += ((o0 > sc) || (o0 == sc && k < c)) + ((o1 > sc) || (o1 == sc && k + 1 < c)) + ((o2 > sc) || (o2 == sc && k + 2 < c)) +
              ((o3 > sc) || (o3 == sc && k + 3 < c));
      }
      for (; k < d.C; ++k) rc += (row[k] > sc) || (row[k] == sc && k < c);
    } else {
      const float* row = d.score + h * d.C;
      for (int k = 0; k < d.C; ++k) {
        const float o = row[k];
        rc += (o > sc) || (o == sc && k < c);
      }
    }
    const int rh = rank_h[h];
    float wm = 0.f, wr = 0.f;
    for (int i = 0; i < d.A0; ++i)
      for (int j = 0; j < d.A1; ++j) {
        const float pij = p[i * d.A1 + j];
        wm += (h < d.head_thr[i] && c < d.chan_thr[j]) ? pij : 0.f;
        wr += (rh < d.head_thr[i] && rc < d.chan_thr[j]) ? pij : 0.f;
      }
    sg = sigmoidf_(sc);
    d.g[e] = d.w_p * sg + (1.0f - d.w_p) * wr;
    d.wr[e] = wr;
    d.wm[e] = wm;
    d.rank[e] = (rh << 16) | rc;
  }
  // per-block partial of sum sigmoid(score) (norm term of the sparsity loss), deterministic
  sg = ofb_wave_sum(sg);
  if (lane == 0) red[w] = sg;
  __syncthreads();
  if (t == 0) d.sig_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];

  if (blockIdx.x == 0 && t < 64) {
    // closed-form sum of the staircase: sum_{ij} p_ij * head_thr[i] * chan_thr[j]
    const bool on = t < cells && d.on[t];
    const float pk = on ? p[t] : 0.f;
    const float ws = ofb_wave_sum(on ? pk * (float)(d.head_thr[t / d.A1] * d.chan_thr[t % d.A1]) : 0.f);
    // adaptive one-hot loss on alpha (base_model.py:58-69): entropy + tan(pi/2 - pi*sigma^2/target)/n
    const int n = (int)ofb_wave_sum(on ? 1.f : 0.f);
    float loss = 0.f, dl_dp = 0.f;
    if (n > 1) {
      const float fn = (float)n, mean = 1.0f / fn, target = 1.0f - 1.0f / fn;
      if (entropy) {
        loss += ofb_wave_sum(on ? -pk * logf(pk) : 0.f);
        dl_dp += on ? -(logf(pk) + 1.0f) : 0.f;
      }
      if (var) {
        const float dev = on ? pk - mean : 0.f;
        const float x = ofb_wave_sum(dev * dev) / target;
        const float ang = 1.57079632679489661923f - 3.14159265358979323846f * x;
        loss += tanf(ang) / fn;
        const float cs = cosf(ang);
        dl_dp += (-3.14159265358979323846f / (cs * cs)) * (2.0f * dev / target) / fn;
      }
    }
    const float dot = ofb_wave_sum(on ? pk * dl_dp : 0.f);
    if (t < cells) {
      d.prob[t] = pk;
      d.dloss_dalpha[t] = (on && n > 1) ? pk * (dl_dp - dot) : 0.f;
    }
    if (t == 0) { d.wsum[0] = ws; d.loss_alpha[0] = loss; }
  }
}

// One block per module.  Upstream: dg (always), dwr / dwm (optional, elementwise), dwsum[m], dspars[m] scalars.
__global__ __launch_bounds__(256) void gates_bwd_kernel(const ofb_gate_desc* __restrict__ descs,
                                                        const ofb_gate_grad* __restrict__ grads) {
  __shared__ float p[64];
  __shared__ float dp[64];
  __shared__ ofb_gate_desc dsh;                              // (descriptor and gradient table entry in LDS: see gates_fwd_kernel)
  __shared__ ofb_gate_grad gsh;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  static_assert(sizeof(ofb_gate_grad) % 4 == 0 && sizeof(ofb_gate_grad) / 4 <= 256, "gradient entry copy");
  if (t < (int)(sizeof(ofb_gate_desc) / 4)) reinterpret_cast<unsigned*>(&dsh)[t] = reinterpret_cast<const unsigned*>(&descs[blockIdx.x])[t];
  if (t < (int)(sizeof(ofb_gate_grad) / 4)) reinterpret_cast<unsigned*>(&gsh)[t] = reinterpret_cast<const unsigned*>(&grads[blockIdx.x])[t];
  __syncthreads();
  const ofb_gate_desc& d = dsh;
  const ofb_gate_grad& gr = gsh;
  const int HC = d.H * d.C, cells = d.A0 * d.A1;
  if (t < 64) p[t] = (t < cells) ? d.prob[t] : 0.f;
  int n_on = 0;
  for (int k = 0; k < cells; ++k) n_on += d.on[k];
  // a module with a single live cell has left the loss (base_model.py:56-57)
  const float dsp = (gr.dspars && n_on > 1) ? gr.dspars[0] : 0.f;   // d(total)/d(this module's sparsity loss)
  const float dws = gr.dwsum ? gr.dwsum[0] : 0.f;     // d(total)/d(sum of staircase)
  // dscore: through sigmoid in the gate and in the norm term
  for (int e = t; e < HC; e += 256) {
    const float sg = sigmoidf_(d.score[e]);
    const float dgv = gr.dg ? gr.dg[e] : 0.f;
    gr.dscore[e] = (dgv * d.w_p + dsp * d.norm_coef) * sg * (1.0f - sg);
  }
  // dp[i][j]: one wave per cell, lanes sweep the elements
  for (int cell = w; cell < cells; cell += 4) {
    const int i = cell / d.A1, j = cell % d.A1;
    const int ht = d.head_thr[i], ct = d.chan_thr[j];
    float s = 0.f;
    if (d.on[cell]) {
      // four sweeps' loads in flight together (one sweep at a time this loop was 24 dependent round trips on a 1536-wide module);
      // the sum keeps its order: element e, then e + 64, ...
      const float wq = 1.0f - d.w_p;
      const int C = d.C;
      const float* dg = gr.dg; const float* dwr = gr.dwr; const float* dwm = gr.dwm; const int* rank = d.rank;
      int e = lane;
      for (; e + 192 < HC; e += 256) {
        int rk[4]; float a[4], b[4], m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int eu = e + 64 * u;
          rk[u] = rank[eu];
          a[u] = dg ? dg[eu] : 0.f; b[u] = dwr ? dwr[eu] : 0.f; m[u] = dwm ? dwm[eu] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int eu = e + 64 * u, rh = rk[u] >> 16, rc = rk[u] & 0xffff;
          const float up = (dg ? wq * a[u] : 0.f) + b[u];
          s += (rh < ht && rc < ct) ? up : 0.f;
          if (dwm) s += ((eu / C) < ht && (eu % C) < ct) ? m[u] : 0.f;
        }
      }
      for (; e < HC; e += 64) {
        const int rk = rank[e], rh = rk >> 16, rc = rk & 0xffff;
        float up = (dg ? wq * dg[e] : 0.f) + (dwr ? dwr[e] : 0.f);
        s += (rh < ht && rc < ct) ? up : 0.f;
        if (dwm) s += ((e / C) < ht && (e % C) < ct) ? dwm[e] : 0.f;
      }
    }
    s = ofb_wave_sum(s);
    if (lane == 0) dp[cell] = s + dws * (float)(ht * ct);
  }
  __syncthreads();
  if (t < 64) {
    const bool on = t < cells && d.on[t];
    const float dot = ofb_wave_sum(on ? p[t] * dp[t] : 0.f);
    if (t < cells) gr.dalpha[t] = on ? p[t] * (dp[t] - dot) + dsp * d.dloss_dalpha[t] : 0.f;
  }
}

// Sums per-module sparsity losses by kind (0 attn, 1 mlp, 2 embed): out[kind] (base_model.py:80-85).
__global__ void spars_finalize_kernel(const ofb_gate_desc* __restrict__ descs, int n, int norm, float* __restrict__ out,
                                      float* __restrict__ per_module) {
  // one thread per module (their loads run side by side: a single thread walking 25 modules took 35 us of dependent round trips),
  // then thread 0 adds the per-module values by kind in module order (the same order as before: deterministic, same rounding)
  __shared__ float lm[256];
  __shared__ int km[256];
  for (int m = threadIdx.x; m < n; m += blockDim.x) {
    const ofb_gate_desc& d = descs[m];
    int on = 0;
    for (int k = 0; k < d.A0 * d.A1; ++k) on += d.on[k];
    float l = 0.f;
    if (on > 1) {
      l = d.loss_alpha[0];
      if (norm) {
        float s = 0.f;
        const int nb = (d.H * d.C + 255) / 256;
        for (int b = 0; b < nb; ++b) s += d.sig_partial[b];
        l += s * d.norm_coef;
      }
    }
    per_module[m] = l;
    if (m < 256) { lm[m] = l; km[m] = d.kind; }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int m = 0; m < n; ++m) {
    const float l = m < 256 ? lm[m] : per_module[m];
    const int kind = m < 256 ? km[m] : descs[m].kind;
    acc[kind] += l;
  }
  out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
}

// FLOPs (MAC) model of vision_transformer.py:759-783 with e = W(0), (sd_l, hid_l) = W(1+2l), W(2+2l), where W(s) is the
// live staircase sum wsum[live_slot[s]] or, for a module compress() has finished, the constant wconst[s].
// One wave.  The 1 + 2 depth staircase sums, their slots and the head counts are fetched side by side into LDS, lane 0 evaluates the
// model in double precision out of LDS (the same operations in the same order as when it walked global memory: 25 dependent loads
// through the slot table, then 25 store -> load round trips to scale its own provisional gradients - 12 us), and the lanes write
// the scaled gradients together.
constexpr int FL_MAXS = 160;                                // slots held in LDS (depth <= 79; deeper models take the serial path)
__global__ __launch_bounds__(64) void flops_loss_kernel(const float* __restrict__ wsum, ofb_flops_cfg c, float* __restrict__ out,
                                                        float* __restrict__ dwsum) {
  __shared__ double Wsh[FL_MAXS];
  __shared__ int slot_sh[FL_MAXS];
  __shared__ float aH_sh[FL_MAXS / 2], dprov[FL_MAXS];
  __shared__ double kde[2];
  const int t = threadIdx.x, ns = 1 + 2 * c.depth;
  const bool lds = ns <= FL_MAXS;
  if (lds) {
    for (int s = t; s < ns; s += 64) {
      const int j = c.live_slot ? c.live_slot[s] : s;
      slot_sh[s] = j;
      Wsh[s] = (double)(j >= 0 ? wsum[j] : c.wconst[s]);
    }
    for (int l = t; l < c.depth; l += 64) aH_sh[l] = c.active_heads ? (float)c.active_heads[l] : (float)c.num_heads;
    for (int j = t; j < c.n_live; j += 64) dprov[j] = 0.f;
  }
  __syncthreads();
  if (t == 0) {
    // n: the ACTIVE patch count of the searched model (vision_transformer.py:768: weighted_mask.sum() once a patch-cell compress()
    // has produced it - a probability-weighted count, so a float - else the full patch count)
    const double N = c.num_patches, n = c.active_patches ? (double)c.active_patches[0] : (double)c.num_patches, D = c.embed_dim, H = c.num_heads, dh = c.head_dim, hid = c.hidden,
                 P2 = c.patch_area, ncls = c.num_classes, Dln = c.ln_dim > 0 ? c.ln_dim : c.embed_dim;
    auto slot = [&](int s) { return lds ? slot_sh[s] : (c.live_slot ? c.live_slot[s] : s); };
    auto W = [&](int s) { if (lds) return Wsh[s]; const int j = slot(s); return (double)(j >= 0 ? wsum[j] : c.wconst[s]); };
    auto prov = [&](int j, float v) { if (lds) dprov[j] = v; else dwsum[j] = v; };
    const double e = W(0);
    double total = N * D * 3.0 * P2, searched = N * e * 3.0 * P2, de = N * 3.0 * P2, dn = 0.0;
    for (int l = 0; l < c.depth; ++l) {
      const double sd = W(1 + 2 * l), hh = W(2 + 2 * l);
      const double aH = lds ? (double)aH_sh[l] : (c.active_heads ? (double)c.active_heads[l] : H);
      total += 2.0 * D * N;
      searched += 2.0 * Dln * n;
      dn += 2.0 * Dln;
      total += N * (H * dh * 3.0 * H * dh) + 3.0 * N * H * dh + H * N * dh * N + H * N * N + 5.0 * H * N * N + H * N * N * dh +
               N * (H * dh * H * dh) + N * H * dh;
      searched += n * (e * 3.0 * sd) + 3.0 * n * sd + n * n * sd + aH * n * n + 5.0 * aH * n * n + n * n * sd + n * (sd * e) + n * e;
      const double dsd = n * e * 3.0 + 3.0 * n + n * n + n * n + n * e;
      dn += e * 3.0 * sd + 3.0 * sd + 2.0 * n * sd + 2.0 * aH * n + 10.0 * aH * n + 2.0 * n * sd + sd * e + e;
      de += n * 3.0 * sd + n * sd + n;
      total += (2.0 * D * hid + D + hid) * N;
      searched += (e * hh + hh * e + e + hh) * n;
      dn += e * hh + hh * e + e + hh;
      const double dhh = (2.0 * e + 1.0) * n;
      de += (2.0 * hh + 1.0) * n;
      const int ja = slot(1 + 2 * l), jm = slot(2 + 2 * l);
      if (ja >= 0) prov(ja, (float)dsd);   // provisional: scaled below
      if (jm >= 0) prov(jm, (float)dhh);
    }
    total += D * ncls;
    searched += e * ncls;
    de += ncls;
    total /= 1e9;
    searched /= 1e9;
    const double diff = (searched - (double)c.target) / total;
    const double k = 2.0 * diff / total / 1e9;      // d loss / d searched_raw
    out[0] = (float)(diff * diff);
    out[1] = (float)total;
    out[2] = (float)searched;
    out[3] = (float)(k * dn);                       // d loss / d active_patches
    kde[0] = k; kde[1] = de;
    if (!lds) {
      const int je = slot(0);
      for (int j = 0; j < c.n_live; ++j)
        if (j != je) dwsum[j] = (float)(k * (double)dwsum[j]);
      if (je >= 0) dwsum[je] = (float)(k * de);
    }
  }
  __syncthreads();
  if (lds) {
    const double k = kde[0], de = kde[1];
    const int je = slot_sh[0];
    for (int j = t; j < c.n_live; j += 64) dwsum[j] = (j == je) ? (float)(k * de) : (float)(k * (double)dprov[j]);
  }
}

}  // namespace

extern "C" int ofb_gates_fwd(const ofb_gate_desc* descs_dev, int32_t n_modules, int32_t max_elems, int32_t entropy,
                             int32_t var, int32_t norm, float* spars_out, float* spars_per_module, void* stream) {
  if (!descs_dev || n_modules <= 0 || max_elems <= 0 || !spars_out || !spars_per_module) return OFB_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gates_fwd_kernel, dim3(ofb_cdiv(max_elems, 256), n_modules), dim3(256), 0, s, descs_dev, entropy, var);
  hipLaunchKernelGGL(spars_finalize_kernel, dim3(1), dim3(64), 0, s, descs_dev, n_modules, norm, spars_out, spars_per_module);
  return ofb_launch_status();
}

extern "C" int ofb_gates_bwd(const ofb_gate_desc* descs_dev, const ofb_gate_grad* grads_dev, int32_t n_modules, void* stream) {
  if (!descs_dev || !grads_dev || n_modules <= 0) return OFB_EINVAL;
  hipLaunchKernelGGL(gates_bwd_kernel, dim3(n_modules), dim3(256), 0, (hipStream_t)stream, descs_dev, grads_dev);
  return ofb_launch_status();
}

extern "C" int ofb_flops_loss(const float* wsum, const ofb_flops_cfg* cfg, float* out4, float* dwsum, void* stream) {
  if (!cfg || !out4 || cfg->depth <= 0 || cfg->n_live < 0 || cfg->n_live > 1 + 2 * cfg->depth) return OFB_EINVAL;
  if (cfg->n_live > 0 && (!wsum || !dwsum)) return OFB_EINVAL;
  if ((cfg->live_slot == nullptr) != (cfg->wconst == nullptr)) return OFB_EINVAL;
  if (!cfg->live_slot && cfg->n_live != 1 + 2 * cfg->depth) return OFB_EINVAL;
  hipLaunchKernelGGL(flops_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, wsum, *cfg, out4, dwsum);
  return ofb_launch_status();
}
