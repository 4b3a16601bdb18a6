/*
 * ofb_hip.h — C ABI of libofb_hip.so, the MI355X (gfx950) implementation of the
 * Once-for-Both search-training hot path.
 *
 * The reference (HankYe/Once-for-Both) is pure Python and has no FFI; its boundary for this
 * path is the PyTorch module API.  Each entry point below replaces a group of
 * ATen calls made by one reference function (cited per entry as file:line into the reference
 * tree) and is what a reference-side ctypes binding would call.
 *
 * Conventions: plain device pointers + sizes, fp32 row-major unless stated, no ownership taken,
 * re-entrant, launches on the caller's `stream` (hipStream_t passed as void*), returns 0 on
 * success, a negative OFB_E* code for rejected arguments, or a positive hipError_t.
 */
#ifndef OFB_HIP_H
#define OFB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFB_OK 0
#define OFB_EINVAL (-1)   /* bad shape / null pointer / misaligned operand */
#define OFB_ELIMIT (-2)   /* shape outside what the kernel supports (see each entry) */

#define OFB_ACT_NONE 0
#define OFB_ACT_GELU 1      /* aux <- pre-activation (if aux), C <- gelu_erf(pre)            */
#define OFB_ACT_DGELU 2     /* C <- value * gelu_erf'(aux[m][n])                              */

/* ---------------------------------------------------------------------------------------------
 * Dense contraction on f32-input MFMA (v_mfma_f32_32x32x2_f32): C[M,N] = A[M,K] * B[K,N], then
 *   v = alpha*acc (+bias[n]) (*colscale[n]); act; (*rowscale[m / rs_div]); (+resid[m*ldr+n]).
 * a_kc / b_kc = 1: operand stored K-contiguous (A[m*lda+k], B[n*ldb+k]); 0: stored
 * MN-contiguous (A[k*lda+m], B[k*ldb+n]).  So (1,1) is x @ W^T (nn.Linear forward,
 * models/layers.py:491,515,845,863; Conv2d patch embed :177; decoder 1x1 conv
 * vision_transformer.py:723; head :744), (1,0) is dY @ W (input gradient) and (0,0) is
 * dY^T @ X (weight gradient, reduction over tokens) of the same Linear layers.
 * kscale (a_kc == 0 only): A's reduction rows are scaled by kscale[k / ks_div] (per-sample
 * DropPath factor inside a weight gradient).
 * split_k > 1: grid.z slices K; raw partial sums go to workspace[split_k][M][N] and the epilogue
 * is skipped — follow with ofb_splitk_reduce.
 * ------------------------------------------------------------------------------------------- */
typedef struct ofb_gemm_args {
  const float* A; const float* B; float* C;
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  int32_t a_kc, b_kc;
  float alpha;
  const float* bias;
  const float* colscale;
  const float* rowscale; int32_t rs_div;
  const float* resid; int32_t ldr;
  float* aux; int32_t ldaux;
  int32_t act;
  const float* kscale; int32_t ks_div;
  int32_t split_k; float* workspace;
} ofb_gemm_args;

int ofb_gemm_f32(const ofb_gemm_args* args, void* stream);

/* out[i] = sum_s workspace[s*count + i] (+ out[i] if accumulate) */
int ofb_splitk_reduce(const float* workspace, int32_t splits, int64_t count, float* out, int32_t accumulate,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-launch HIP-event timing on the launch stream (used by bench.py for the roofline object).
 * Tags: 0 gemm, 1 attention fwd, 2 layernorm fwd, 3 layernorm bwd, 4 attention bwd.
 * ofb_prof_collect synchronises the recorded events and fills out[tag*3 + {0 launches, 1 ms, 2 work}].
 * ------------------------------------------------------------------------------------------- */
int ofb_prof_enable(int32_t on);
int ofb_prof_collect(double* out, int32_t ntags);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (eps inside the sqrt), one wavefront per token row; D <= 1024.
 * Replaces F.layer_norm in models/layers.py:96-98 (LayerNorm.forward) as used by MAEBlock
 * (models/vision_transformer.py:193-204) and the final norm (:663-668).
 * bwd: dx = rstd*(dy*g - mean(dy*g) - xhat*mean(dy*g*xhat)) (+ dres); per-block partial
 * dgamma/dbeta go to partials[ofb_layernorm_bwd_blocks(rows)][2][D] (sum them with ofb_colsum).
 * ------------------------------------------------------------------------------------------- */
int ofb_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                      int32_t rows, int32_t D, float eps, void* stream);
int32_t ofb_layernorm_bwd_blocks(int32_t rows);
int ofb_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                      const float* dres, float* dx, float* partials, int32_t rows, int32_t D, void* stream);

/* out[N] = column sums of x[M][ld] (optionally rows scaled by rowscale[m / rs_div]): bias gradients of every
 * Linear on the path.  scratch: ofb_colsum_slabs(M, N) * N floats. */
int32_t ofb_colsum_slabs(int32_t M, int32_t N);
int ofb_colsum(const float* x, int32_t ld, int32_t M, int32_t N, const float* rowscale, int32_t rs_div, float* out,
               float* scratch, void* stream);

/* Bi-mask gate folded into a Linear layer (q,k,v *= g: models/layers.py:507-509; fc1 out *= g: :858;
 * conv out *= g: :191).  out = g[n] * W[n][:]; and the matching backward: from the UNGATED raw gradients
 * dWraw = dY^T x, dbraw = colsum(dY): dW = g*dWraw, db = g*dbraw, dg[n] = <dWraw[n], W[n]> + dbraw[n]*b[n]. */
int ofb_scale_rows(const float* W, const float* g, float* out, int32_t N, int32_t K, void* stream);
int ofb_gate_fold_bwd(const float* dWraw, const float* W, const float* g, const float* dbraw, const float* b, float* dW,
                      float* db, float* dg, int32_t N, int32_t K, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention core softmax(q k^T * scale) v with probabilities kept on chip
 * (models/layers.py:510-514; plain Attention.forward :387-391 for the finetune path).
 * qkv: [B*N][3*H*dh] exactly as the qkv Linear writes it (q | k | v, head-major); out: [B*N][H*dh]
 * (the transpose(1,2).reshape of :514 is folded into the store); lse: [B*H][N] row log-sum-exp.
 * Limits: N <= 224, dh <= 64, dh % 4 == 0 (covers DeiT-T/S/B and every pruned d' in {16,24,..,64}).
 * bwd writes dqkv in the same packing (dq | dk | dv).
 * ------------------------------------------------------------------------------------------- */
int ofb_attention_fwd(const float* qkv, float* out, float* lse, int32_t B, int32_t N, int32_t H, int32_t dh, float scale,
                      void* stream);
int ofb_attention_bwd(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv, int32_t B,
                      int32_t N, int32_t H, int32_t dh, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OFB_HIP_H */
