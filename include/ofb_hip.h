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

#ifdef __cplusplus
}
#endif
#endif /* OFB_HIP_H */
