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
/* the forward saves the DERIVATIVE (it has Phi and phi in hand already), the backward epilogue is one multiply */
#define OFB_ACT_GELU_GRAD 3 /* aux <- gelu_erf'(pre) (aux required), C <- gelu_erf(pre)       */
#define OFB_ACT_MULAUX 4    /* C <- value * aux[m][n]                                         */
/* the same pair with aux in "T-layout" (ofb_gemm_h_aux_t_floats(M, N) floats, 16-byte aligned; ldaux unused): the saved derivative only
 * ever travels from the epilogue of the product that computes gelu(..) to the epilogue of the product that forms the gradient of the
 * pre-activation - two ofb_gemm_h launches with the same [M][N] - so it may be stored the way a wave of the 128 x 192 tile holds its
 * accumulators (csrc/gemm_h.hip: aux_t_index).  With an H-format-only output and no row scale / residual these forms take the DIRECT
 * epilogue: planes and aux move straight between the accumulator registers and memory (no LDS pass) and the next tile's first
 * stages are requested underneath.  The layout is private to ofb_gemm_h: pass the buffer from one launch to the other, nothing else. */
#define OFB_ACT_GELU_GRAD_T 5
#define OFB_ACT_MULAUX_T 6

/* ---------------------------------------------------------------------------------------------
 * Dense f32-class contraction on the f16 matrix pipe (csrc/gemm_h.hip, csrc/hformat.h):
 *   C[M,N] = A[M,K] * B[K,N], then  v = alpha*acc (+bias[n]) (*colscale[n]); act; (*rowscale[m / rs_div]); (+resid[m*ldr+n]).
 * Every operand value is held as TWO f16 numbers of a power-of-two scaled copy, X 2^e = h1 + h2 ("H-format": two 11-bit
 * significands, the second signed against the first: 23 significant bits, |X 2^e - h1 - h2| <= 2^-23 |X 2^e|), and every product is
 * three v_mfma_f32_32x32x16_f16 terms (h2 h1, h1 h2, h1 h1; the dropped h2 h2 is a zero-mean 2^-25 of the product in RMS, <= 2^-22
 * worst case) with f32 accumulation: per-product error ~2^-24 RMS (f32's own product rounding), measured against fp64 3-10x below a
 * k-ordered f32 fma chain on every operand class of tests/test_gpu_accuracy_class.py.  (Rounds 1-3 used an exact three-way bf16 split with six
 * terms: twice the matrix-pipe work and 1.5x the operand bytes - on a power-limited chip 1.4-1.7x the time,
 * profiles/r04_gemm_h_vs_p_step_shapes_v1.txt; scripts/lab/gemm_p_bf16x3_round3.hip.)
 * The operands arrive ALREADY split: the split is done ONCE by the producer of each tensor (LayerNorm, attention, GEMM epilogues,
 * ofb_to_hformat* for everything else) instead of inside every GEMM tile that touches it, and the K loop is LDS-DMA + fragment reads
 * + MFMAs only.  Every nn.Linear / Conv2d-as-GEMM of the path and its autograd runs here: x @ W^T (models/layers.py:491,515,845,863;
 * patch embed :177; decoder 1x1 conv vision_transformer.py:723; head :744), dY @ W (input gradient) and dY^T @ X (weight gradient,
 * reduction over tokens).
 * Scheduling is hybrid stream-K over persistent workgroups (csrc/gemm_plan.h): tiles that do not fill a whole round of
 * workgroups are cut along K and summed in a fixed order by a fix-up launch (deterministic).  `workspace` must hold
 * ofb_gemm_h_workspace_bytes(args) bytes (0 when no tile is streamed); successive calls on ONE stream may share it.
 *
 * H-format of X[R][C] (ofb_hformat_bytes(R, C) bytes): [256-B header {int32 e; f32 amax, rn2sq, cn2sq}][granules of 4 rows x 16
 * columns, 256 B, stored [ceil(R/16)*4][ncb = ceil(C/16)]; a granule holds [plane h1|h2][c % 16][r % 4] f16].  Rows >= R / columns
 * >= C inside the last granules are ZERO (the reduction axis relies on it); tile-granular reads run past the matrix into slack that
 * is never initialised and only reaches accumulators that are not stored.  The header is written and read on the DEVICE only: e is
 * chosen by the producer from an upper bound b >= max|X| so that b 2^e lies in [2^14, 2^15); elements >= 2^-18 b keep the full
 * relative accuracy, smaller ones an absolute accuracy of 2^-39 b.  amax / rn2sq / cn2sq (bounds of max|X| and of the largest
 * squared row / column 2-norm; 0 = unknown) feed the Cauchy-Schwarz bound with which a GEMM that WRITES H-format chooses its output's
 * exponent before its first tile is finished.
 * a_kc / b_kc = 1: the reduction runs along the columns C of that operand's matrix (x[M][K], W[N][K]: nn.Linear forward,
 * models/layers.py:491,515,845,863); 0: along its rows R (W[K..][N] in dY @ W; dY[tokens][N], x[tokens][K] in dY^T @ x).
 * So the SAME H-format copy of an activation or weight feeds its forward, input-gradient and weight-gradient products.
 * Output: f32 C (ldc) and / or H-format Cp ([R = M][C = N], c_ncb granule columns) - e.g. gelu(fc1) leaves the kernel as the
 * H-format operand of fc2 plus the f32 gelu'(pre-activation) in aux.  An H-format output needs its exponent first: a one-block
 * pre-kernel bounds |output| from the operand headers and the epilogue inputs (bias, colscale, rowscale are scanned; aux_bound
 * bounds |aux| of the multiplying activations, default 1.13 = max gelu'); out_bound (device scalar) overrides that bound and is
 * required with resid (for an H-format output AND for cbound_out: the analytic bound does not see the residual).  cbound_out (optional
 * device scalar) receives the bound of an f32 output for the consumer that will split
 * it (attention).  colpart: optional [ofb_gemm_h_colpart_rows(args)][N] partial column sums of the OUTPUT (one row per 128-row
 * tile row; 32 rows per tile row that runs in the streamed tail), each summed in a fixed order: the bias gradient of an
 * H-format-only result, e.g. d(pre-activation) of fc1; add the rows with ofb_colsum.
 * ------------------------------------------------------------------------------------------- */
/* one conversion job of ofb_to_hformat_multi: X (f32 [R][C], row stride ld) -> P (ofb_hformat_bytes(R, C) bytes), row r
 * optionally multiplied by rowscale[r] */
typedef struct ofb_hformat_job { const float* X; void* P; const float* rowscale; int32_t R, C, ld, pad_; } ofb_hformat_job;
typedef struct ofb_gemm_h_args {
  const void* A; const void* B;
  int32_t a_kc, b_kc;
  int32_t a_ncb, b_ncb;
  int32_t M, N, K;
  float* C; int32_t ldc;
  void* Cp; int32_t c_ncb;
  float alpha;
  const float* bias;
  const float* colscale;
  const float* rowscale; int32_t rs_div;
  const float* resid; int32_t ldr;
  float* aux; int32_t ldaux;
  int32_t act;
  float* workspace; int64_t workspace_bytes;
  float* colpart;
  float aux_bound;
  const float* out_bound;
  float* cbound_out;
  /* optional (all three or none): rn_out[ofb_gemm_h_rn_tiles(args)] receives, per output tile, the maximum over the tile's rows m of
   * rn_rowfac[m] * | rn_gamma[n] * C[m][n], n in the tile |_2 - with rn_gamma / rn_rowfac = the weight and the saved 1 / std of the
   * LayerNorm whose backward consumes this output (an input gradient), the bound of that backward's result is
   * sqrt(column tiles) * max(rn_out): ofb_layernorm_bwd_h_rn takes it instead of running its own pass over the gradient */
  const float* rn_gamma;
  const float* rn_rowfac;
  float* rn_out;
} ofb_gemm_h_args;
int32_t ofb_gemm_h_rn_tiles(const ofb_gemm_h_args* args, int32_t* col_tiles);   /* entries of rn_out (0: this shape should not ask for it); *col_tiles = tiles along N */
int32_t ofb_gemm_h_colpart_rows(const ofb_gemm_h_args* args);
int64_t ofb_gemm_h_aux_t_floats(int32_t M, int32_t N);       /* floats of a T-layout aux tensor for an [M][N] output (OFB_ACT_*_T) */
int64_t ofb_hformat_bytes(int32_t R, int32_t C);
/* bound (optional device scalar >= max |X * rowscale|): skips the statistics pass that otherwise measures amax / row norms first */
int ofb_to_hformat(const float* X, int32_t R, int32_t C, int32_t ld, void* P, const float* rowscale, int32_t rs_div, const float* bound,
                   void* stream);
int ofb_patchify_hformat(const float* img, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t patch, void* P, void* stream);
/* scratch: n_jobs * 64 floats (two-stage maxima of every job) */
int ofb_to_hformat_multi(const ofb_hformat_job* jobs_dev, int32_t n_jobs, int32_t max_R, int32_t max_C, float* scratch, void* stream);
int ofb_to_hformat_colsum(const float* X, int32_t R, int32_t C, int32_t ld, void* P, const float* rowscale, int32_t rs_div,
                          float* partial, const float* bound, void* stream);
/* the same with the bound given as n_bound (>= 1) device floats whose MAXIMUM bounds |X * rowscale|: the per-workgroup maxima that
 * ofb_attention_bwd_wgmax leaves (every block of the conversion reduces the same words in the same order; no atomics, no memset node) */
int ofb_to_hformat_colsum_nb(const float* X, int32_t R, int32_t C, int32_t ld, void* P, const float* rowscale, int32_t rs_div,
                             float* partial, const float* bound, int32_t n_bound, void* stream);
int ofb_from_hformat(const void* P, int32_t R, int32_t C, float* X, int32_t ld, void* stream);   /* (h1 + h2) 2^-e */
int32_t ofb_colsum_h_slabs(int32_t R);
int ofb_colsum_h(const void* P, int32_t R, int32_t C, float* partial, void* stream);
int64_t ofb_gemm_h_workspace_bytes(const ofb_gemm_h_args* args);
int ofb_gemm_h(const ofb_gemm_h_args* args, void* stream);
/* Run-time switches of the GEMM (same-process A/B of kernel variants; results are identical to rounding, only speed differs).
 * Unset keys take the environment variable named below, read once, else the default.  Not thread-safe against running calls. */
#define OFB_TUNE_GEMM_MFMA 0   /* OFB_GEMM_H_MFMA: 16 (default) = v_mfma_f32_16x16x32_f16, 32 = v_mfma_f32_32x32x16_f16 on the 128 x 192 tile */
#define OFB_TUNE_GEMM_SCHED 1  /* OFB_GEMM_H_SPREAD: 2 (default) = 1 + a launch of ONE partial round gives every XCD an equal share of its tiles; 1 = a partial last round of a multi-round launch is spread over all XCDs; 0 = contiguous */
#define OFB_TUNE_GEMM_TILE 2   /* OFB_GEMM_H_TILE: 0 (default) = the 128 x 192 tile everywhere, 96 = the 256 x 96 tile wherever it is legal, 97 = by the
                                  padded-columns model of rounds 3-4, 128 = forced */
#define OFB_TUNE_GEMM_T112 3   /* OFB_GEMM_H_T112: 1 = the 112 x 192 tile for token-row products whose 128-row tiles fill between half a round and one round; 0 (default) = off: measured slower, profiles/r05_gemm_tile_112.txt */
#define OFB_TUNE_GEMM_YIELD 4  /* OFB_GEMM_H_YIELD: n in 1..7 = in single-round launches the first-dispatched workgroup of a CU that holds two sleeps 128 n cycles at each stage hand-over (default 4; 0 = off) */
#define OFB_TUNE_GEMM_DIRECT 5 /* OFB_GEMM_H_DIRECT: 1 (default) = the T-layout activation forms take the direct epilogue on interior tiles; 0 = every tile parks in LDS (same results) */
#define OFB_TUNE_GEMM_CUS 6    /* OFB_GEMM_H_CUS: n > 0 = the GEMM plans its persistent workgroups for n CUs instead of all of them (a data-parallel job whose exchange kernels hold CUs during backward: ofb_amd.dp); 0 (default) = every CU */
#define OFB_TUNE_COUNT 7
/* A value set here - 0 included - beats the environment variable of the same key.  OFB_EINVAL for a value outside the key's set
 * (MFMA: 16 | 32; SCHED: 0..2; TILE: 0 | 96 | 97 | 128; T112: 0 | 1; YIELD: 0..32; DIRECT: 0 | 1; CUS: 0..1024).  Process-wide and unsynchronised: do not change a
 * switch between ofb_gemm_h_rn_tiles / ofb_gemm_h_workspace_bytes and the ofb_gemm_h call they size buffers for (the tile choice
 * decides both), nor from a second thread while GEMMs are being launched. */
int ofb_tune(int32_t key, int32_t value);

/* out[i] = sum_s workspace[s*count + i] (+ out[i] if accumulate): sums per-chunk partial buffers (embed assembly) */
int ofb_splitk_reduce(const float* workspace, int32_t splits, int64_t count, float* out, int32_t accumulate,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-launch HIP-event timing on the launch stream (used by bench.py for the roofline object).
 * Tags: 0 gemm, 1 attention fwd, 2 layernorm fwd, 3 layernorm bwd, 4 attention bwd.
 * ofb_prof_collect synchronises the recorded events and fills out[tag*3 + {0 launches, 1 ms, 2 work}].
 * ------------------------------------------------------------------------------------------- */
int ofb_prof_enable(int32_t on);   /* on: bit t set -> bracket the launches of tag t; 0: off */
int ofb_prof_collect(double* out, int32_t ntags);
/* diagnostic: `blocks` workgroups x 4 waves each issue 4*iters back-to-back f32 MFMAs (measures the sustained roof) */
int ofb_diag_mfma_peak(float* out, int32_t blocks, int32_t iters, void* stream);
/* measurement only: `blocks` 256-thread workgroups that occupy their CU slots (and lds_bytes of LDS each) for usec microseconds on
 * `stream` - a stand-in for another stream's resident kernels (RCCL channels) while a step is timed (scripts/cu_thief.py) */
int ofb_diag_cu_thief(int32_t blocks, int32_t usec, int32_t lds_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (eps inside the sqrt), one wavefront per token row; D <= 1024.
 * Replaces F.layer_norm in models/layers.py:96-98 (LayerNorm.forward) as used by MAEBlock
 * (models/vision_transformer.py:193-204) and the final norm (:663-668).
 * bwd: dx = rstd*(dy*g - mean(dy*g) - xhat*mean(dy*g*xhat)) (+ dres); per-block partial
 * dgamma/dbeta go to partials[ofb_layernorm_bwd_blocks(rows)][2][D] (sum them with ofb_colsum).
 * ------------------------------------------------------------------------------------------- */
int ofb_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                      int32_t rows, int32_t D, float eps, void* stream);
/* y (optional, may be NULL) and the same rows as H-format planes y_h[rows][D] (the operand form of the following GEMM, see
 * ofb_gemm_h; ofb_hformat_bytes(rows, D) bytes, header included: its exponent follows from gamma / beta alone, |y_i| <= sqrt(D)
 * |gamma_i| + |beta_i|).  Row groups 0 .. ceil(rows/4)-1 are written whole (padding as zeros); the caller zeroes what is left of
 * the last 16-row group when rows % 16 is in 1..12. */
int ofb_layernorm_fwd_h(const float* x, const float* gamma, const float* beta, float* y, void* y_h, float* mean, float* rstd,
                        int32_t rows, int32_t D, float eps, void* stream);
int32_t ofb_layernorm_bwd_blocks(int32_t rows);
int ofb_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                      const float* dres, float* dx, float* partials, int32_t rows, int32_t D, void* stream);
/* Same, and dx * rowscale[row / rs_div] (rowscale optional: the DropPath factor of the branch this gradient flows into) also as
 * H-format planes dx_h[rows][D]; partials is then [ofb_layernorm_bwd_blocks(rows)][3][D]: dgamma | dbeta | column sums of the
 * scaled dx rows (that branch's output-bias gradient).  The planes' exponent comes from a bound formed by a pass over dy before
 * the main kernel: |dx_row|_2 <= rstd |gamma * dy_row|_2 (LayerNorm's Jacobian is rstd times an orthogonal projection); the pass
 * parks its per-block maxima in the LAST 8 KB of dx_h (ofb_hformat_bytes(rows, D) bytes: slack past the matrix that no consumer reads
 * as values), so dx_h must be a buffer of exactly that size. */
int ofb_layernorm_bwd_h(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                        const float* dres, float* dx, float* partials, void* dx_h, const float* rowscale, int32_t rs_div,
                        int32_t rows, int32_t D, void* stream);
/* The same with the bound of the result taken from rn[n_rn] - the rn_out of the ofb_gemm_h call that produced dy, rn_fac = sqrt(its
 * column tiles) - instead of a bound pass over dy; dy must be the only gradient of the LayerNorm's output (no dres). */
int ofb_layernorm_bwd_h_rn(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                           float* partials, void* dx_p, const float* rowscale, int32_t rs_div, int32_t rows, int32_t D, const float* rn,
                           int32_t n_rn, float rn_fac, void* stream);

/* out[N] = column sums of x[M][ld] (optionally rows scaled by rowscale[m / rs_div]): bias gradients of every
 * Linear on the path.  scratch: ofb_colsum_slabs(M, N) * N floats. */
int32_t ofb_colsum_slabs(int32_t M, int32_t N);
int ofb_colsum(const float* x, int32_t ld, int32_t M, int32_t N, const float* rowscale, int32_t rs_div, float* out,
               float* scratch, void* stream);
/* Many such sums in ONE launch: out[N] = column sums of x[M][ld] per job (jobs_dev[n_jobs] in device memory; max_N = the widest).
 * The LayerNorm parameter gradients of a whole backward pass (25 x [1024 partial rows][3 D]) are reduced together after it. */
typedef struct ofb_colsum_job { const float* x; float* out; int32_t ld, M, N, pad_; } ofb_colsum_job;
int ofb_colsum_multi(const ofb_colsum_job* jobs_dev, int32_t n_jobs, int32_t max_N, void* stream);

/* Bi-mask gate folded into a Linear layer (q,k,v *= g: models/layers.py:507-509; fc1 out *= g: :858;
 * conv out *= g: :191).  out = g[n] * W[n][:]; and the matching backward: from the UNGATED raw gradients
 * dWraw = dY^T x, dbraw = colsum(dY): dW = g*dWraw, db = g*dbraw, dg[n] = <dWraw[n], W[n]> + dbraw[n]*b[n].
 * dbraw: [dbraw_rows][N]; with dbraw_rows > 1 the rows are partial column sums (per image from the attention backward, per
 * tile from a GEMM epilogue) that this kernel adds up itself.  fold > 1: ONE gate vector of N / fold values serves `fold` row
 * groups (q | k | v, layers.py:507-509): g holds the tiled N values, dg receives the N / fold sums over the groups. */
int ofb_scale_rows(const float* W, const float* g, float* out, int32_t N, int32_t K, void* stream);
int ofb_gate_fold_bwd(const float* dWraw, const float* W, const float* g, const float* dbraw, int32_t dbraw_rows, const float* b,
                      float* dW, float* db, float* dg, int32_t N, int32_t K, int32_t fold, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Attention core softmax(q k^T * scale) v with probabilities kept on chip
 * (models/layers.py:510-514; plain Attention.forward :387-391 for the finetune path).
 * qkv: [B*N][3*H*dh] exactly as the qkv Linear writes it (q | k | v, head-major); out: [B*N][H*dh]
 * (the transpose(1,2).reshape of :514 is folded into the store); lse: [2][B*H][N]: row log-sum-exp fl(m + log l), then (B*H*N floats
 * further) its rounding residue (m - lse) + log l - the backward recomputes P = exp((S - lse) - residue) at fp32-softmax accuracy.
 * Limits: dh <= 64, dh % 4 == 0 (covers DeiT-T/S/B and every pruned d in {16,24,..,64}); N <= 4096.  N <= 208 (13 tiles of 16 tokens;
 * DeiT at 224 px) is the tuned case: one workgroup per (image, head).  Longer sequences (384 px: N = 577; patch 8: N = 785) run the same
 * kernels chunked: the forward spreads the query tiles over ceil(N / 208) workgroups per (image, head), the backward is launched once
 * per 224 keys and accumulates dq across the launches in stream order (deterministic).
 * bwd writes dqkv in the same packing (dq | dk | dv).
 * Both kernels split their operands into two f16 planes of a power-of-two scaled copy (csrc/hformat.h) and therefore need upper
 * bounds of |qkv| and |dout| as DEVICE scalars: the cbound_out of the GEMMs that produced them (ofb_gemm_h), or ofb_amax.
 * ------------------------------------------------------------------------------------------- */
int ofb_amax(const float* x, int64_t n, float* out, void* stream);     /* out[0] = max |x[i]| */
int ofb_attention_fwd(const float* qkv, float* out, float* lse, int32_t B, int32_t N, int32_t H, int32_t dh, float scale,
                      const float* qkv_bound, void* stream);
/* dqkv_amax (optional device scalar) receives max |dqkv|: the bound for the H-format copy ofb_to_hformat_colsum makes of it */
int ofb_attention_bwd(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv, int32_t B,
                      int32_t N, int32_t H, int32_t dh, float scale, const float* qkv_bound, const float* dout_bound,
                      float* dqkv_amax, void* stream);
/* The same with the maximum left as ONE WORD PER WORKGROUP: wg_amax[B * H] (required) is plainly written - no atomic max, and nothing to
 * zero ahead of the launch (ofb_attention_bwd enqueues a 4-byte memset node for its scalar); hand the vector to ofb_to_hformat_colsum_nb */
int ofb_attention_bwd_wgmax(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv, int32_t B,
                            int32_t N, int32_t H, int32_t dh, float scale, const float* qkv_bound, const float* dout_bound,
                            float* wg_amax, void* stream);
/* Forward that also writes the output rows as H-format planes out_h[B*N][H*dh] (the operand of the projection GEMM; the values
 * of `out`; |out| <= max |v|, so the planes take the qkv exponent).  The caller zeroes out_h beforehand when B*N or H*dh is not a
 * multiple of 16. */
int ofb_attention_fwd_h(const float* qkv, float* out, void* out_h, float* lse, int32_t B, int32_t N, int32_t H, int32_t dh,
                        float scale, const float* qkv_bound, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Bi-mask gates of ALL searchable modules in one launch + adaptive one-hot (sparsity) loss + FLOPs loss.
 *   g[h][c] = w_p*sigmoid(score[h][c]) + (1-w_p)*wm[rank_h[h]][rank_c[h][c]],
 *   wm[h][c] = sum_{cells(i,j) on} softmax(alpha)[i][j] * [h < head_thr[i]] * [c < chan_thr[j]]
 * Replaces the per-forward gate micro-ops of MAEPatchEmbed.forward (models/layers.py:179-191),
 * MAESparseAttention.forward (:494-509), MAESparseMlp.forward (:847-858), the get_weight() re-computation
 * (:211-216, :548-557, :876-881), MAEBaseModel.get_sparsity_loss / get_flops_loss (models/base_model.py:31-86)
 * and MIMVisionTransformer.get_flops (models/vision_transformer.py:759-783).
 * The host keeps one descriptor per module in a device array.  1-D modules (mlp, embed) use H = 1, A0 = 1,
 * head_thr = {1}.  Limits: A0 <= 8, A1 <= 40, A0*A1 <= 64, H <= 16.
 * ------------------------------------------------------------------------------------------- */
typedef struct ofb_gate_desc {
  const float* alpha;        /* [A0*A1] */
  const float* score;        /* [H*C]   */
  float* g; float* wr; float* wm;   /* [H*C] gate, restored staircase (weight_restore), staircase (weighted_mask) */
  float* prob;               /* [A0*A1] softmax(alpha) over the "on" cells */
  float* wsum;               /* [1] sum of wm (input of the FLOPs model) */
  float* loss_alpha;         /* [1] entropy + variance terms of this module's one-hot loss */
  float* dloss_dalpha;       /* [A0*A1] their gradient wrt alpha */
  float* sig_partial;        /* [ceil(H*C/256)] per-block sums of sigmoid(score) */
  int32_t* rank;             /* [H*C] (rank_h << 16) | rank_c */
  int32_t H, C, A0, A1;
  int32_t kind;              /* 0 attention, 1 mlp, 2 embed (loss bucket) */
  float w_p;
  float norm_coef;           /* 4e-4 attention, 1e-4 otherwise (base_model.py:74-78) */
  int32_t head_thr[8];
  int32_t chan_thr[40];
  uint8_t on[64];            /* switch_cell */
} ofb_gate_desc;

typedef struct ofb_gate_grad {
  const float* dg;           /* [H*C] or null */
  const float* dwr;          /* [H*C] or null */
  const float* dwm;          /* [H*C] or null */
  const float* dwsum;        /* [1] or null */
  const float* dspars;       /* [1] or null: d total / d (this module's sparsity loss) */
  float* dalpha;             /* [A0*A1] */
  float* dscore;             /* [H*C] */
} ofb_gate_grad;

typedef struct ofb_flops_cfg {
  int32_t num_patches, embed_dim, num_heads, head_dim, hidden, patch_area, num_classes, depth;   /* ORIGINAL architecture */
  float target;              /* target GMACs */
  int32_t ln_dim;            /* current embedding width (norm1.normalized_shape[0]); 0 = embed_dim */
  const int32_t* active_heads;   /* [depth] device array (head_num after compress) or null (= num_heads) */
  /* modules finished by compress() have a constant staircase sum: slot s of {embed, attn_0, mlp_0, ...} reads
   * wsum[live_slot[s]] when live_slot[s] >= 0, else wconst[s].  Both null: wsum already has all 1+2*depth slots. */
  const int32_t* live_slot;  /* [1+2*depth] device array or null */
  const float* wconst;       /* [1+2*depth] device array or null */
  int32_t n_live;            /* entries of wsum / dwsum (1+2*depth when live_slot is null) */
  const float* active_patches;   /* device scalar: the searched model's patch count (vision_transformer.py:768: weighted_mask.sum()
                                    once a patch-cell compress() has run, :789-820) or null (= num_patches) */
} ofb_flops_cfg;

/* spars_out[3] = {attn, mlp, embed} sums, spars_per_module[n_modules]; max_elems = max H*C over modules. */
int ofb_gates_fwd(const ofb_gate_desc* descs_dev, int32_t n_modules, int32_t max_elems, int32_t entropy, int32_t var,
                  int32_t norm, float* spars_out, float* spars_per_module, void* stream);
int ofb_gates_bwd(const ofb_gate_desc* descs_dev, const ofb_gate_grad* grads_dev, int32_t n_modules, void* stream);
/* wsum[n_live] = staircase sums of the live modules in {embed, attn_0, mlp_0, ...} order; out4 =
 * {((searched-target)/total)^2, total, searched, d out4[0] / d active_patches}; dwsum[n_live] = d out4[0] / d wsum. */
int ofb_flops_loss(const float* wsum, const ofb_flops_cfg* cfg, float* out4, float* dwsum, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Token assembly after the patch-embedding conv (models/vision_transformer.py:615-651, gate of
 * models/layers.py:191 factored out):
 *   tokens[b][0]   = g * (cls + pos[0]);  tokens[b][1+l] = g * ((conv[b][l] + pos[1+l])*(1-m[b][l]) + m[b][l]*mask_token)
 * conv: [B*L][D] UNGATED conv output (+bias); g/mask_token/mask may be null (no gate / no masking).
 * bwd writes dconv [B*L][D] and per-chunk partial sums ppos/pg/pmt, each [chunks][L+1][D] with
 * chunks = ofb_embed_assemble_chunks(B): dpos = sum_z ppos, dcls = dpos[0], dg = sum_{z,t} pg, dmask_token = sum pmt.
 * ------------------------------------------------------------------------------------------- */
int ofb_embed_assemble_fwd(const float* conv, const float* g, const float* pos, const float* cls, const float* mask_token,
                           const float* mask, float* tokens, int32_t B, int32_t L, int32_t D, void* stream);
int32_t ofb_embed_assemble_chunks(int32_t B);
int ofb_embed_assemble_bwd(const float* dtokens, const float* conv, const float* g, const float* pos, const float* cls,
                           const float* mask_token, const float* mask, float* dconv, float* ppos, float* pg, float* pmt,
                           int32_t B, int32_t L, int32_t D, void* stream);

/* norm_targets(imgs, 47) (models/vision_transformer.py:121-141): (x-mean)/sqrt(max(var*cnt/(cnt-1),0)+1e-6) with
 * 47x47 box statistics, count_include_pad=False.  imgs/out/scratch1/scratch2: [planes][H][W]. */
int ofb_norm_targets(const float* imgs, float* out, float* scratch1, float* scratch2, int32_t planes, int32_t Hh, int32_t Ww,
                     int32_t ksize, void* stream);
/* The same values for the pixels of the listed patches only (the PMIM loss reads nothing else: M = 0 on unmasked patches):
 * patch_ids[n_ids] = b*L + l (L = gw*gw patches of P x P pixels per plane, P <= 16); one fused kernel, the patch's 62 x 62 window
 * staged in LDS; out [B*C][H][W] receives those pixels, the rest of it is left untouched. */
int ofb_norm_targets_masked(const float* imgs, const int32_t* patch_ids, int32_t n_ids, float* out, int32_t B, int32_t C, int32_t L,
                            int32_t P, int32_t Hh, int32_t Ww, int32_t ksize, void* stream);

/* PMIM masked L1 loss (models/vision_transformer.py:724-729) evaluated in PATCH layout: rec [B*L][C*P*P] is the
 * decoder 1x1-conv output before PixelShuffle (channel c*P*P+i*P+j <-> pixel (c, P*py+i, P*px+j)), targets
 * [B][C][P*gw][P*gw], mask [B*L] in {0,1}.  out2 = {loss, 1/((sum(mask)*P*P+1e-5)*C)}; partial: [B*L] scratch.
 * bwd: drec = upstream[0] * out2[1] * sign(rec - target) * mask.
 * patch_ids (optional): rec holds only n_rows decoded patches, row i = global patch patch_ids[i] (unmasked patches add
 * exactly 0 to the loss and its gradient, so decoding only the masked ones is exact); else n_rows = B*L. */
int ofb_pmim_loss_fwd(const float* rec, const float* targets, const float* mask, const int32_t* patch_ids, int32_t n_rows,
                      float* partial, float* out2, int32_t B, int32_t L, int32_t P, int32_t C, void* stream);
int ofb_pmim_loss_bwd(const float* rec, const float* targets, const float* mask, const int32_t* patch_ids, int32_t n_rows,
                      const float* out2, const float* upstream, float* drec, int32_t B, int32_t L, int32_t P, int32_t C,
                      void* stream);

/* Label-smoothing cross entropy (timm LabelSmoothingCrossEntropy as used by search.py:584 / losses.py:38):
 * loss[0] = mean_b[(1-s)*nll + s*mean_c(-logp)]; grad [B][C] = d loss / d logits; row_loss [B] scratch. */
int ofb_ls_cross_entropy(const float* logits, const int64_t* labels, float* row_loss, float* loss, float* grad, int32_t B,
                         int32_t C, float smoothing, void* stream);

/* Per-sample random patch masking (models/vision_transformer.py:586-612): mask[b][l] = 0 for the len_keep patches
 * with the smallest noise[b][.], 1 for the rest (== gather(mask, argsort(argsort(noise)))).  masked_ids (optional,
 * [B][L-len_keep]): global ids b*L + l of the removed patches, so the decoder can run on those rows only. */
int ofb_patch_mask(const float* noise, float* mask, int32_t* masked_ids, int32_t B, int32_t L, int32_t len_keep, void* stream);

/* Scalar mixing of one search micro-step's losses (reference losses.py:97-104 `w1*attn + w2*mlp + w4*embed + w5*flops`, engine.py:134-144
 * `base + arch + stopgrad(base / decoder_loss) * decoder_loss`) in one launch; every input is a device scalar (spars3: three), any may be null:
 *   out3[0] = arch  = (w0 spars3[0] + w1 spars3[1]) + w2 spars3[2] + w3 flops[0]      (null terms: 0)
 *   out3[1] = coef  = base / dec                                                       (base or dec null: 0)
 *   out3[2] = total = (base + arch) + coef * dec
 * The criterion calls it with (spars3, flops) for `arch`, the engine with (base, flops = arch, w3 = 1, dec) for `total`; the gradients are
 * the constants w0..w3, 1 and coef (ofb_scale_by_scalar chains them). */
int ofb_loss_mix(const float* base, const float* spars3, const float* flops, const float* dec, float w0, float w1, float w2, float w3,
                 float* out3, void* stream);

/* The two readers of the final token stream stream_rows [B][T][D], T = L + 1 (reference vision_transformer.py:735-744: `x[:, 0]` for the head,
 * the masked patch tokens for the decoder): cls_out [B][D] = row b T; z_out [n_ids][D] = token row of global patch id p = b L + l, i.e.
 * stream row p + p / L + 1.  bwd: dstream [B][T][D] = 0 everywhere except those rows (dcls / dz may be null); the rows are disjoint and
 * unique. */
int ofb_token_taps_fwd(const float* stream_rows, const int32_t* patch_ids, int32_t n_ids, int32_t B, int32_t T, int32_t D, float* cls_out,
                       float* z_out, void* stream);
int ofb_token_taps_bwd(const float* dcls, const float* dz, const int32_t* patch_ids, int32_t n_ids, int32_t B, int32_t T, int32_t D,
                       float* dstream, void* stream);

/* timm DropPath factors (reference vision_transformer.py:152 `DropPath`): out[r][b] = floor(keep[r] + u[r][b]) / keep[r] for R residual
 * branches x B samples, u uniform in [0, 1). */
int ofb_droppath_scales(const float* u, const float* keep, float* out, int32_t R, int32_t B, void* stream);

/* out = x * scalar_dev[0] (chains a device-resident upstream gradient without a host sync) */
int ofb_scale_by_scalar(const float* x, const float* scalar_dev, float* out, int64_t n, void* stream);

/* dst[o][i][k] = src[o][idx[i]][k] for a tensor viewed as [outer][n_src][inner]: the physical cut compress() applies to
 * weights and to AdamW moments (models/layers.py:272-293 `weight.data.clone()[keep_index, ...]`, optim.py:129-139).
 * idx: int32 device array [n_idx], every entry in [0, n_src) (entries outside are rejected on the device: the
 * row is zero-filled and the call reports OFB_EINVAL through `bad`, a 1-int device flag, when given). */
int ofb_index_select(const float* src, const int32_t* idx, float* dst, int64_t outer, int64_t n_src, int64_t n_idx, int64_t inner,
                     int32_t* bad, void* stream);

/* hipMemcpyAsync(host -> device) of a small pointer table from PINNED host memory on `stream` (capturable: a memcpy node) */
int ofb_upload(void* dst_dev, const void* src_pinned, int64_t nbytes, void* stream);

/* Non-finite loss watch (engine.py:146-150: the reference reads the loss on the host every micro-step and exits BEFORE backward when
 * it is not finite).  This path never syncs per step: ofb_nonfinite_watch adds 1 to the device counter flag[0] when any of the n
 * values is NaN / +-inf, and the parameter-changing kernels below (AdamW, EMA) take that counter as `skip`: once it is non-zero
 * they leave every tensor untouched, so no optimizer / EMA update is ever applied after a non-finite loss; the host reads the
 * counter at its print points and stops. */
int ofb_nonfinite_watch(const float* values, int32_t n, int32_t* flag, void* stream);

/* Multi-tensor AdamW, one launch per parameter group (optim.py:56-120): decoupled decay, bias-corrected Adam.
 * skip (optional, device): when *skip != 0 the launch changes nothing (see ofb_nonfinite_watch). */
typedef struct ofb_adamw_tensor { float* p; const float* g; float* m; float* v; int64_t n; } ofb_adamw_tensor;
int ofb_adamw_step(const ofb_adamw_tensor* table_dev, int32_t n_tensors, int64_t max_numel, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int32_t step, const int32_t* skip, void* stream);
/* The same update with hyper_dev[3] = {lr, 1 - beta1^step, 1 / sqrt(1 - beta2^step)} read from device memory: the form a step
 * captured in a hipGraph replays (the host refreshes the three floats before each replay). */
int ofb_adamw_step_dev(const ofb_adamw_tensor* table_dev, int32_t n_tensors, int64_t max_numel, const float* hyper_dev, float beta1,
                       float beta2, float eps, float weight_decay, const int32_t* skip, void* stream);

/* dst[i] = src[i] (src != NULL) or 0 for every job, ONE launch: the small gradients (biases, LayerNorm, alpha / score) and the
 * zero fill of absent ones on their way into a flat all-reduce bucket - the per-parameter copies DistributedDataParallel makes
 * when it wraps the model (search.py:617-620). */
typedef struct ofb_copy_job { const float* src; float* dst; int64_t n; } ofb_copy_job;
int ofb_multi_copy(const ofb_copy_job* jobs_dev, int32_t n_jobs, int64_t max_n, void* stream);

/* Multi-tensor weight EMA (utils.py:430-441 `ema_v.copy_(ema_v * decay + (1 - decay) * model_v)`), one launch for the
 * whole state_dict; products and the sum are rounded separately (no FMA contraction) so the result is bit-identical to
 * the reference's three elementwise ops.  one_minus_decay = (float)(1.0 - (double)decay), as Python computes it. */
typedef struct ofb_ema_tensor { float* ema; const float* src; int64_t n; } ofb_ema_tensor;
int ofb_ema_update(const ofb_ema_tensor* table_dev, int32_t n_tensors, int64_t max_numel, float decay, float one_minus_decay,
                   const int32_t* skip, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input side of the step.
 * Mixup / CutMix (timm `Mixup` as called by engine.py:35-36,99-100; built at search.py:651-655, finetune.py:310):
 * one parameter record per sample, so the 'batch', 'pair' and 'elem' modes share the kernels.  Sample b is mixed with
 * the ORIGINAL sample B-1-b (x.flip(0)) in place:  x[b] = x[b]*lam + x[B-1-b]*one_minus_lam  (products and sum rounded
 * separately, as torch's mul_/add_ do), or, for use_cutmix, x[b][:, yl:yh, xl:xh] = x[B-1-b][:, yl:yh, xl:xh].
 * ofb_mixup_targets: out[b][c] = y1*lam + y2*one_minus_lam with y1/y2 the smoothed one-hot rows of labels[b] /
 * labels[B-1-b] (on_value = 1 - smoothing + smoothing/C, off_value = smoothing/C; timm mixup_target).
 * ofb_soft_cross_entropy: timm SoftTargetCrossEntropy (search.py:581-583, finetune.py:390-393):
 * loss[0] = mean_b sum_c -target[b][c]*log_softmax(logits[b])[c]; grad = d loss / d logits; row_loss [B] scratch.
 * ------------------------------------------------------------------------------------------- */
typedef struct ofb_mix_param {
  float lam, one_minus_lam;
  int32_t use_cutmix;
  int32_t yl, yh, xl, xh;
} ofb_mix_param;
int ofb_mixup_batch(float* x, const ofb_mix_param* params_dev, int32_t B, int32_t C, int32_t H, int32_t W, void* stream);
int ofb_mixup_targets(const int64_t* labels, const ofb_mix_param* params_dev, float* out, int32_t B, int32_t num_classes,
                      float on_value, float off_value, void* stream);
int ofb_soft_cross_entropy(const float* logits, const float* target, float* row_loss, float* loss, float* grad, int32_t B,
                           int32_t C, void* stream);

/* RandomResizedCrop(+interpolation) -> RandomHorizontalFlip -> ToTensor -> Normalize of the reference's build_transform
 * (datasets.py:127-163; eval: Resize + CenterCrop is the same call with a fixed box) on DECODED uint8 HWC RGB images that
 * sit in one device buffer.  Per sample: byte offset and size of the source image, the crop box (top, left, height,
 * width in source pixels), flip, filter (0 = bilinear, 1 = bicubic).  Resampling follows PIL's Image.resize(box=...)
 * (separable, support stretched by the down-scale factor, 8-bit intermediate rounded half up), horizontal pass first.
 * out: f32 [B][3][S][S] = ((v/255) - mean[c]) / std[c] (mean3/std3: HOST arrays of 3 floats); out_u8 (optional):
 * the resized uint8 pixels in the same CHW layout.  scratch: ofb_crop_resize_scratch_bytes(B, S, max_src_h) bytes,
 * max_src_h >= every sample's src_h. */
typedef struct ofb_crop_param {
  int64_t offset;
  int32_t src_h, src_w;
  int32_t top, left, height, width;
  int32_t flip, cubic;
} ofb_crop_param;
int64_t ofb_crop_resize_scratch_bytes(int32_t B, int32_t out_size, int32_t max_src_h);
int ofb_crop_resize_norm(const uint8_t* src, const ofb_crop_param* params_dev, int32_t B, int32_t out_size, int32_t max_src_h,
                         const float* mean3, const float* std3, float* out, uint8_t* out_u8, uint8_t* scratch, void* stream);

/* RandomErasing, mode 'pixel' (timm RandomErasing inside datasets.build_transform: --reprob 0.25 --remode pixel --recount 1,
 * search.py:135-139): every element of x[b][:, top:top+h, left:left+w] becomes N(0, 1) noise (h == 0: sample untouched).
 * Noise = Box-Muller on Philox4x32-10 words, key = seed, counter = (element index / 4, 0, b, 0); element index runs over
 * (c, y, x) of the rectangle. */
typedef struct ofb_erase_param { int32_t top, left, h, w; } ofb_erase_param;
int ofb_random_erase(float* x, const ofb_erase_param* params_dev, int32_t B, int32_t C, int32_t H, int32_t W, uint64_t seed,
                     void* stream);

/* RandAugment (timm rand_augment_transform('rand-m9-mstd0.5-inc1'): the reference's default --aa, search.py:123, applied by
 * datasets.build_transform between the flip and ToTensor): ONE layer of it on uint8 CHW images [B][3][H][W], one op per image
 * (Pillow semantics: ImageOps / ImageEnhance / Image.transform), in != out.
 *   op 0 none, 1 AutoContrast, 2 Equalize, 3 Invert, 4 Posterize (iarg = bits), 5 Solarize (iarg = threshold), 6 SolarizeAdd (iarg = add,
 *   threshold 128), 7 Color, 8 Contrast, 9 Brightness, 10 Sharpness (farg = factor of Image.blend(degenerate, image, factor)),
 *   11 affine warp Image.transform(size, AFFINE, m, resample = iarg (0 nearest, 2 bilinear, 3 bicubic), fillcolor = fill): Rotate,
 *   ShearX / ShearY, TranslateXRel / TranslateYRel.
 * hist_scratch: B*768 int32, lsum_scratch: B uint64 (per-image channel histograms and luma sum, recomputed by every call).
 * ofb_normalize_u8: ToTensor + Normalize, out[b][c][y][x] = ((v / 255) - mean[c]) / std[c] (mean3 / std3: HOST arrays). */
typedef struct ofb_aug_op {
  int32_t op, iarg;
  float farg;
  int32_t fill[3];
  double m[6];
} ofb_aug_op;
int ofb_randaug_layer(const uint8_t* in, uint8_t* out, const ofb_aug_op* ops_dev, int32_t B, int32_t H, int32_t W, int32_t* hist_scratch,
                      uint64_t* lsum_scratch, void* stream);
int ofb_normalize_u8(const uint8_t* in, float* out, int32_t B, int32_t H, int32_t W, const float* mean3, const float* std3, void* stream);

/* ---------------------------------------------------------------------------------------------
 * JPEG decode in front of the input pipeline (reference datasets.py:90-125: torchvision ImageFolder + PIL default_loader, i.e.
 * libjpeg-turbo).  HOST stage (plain C++, re-entrant: call it from the loader's threads): ofb_jpeg_parse reads the frame header;
 * ofb_jpeg_decode_coefficients Huffman-decodes every scan into int16 coefficient blocks, de-zigzagged, component c at
 * coef_off[c] as [blocks_h][blocks_w][64].  DEVICE stage: ofb_jpeg_decode_pixels turns the coefficients of a batch of images into
 * uint8 HWC RGB pixels in two launches (dequantise + 8x8 inverse DCT per block; chroma upsampling + YCbCr -> RGB per pixel),
 * restating libjpeg's default path bit for bit: jidctint.c (JDCT_ISLOW), jdsample.c fancy upsampling (h2v1 / h2v2 / h1v2),
 * jdcolor.c.  Baseline / extended-sequential Huffman files with 1 or 3 components; progressive, arithmetic-coded, CMYK and
 * 12-bit files return OFB_ELIMIT.  Buffers of ofb_jpeg_decode_pixels per image i: planes at plane_off[c] (blocks_h*8 rows of
 * blocks_w*8 bytes), pixels at out_off (height*width*3 bytes).
 * ------------------------------------------------------------------------------------------- */
typedef struct ofb_jpeg_info {
  int32_t width, height, ncomp;
  int32_t hs[3], vs[3];              /* sampling factors */
  int32_t hmax, vmax, mcu_x, mcu_y;
  int32_t blocks_w[3], blocks_h[3];  /* blocks per component (whole MCUs) */
  int32_t pad_;
  int64_t coef_off[3], coef_count;   /* in int16 elements */
  uint16_t quant[3][64];             /* natural order */
} ofb_jpeg_info;
typedef struct ofb_jpeg_job {
  int32_t width, height, ncomp;
  int32_t hs[3], vs[3];
  int32_t hmax, vmax;
  int32_t blocks_w[3], blocks_h[3];
  int32_t pad_;
  int64_t coef_off[3], plane_off[3], out_off;
  uint16_t quant[3][64];
} ofb_jpeg_job;
int ofb_jpeg_parse(const uint8_t* data, int64_t nbytes, ofb_jpeg_info* info);
int ofb_jpeg_decode_coefficients(const uint8_t* data, int64_t nbytes, const ofb_jpeg_info* info, int16_t* coef);
/* Whole-batch host stage: plan = parse every header and lay the batch out (infos[n], jobs[n] with absolute offsets, totals[6] =
   {coefficient elements, plane bytes, pixel bytes, max blocks of a component, max width, max height}); decode = the Huffman stage of all
   files on `threads` host threads into the staging buffer the plan sized (pinned memory; one H2D copy follows). */
int ofb_jpeg_plan_batch(const uint8_t* const* files, const int64_t* nbytes, int32_t n, ofb_jpeg_info* infos, ofb_jpeg_job* jobs,
                        int64_t* totals);
int ofb_jpeg_decode_batch(const uint8_t* const* files, const int64_t* nbytes, int32_t n, const ofb_jpeg_info* infos,
                          const ofb_jpeg_job* jobs, int16_t* coef, int32_t threads);
int ofb_jpeg_decode_pixels(const ofb_jpeg_job* jobs_dev, int32_t n_images, int32_t max_blocks, int32_t max_w, int32_t max_h,
                           const int16_t* coef_dev, uint8_t* planes_dev, uint8_t* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OFB_HIP_H */
