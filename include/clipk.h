/*
 * clipk.h — C ABI of libclipk.so: the MI355X (gfx950 / CDNA4) kernels behind the CLIP-style
 * dual-encoder contrastive path of SrikarK-code/clip-dplm.
 *
 * The reference has NO FFI / operator boundary of its own: its boundary is the Python
 * nn.Module API of old/clip.py.  Each entry point below therefore cites the ATen call site(s) in the
 * reference that it replaces; the Python mirror of the module API lives in clip_dplm_amd/ and binds these
 * symbols through ctypes (clip_dplm_amd/_ffi.py).  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions (every entry point):
 *   - extern "C", plain device pointers + sizes, no torch types;
 *   - returns CLIPK_OK (0) or a negative clipk_status; never throws, never allocates, never syncs;
 *   - enqueues on the caller's hipStream_t (passed as void*), so ordering is the caller's stream order;
 *   - every buffer (outputs, workspaces) is owned by the caller; workspace sizes come from
 *     clipk_*_workspace() helpers; stateless and re-entrant;
 *   - bf16 tensors are raw uint16 storage ("bf16"), f32 are float; row-major with explicit leading
 *     dimensions in ELEMENTS.
 */
#ifndef CLIPK_H
#define CLIPK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum clipk_status {
  CLIPK_OK = 0,
  CLIPK_ERR_BAD_ARG = -1,       /* null pointer / non-positive dim / misaligned pointer            */
  CLIPK_ERR_UNSUPPORTED = -2,   /* shape outside what the kernels tile (see each entry point)       */
  CLIPK_ERR_LAUNCH = -3         /* hipGetLastError() != hipSuccess after the launch                 */
} clipk_status;

typedef enum clipk_dtype { CLIPK_BF16 = 0, CLIPK_F32 = 1, CLIPK_U8 = 2 } clipk_dtype;
typedef enum clipk_act { CLIPK_ACT_NONE = 0, CLIPK_ACT_RELU = 1, CLIPK_ACT_GELU = 2, CLIPK_ACT_CELU = 3,
                         CLIPK_ACT_SOFTPLUS = 4 } clipk_act;

int clipk_version(void);          /* ABI version, bumped on any signature change */
const char* clipk_arch(void);     /* "gfx950" */
const char* clipk_status_string(int status);

/* Kernel-selection options (process-wide, explicit; the library never reads environment variables).  Every value
 * of every option computes the same results: they pick between kernels / schedules that tests and tools/ compare.
 * Names: gemm_kernel (-1 auto, 1 generic, 2 128x128, 3 persistent 256x256, 4 persistent 128x256 with two workgroups per CU), gemm_epi_generic, gemm_bm, gemm_stages,
 * gemm_nwg, gemm_stagger, epi_nt, wgrad_kernel (-1 auto, 2, 3), attn_whole_fwd, attn_fused_bwd (-1 auto, 0, 1),
 * attn_fused_waves (0 auto: 4 waves for head dims <= 32, 8 for 96; 4; 8), attn_row_stores (0, 2: forward write-back of rotated rows from LDS), gemm_f32_splits (0 auto, 1 .. 8: cross-workgroup splits of the skinny f32 Linear), simce_kernel (-1 auto, 1 first-generation, 2 tiled LSE pass).  Unknown name -> CLIPK_ERR_BAD_ARG.  (The reference has no counterpart: its kernels are
 * ATen's.) */
int clipk_set_option(const char* name, int value);
int clipk_get_option(const char* name, int* value);
int clipk_reset_options(void);

/* ------------------------------------------------------------------------------------------------
 * Linear (GEMM + fused epilogue), bf16 MFMA, f32 accumulate.
 *   C[M,N] = epilogue( A[M,K] · B[N,K]^T )
 *   epilogue(v) : v += bias[n];  if out_preact: out_preact = v (bf16);  v = act(v);
 *                 if dact_aux: v *= act'(dact_aux[m,n]);  if residual: v += residual[m,n];  C = v
 * Replaces nn.Linear (+ReLU/GELU, + residual add) at old/clip.py:11,16,27,29,31; the QKV / out-proj /
 * FFN Linear layers of nn.TransformerEncoderLayer (current/rna_clip_codes.ipynb:1915) and of the
 * third-party EsmLayer (transformers modeling_esm.py:362-374,517-521) called at
 * triple_flow/3_esm_integration.py:118-119.  With B = W^T (a [K_out,N_in] copy) it is the input
 * gradient dX = dY·W of the same layers.
 * Requirements: K % 8 == 0, N % 8 == 0, lda/ldb/ldc/... % 8 == 0, pointers 16-byte aligned.
 */
typedef struct clipk_gemm_args {
  const void* A; int64_t lda;          /* bf16 [M,K]                                       */
  const void* B; int64_t ldb;          /* bf16 [N,K]  (nn.Linear weight layout)            */
  void* C; int64_t ldc; int c_dtype;   /* bf16 or f32 [M,N]                                */
  int M, N, K;
  const float* bias;                   /* f32 [N] or NULL                                  */
  int act;                             /* clipk_act applied after bias                     */
  void* out_preact; int64_t ldp;       /* optional bf16 [M,N]: value before activation     */
  const void* dact_aux; int64_t ldd;   /* optional bf16 [M,N]: multiply by act'(aux)       */
  int dact;                            /* clipk_act whose derivative is applied to aux     */
  const void* residual; int64_t ldr; int r_dtype; /* optional [M,N] bf16/f32, added last   */
  float alpha;                         /* scale applied to the raw product before bias     */
  /* dropout on the value after the activation (and before act'(aux) / the residual add): v *= keep / (1 - p), with
   * keep = hash(drop_seed, m * N + n) >= drop_p * 2^32 — nn.Dropout after out_proj / linear2 / the FFN activation of
   * nn.TransformerEncoderLayer(dropout = p) (current/rna_clip_codes.ipynb:1915).  The same (seed, p) on the matching
   * backward GEMM reproduces the mask; nothing is stored.  drop_p = 0: off. */
  float drop_p; uint32_t drop_seed;
  /* rotary position embedding on the first rope_cols output columns (ESM-2's fused [q | k | v] projection: the q and k
   * thirds), applied to the f32 value (product + bias) before the single bf16 rounding: heads are rope_hd consecutive
   * columns, out[d] = x[d] cos[pos, d mod hd/2] -/+ x[d +/- hd/2] sin[pos, d mod hd/2] (rotate-half, transformers
   * modeling_esm.py:48-52,74-79), pos = (row + rope_row0) mod rope_L, tables f32 [rope_L, rope_hd/2].  Replaces the
   * separate clipk_rope_qk pass after the projection (one read + one write of q and k per layer).  rope_cos = NULL:
   * off.  Requirements: bf16 output, no activation / residual / aux / dropout, rope_hd in {16, 32, 64},
   * rope_cols % rope_hd == 0, K % 32 == 0; anything else returns CLIPK_ERR_UNSUPPORTED (never silently unrotated). */
  const float* rope_cos; const float* rope_sin; int rope_L, rope_hd, rope_cols, rope_row0;
  /* aux_dtype = CLIPK_U8 (act / dact must be GELU): the auxiliary tensor of the FFN pair is the DERIVATIVE GELU'(v) as an
   * 8-bit code instead of the bf16 pre-activation v - out_preact receives u8 [M, N] codes (ldp in bytes, % 8 == 0),
   * dact_aux is read as such codes and the product is multiplied by the decoded value: code = round(GELU'(v) * 200 + 26),
   * decoded as -0.13 + 0.005 code: 256 levels from -0.13 to 1.145 (GELU' lies in [-0.129, 1.129]) with GELU' = 0 and
   * GELU' = 1 - what dead and saturated units take - code points themselves, error <= 0.0025 - half the bytes of the
   * pre-activation in the two store-bound FFN epilogues of EsmLayer / nn.TransformerEncoderLayer(activation = gelu)
   * (modeling_esm.py:517-521; run1/configuration_hybrid_clip.py:75 hidden_act), nothing but the backward's GELU' factor
   * is affected.  CLIPK_BF16 (0): the pre-activation itself, as before. */
  int aux_dtype;
  /* rope_interleaved != 0 (with rope_cos / rope_sin / rope_L / rope_hd / rope_cols): the heads of the first rope_cols output
   * columns are in PAIR-INTERLEAVED order - columns 2 j and 2 j + 1 of a head hold what rotate-half calls x[j] and
   * x[j + hd/2] (the B operand's rows were permuted accordingly: clipk_cast_transpose il_hd / il_rows) - and are rotated as
   * neighbours: out[2j] = x1 cos[pos, j] - x2 sin[pos, j], out[2j+1] = x2 cos[pos, j] + x1 sin[pos, j].  Any rope_hd % 8 == 0
   * (ESM-2-35M's 24, which does not tile the kernels' 64-column slices).  q . k does not depend on a common order of the head
   * dim, so attention runs on such q / k as it is (clipk_attn_fwd without tables; clipk_attn_bwd with prerotated = 2). */
  int rope_interleaved;
} clipk_gemm_args;
int clipk_gemm_nt(const clipk_gemm_args* args, void* stream);

/* Weight gradient: dW[N,K] (+)= dY[M,N]^T · X[M,K]   (contraction over the M tokens), f32 output.
 * Replaces autograd's mm(dY^T, X) for every nn.Linear above.  Split over M with per-split f32 slabs in
 * `workspace` followed by a deterministic reduce (no float atomics).  Also emits db[N] = colsum(dY)
 * when dbias != NULL.  accumulate != 0 adds into dW/dbias instead of overwriting.
 * Requirements: N % 8 == 0, K % 8 == 0. */
size_t clipk_gemm_wgrad_workspace(int M, int N, int K);
/* il_rows > 0: dY's first il_rows columns are in pair-interleaved head order (heads of il_hd columns: clipk_gemm_nt
 * rope_interleaved, clipk_attn_bwd prerotated = 2); their gradient rows are written to the rows of dW / entries of dbias of
 * the ORIGINAL order, so the master weights and their gradients never see the permutation. */
int clipk_gemm_wgrad(const void* dY, int64_t lddy, const void* X, int64_t ldx,
                     float* dW, int64_t lddw, float* dbias,
                     int M, int N, int K, int accumulate, int il_hd, int il_rows,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused similarity + cross-entropy ("simce"), exact-f32 MFMA (v_mfma_f32_32x32x2_f32), never
 * materialises the logits.
 *   S[i,j] = scale * <X[i,:], Y[j,:]>,  label(i) = label_offset + i
 *   lse[i] = log sum_j exp(S[i,j])  over the Ny keys of Y plus the Nc keys of Yc (cache negatives)
 *   pos[i] = S[i, label(i)]
 * Replaces matmul(a, b.t()) * logit_scale + cross_entropy at old/clip.py:66-67 + old/ablation.py:16,
 * current/rna_clip_codes.ipynb:1950-1953 (both directions: call twice with X/Y swapped) and
 * old/clip_opt.py:115-121,130-151 (cache columns).  Rows of X are this rank's samples, Y holds the
 * all-gathered batch (old/clip_opt.py:102-112).
 * Requirements: P % 4 == 0, P <= 768.  workspace: clipk_simce_workspace(Mx, Ny + Nc, P) bytes (covers both
 * clipk_simce_lse and clipk_simce_grad). */
size_t clipk_simce_workspace(int Mx, int Nkeys, int P);
int clipk_simce_lse(const float* X, int Mx, const float* Y, int Ny, const float* Yc, int Nc,
                    int P, const float* scale /* device scalar = exp(logit_scale), clamped by caller */,
                    int label_offset, float* lse /*[Mx]*/, float* pos /*[Mx]*/,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Gradient of  L = (w_row * sum_i (lse_row[i] - pos[i]) + w_col * sum_j (lse_col[j] - pos[j])) / Bg
 * with respect to X rows (this rank's rows of one modality):
 *   G[i,j] = ( w_row * exp(S[i,j]-lse_x[i]) + w_col * exp(S[i,j]-lse_y[j]) - (w_row+w_col)*[j==label(i)] ) / Bg
 *   dX[i,:] = scale * sum_j G[i,j] * Y[j,:]  (+ cache keys: only the w_row term, no positives)
 *   dscale_partial[i] = sum_j G[i,j] * <X[i],Y[j]>       (d/d scale; caller multiplies by scale for
 *                                                        d/d logit_scale and halves the double count)
 * lse_x: [Mx] LSE of the rows of X over all keys;  lse_y: [Ny] LSE of each key over all queries
 * (the other direction), all-gathered across ranks by the caller. */
int clipk_simce_grad(const float* X, int Mx, const float* Y, int Ny, const float* Yc, int Nc,
                     int P, const float* scale, int label_offset,
                     const float* lse_x, const float* lse_y, float w_row, float w_col, float inv_bg,
                     float* dX /*[Mx,P]*/, float* dscale_partial /*[Mx]*/,
                     void* workspace, size_t workspace_bytes, void* stream);
/* The same with the loss' incoming gradient folded in: `upstream` (device scalar, or NULL = 1) multiplies inv_bg inside the
 * kernel - autograd's `grad_output * dX` after the fact was three more launches on [B, P] / [1] tensors per step
 * (loss.backward() hands 1.0; upstream = 1 gives clipk_simce_grad's bits). */
int clipk_simce_grad_scaled(const float* X, int Mx, const float* Y, int Ny, const float* Yc, int Nc,
                            int P, const float* scale, int label_offset,
                            const float* lse_x, const float* lse_y, float w_row, float w_col, float inv_bg,
                            const float* upstream, float* dX /*[Mx,P]*/, float* dscale_partial /*[Mx]*/,
                            void* workspace, size_t workspace_bytes, void* stream);
/* loss[0] = (w_row * sum_i (lse_r[i] - pos_r[i]) + w_col * sum_i (lse_c[i] - pos_c[i])) / bg from the two clipk_simce_lse
 * results in one launch, fixed summation order (the two F.cross_entropy means and their average of
 * rna_clip_codes.ipynb:1952-1953; old/ablation.py:16 with w_col = 0 and lse_c = pos_c = NULL). */
int clipk_ce_combine(const float* lse_r, const float* pos_r, const float* lse_c, const float* pos_c, int n,
                     float w_row, float w_col, float bg, float* loss, void* stream);

/* Batched form for several same-shape contrastive problems on one logit scale — the tri-modal ContrastiveModel of
 * current/tf_clip_codes (1).ipynb:13150-13163 (cell x pert, cell x protein, pert x protein, each symmetric) is six
 * directed problems (X = E[pairs[2i]], Y = E[pairs[2i+1]]), computed by ONE launch per pass (grid z = problem):
 *   lse_pairs : lse[i][b] = LSE_j scale <X_b, Y_j>, pos[i][b] = scale <X_b, Y_b>
 *   grad_pairs: dX[i] = complete gradient of (w_row CE_rows + w_col CE_cols)(X, Y) * inv_bg w.r.t. the rows of X,
 *               using lse[i] for the rows and lse[reverse[i]] (the problem with X and Y exchanged) for the keys.
 * E: f32 [nmod][B][P] contiguous; pairs / reverse: HOST int arrays; npairs <= 6. */
size_t clipk_simce_pairs_workspace(int npairs, int B, int P);
int clipk_simce_lse_pairs(const float* E, int nmod, int B, int P, const int* pairs, int npairs, const float* scale,
                          float* lse /*[npairs][B]*/, float* pos /*[npairs][B]*/, void* workspace,
                          size_t workspace_bytes, void* stream);
int clipk_simce_grad_pairs(const float* E, int nmod, int B, int P, const int* pairs, const int* reverse, int npairs,
                           const float* scale, const float* lse /*[npairs][B]*/, float w_row, float w_col,
                           float inv_bg, float* dX /*[npairs][B][P]*/, float* dscale_partial /*[npairs][B]*/,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Materialised logits for the drop-in module API (old/clip.py:67 returns them):
 *   S[Mx,Ny] = scale * X·Y^T, exact f32. */
int clipk_sim_logits(const float* X, int Mx, const float* Y, int Ny, int P, const float* scale,
                     float* S, int64_t lds, void* stream);

/* Cross-entropy on MATERIALISED logits — the reference's loss call sites take the logits tensor its modules return:
 * F.cross_entropy(logits, arange(B)) at old/ablation.py:16 / run1/full.py:133, the symmetric pair at
 * current/rna_clip_codes.ipynb:1952-1953, and (F.cross_entropy(cat([S, S_cache], 1)) + F.cross_entropy(S^T)) / 2
 * at old/clip_opt.py:130-151.  (Training should use clipk_simce_*: there the logits never reach HBM.)
 *   lse:  columns == 0: lse[i] = logsumexp_j [S | S2][i, j], pos[i] = S[i, i + label_offset]          (i < M)
 *         columns == 1: lse[j] = logsumexp_i S[i, j],        pos[j] = S[j + label_offset, j]          (j < N; N2 == 0)
 *   bwd:  dS[i,j]  = g * ( w_row (exp(S_ij - lse_row[i]) - [j == i + off_row])
 *                        + w_col (exp(S_ij - lse_col[j]) - [i == j + off_col]) ),
 *         dS2[i,j] = g * w_row exp(S2_ij - lse_row[i]);   lse_row / lse_col may be NULL (direction unused);
 *         w_row / w_col carry the 1 / batch factors; g = device scalar (upstream gradient). */
int clipk_ce_logits_lse(const float* S, int64_t ld, int M, int N, const float* S2, int64_t ld2, int N2,
                        int columns, int label_offset, float* lse, float* pos, void* stream);
int clipk_ce_logits_bwd(const float* S, int64_t ld, int M, int N, const float* S2, int64_t ld2, int N2,
                        const float* lse_row, const float* lse_col, float w_row, float w_col,
                        int label_offset_row, int label_offset_col, const float* gscale,
                        float* dS, int64_t ldd, float* dS2, int64_t ldd2, void* stream);

/* out[cols, rows] = scale_dev[0] * in[rows, cols]^T (f32; scale_dev NULL = 1).  Operand preparation of the exact-f32
 * products that differentiate the materialised logits (d/dA = scale * dS · B, d/dB = scale * dS^T · A of
 * old/clip.py:67) and of the ICNN's transposed weights (triple_flow/2_icnn_core.py:181-211). */
int clipk_transpose_scale_f32(const float* in, int rows, int cols, const float* scale_dev, float* out, void* stream);

/* Exact-f32 Linear for the ICNN transport maps (the reference forces f32 there: triple_flow/2_icnn_core.py:195):
 *   out[M,N] = X[M,K] · W[N,K]^T (+ bias[N]) (+ addend_scale[0] * addend[M,N])
 * Replaces self.linear(x) + scale * F.linear(z, softplus(W+)) at triple_flow/2_icnn_core.py:102-119 and the
 * matching products of the analytic input gradient T(x) = dPsi/dx (:181-211).  Same f32-MFMA kernel as
 * clipk_sim_logits.  K % 4 == 0, K <= 768. */
int clipk_gemm_f32_nt(const float* X, int M, const float* W, int N, int K, const float* bias,
                      const float* addend, const float* addend_scale /* device scalar or NULL (=1) */,
                      float* out, void* stream);

/* Tiled exact-f32 GEMM (v_mfma_f32_32x32x2_f32, 128 x 64 tiles, LDS-staged) with all four operand layouts:
 *   out[M,N] = alpha[0] * opA(A) · opB(B) (+ bias[N]) (+ addend_scale[0] * addend[M,N])     (alpha NULL = 1)
 *   transA == 0: A is [M,K] (lda);  transA == 1: A is stored [K,M] (contraction-major: dW = dY^T X)
 *   transB == 0: B is [N,K] (nn.Linear weight, out = A·B^T);  transB == 1: B is stored [K,N] (dA = dZ · W)
 * Every product of the ICNN potential, of its input gradient T(x) = dPsi/dx and of the training path's double backward
 * (triple_flow/2_icnn_core.py:102-119,181-211 under autocast(enabled=False)), and the gradients of the materialised
 * logits (old/clip.py:67): no transposed copies, any K.  lda / ldb % 4 == 0, A / B 16-byte aligned.
 * M <= 64 rows against a large weight (K >= 256, A not transposed: forward and input gradient of every Linear of the
 * position-0-sliced notebook models, M = batch) take the skinny form: one bandwidth-bound pass over B, the contraction
 * split 16 ways inside a workgroup and - given a workspace - across workgroups so that every CU streams its share.
 * workspace: clipk_gemm_f32_workspace(...) bytes (0 = none needed; scratch for the partial tiles of a split launch, summed
 * in split order by a second kernel); NULL or too small: no cross-workgroup split. */
size_t clipk_gemm_f32_workspace(int M, int N, int K, int transA, int transB);
int clipk_gemm_f32(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB,
                   int M, int N, int K, const float* alpha /* device scalar or NULL */, const float* bias,
                   const float* addend, int64_t ldadd, const float* addend_scale /* device scalar or NULL (=1) */,
                   float* out, int64_t ldo, void* workspace, size_t workspace_bytes, void* stream);

/* out[c] (+)= sum_r x[r][c] (x f32 [rows, cols] contiguous): the bias gradient dY.sum(0) of the exact-f32 Linear layers
 * (old/clip.py:11,27,31 under autograd), accumulated straight into the parameter's .grad; fixed summation order. */
int clipk_colsum_f32(const float* x, int rows, int cols, float* out, int accumulate, void* stream);
/* Deferred parameter gradients of several LayerNorms in one launch.  clipk_layernorm_bwd called with dgamma = dbeta = NULL
 * leaves its per-block partial rows [blocks][2][cols] (blocks = clipk_layernorm_bwd_workspace(rows, cols) / (2 cols 4)) in the
 * workspace it was given; with one such buffer per LayerNorm, this reduces all of them - what nn.LayerNorm's weight.grad /
 * bias.grad receive from autograd (old/clip.py:12,28,32; rna_clip_codes.ipynb:1911-1923) - when the backward pass is over:
 * desc_dev = n x 6 int64 in device memory {partial rows, blocks, cols, dgamma, dbeta, accumulate}; max_cols = the widest. */
int clipk_colreduce_batched(const int64_t* desc_dev, int n, int max_cols, void* stream);

/* Both parameter gradients of an exact-f32 Linear in one call: dW[N, K] (+)= dY[M, N]^T X[M, K], dbias[N] (+)= dY.sum(0)
 * (what autograd computes for nn.Linear under the reference's fp32 callers: old/clip.py:11,27,31; the Linear layers of
 * RNARBPCLIPModel / ContrastiveModel, current/rna_clip_codes.ipynb:1911-1954) - the f32 sibling of clipk_gemm_wgrad.
 * dW or dbias may be NULL (not both).  accumulate != 0 adds into dW / dbias (a parameter's .grad).  M <= 64 (the models
 * sliced to the one position they pool: M = batch rows): ONE launch, one pass over dW with dbias from the same operand
 * registers; more rows: the tiled clipk_gemm_f32 (contraction-major operands) + clipk_colsum_f32 (needs lddy == N).
 * dW is bit-identical to clipk_gemm_f32(dY, transA = 1, X, transB = 1, addend = dW) either way. */
int clipk_gemm_wgrad_f32(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw,
                         float* dbias, int M, int N, int K, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row-wise normalisation kernels (one wave per row, f32 statistics).
 * LayerNorm forward:  y = (x-mean)*rstd*gamma + beta, optional activation fused after it
 * (old/clip.py:12,28-29,32; transformer / ESM LayerNorms).  x is f32 or bf16; writes any of
 * y_f32 / y_bf16 (NULL to skip) and mean/rstd [rows] for the backward.
 */
int clipk_layernorm_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta,
                        float eps, int act, float* y_f32, void* y_bf16, int64_t ldy,
                        float* mean, float* rstd, int rows, int cols, void* stream);
/* LayerNorm backward: dx (f32 and/or bf16), and per-block partial dgamma/dbeta in workspace followed
 * by a deterministic column reduce into dgamma/dbeta (accumulate flag as above).  If act != NONE the
 * incoming dy is first multiplied by act'(ln_out) where ln_out is recomputed from x, mean, rstd.
 * dx_add: optional [rows,cols] f32 or bf16 (dx_add_dtype) added to the result (residual-stream gradient).
 * drop_p > 0: the bf16 output (only) is multiplied by the dropout mask keep(drop_seed, row * cols + col) / (1 - p): it is
 * the gradient of the dropped-out Linear output that was added to the residual stream in front of this LayerNorm
 * (dropout1 / dropout2 of nn.TransformerEncoderLayer); the f32 output stays the residual-path gradient.
 * dgamma == dbeta == NULL: input gradient only. */
size_t clipk_layernorm_bwd_workspace(int rows, int cols);
int clipk_layernorm_bwd(const void* dy, int dy_dtype, int64_t lddy, const void* x, int x_dtype, int64_t ldx,
                        const float* gamma, const float* beta, const float* mean, const float* rstd, int act,
                        const void* dx_add, int dx_add_dtype, float* dx_f32, void* dx_bf16, int64_t lddx,
                        float* dgamma, float* dbeta, int accumulate,
                        int rows, int cols, float drop_p, uint32_t drop_seed, void* workspace, size_t workspace_bytes, void* stream);

/* Backward OF clipk_layernorm_bwd (second order), f32: the ICNN transport map is T(x) = dPsi/dx
 * (triple_flow/2_icnn_core.py:181-211: torch.autograd.grad(..., create_graph=True)) and the training loss of
 * triple_flow/4_transport_maps.py:113-145 is a function of T, so autograd differentiates the first backward
 *   da = LNact_bwd(dy; a, gamma, beta)   (what clipk_layernorm_bwd computes, act in {NONE, CELU, SOFTPLUS})
 * with respect to dy, a, gamma and beta.  g = cotangent of da [rows, cols]; outputs d_dy, d_a [rows, cols] (either may
 * be NULL) and d_gamma / d_beta [cols] (both or neither; overwritten or accumulated).  mean / rstd as saved by
 * clipk_layernorm_fwd for `a`.  One leading dimension ld for g, dy, a, d_dy, d_a.  workspace:
 * clipk_layernorm_bwd_workspace(rows, cols) bytes. */
int clipk_layernorm_bwd2(const float* g, const float* dy, const float* a, int64_t ld, const float* gamma,
                         const float* beta, const float* mean, const float* rstd, int act, float* d_dy, float* d_a,
                         float* d_gamma, float* d_beta, int accumulate, int rows, int cols,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Final LayerNorm of an encoder + masked mean over the L token rows of each sample in one pass: pooled[b] =
 * mean_{l valid} LN(x[b, l]) (the pooling of run1/configuration_hybrid_clip.py:109,148 `use_mean_pooling`, fair-esm
 * mean over residues current/tf_clip_codes (1).ipynb:1188, behind the encoders' last LayerNorm modeling_esm.py:552-553 /
 * rna_clip_codes.ipynb:1923).  The normalised rows are never written.  x f32 or bf16 [B*L, cols], mask u8 [B*L] (1 = valid) or
 * NULL; outputs pooled f32 [B, cols], mean / rstd f32 [B*L] (for the backward), row_weight f32 [B*L] = the row's weight in
 * its sample's mean (1 / #valid rows, 0 for a masked row or an empty sample).
 * Backward: row r gets dy = dpooled[r / L] * row_weight[r] and goes through the LayerNorm backward
 * (outputs / workspace / dgamma / dbeta as clipk_layernorm_bwd, workspace size clipk_layernorm_bwd_workspace(B*L, cols)). */
int clipk_layernorm_meanpool_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, float eps,
                                 const uint8_t* mask, int B, int L, int cols, float* pooled, float* mean, float* rstd,
                                 float* row_weight, void* stream);
int clipk_layernorm_meanpool_bwd(const float* dpooled, const float* row_weight, int B, int L,
                                 const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* mean, const float* rstd,
                                 float* dx_f32, void* dx_bf16, int64_t lddx, float* dgamma, float* dbeta, int accumulate,
                                 int cols, void* workspace, size_t workspace_bytes, void* stream);

/* F.normalize(x, dim=-1) with eps=1e-12 (old/clip.py:63-64): y = x / max(||x||, eps); f32. */
int clipk_l2norm_fwd(const float* x, float* y, float* norm, int rows, int cols, float eps, void* stream);
int clipk_l2norm_bwd(const float* dy, const float* y, const float* norm, float* dx,
                     int rows, int cols, float eps, void* stream);

/* Elementwise helpers on f32/bf16 buffers (n elements, n % 8 == 0 not required). */
int clipk_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);
int clipk_cast_bf16_to_f32(const void* x, float* y, int64_t n, void* stream);
/* W f32 [rows,cols] -> bf16 copy and bf16 transposed copy [cols,rows] (either may be NULL).
 * il_rows > 0: rows [0, il_rows) of both copies are written in PAIR-INTERLEAVED head order for heads of il_hd rows - copy
 * row 2 j (+1) of a head holds W row j (+ il_hd / 2) of that head - the order in which clipk_gemm_nt's interleaved RoPE
 * epilogue (rope_interleaved) expects the q and k sections of ESM-2's fused qkv projection (modeling_esm.py:362-374). */
int clipk_cast_transpose(const float* w, void* w_bf16, void* wt_bf16, int rows, int cols, int il_hd, int il_rows, void* stream);
/* The same for n weights in one launch (what `optimizer.step()` leaves to do before the next forward: the reference
 * re-reads its f32 nn.Linear weights under autocast every step, old/clip.py:11).  desc_dev: device array of n
 * records {w, w_bf16, wt_bf16, rows, cols, il_hd, il_rows}, seven int64 each (pointers as integers, either output may be 0). */
int clipk_cast_transpose_batched(const void* desc_dev, int n, void* stream);
/* y = act(x) / dx = dy * act'(x) on f32. */
int clipk_act_fwd(const float* x, float* y, int act, int64_t n, void* stream);
int clipk_act_bwd(const float* dy, const float* x, float* dx, int act, int64_t n, void* stream);
/* out_bf16 = dy * act'(aux_bf16): activation backward between two Linear layers; dy f32 or bf16. */
int clipk_dact(const void* dy, int dy_dtype, const void* aux_bf16, int act, void* out_bf16, int64_t n, void* stream);
/* y = a + s[0] * b  (skip + layer_scale * projected, old/clip_opt.py:41-44), f32; a == NULL: y = s[0] * b (its backward). */
int clipk_axpby_dev(const float* a, const float* b, const float* s, float* y, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-head self-attention, flash style (no LxL matrix in HBM), bf16 MFMA, f32 softmax.
 * qkv: bf16 [B*L, 3*H*D] rows = tokens (b-major), columns = [q heads | k heads | v heads];
 * key_mask: uint8 [B, L], 1 = valid key, or NULL; rope_cos/sin: f32 [L, D/2] or NULL (ESM-2 rotary,
 * rotate-half, applied to q and k after q *= q_scale — transformers modeling_esm.py:48-52,74-79,374);
 * out: bf16 [B*L, H*D]; lse: f32 [B, H, L] (log-sum-exp of the scaled scores, saved for backward).
 * Replaces nn.MultiheadAttention inside nn.TransformerEncoderLayer (rna_clip_codes.ipynb:1915) and
 * EsmSelfAttention (modeling_esm.py:306-314,362-384).  D in {8..160}, D % 8 == 0.
 * dropout_p > 0: dropout on the attention probabilities (P~ = P * keep / (1 - p) feeds P·V, the normaliser uses P),
 * mask = hash(dropout_seed, ((token row * H + h) * L + key)); pass the same (p, seed) to the backward.
 */
int clipk_attn_fwd(const void* qkv, const uint8_t* key_mask, const float* rope_cos, const float* rope_sin,
                   void* out, float* lse, int B, int L, int H, int D, float q_scale, float dropout_p,
                   uint32_t dropout_seed, void* stream);
/* Backward: dqkv bf16 [B*L, 3*H*D] from dout bf16 [B*L, H*D]; recomputes P from qkv + lse.
 * delta: f32 [B,H,L] scratch (rowsum(dout*out)) provided by the caller.
 * prerotated: 0 = q / k in qkv are un-rotated (the kernels rotate at staging when given tables); 1 = already rotated
 * (clipk_rope_qk / clipk_attn_fwd_rot / clipk_gemm_nt's RoPE epilogue), rotate-half column order; 2 = already rotated, heads in
 * PAIR-INTERLEAVED column order (clipk_gemm_nt rope_interleaved): the gradients leave through the matching RoPE^T. */
int clipk_attn_bwd(const void* qkv, const uint8_t* key_mask, const float* rope_cos, const float* rope_sin,
                   const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                   int B, int L, int H, int D, float q_scale, int prerotated, float dropout_p, uint32_t dropout_seed,
                   void* stream);
/* Rotate-half RoPE (transformers modeling_esm.py:88-110) applied ONCE, in place, to the q and k sections of
 * qkv bf16 [B*L, 3*H*D].  Afterwards call clipk_attn_fwd WITHOUT rope tables and clipk_attn_bwd with the tables and
 * prerotated = 1: q / k are then staged as they are and only the gradients go through RoPE^T. */
int clipk_rope_qk(void* qkv, const float* rope_cos, const float* rope_sin, int B, int L, int H, int D, void* stream);
/* clipk_rope_qk followed by clipk_attn_fwd(rope = NULL) in one call: q / k in `qkv` are rotated IN PLACE and the
 * attention output / lse computed from the rotated values (same bits as the two calls).  Short heads (D in
 * {16, 24, 32}, 128 < L <= 256) do both in one kernel - one workgroup owns every row of a head, rotates it while
 * staging and writes it back; other shapes make the two calls.  Backward: clipk_attn_bwd(..., prerotated = 1). */
int clipk_attn_fwd_rot(void* qkv, const uint8_t* key_mask, const float* rope_cos, const float* rope_sin,
                       void* out, float* lse, int B, int L, int H, int D, float q_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Exact-f32 multi-head self-attention (f32 q / k / v, f32 scores, f32 softmax, f32 P·V; VALU fmas in a fixed order) for
 * the models the reference runs WITHOUT autocast: nn.MultiheadAttention inside the nn.TransformerEncoderLayer stacks of
 * RNARBPCLIPModel (current/rna_clip_codes.ipynb:1911-1954) and of the tri-modal ContrastiveModel
 * (current/tf_clip_codes (1).ipynb:13056-13072,13113-13176).  Layouts as clipk_attn_fwd / clipk_attn_bwd with f32
 * tensors: qkv [B*L, 3*H*D], out / dout [B*L, H*D], dqkv [B*L, 3*H*D], lse / delta [B, H, L]; key_mask u8 [B, L]
 * (1 = valid) or NULL; same dropout mask (hash, element index) as the bf16 kernels.  Any D in 1..192 (no multiple-of-8
 * rule: the notebook's 120 / 8 = 15 runs unpadded).  A row whose keys are all masked gives a zero output row and
 * lse = -inf, as clipk_attn_fwd. */
int clipk_attn_f32_fwd(const float* qkv, const uint8_t* key_mask, float* out, float* lse, int B, int L, int H, int D,
                       float q_scale, float dropout_p, uint32_t dropout_seed, void* stream);
int clipk_attn_f32_bwd(const float* qkv, const uint8_t* key_mask, const float* out, const float* dout, const float* lse,
                       float* delta /* scratch [B,H,L] */, float* dqkv, int B, int L, int H, int D, float q_scale,
                       float dropout_p, uint32_t dropout_seed, void* stream);
/* y[i] = x[i] * keep(seed, i) / (1 - p) (+ addend[i]): nn.Dropout on an f32 tensor (out_proj / linear2 outputs in front
 * of their residual add, the FFN activation: nn.TransformerEncoderLayer(dropout = p), rna_clip_codes.ipynb:1915) with
 * the counter-based mask of the GEMM epilogues (element index = row-major position); the backward is the same call on
 * the gradient.  addend may be NULL; y may alias x. */
int clipk_dropout_f32(const float* x, const float* addend, float* y, int64_t n, float p, uint32_t seed, void* stream);
/* Dropout under hipGraph replay (training.GraphedTrainStep with nn.TransformerEncoderLayer's dropout = 0.1 active,
 * rna_clip_codes.ipynb:1915, :2061-2089): a captured launch carries its seed as a constant, so every replay would repeat
 * the mask.  While a device word is registered here, EVERY dropout site of the library (clipk_gemm_nt's drop_p,
 * clipk_attn_*'s dropout_p, clipk_attn_f32_*, clipk_dropout_f32, clipk_layernorm_bwd's drop_p) uses
 * seed + *epoch_dev * 0x9E3779B9 instead of seed, read when the kernel RUNS: the captured step increments the word once per
 * replay, after its backward (which re-draws the forward's masks from the same seeds).  NULL (the default) restores the
 * plain seeds; launches made while nothing is registered are unaffected, bit for bit.  Process-wide. */
int clipk_set_dropout_epoch(const uint32_t* epoch_dev);

/* ------------------------------------------------------------------------------------------------
 * Token embedding (ESM-2): x[t,:] = table[ids[t],:] * scale[b] * mask[t], with the token-dropout
 * rescale (modeling_esm.py:252-268) folded into row_scale[B] by the caller.  f32 out.
 */
int clipk_embed_fwd(const int64_t* ids, const float* table, const float* row_scale /*[B] or NULL*/,
                    const uint8_t* mask /*[B*L] or NULL*/, int mask_token_id,
                    float* x, int B, int L, int d, int V /* table rows: ids outside [0, V) give NaN rows */, void* stream);
/* dtable[V,d] += sum over tokens (torch embedding backward).  V <= 64 (ESM-2: 33): one-hot product on the exact-f32
 * matrix pipe with fixed-order reductions, bitwise reproducible, needs the workspace below; larger vocabularies (or
 * workspace == NULL): per-block LDS tables + float atomics. */
size_t clipk_embed_bwd_workspace(int B, int L, int d, int V);
int clipk_embed_bwd(const int64_t* ids, const float* dx, const float* row_scale, const uint8_t* mask,
                    int mask_token_id, float* dtable, int B, int L, int d, int V,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Pooling over the sequence: mode 0 = position 0 (rna_clip_codes.ipynb:1948), 1 = masked mean
 * (configuration_hybrid_clip.py:109 use_mean_pooling).  x f32 [B,L,d] -> y f32 [B,d]. */
int clipk_pool_fwd(const float* x, const uint8_t* mask, float* y, int B, int L, int d, int mode, void* stream);
int clipk_pool_bwd(const float* dy, const uint8_t* mask, float* dx, int B, int L, int d, int mode, void* stream);

/* Pooling over packed variable-length batches (sequence b = rows [cu[b], cu[b+1]) of x f32 [T, d]): mode 0 = the first
 * row, 1 = mean over the sequence's rows; backward writes every row of dx [T, d].  cu_seqlens: DEVICE int32 [B+1]. */
int clipk_pool_varlen_fwd(const float* x, const int* cu_seqlens, float* y, int B, int d, int mode, void* stream);
int clipk_pool_varlen_bwd(const float* dy, const int* cu_seqlens, float* dx, int B, int d, int mode, void* stream);

/* Attention over PACKED variable-length batches.  The reference pads every batch to its longest
 * sequence with NaN rows and masks them as keys (current/rna_clip_codes.ipynb:1824-1857 collate_fn /
 * create_padding_mask, :1936-1946; lengths 30..2542), so padded rows still run through every Linear, LayerNorm and
 * attention row.  Here sequence b is rows [cu_seqlens[b], cu_seqlens[b+1]) of the packed tensors:
 *   qkv bf16 [T, 3*H*D], out / dout bf16 [T, H*D], dqkv bf16 [T, 3*H*D], lse / delta f32 [H, T];
 *   cu_seqlens: DEVICE int32 [B+1] (cu[0] = 0, cu[B] = T); max_len = longest sequence (grid sizing only);
 *   rope tables (optional, ESM head dims): f32 [>= max_len, D/2], indexed by the position inside the sequence.
 * Arithmetic is that of clipk_attn_fwd / clipk_attn_bwd on each sequence alone. */
int clipk_attn_varlen_fwd(const void* qkv, const int* cu_seqlens, const float* rope_cos, const float* rope_sin,
                          void* out, float* lse, int B, int T, int max_len, int H, int D, float q_scale, float dropout_p,
                          uint32_t dropout_seed, void* stream);
/* clipk_attn_fwd_rot for a packed batch: rotates q / k of every sequence IN PLACE (positions count from the sequence's
 * first row) while the whole-head kernel stages them, for D in {16, 24, 32} and 128 < max_len <= 256 only
 * (CLIPK_ERR_UNSUPPORTED otherwise: use clipk_attn_varlen_fwd, which leaves qkv alone).  Its backward is
 * clipk_attn_varlen_bwd(..., prerotated = 1): one whole-head kernel per (sequence, head) instead of the general pair. */
int clipk_attn_varlen_fwd_rot(void* qkv, const int* cu_seqlens, const float* rope_cos, const float* rope_sin,
                              void* out, float* lse, int B, int T, int max_len, int H, int D, float q_scale, void* stream);
int clipk_attn_varlen_bwd(const void* qkv, const int* cu_seqlens, const float* rope_cos, const float* rope_sin,
                          const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                          int B, int T, int max_len, int H, int D, float q_scale, int prerotated, float dropout_p,
                          uint32_t dropout_seed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser step on flat f32 buffers: AdamW (decoupled weight decay, torch.optim.AdamW semantics,
 * rna_clip_codes.ipynb:2033) with the global-norm clip of clip_grad_norm_ (ipynb:2076) folded in:
 *   sumsq kernel -> grad_norm_sq[0] (device), then the update reads it to compute the clip factor.
 * wd_mask: per-element 0/1 float or NULL (all decayed).  Also refreshes the bf16 weight copy.
 */
size_t clipk_sumsq_workspace(int64_t n);
int clipk_sumsq(const float* g, int64_t n, float* out /* device scalar, overwritten */,
                void* workspace, size_t workspace_bytes, void* stream);
/* hyper_dev (optional): DEVICE array {lr, 1 - beta1^t, sqrt(1 - beta2^t)} read by the kernel instead of `lr` / `step` -
 * what changes from step to step lives in memory, so that a whole training step (rna_clip_codes.ipynb:2061-2089) can be
 * captured once in a hipGraph and replayed (clip_dplm_amd.training.GraphedTrainStep); NULL: the scalars. */
int clipk_adamw_step(float* w, const float* g, float* m, float* v, void* w_bf16 /* or NULL */,
                     int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                     int step, const float* grad_norm_sq /* device scalar or NULL */, float max_norm,
                     float grad_scale, const float* hyper_dev /* or NULL */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLIPK_H */
