/*
 * libuniter_hip.so -- C ABI of the MI355X-native UNITER fine-tuning hot path.
 *
 * The reference (Nithin-Holla/meme_challenge) has no FFI of its own: its hot
 * path is the nn.Module surface of model/model.py, model/layer.py and
 * model/meme_uniter.py dispatching to ATen/cuBLAS/Apex kernels.  Every entry
 * point below names the reference code (file:line, relative to the reference
 * root) whose device work it replaces.  A maintainer binds them with ctypes
 * (see INTEGRATION.md); `meme_challenge_amd/_lib.py` is that binding.
 *
 * Conventions
 *  - all pointers are CALLER-OWNED DEVICE pointers (fp32 unless stated), row-major,
 *    contiguous unless a leading dimension is given; the library never allocates
 *    or frees user tensors; scratch comes from a caller-provided workspace
 *  - every call takes the hipStream_t to launch on (as void*) and is asynchronous,
 *    stream-ordered and re-entrant; no host synchronisation inside
 *  - return value: 0 = ok, <0 = invalid argument (UNITER_E_*), >0 = hipError_t
 *  - dropout: counter-based Philox4x32-10 keyed by (seed, offset, site); the same
 *    (seed, offset) passed to forward and backward regenerates identical masks
 *    (specification: oracle/philox.py; device code: csrc/philox.h)
 *  - one process = one device = one rank
 */
#ifndef UNITER_HIP_H
#define UNITER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNITER_ABI_VERSION 1

#define UNITER_E_ARG   (-1)   /* bad pointer / size / alignment            */
#define UNITER_E_SHAPE (-2)   /* unsupported shape (e.g. head_dim != 64)   */
#define UNITER_E_WS    (-3)   /* workspace too small                       */
#define UNITER_E_STATE (-4)   /* call order violated                       */

int         uniter_abi_version(void);
const char* uniter_last_error(void);          /* thread-local message of the last failure */
const char* uniter_build_info(void);          /* "gfx950 ..." */

/* ------------------------------------------------------------------------- *
 * Dense contraction (replaces nn.Linear / torch.matmul dispatch to cuBLAS:
 * model/layer.py:76-78,112,140,153; model/model.py:267; and their autograd).
 *   C[M,N] (+)= epilogue( sum_k A(m,k) * B(k,n) )      fp32 MFMA (v_mfma_f32_32x32x2_f32)
 *   a_kmajor = 0: A(m,k) = A[m*lda + k]      1: A(m,k) = A[k*lda + m]
 *   b_kmajor = 0: B(k,n) = B[n*ldb + k]      1: B(k,n) = B[k*ldb + n]
 *     (a_kmajor=0,b_kmajor=0 is x @ W^T with W stored [out,in] like nn.Linear)
 *   epilogue: UNITER_EPI_*; beta = 0 overwrite, 1 accumulate into C.
 * ------------------------------------------------------------------------- */
enum {
  UNITER_EPI_NONE      = 0,
  UNITER_EPI_BIAS      = 1,   /* + bias[n]                                             */
  UNITER_EPI_BIAS_GELU = 2,   /* u = acc + bias[n]; aux_out = u; C = gelu_erf(u)       */
  UNITER_EPI_DGELU     = 3,   /* C = acc * gelu_erf'(aux_in[m,n])                      */
  UNITER_EPI_ADD       = 4,   /* C = acc + aux_in[m,n]                                 */
  /* the forward pass hands the backward pass gelu'(u) instead of u, so the
   * backward epilogue is one multiply (the erf / exp are evaluated once per element, not twice) */
  UNITER_EPI_BIAS_GELU_D = 5, /* u = acc + bias[n]; aux_out = gelu_erf'(u); C = gelu_erf(u) */
  UNITER_EPI_MUL       = 6    /* C = acc * aux_in[m,n]                                 */
};
int uniter_gemm_f32(int a_kmajor, int b_kmajor, int M, int N, int K,
                    const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, int epilogue, const float* bias,
                    const float* aux_in, float* aux_out, int ld_aux,
                    int beta, void* stream);
/* dW_p[M_p, N_p] (+)= A_p^T B_p for up to four products of ONE reduction length K (A_p [K, M_p], B_p [K, N_p], dW_p with
 * leading dimension N_p; overwrite != 0: `=`), one persistent launch of whole-K 64 x 64 tiles: the four weight gradients of an
 * encoder layer (autograd of the nn.Linear of model/layer.py:76-78,112,140,153), no partial tiles, no float atomics between
 * workgroups.  UNITER_E_SHAPE: a product outside the kernel's range (M, N % 4, 31-bit offsets) -- launch them one by one. */
int uniter_wgrad_f32_group(int n, const int* M, const int* N, int K, const float* const* A, const float* const* B,
                           float* const* dW, int overwrite, void* stream);
/* tile-configuration override for tuning (0 = heuristic). */
int uniter_gemm_f32_cfg(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K,
                    const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, int epilogue, const float* bias,
                    const float* aux_in, float* aux_out, int ld_aux,
                    int beta, void* stream);

/* Mixed precision (BASELINE config 3, "bf16 MFMA for the dense GEMMs"): same contract and fp32
 * operands/outputs as uniter_gemm_f32_cfg, but A and B are rounded to bf16 (RNE) on their way
 * into LDS and multiplied with v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  Shapes with
 * K % 64 != 0 run on the exact fp32 kernel. */
int uniter_gemm_bf16_cfg(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K,
                    const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, int epilogue, const float* bias,
                    const float* aux_in, float* aux_out, int ld_aux,
                    int beta, void* stream);
/* All-bf16 operands (resident bf16 activations / the bf16 mirror of the weights), fp32 accumulate;
 * writes C (fp32, may be NULL) and / or C_bf16 (bf16 copy for the next GEMM, may be NULL).  lda / ldb /
 * ldcb count bf16 elements.  Layouts: (0,0) x @ W^T, (0,1) dgrad, (1,1) wgrad.  cfg 0 / 1 / 4.
 * K % 64 == 0 (any K for (1,1): rows beyond K read as zero), leading dimensions % 8, 16-byte aligned operands.
 * beta = 1 (C += product) is formed with fp32 atomic adds -- no read of C in the kernel -- and has no bf16 copy
 * (C_bf16 must be NULL).  In every epilogue the loads of a 32x32 output block (bias, aux_in) are issued before its
 * first store: loads and stores share the vector-memory counter on gfx9. */
int uniter_gemm_bf16res_cfg(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K,
                            const void* A, int lda, const void* B, int ldb,
                            float* C, int ldc, void* C_bf16, int ldcb, int epilogue, const float* bias,
                            const float* aux_in, float* aux_out, int ld_aux, int beta, void* stream);
/* Second-generation bf16-resident product for the forward and input-gradient GEMMs (csrc/gemm_bf16_dma.hip;
 * replaces cuBLAS behind nn.Linear of model/layer.py:76-78,112,140,153 in the bf16 mode): operands reach LDS by
 * LDS-DMA through a ring of stages with counted vmcnt waits; the accumulator is held transposed so outputs leave
 * as 16-byte row stores.  Layouts (a_kmajor, b_kmajor): (0,0) x @ W^T, (0,1) dgrad.  cfg 1 = 128x128 tile
 * (2 stages, two workgroups per CU), 2 = 128x256, 3 = 256x128 (8 waves, 3 stages), 4 = 128x128 (3 stages),
 * 5 = 64x128 (2 waves; the few-tile shapes), 0 = choose.
 * nsplit > 1 cuts K into pieces computed by different workgroups: piece s stores its fp32 partial tile at
 * C + s * c_split_stride (piece 0 applies the epilogue) and the CONSUMER adds the slabs (no bf16 output then).
 * aux_in / aux_out are fp32 or bf16 ([M, ld_aux] elements) as the *_bf16 flags say.  beta = 1: C += product
 * by fp32 atomics (epilogue NONE, no bf16 copy).  N % 8 == 0, K % 64 == 0, 16-byte aligned buffers. */
int uniter_gemm_bf16v2_cfg(int cfg, int nsplit, int a_kmajor, int b_kmajor, int M, int N, int K,
                           const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                           long c_split_stride, void* C_bf16, int ldcb, int epilogue, const float* bias,
                           const void* aux_in, int aux_in_bf16, void* aux_out, int aux_out_bf16,
                           int ld_aux, int beta, void* stream);
/* dst[i] = bf16(src[i]) (round to nearest even); n % 4 == 0 */
int uniter_cast_bf16(const float* src, void* dst, size_t n, void* stream);

/* out[n] (+)= sum_m X[m*ld + n]   (bias gradients of every nn.Linear) */
int uniter_colsum_f32(const float* X, int M, int N, int ld, float* out, int beta,
                      void* ws, size_t ws_bytes, void* stream);
size_t uniter_colsum_ws_bytes(int M, int N);

/* ------------------------------------------------------------------------- *
 * Fused dropout + residual + LayerNorm (replaces nn.Dropout + add + Apex
 * FusedLayerNorm: model/layer.py:113-114,154-155; eps=1e-12, biased variance).
 *   z = dropout_p(x) + res ;  y = LN(z) * gamma + beta
 * z_out/mean/rstd may be NULL in inference.  res may be NULL.
 * ------------------------------------------------------------------------- */
int uniter_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta,
                  float* z_out, float* y, float* mean, float* rstd, int M, int H,
                  float p_drop, uint64_t seed, uint32_t offset, uint32_t site,
                  void* stream);
/*   dz = dLN(dy) (gradient w.r.t. z, i.e. w.r.t. the residual input)
 *   dx = dropout-masked dz (gradient w.r.t. x); dx may equal dz when p_drop == 0
 *   dgamma/dbeta += column reductions (two-stage, deterministic)
 *   dbias (optional) += column sum of dx = bias gradient of the Linear that produced x
 *   (model/layer.py:112,153), fused here to save a pass over dx                         */
int uniter_ln_bwd(const float* dy, const float* z, const float* mean, const float* rstd,
                  const float* gamma, float* dz, float* dx, float* dgamma, float* dbeta,
                  float* dbias, int M, int H, float p_drop, uint64_t seed, uint32_t offset,
                  uint32_t site, void* ws, size_t ws_bytes, void* stream);
/* As uniter_ln_fwd / uniter_ln_bwd with an additional bf16 (round-to-nearest-even) copy of y / dx for the
 * bf16-resident GEMMs (NULL = none): saves a separate uniter_cast_bf16 pass over the activation. */
int uniter_ln_fwd_b16(const float* x, const float* res, const float* gamma, const float* beta,
                      float* z_out, float* y, void* y_bf16, float* mean, float* rstd, int M, int H,
                      float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* stream);
/* uniter_ln_fwd_b16 whose input x is the SUM of nslab fp32 slabs x + s * slab_stride (split-K partial sums left by
 * uniter_gemm_bf16v2_cfg: the reduction happens here, in the consumer's row pass, instead of in the GEMM). */
int uniter_ln_fwd_slabs(const float* x, int nslab, size_t slab_stride, const float* res, const float* gamma,
                        const float* beta, float* z_out, float* y, void* y_bf16, float* mean, float* rstd,
                        int M, int H, float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* stream);
int uniter_ln_bwd_b16(const float* dy, const float* z, const float* mean, const float* rstd,
                      const float* gamma, float* dz, float* dx, void* dx_bf16, float* dgamma, float* dbeta,
                      float* dbias, int M, int H, float p_drop, uint64_t seed, uint32_t offset,
                      uint32_t site, void* ws, size_t ws_bytes, void* stream);
/* uniter_ln_bwd_b16 in two calls: _rows writes dz / dx (what the backward chain needs) and the per-block
 * column partials into ws; _finalize reduces them into dgamma / dbeta / dbias (+=) and may run later on
 * another stream (ws must stay untouched in between). */
int uniter_ln_bwd_rows(const float* dy, const float* z, const float* mean, const float* rstd,
                       const float* gamma, float* dz, float* dx, void* dx_bf16, int want_dbias, int M, int H,
                       float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* ws, size_t ws_bytes,
                       void* stream);
/* uniter_ln_bwd_rows whose upstream gradient dy is the sum of nslab fp32 slabs (see uniter_ln_fwd_slabs). */
int uniter_ln_bwd_rows_slabs(const float* dy, int nslab, size_t slab_stride, const float* z, const float* mean,
                             const float* rstd, const float* gamma, float* dz, float* dx, void* dx_bf16,
                             int want_dbias, int M, int H, float p_drop, uint64_t seed, uint32_t offset,
                             uint32_t site, void* ws, size_t ws_bytes, void* stream);
/* The weight gradients of one encoder layer in ONE launch (csrc/gemm_bf16_dma.hip; replaces the autograd products
 * dW = dY^T X behind nn.Linear of model/layer.py:76-78,112,140,153 in the bf16 mode): for p < n (n <= 4)
 * dW[p] [M[p], N[p]] (fp32, leading dimension N[p]) += A[p]^T B[p] with A[p] [K, M[p]] and B[p] [K, N[p]] bf16
 * row-major (both operands k-major, any K).  The 128 x 128 tiles of all products are numbered through; every tile is
 * owned by one workgroup over the whole K, so dW += is a plain read-modify-write -- no atomics, no partial sums, and
 * the result does not depend on the launch (bit-reproducible).  cfg 0/1 = two LDS stages, 4 = three.
 * M[p] % 8 == 0, N[p] % 8 == 0, 16-byte aligned buffers; the dW[p] must not overlap. */
int uniter_wgrad_bf16_group(int cfg, int n, const int* M, const int* N, int K, const void* const* A,
                            const void* const* B, float* const* dW, void* stream);
/* ---- fp32-accurate products on the bf16 matrix pipe: "x3" operands (csrc/gemm_split3.hip) -----------------------
 * An fp32 value is exactly the sum of three bf16 pieces (round-to-nearest residuals).  The x3 form of a [rows][cols]
 * fp32 tensor holds piece p of element (r, c) at  base + r * row_stride + p * piece_stride + c  (bf16 elements):
 * activations as [rows][3][ld] (row_stride 3 ld, piece_stride ld), weights piece-major (row_stride ld, piece_stride =
 * the flat buffer's length).  A product of two x3 operands runs as SIX bf16 MFMA products per block (a1 b1, a1 b2,
 * a2 b1, a1 b3, a2 b2, a3 b1; fp32 accumulate; the dropped terms are <= 2^-24 |a b|): fp32 accuracy at 6/16 of the fp32
 * matrix pipe's time.  Replaces cuBLAS behind nn.Linear of model/layer.py:76-78,112,140,153 (forward, input gradient,
 * weight gradient) in the fp32 mode `fp32x3` (uniter_model_set_precision 3).
 *
 * uniter_split3: fp32 [rows][ld] -> x3;  uniter_join3: the exact sum of the pieces back to fp32.
 * cols % 8 == 0, ld % 4 == 0, strides % 8 == 0, 16-byte aligned buffers. */
int uniter_split3(const float* x, int rows, int cols, int ld, void* x3, size_t row_stride, size_t piece_stride,
                  void* stream);
int uniter_join3(const void* x3, int rows, int cols, size_t row_stride, size_t piece_stride, float* x, int ld,
                 void* stream);
/* C [M, N] (fp32, optional; slab s of a split-K launch at C + s * c_split_stride) and / or C_x3 (optional; row stride
 * ldcx, piece stride pscx) = epilogue(A . B^T):  A x3 of M rows (a_kmajor: K rows), row stride lda, piece stride psa;
 * B x3 of N rows (b_kmajor: K rows).  Layouts and epilogues: forward (a_kmajor = b_kmajor = 0): NONE, BIAS, BIAS_GELU_D
 * (aux_out = gelu'(u) fp32, outputs = gelu(u)); input gradient (b_kmajor = 1): NONE, ADD, MUL (aux_in fp32 [M, ld_aux]);
 * weight gradient (both 1, any K; operands [K][3][.]): NONE, ADD.  K % 32 == 0 unless both operands are k-major;
 * N % 8 == 0.  Rows beyond an operand's row count read as zeros only in the [rows][3][ld] form (what a k-major
 * operand with a ragged K needs).  cfg: 0 = choose, 1..3 = wave geometry and MFMA shape of 128 x 128 tiles, 4 = 128 x 256 tiles (not for weight
 * gradients), 5 = 128 x 192 tiles (forward layout, fp32 output only) (gemm_split3.hip).  nsplit > 1: fp32 slabs only;
 * piece 0 applies the epilogue, the consumer adds the slabs. */
int uniter_gemm_x3_cfg(int cfg, int nsplit, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int lda,
                       int psa, const void* B, int ldb, int psb, float* C, int ldc, long c_split_stride, void* C_x3,
                       int ldcx, int pscx, int epilogue, const float* bias, const float* aux_in, float* aux_out,
                       int ld_aux, void* stream);
/* What uniter_gemm_x3_cfg (cfg 0) and the model's plan choose for a forward / input-gradient product on `avail_cus` CUs (0 = the
 * chip's): tile geometry (3 = 128 x 128 persistent, 4 = 128 x 256) and k-pieces (nsplit_fixed > 0: the caller's; 0: 1..4 compete
 * where N <= 1024).  Host-only arithmetic (no launch): the cost model of DESIGN.md section 4 -- rounds x k-tiles x time per
 * k-tile + 4 us per slab.  With a data-parallel exchange holding CUs (uniter_model_set_cu_reserve) the backward pass is planned
 * with it: 252-item forms that exactly fit 256 CUs would run two rounds on 240. */
int uniter_gemm_x3_plan(int M, int N, int K, int avail_cus, int nsplit_fixed, int* cfg, int* nsplit);
/* The same for a forward product (weights k-contiguous) with an fp32 output -- the query|key|value projection of model/layer.py:76-78:
 * 128 x 192 tiles (cfg 5) compete too (what uniter_gemm_x3_cfg's cfg 0 picks there); bench.py prices the launch's staging floor from it. */
int uniter_gemm_x3_plan_fwd32(int M, int N, int K, int avail_cus, int nsplit_fixed, int* cfg, int* nsplit);
/* The weight gradients of one encoder layer in one launch on x3 operands (as uniter_wgrad_bf16_group): for p < n <= 4
 * dW[p] [M[p], N[p]] (fp32) (+)= A[p]^T B[p], A[p] x3 [K][3][M[p]], B[p] x3 [K][3][N[p]]; whole-K 128 x 128 tiles, no
 * atomics, bit-reproducible.  overwrite = 1 stores instead of adding; max_wgs > 0 caps the grid (the persistent
 * workgroups walk the tiles; default one per CU); cfg as uniter_gemm_x3_cfg. */
int uniter_wgrad_x3_group(int cfg, int n, const int* M, const int* N, int K, const void* const* A,
                          const void* const* B, float* const* dW, int overwrite, int max_wgs, void* stream);
/* The same launch with RIDERS: the side work a layer's backward pass would otherwise launch separately, done by the workgroups
 * of the weight-gradient launch itself (round 5; model/layer.py:76-78,112,140,153 backward, train_template.py:104 clip norm):
 *  - colsum_out[m] += sum_k A[0][k][m]: the column sums of product 0's A operand (A = dY: the bias gradient that belongs to
 *    this weight gradient), by three more MFMAs per row block and k-tile in the tiles of the first tile column;
 *  - up to 4 column-reduction jobs out[j][c / seg][c % seg] += sum_p part[j][p * stride + c] for c < n[j] (the LayerNorm
 *    backward passes' [dgamma | dbeta | dbias] partial rows, the attention backward's per-sample query|key|value bias
 *    partials), 64 columns per item, run by the workgroups with one tile less while the first k-tiles are staged;
 *  - ssq[4 * workgroup + wave] = the sum of squares of EVERYTHING that wave wrote (weight-gradient tiles after the add,
 *    colsum_out, reduced vectors), in double: the clip norm's share of this launch as unreduced partial sums in fixed slots
 *    (no atomics: bit-reproducible); uniter_sumsq_combine joins them.  uniter_wgrad_x3_group_slots = slots written.
 * `grid`, `nred`, `first_item` are filled in by the call.  cfg 0 (choose), 3 (128 x 128 tiles, 4 slots per workgroup) or 4 (128 x 256
 * tiles, 8 slots per workgroup: one per compute wave). */
typedef struct uniter_x3_riders {
  double* ssq;
  float* colsum_out;
  int grid, njobs, nred;
  const float* part[4];
  int nparts[4], stride[4], n[4], seg[4];
  float* out[4][3];
  int first_item[5];
} uniter_x3_riders_t;
/* uniter_gemm_x3_cfg with epilogue UNITER_EPI_MUL, one k-piece, that also leaves partial column sums of its output in
 * colsum_part [(M + 63) / 64][N] (row i = the sum of output rows 64 i .. 64 i + 63): the bias gradient of a dense layer from the
 * product that WRITES its dY (dU of model/layer.py:140's backward), finished by a column-reduction job of the riders above. */
int uniter_gemm_x3_colpart(int cfg, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int lda, int psa,
                           const void* B, int ldb, int psb, float* C, int ldc, void* C_x3, int ldcx, int pscx,
                           const float* aux_in, int ld_aux, float* colsum_part, void* stream);
int uniter_wgrad_x3_group_riders(int cfg, int n, const int* M, const int* N, int K, const void* const* A,
                                 const void* const* B, float* const* dW, int overwrite, int max_wgs,
                                 uniter_x3_riders_t* riders, void* stream);
int uniter_wgrad_x3_group_slots(int cfg, int n, const int* M, const int* N, int max_wgs);
/* The BALANCED WALK of the 128 x 256-tile launches (round 5).  A persistent launch whose tiles do not fill whole rounds of the chip
 * (216 weight-gradient tiles or the QKV product's 189 on 256 CUs) leaves CUs idle for the length of a tile; given a workspace these
 * entry points cut the k-tiles of all tiles, in tile order, into one equal run per workgroup instead: a tile whose k-range is cut is
 * finished by the workgroup that holds its first part, which adds the partial sums the following workgroups stored (fixed order:
 * reproducible run to run; the sum differs from the whole-tile walk's by fp32 rounding of the regrouped additions).
 * Workspace: uniter_gemm_x3_balanced_ws_bytes() bytes, 256-byte aligned; its first 16 KB (flag words) ZERO before the first launch --
 * every launch leaves them zero; launches that may run at the same time (two streams) need a workspace each.  ws = NULL, a
 * geometry other than 128 x 256 or rounds that are full anyway: the classic walk.  (The model passes a workspace only under
 * UNITER_X3_BALANCED: in the training step the balanced walk loses -- DESIGN.md section 9.)
 *  - uniter_gemm_x3_cfg_ws: uniter_gemm_x3_cfg + workspace (balanced: forward layout, UNITER_EPI_BIAS, one k-piece, cfg 0 / 4);
 *  - uniter_wgrad_x3_group_ws: uniter_wgrad_x3_group_riders + workspace (riders may be NULL); a balanced launch runs on every CU it
 *    may use, so it writes more sum-of-squares slots: uniter_wgrad_x3_group_slots_ws(…, K, max_wgs, ws_bytes) counts them. */
size_t uniter_gemm_x3_balanced_ws_bytes(void);
int uniter_gemm_x3_cfg_ws(int cfg, int nsplit, int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int lda,
                          int psa, const void* B, int ldb, int psb, float* C, int ldc, long c_split_stride, void* C_x3,
                          int ldcx, int pscx, int epilogue, const float* bias, const float* aux_in, float* aux_out,
                          int ld_aux, void* ws, size_t ws_bytes, void* stream);
int uniter_wgrad_x3_group_ws(int cfg, int n, const int* M, const int* N, int K, const void* const* A,
                             const void* const* B, float* const* dW, int overwrite, int max_wgs,
                             uniter_x3_riders_t* riders, void* ws, size_t ws_bytes, void* stream);
int uniter_wgrad_x3_group_slots_ws(int cfg, int n, const int* M, const int* N, int K, int max_wgs, size_t ws_bytes);
/* The same riders on the grouped bf16 weight-gradient launch (uniter_wgrad_bf16_group; precision bf16): overwrite = 1 stores
 * instead of adding, max_wgs > 0 caps the grid; cfg must be 0 or 1 (two LDS stages). */
int uniter_wgrad_bf16_group_riders(int cfg, int n, const int* M, const int* N, int K, const void* const* A,
                                   const void* const* B, float* const* dW, int overwrite, int max_wgs,
                                   uniter_x3_riders_t* riders, void* stream);
int uniter_wgrad_bf16_group_slots(int n, const int* M, const int* N, int max_wgs);
/* Round 6: the PERSISTENT loader / compute form of the bf16-resident products (csrc/gemm_bf16_p.hip: the structure of the x3 kernels
 * for one-piece operands -- 4 loader waves run LDS-DMA two 64-deep k-tiles ahead of 4 or 8 compute waves across the work items of a
 * persistent workgroup, v_mfma_f32_16x16x32_bf16, XCD-chunked banded tile walk).  Reached through the entry points above:
 *  - uniter_gemm_bf16v2_cfg with cfg 6 (128 x 128 tiles), 7 (128 x 256: 48 instead of 64 KB staged per k-tile and pair of 128 x 128
 *    products) or 8 (128 x 192, k-contiguous weights only: the query|key|value projection's 252 tiles); epilogues none / bias /
 *    + aux / bias + GELU + gelu' / x aux, no accumulate (beta = 0), forward and input-gradient layouts;
 *  - uniter_wgrad_bf16_group(_riders) with cfg 7: a layer's four weight gradients as 216 whole-K tiles of 128 x 256 (one round of
 *    the chip); riders without colsum_out (the bias gradient of intermediate.dense comes from the producing product's column
 *    partials as a reduction job), 8 sum-of-squares slots per workgroup: uniter_wgrad_bf16_group_slots_cfg.
 * uniter_gemm_bf16p_plan: the geometry (6 / 7 / 8) and k-pieces the model's plan picks for a product on `avail_cus` CUs (0 = the
 * chip's; nsplit_fixed > 0: the caller's k-pieces) -- host arithmetic; bench.py prices the launch's staging floor from it. */
int uniter_wgrad_bf16_group_slots_cfg(int cfg, int n, const int* M, const int* N, int max_wgs);
int uniter_gemm_bf16p_plan(int M, int N, int K, int avail_cus, int nsplit_fixed, int b_kmajor, int* cfg, int* nsplit);
/* out[c] += sum_r X[r, c] for an x3 tensor X [rows][3][ldx] (bias gradient of intermediate.dense from dU). */
int uniter_colsum_x3_add(const void* x3, int rows, int cols, int ldx, float* out, void* stream);
/* uniter_ln_fwd_slabs / uniter_ln_bwd_rows_slabs whose operand copy for the next GEMM is x3 [M][3][H] instead of bf16 */
int uniter_ln_fwd_slabs_x3(const float* x, int nslab, size_t slab_stride, const float* res, const float* gamma,
                           const float* beta, float* z_out, float* y, void* y_x3, float* mean, float* rstd,
                           int M, int H, float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* stream);
int uniter_ln_bwd_rows_slabs_x3(const float* dy, int nslab, size_t slab_stride, const float* z, const float* mean,
                                const float* rstd, const float* gamma, float* dz, float* dx, void* dx_x3,
                                int want_dbias, int M, int H, float p_drop, uint64_t seed, uint32_t offset,
                                uint32_t site, void* ws, size_t ws_bytes, void* stream);
/* out[n] += sum_m X[m, n] for a bf16 matrix X [M, ld] (bias gradient of a dense layer from the bf16 gradient of its
 * output; replaces the autograd sum of model/layer.py:140 in the bf16 mode).  N % 8 == 0, ld % 8 == 0. */
int uniter_colsum_bf16_add(const void* X, int M, int N, int ld, float* out, void* stream);
/* out[i] += sum_s slabs[s * slab_stride + i], i < n: folds the fp32 k-piece slabs of a split-K weight-gradient product
 * (uniter_gemm_bf16v2_cfg with a_kmajor = b_kmajor = 1, nsplit > 1) into the gradient buffer -- what autograd's
 * accumulation into .grad does for model/layer.py's nn.Linear weights.  n, slab_stride multiples of 4. */
int uniter_slab_reduce_add(const float* slabs, int nslab, size_t slab_stride, float* out, size_t n, void* stream);
int uniter_ln_bwd_finalize(const void* ws, size_t ws_bytes, int M, int H, float* dgamma, float* dbeta,
                           float* dbias, void* stream);
size_t uniter_ln_bwd_ws_bytes(int M, int H);

/* ------------------------------------------------------------------------- *
 * Fused self-attention over the joint [text|region] sequence (replaces
 * BertSelfAttention.forward model/layer.py:85-100: QK^T / sqrt(d) + mask,
 * softmax, dropout on probabilities, P.V, head merge).
 *   qkv : [B*L, 3*H] rows = (b,l), columns = [Q | K | V], head h at h*64..h*64+63
 *   attn_mask : [B, L] 1 = attend, 0 = padded (additive (1-m)*-10000 as model/model.py:345)
 *   ctx : [B*L, H] merged heads;  lse : [B, nh, L] log-sum-exp of the scaled+masked scores
 * head_dim must be 64.
 * ------------------------------------------------------------------------- */
int uniter_attn_fwd(const float* qkv, const float* attn_mask, float* ctx, float* lse,
                    int B, int L, int nh, float p_drop, uint64_t seed, uint32_t offset,
                    uint32_t site, void* stream);
/*   dqkv : [B*L, 3*H] gradients w.r.t. qkv;  delta : [B, nh, L] scratch;
 *   ws : device scratch of uniter_attn_bwd_ws_bytes(B, L, nh) bytes (the dropped probabilities and
 *        score gradients [B*nh, Lr, Lr] x 2 that the dQ kernel hands to the dK/dV kernel when
 *        L <= 192; 0 bytes / may be NULL for longer sequences) */
size_t uniter_attn_bwd_ws_bytes(int B, int L, int nh);
int uniter_attn_bwd(const float* qkv, const float* attn_mask, const float* ctx,
                    const float* lse, const float* dctx, float* dqkv, float* delta,
                    int B, int L, int nh, float p_drop, uint64_t seed, uint32_t offset,
                    uint32_t site, void* ws, size_t ws_bytes, void* stream);
/* Packed (varlen) batches -- SURVEY 8(f) N3.  Sample b owns rows cu_seqlens[b] .. cu_seqlens[b+1]-1
 * of qkv / ctx / dqkv ([Mp, 3H] / [Mp, H], Mp = cu_seqlens[B]); every position of a sample is valid,
 * so there is no mask: the result equals uniter_attn_fwd/bwd on the right-padded layout at the valid
 * rows (padded keys carry weight exp(-10000) = 0 in fp32 there).  lse / delta stay [B, nh, Lmax];
 * ws as uniter_attn_bwd_ws_bytes(B, Lmax, nh).  Lmax <= uniter_attn_varlen_max_len() (192). */
int uniter_attn_varlen_max_len(void);
int uniter_attn_fwd_varlen(const float* qkv, const int32_t* cu_seqlens, float* ctx, float* lse,
                           int B, int Lmax, int nh, float p_drop, uint64_t seed, uint32_t offset,
                           uint32_t site, void* stream);
int uniter_attn_bwd_varlen(const float* qkv, const int32_t* cu_seqlens, const float* ctx,
                           const float* lse, const float* dctx, float* dqkv, float* delta,
                           int B, int Lmax, int nh, float p_drop, uint64_t seed, uint32_t offset,
                           uint32_t site, void* ws, size_t ws_bytes, void* stream);
/* General forms of the L <= uniter_attn_varlen_max_len() kernels: exactly one of attn_mask / cu_seqlens,
 * plus optional bf16 copies of ctx / dqkv (operands of the bf16-resident GEMMs; NULL = none) and, for the
 * backward pass, bias_part [B, 3H] (NULL = none): per-sample column sums of dqkv, i.e. the gradient of the
 * fused query|key|value bias before the sum over the batch -- saves a 24 MB re-read of dqkv. */
/* keep_bits (NULL = none): uniter_attn_keep_bits_bytes(B, L, nh) bytes; the forward pass stores the dropout
 * keep flags it drew (one bit per probability) and the backward pass reads them instead of evaluating
 * Philox a second time -- same masks, ~0.75 MB per layer at B = 16. */
size_t uniter_attn_keep_bits_bytes(int B, int L, int nh);
int uniter_attn_fwd_ex(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, float* ctx,
                       void* ctx_bf16, float* lse, void* keep_bits, int B, int L, int nh, float p_drop,
                       uint64_t seed, uint32_t offset, uint32_t site, void* stream);
/* The keep flags drawn AHEAD of the forward pass, for `nlayers` layers in one launch (layer l at keep_bits +
 * l * layer_stride_bytes, dropout site site0 + l * site_step): ten dependent Philox rounds per four keys stall an attention
 * kernel's three waves per SIMD, a full-occupancy elementwise pass hides them.  The _pre forms of the forward pass with
 * keep_bits_ready = 1 READ keep_bits instead of drawing (and storing) them: identical masks, identical results. */
int uniter_attn_keep_bits_gen(void* keep_bits, size_t layer_stride_bytes, int nlayers, int B, int L, int nh, float p_drop,
                              uint64_t seed, uint32_t offset, uint32_t site0, uint32_t site_step, void* stream);
int uniter_attn_fwd_pre(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, float* ctx, void* ctx_bf16,
                        float* lse, void* keep_bits, int keep_bits_ready, int B, int L, int nh, float p_drop, uint64_t seed,
                        uint32_t offset, uint32_t site, void* stream);
int uniter_attn_bwd_ex(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens,
                       const float* ctx, const float* lse, const float* dctx, float* dqkv, void* dqkv_bf16,
                       float* bias_part, const void* keep_bits, float* delta, int B, int L, int nh, float p_drop,
                       uint64_t seed, uint32_t offset, uint32_t site, void* ws, size_t ws_bytes, void* stream);
/* uniter_attn_fwd_pre / uniter_attn_bwd_ex whose operand copies are x3 pieces (csrc/gemm_split3.hip) instead of bf16:
 * ctx_x3 [B*L][3][H], dqkv_x3 [B*L][3][3H]; with dqkv_x3 given the fp32 dqkv may be NULL (nothing else reads it in the
 * fp32x3 mode: the bias partials come from the kernel).  model/layer.py:85-100. */
int uniter_attn_fwd_pre_x3(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, float* ctx,
                           void* ctx_x3, float* lse, void* keep_bits, int keep_bits_ready, int B, int L, int nh,
                           float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* stream);
int uniter_attn_bwd_ex_x3(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, const float* ctx,
                          const float* lse, const float* dctx, float* dqkv, void* dqkv_x3, float* bias_part,
                          const void* keep_bits, float* delta, int B, int L, int nh, float p_drop, uint64_t seed,
                          uint32_t offset, uint32_t site, void* ws, size_t ws_bytes, void* stream);
/* The fp32x3 mode's attention with its PRODUCTS on the bf16 matrix pipe too (csrc/attention_x3.hip): Q, K, V, the
 * probabilities, dO and the score gradients each as three bf16 pieces, six v_mfma_f32_16x16x32_bf16 products per block with
 * fp32 results -- fp32 arithmetic like uniter_attn_fwd_pre_x3 / uniter_attn_bwd_ex_x3 (same results to fp32 rounding), whose
 * fp32 MFMAs run at 1/16 of that pipe's rate.  L <= uniter_attn_x3_max_len() (192); dropout reads the keep flags that
 * uniter_attn_keep_bits_gen drew (keep_bits may be NULL only with p_drop == 0); the backward pass needs no workspace: it
 * recomputes the scores in its dK / dV pass instead of handing probabilities over through memory.  Outputs: ctx and / or
 * ctx_x3 [rows][3][H]; dqkv (may be NULL) and / or dqkv_x3 [rows][3][3H]; bias_part [B, 3H] optional; lse, delta [B, nh, L].
 * dctx may arrive as dctx_slabs k-pieces of the attention-output input gradient (slab s at dctx + s * dctx_slab_stride
 * elements; the product then fills the chip with two k-pieces per tile): the kernel sums them while it reads them.
 * model/layer.py:80-100. */
int uniter_attn_x3_max_len(void);
int uniter_attn_x3_fwd(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, float* ctx, void* ctx_x3,
                       float* lse, const void* keep_bits, int B, int L, int nh, float p_drop, void* stream);
int uniter_attn_x3_bwd(const float* qkv, const float* attn_mask, const int32_t* cu_seqlens, const float* ctx,
                       const float* lse, const float* dctx, int dctx_slabs, size_t dctx_slab_stride, float* dqkv,
                       void* dqkv_x3, float* bias_part, const void* keep_bits, float* delta, int B, int L, int nh,
                       float p_drop, void* stream);
/* The bf16 mode's attention (uniter_attn_bf16_fwd_pre / uniter_attn_bf16_bwd below: same arithmetic, same rounding points) in the
 * decomposition of csrc/attention_x3.hip: one wave per 16 rows, one LDS image per operand read row-wise and transposed, backward in
 * ONE launch without the Pd / dS scratch (its dK / dV pass recomputes the scores).  qkv fp32 or bf16 (qkv_is_bf16); L <= 192; keep
 * flags drawn ahead (uniter_attn_keep_bits_gen) when p_drop > 0; ctx_bf16 [rows][H], dqkv_bf16 [rows][3H]; dqkv fp32 optional. */
int uniter_attn_b16x_fwd(const void* qkv, int qkv_is_bf16, const float* attn_mask, const int32_t* cu_seqlens, float* ctx,
                         void* ctx_bf16, float* lse, const void* keep_bits, int B, int L, int nh, float p_drop, void* stream);
int uniter_attn_b16x_bwd(const void* qkv, int qkv_is_bf16, const float* attn_mask, const int32_t* cu_seqlens,
                         const float* ctx, const float* lse, const float* dctx, float* dqkv, void* dqkv_bf16, float* bias_part,
                         const void* keep_bits, float* delta, int B, int L, int nh, float p_drop, void* stream);
/* Optimal-transport distance of the ITM pretraining task (IPOT): optimal_transport_dist of model/ot.py:69-85 (cosine cost matrix
 * :11-21, `iteration` proximal steps of ipot :36-66 with k = 1 -- the reference's ipot fails for k > 1 --, trace(C T) :84), called
 * from model/pretrain.py:168-193 on the text rows txt_emb [B, M, D] and region rows img_emb [B, N, D] of the encoder output;
 * txt_pad [B, M] / img_pad [B, N]: 1 = padding.  fp32 throughout like the reference.  One workgroup per sample with the cost
 * matrix and the plan in LDS: M * N <= 12288 and (3 M N + 35 (M + N)) * 4 bytes <= 160 KB (128 x 64 fits, 128 x 96 does not:
 * UNITER_E_SHAPE).  T [B, N, M] (optional in the forward call) is the transport plan the backward
 * call needs: the gradient reaches the embeddings through the cost matrix only (the reference detaches T). */
int uniter_ot_dist_fwd(const float* txt_emb, const float* img_emb, const unsigned char* txt_pad, const unsigned char* img_pad,
                       float* dist, float* T, int B, int M, int N, int D, float beta, int iteration, void* stream);
int uniter_ot_dist_bwd(const float* txt_emb, const float* img_emb, const float* T, const float* grad_dist, float* d_txt,
                       float* d_img, int B, int M, int N, int D, void* stream);
/* The same two operations on the bf16 matrix pipe (precision mode 2): Q, K, V rounded to bf16 while
 * staged, fp32 scores / softmax / dropout / LSE, probabilities and score gradients rounded to bf16 as
 * MFMA operands.  Same arguments and Philox element indices as the _ex forms; L <= 192;
 * qkv is fp32, or -- qkv_is_bf16 = 1 -- the bf16 output of the QKV GEMM read as stored (what the model passes).
 * ws: uniter_attn_bf16_bwd_ws_bytes (bf16 Pd / dS scratch, half of the fp32 kernels').  The backward pass
 * writes whichever of dqkv (fp32) / dqkv_bf16 is not NULL (at least one). */
size_t uniter_attn_bf16_bwd_ws_bytes(int B, int L, int nh);
int uniter_attn_bf16_fwd(const void* qkv, int qkv_is_bf16, const float* attn_mask, const int32_t* cu_seqlens,
                         float* ctx, void* ctx_bf16, float* lse, void* keep_bits, int B, int L, int nh, float p_drop,
                         uint64_t seed, uint32_t offset, uint32_t site, void* stream);
int uniter_attn_bf16_fwd_pre(const void* qkv, int qkv_is_bf16, const float* attn_mask, const int32_t* cu_seqlens,
                             float* ctx, void* ctx_bf16, float* lse, void* keep_bits, int keep_bits_ready, int B, int L,
                             int nh, float p_drop, uint64_t seed, uint32_t offset, uint32_t site, void* stream);
int uniter_attn_bf16_bwd(const void* qkv, int qkv_is_bf16, const float* attn_mask, const int32_t* cu_seqlens,
                         const float* ctx, const float* lse, const float* dctx, float* dqkv, void* dqkv_bf16,
                         float* bias_part, const void* keep_bits, float* delta, int B, int L, int nh, float p_drop,
                         uint64_t seed, uint32_t offset, uint32_t site, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Embeddings (replace UniterTextEmbeddings.forward model/model.py:232-245,
 * UniterImageEmbeddings.forward :261-272 after the img_linear GEMM, and the
 * cat+gather of :330-333).
 * ------------------------------------------------------------------------- */
/* out rows [b, t] of `cat` (row stride: (T+R)*H per sample, text at offset 0):
 *   LN(word[ids] + pos[position_ids] + type[type_ids or 0]) then dropout      */
int uniter_txt_embed_fwd(const int64_t* input_ids, const int64_t* position_ids,
                         const int64_t* type_ids /*NULL=0*/, const float* word, const float* pos,
                         const float* type, const float* gamma, const float* beta,
                         float* cat, int B, int T, int S /*rows per sample in cat*/, int H,
                         int vocab, int max_pos, int type_vocab, int pos_bcast /*position_ids is [1,T]*/,
                         float p_drop, uint64_t seed, uint32_t offset, void* stream);
/* image rows: cat[b, T0 + r] = dropout(LN_f(LN_i(imgfc[b,r]) + LN_p(pos7[b,r] @ Wp^T + bp) + type[tid]))
 * imgfc = img_feat @ W_img^T + b_img is produced by uniter_gemm_f32 beforehand.
 * stats (optional, training): [B*R, 6] = mean/rstd of LN_i, LN_p, LN_f      */
int uniter_img_embed_fwd(const float* imgfc, const float* pos7, const int64_t* img_type_ids /*NULL=1*/,
                         const float* Wp, const float* bp, const float* type,
                         const float* g_i, const float* b_i, const float* g_p, const float* b_p,
                         const float* g_f, const float* b_f, float* cat, float* stats,
                         int B, int R, int T0, int S, int H, int type_vocab,
                         float p_drop, uint64_t seed, uint32_t offset, void* stream);
/* out[b, j] = cat[b, gather_index[b, j]]   (gather_index NULL = identity copy) */
int uniter_gather_rows(const float* cat, const int64_t* gather_index, float* out,
                       int B, int S, int Lout, int H, void* stream);
/* uniter_gather_rows with the gathered rows' operand copy for the first encoder product written by the same launch (the
 * concatenation + gather of model/model.py:327-334 and the copy the precision modes add): mode 1 = three bf16 pieces per row,
 * out_b16[row][3][H] (= uniter_split3 with row stride 3 H, piece stride H), mode 2 = one bf16 copy out_b16[row][H]
 * (= uniter_cast_bf16).  H % 4 == 0, H <= 1024 (UNITER_E_SHAPE beyond: run the two launches). */
int uniter_gather_rows_ex(const float* cat, const int64_t* gather_index, float* out, void* out_b16, int mode,
                          int B, int S, int Lout, int H, void* stream);
/* dcat[b, s] = sum_{j : gather_index[b,j] == s} dout[b, j]   (deterministic) */
int uniter_gather_rows_bwd(const float* dout, const int64_t* gather_index, float* dcat,
                           int B, int S, int Lout, int H, void* stream);
/* feat_out = feat + mask_emb[img_masks] with mask_emb row 0 treated as zero (model/model.py:262-265) */
/* out[m][:] = bias (m < M; N % 4 == 0): the starting value of x @ W^T + b where the product is accumulated on top by the
 * stream-K form of uniter_gemm_f32 (beta = 1), which has no bias epilogue -- the image projection of model/model.py:267,
 * too few tiles (108) for the tile-per-workgroup form to fill the chip. */
int uniter_bias_rows(const float* bias, float* out, int M, int N, void* stream);
int uniter_img_mask_add(const float* feat, const int64_t* img_masks, const float* mask_emb,
                        float* feat_out, int rows, int D, void* stream);
/* backward of the text embedding: accumulates into dword/dpos/dtype/dgamma/dbeta */
int uniter_txt_embed_bwd(const float* dcat, const int64_t* input_ids, const int64_t* position_ids,
                         const int64_t* type_ids, const float* word, const float* pos,
                         const float* type, const float* gamma,
                         float* dword, float* dpos, float* dtype, float* dgamma, float* dbeta,
                         int B, int T, int S, int H, int vocab, int max_pos, int type_vocab, int pos_bcast,
                         float p_drop, uint64_t seed, uint32_t offset,
                         void* ws, size_t ws_bytes, void* stream);
/* backward of the image embedding epilogue: writes d_imgfc, d_posfc [B*R,H]; accumulates the
 * six LN affine grads and dtype; dWp/dbp from d_posfc */
int uniter_img_embed_bwd(const float* dcat, const float* imgfc, const float* pos7,
                         const int64_t* img_type_ids, const float* Wp, const float* bp,
                         const float* type, const float* g_i, const float* b_i,
                         const float* g_p, const float* b_p, const float* g_f,
                         const float* stats, float* d_imgfc, float* d_posfc,
                         float* dWp, float* dbp, float* dtype,
                         float* dg_i, float* db_i, float* dg_p, float* db_p, float* dg_f, float* db_f,
                         int B, int R, int T0, int S, int H, int type_vocab,
                         float p_drop, uint64_t seed, uint32_t offset,
                         void* ws, size_t ws_bytes, void* stream);
size_t uniter_embed_bwd_ws_bytes(int rows, int H);

/* ------------------------------------------------------------------------- *
 * Pooler + classification head (replaces BertPooler.forward model/layer.py:179-185
 * and MemeUniter.linear model/meme_uniter.py:19-20) and the loss
 * (nn.BCEWithLogitsLoss(pos_weight), train_template.py:65,98-99).
 * ------------------------------------------------------------------------- */
/* pooled[b,:] = tanh(hidden[b,0,:] @ Wp^T + bp) ; hidden row stride per sample = L*H */
int uniter_pooler_fwd(const float* hidden, const float* Wp, const float* bp, float* pooled,
                      int B, int L, int H, void* stream);
/* dpre = dpooled*(1-pooled^2); dWp += dpre^T h0; dbp += colsum(dpre);
 * dhidden[b,0,:] (+)= dpre @ Wp   (other rows untouched) */
int uniter_pooler_bwd(const float* dpooled, const float* pooled, const float* hidden,
                      const float* Wp, float* dWp, float* dbp, float* dhidden,
                      int B, int L, int H, int beta_dhidden, void* stream);
/* logits[b,c] = pooled[b,:] . Wc[c,:] + bc[c]   (any small n_classes) */
int uniter_linear_small_fwd(const float* x, const float* W, const float* b, float* y,
                            int B, int H, int C, void* stream);
int uniter_linear_small_bwd(const float* dy, const float* x, const float* W,
                            float* dx, float* dW, float* db, int B, int H, int C, void* stream);
/* Pooler + classifier as ONE launch each way (model/layer.py:179-185 followed by model/meme_uniter.py:19-21): results equal
 * uniter_pooler_fwd + uniter_linear_small_fwd, and uniter_linear_small_bwd + uniter_pooler_bwd.  `ticket`:
 * UNITER_POOL_HEAD_TICKET_WORDS zero-initialised unsigneds in device memory, left zero by every launch (the workgroups count
 * themselves there; the last one to finish computes the logits).  Hidden sizes up to 4096.  The backward ACCUMULATES
 * into dWp, dbp, dWl, dbl; dhidden (optional) gets row 0 of every sample (assigned, or added with beta_dhidden != 0). */
#define UNITER_POOL_HEAD_TICKET_WORDS (17 * 64)
int uniter_pool_head_fwd(const float* hidden, const float* Wp, const float* bp, const float* Wl, const float* bl,
                         float* pooled, float* logits, unsigned* ticket, int B, int L, int H, int C, void* stream);
int uniter_pool_head_bwd(const float* dlogits, const float* pooled, const float* hidden, const float* Wp, const float* Wl,
                         float* dWp, float* dbp, float* dWl, float* dbl, float* dhidden,
                         int B, int L, int H, int C, int beta_dhidden, void* stream);
/* loss (scalar, mean over B), probs = sigmoid(logits), dlogits = dloss/dlogits * grad_scale */
int uniter_bce_logits(const float* logits, const int64_t* labels, float pos_weight,
                      float* loss, float* probs, float* dlogits, float grad_scale,
                      int B, void* stream);

/* ------------------------------------------------------------------------- *
 * Pretraining heads (BASELINE config 5: UniterForPretraining.forward_{mlm,mrfr,itm},
 * model/pretrain.py:107-203; and forward_mrc, :205-233).  Dense / tied-decoder products are uniter_gemm_f32, the
 * LayerNorm uniter_ln_*; these are the remaining pieces.
 * ------------------------------------------------------------------------- */
/* dst[r,:] = src[idx[r],:]           (_compute_masked_hidden, model/pretrain.py:129-133) */
int uniter_row_gather(const float* src, const int64_t* idx, float* dst, int n, int H, int nsrc, void* stream);
/* dst[idx[r],:] += src[r,:]          (its backward; idx must be unique) */
int uniter_row_scatter_add(const float* src, const int64_t* idx, float* dst, int n, int H, int ndst, void* stream);
/* F.cross_entropy(logits, targets, reduction='none'): loss[r] = lse[r] - logits[r,targets[r]] */
int uniter_cross_entropy_fwd(const float* logits, const int64_t* targets, float* loss, float* lse,
                             int n, int C, int ld, void* stream);
/* dlogits[r,c] = (softmax(logits[r])[c] - [c == targets[r]]) * dloss[r]   (dlogits may alias logits) */
int uniter_cross_entropy_bwd(const float* logits, const int64_t* targets, const float* lse,
                             const float* dloss, float* dlogits, int n, int C, int ld, void* stream);
/* MRC-kl (model/pretrain.py:222-226): F.kl_div(F.log_softmax(logits, -1), target, reduction='none'):
 * loss[r,c] = target > 0 ? target * (log target - (logits[r,c] - lse[r])) : 0;
 * backward  dlogits[r,j] = softmax[r,j] * sum_c(dloss[r,c] target[r,c]) - dloss[r,j] target[r,j] */
int uniter_kl_div_fwd(const float* logits, const float* target, float* loss, float* lse, int n, int C, int ld,
                      void* stream);
int uniter_kl_div_bwd(const float* logits, const float* target, const float* lse, const float* dloss,
                      float* dlogits, int n, int C, int ld, void* stream);
/* MRC (model/pretrain.py:227-228): out[r] = index of the first maximum of x[r, c0:C] (absolute column) */
int uniter_row_argmax(const float* x, int n, int C, int ld, int c0, int64_t* out, void* stream);
/* F.mse_loss(pred, target, reduction='none') and its backward dpred = 2 (pred - target) dloss */
int uniter_mse_fwd(const float* pred, const float* target, float* loss, size_t n, void* stream);
int uniter_mse_bwd(const float* pred, const float* target, const float* dloss, float* dpred, size_t n,
                   void* stream);
/* out = dy * gelu_erf'(u)   (backward through the heads' GELU, model/layer.py:31-37) */
int uniter_dgelu_mul(const float* dy, const float* u, float* out, size_t n, void* stream);

/* ------------------------------------------------------------------------- *
 * Optimizer step over FLAT fp32 buffers (replaces average_gradients +
 * clip_grad_norm_ + torch.optim.Adam/AdamW.step + zero_grad:
 * train_template.py:89-92,103-107; utils/optim_utils.py:9-46).
 *   chunk_flags[i] for elements [64*i, 64*i+64): 0 = skip (no gradient this step),
 *   1 = no weight decay, 2 = weight decay; + 4 = zero_grads leaves this chunk's gradient alone (the next backward pass
 *   overwrites it: uniter_model_set_wgrad_overwrite) -- 28 instead of 32 bytes per parameter.
 * ------------------------------------------------------------------------- */
/* sumsq[0] = sum g^2 over flagged chunks (device scalar, double) */
int uniter_grad_sumsq(const float* grads, const uint8_t* chunk_flags, size_t n,
                      double* sumsq, void* ws, size_t ws_bytes, void* stream);
size_t uniter_grad_sumsq_ws_bytes(size_t n);
/* g' = g * grad_scale * min(1, max_norm / (sqrt(sumsq)*grad_scale + 1e-6))   (max_norm<=0: no clip)
 * adamw = 0: g' += wd*p (torch.optim.Adam L2);  1: p *= 1 - lr*wd (AdamW)
 * then m,v update, p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps); grads zeroed if zero_grads */
int uniter_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                     const uint8_t* chunk_flags, size_t n, const double* sumsq,
                     float grad_scale, float max_norm, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, int adamw, int zero_grads,
                     void* stream);
/* as uniter_adam_step; additionally mirror_bf16[i] = bf16(params[i]) for every updated element (the bf16
 * weight mirror of precision mode 2 stays in step without a separate cast pass); NULL = none */
int uniter_adam_step_mirror(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                            const uint8_t* chunk_flags, size_t n, const double* sumsq,
                            float grad_scale, float max_norm, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, int adamw, int zero_grads,
                            void* mirror_bf16, void* stream);
/* as uniter_adam_step_mirror with the grid set to max_workgroups (0 = the default, at most 2048 workgroups, which fill every
 * wave slot of the chip; an explicit value may go up to the grid in which each thread takes its two 16-byte items once): a launch that shares the GPU with the next forward (trainer.FusedAdam's per-layer blocks on the side
 * stream) leaves the wave slots the forward's kernels need -- at full occupancy it slows them three-fold. */
int uniter_adam_step_ex(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                        const uint8_t* chunk_flags, size_t n, const double* sumsq,
                        float grad_scale, float max_norm, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int step, int adamw, int zero_grads,
                        void* mirror_bf16, int max_workgroups, void* stream);
/* Data parallel with a bf16 gradient payload (dp.GradSync): the reduced gradients ARE a bf16 buffer; these two read them
 * from there (widened on the fly) instead of from an fp32 copy somebody would have to write first.  uniter_adam_step_g16 still
 * zeroes the fp32 `grads` (the buffer the next backward accumulates into) when zero_grads is set; grads_bf16 NULL = read
 * `grads` (= uniter_adam_step_ex).  Replaces nothing in the reference (its nn.DataParallel reduces fp32 gradients). */
int uniter_grad_sumsq_bf16(const void* grads_bf16, const uint8_t* chunk_flags, size_t n, double* sumsq, void* ws,
                           size_t ws_bytes, void* stream);
/* out[0] = parts[0] + .. + parts[n-1]: the clip norm (train_template.py:104, clip_grad_norm_ over ALL parameters) reduced
 * slice by slice -- one uniter_grad_sumsq per data-parallel collective as it lands, each into its own slot -- and joined here */
int uniter_sumsq_combine(const double* parts, int n, double* out, void* stream);
/* the same norm taken DURING the backward pass: as soon as a slice of the gradient buffer is final (a layer's weight
 * gradients, on the stream that wrote them) `nblocks` workgroups leave their unreduced partial sums of it in
 * parts[0 .. nblocks) (chunk_flags NULL = every element counts); one uniter_sumsq_combine over all slices' partials
 * ends it.  Behind the backward pass only the embeddings' slice is left instead of a pass over every gradient. */
int uniter_grad_sumsq_part(const float* grads, const uint8_t* chunk_flags, size_t n, double* parts, int nblocks,
                           void* stream);
int uniter_adam_step_g16(float* params, float* grads, const void* grads_bf16, float* exp_avg, float* exp_avg_sq,
                         const uint8_t* chunk_flags, size_t n, const double* sumsq, float grad_scale,
                         float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                         int step, int adamw, int zero_grads, void* mirror_bf16, int max_workgroups, void* stream);
/* uniter_adam_step_g16 whose mirror holds the three bf16 pieces of every updated parameter (x = x1 + x2 + x3 exactly), piece p
 * at mirror + p * mirror_piece_stride elements: the piece-major x3 weights of the fp32x3 mode (uniter_gemm_x3_cfg), written
 * by the update itself.  mirror_piece_stride = 0: one bf16 copy, as uniter_adam_step_g16.
 * Replaces the optimizer step of train_template.py:95-109 + utils/optim_utils.py:9-46. */
int uniter_adam_step_x3(float* params, float* grads, const void* grads_bf16, float* exp_avg, float* exp_avg_sq,
                        const uint8_t* chunk_flags, size_t n, const double* sumsq, float grad_scale, float max_norm,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int step, int adamw,
                        int zero_grads, void* mirror, size_t mirror_piece_stride, int max_workgroups, void* stream);
/* The update of ONE table split by rows (round 6; utils/optim_utils.py:33-40 + train_template.py:104-107 for
 * model/model.py:220-221's word-embedding table).  torch.optim.Adam with weight_decay updates every row of the table (g += wd p),
 * 0.62 of the step's 3.18 GB, and the next forward's text branch waits for it; but a row WITHOUT a gradient is updated from g = 0,
 * which no clip coefficient changes -- its update depends on nothing this step's backward pass computes.  So: rows_touched = 0 updates
 * the rows whose row_mask byte is 0, g taken as zero (never read, never cleared; sumsq / max_norm ignored), at any point after the
 * previous step's update -- e.g. beside the backward pass; rows_touched = 1 updates the masked rows with their gradients, clipped,
 * behind the backward pass.  Per-element arithmetic unchanged: the parameters are bit-identical to one launch over the table.
 * params / grads / exp_avg / exp_avg_sq / chunk_flags point at the table; n = rows x row_len, row_len % 64 == 0. */
/* Round 6: the PAIRED-ROW layout of the x3 weight mirror.  The forward products read a weight [N][K] k-contiguous, 32 elements (64 bytes)
 * of a row per k-tile: half a cache line per request, and the loaders' cost is per request.  In the paired layout
 * rows 2 q and 2 q + 1 are interleaved in 64-byte units -- element (n, k) of the tensor at (n >> 1) * 2 K + (k >> 5) * 64 + (n & 1) * 32 +
 * (k & 31) in every piece -- so a k-tile of a row pair is ONE 128-byte line (forward products -3.5 .. -4.7 %); the input-gradient
 * products read the same weight k-major and find whole lines too (two rows of a pair per request).  Who knows the layout:
 *  - the writers: uniter_adam_step_x3p -- uniter_adam_step_x3 walking the buffer in the MIRROR's order: pair_src holds two int32 per
 *    64-element chunk of the mirror, the flat-buffer offsets of the parameters behind its two 32-element units (one unit of row 2 q, the
 *    same unit of row 2 q + 1; first < 0 = the chunk is its own source), so the fp32 streams and the mirror are both read / written in
 *    whole 128-byte lines -- and uniter_mirror_refresh_x3 (a refresh from the fp32 parameters through the inverse table: pair_dst[c] >= 0 =
 *    mirror offset of parameter chunk c's first unit, the second at + 64);
 *  - the readers: uniter_gemm_x3_cfg with cfg | 64 (B = the weight, forward or input-gradient layout) and the model once
 *    uniter_model_set_weight_pairing(m, 1) is set (the caller's mirror must then hold the encoder layers' weights -- query|key|value
 *    as ONE [3 H][H] tensor, attention.output.dense, intermediate.dense, output.dense -- in this layout). */
int uniter_adam_step_x3p(float* params, float* grads, const void* grads_bf16, float* exp_avg, float* exp_avg_sq,
                         const uint8_t* chunk_flags, size_t n, const double* sumsq, float grad_scale, float max_norm,
                         float lr, float beta1, float beta2, float eps, float weight_decay, int step, int adamw,
                         int zero_grads, void* mirror, size_t mirror_piece_stride, const int* pair_src, size_t first_element,
                         int max_workgroups, void* stream);
int uniter_mirror_refresh_x3(const float* params_base, size_t first, size_t n, void* mirror, size_t piece_stride,
                             const int* pair_dst, void* stream);
int uniter_adam_step_rows(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* chunk_flags, size_t n,
                          const double* sumsq, float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, int adamw, int zero_grads, const uint8_t* row_mask, int row_len,
                          int rows_touched, int max_workgroups, void* stream);

/* ------------------------------------------------------------------------- *
 * Whole-model schedule: the library owns the kernel sequence of
 * UniterModel.forward (model/model.py:336-367) + pooler/head and of its
 * backward; Python makes a handful of calls per step.
 * ------------------------------------------------------------------------- */
typedef struct {
  int32_t hidden_size, num_hidden_layers, num_attention_heads, intermediate_size;
  int32_t vocab_size, max_position_embeddings, type_vocab_size, img_dim;
  float   hidden_dropout_prob, attention_probs_dropout_prob;
} uniter_config_t;

/* canonical parameter order = reference UniterModel.state_dict() order
 * (see uniter_param_name); query/key/value weights (and biases) of a layer
 * must be contiguous in memory (they are one [3H,H] operand).           */
int         uniter_num_params(const uniter_config_t* cfg);
const char* uniter_param_name(const uniter_config_t* cfg, int i);   /* without prefix */
int         uniter_param_shape(const uniter_config_t* cfg, int i, int64_t* rows, int64_t* cols);

typedef struct uniter_model uniter_model_t;
int  uniter_model_create(const uniter_config_t* cfg, float* const* params, float* const* grads,
                         int n_params, uniter_model_t** out);
void uniter_model_destroy(uniter_model_t* m);

typedef struct {
  const int64_t* input_ids;      /* [B,T] or NULL (image only)  */
  const int64_t* position_ids;   /* [B,T] or [1,T] (pos_bcast=1) */
  const int64_t* txt_type_ids;   /* [B,T] or NULL */
  const float*   img_feat;       /* [B,R,img_dim] or NULL (text only) */
  const float*   img_pos_feat;   /* [B,R,7] */
  const int64_t* img_type_ids;   /* [B,R] or NULL */
  const int64_t* img_masks;      /* [B,R] or NULL */
  const float*   attention_mask; /* [B,L] */
  const int64_t* gather_index;   /* [B,L] or NULL */
  int32_t B, T, R, L;            /* L = output sequence length */
  int32_t pos_bcast;
  /* Packed mode (all NULL / 0 otherwise; needs attention_mask rows of the form 1..1 0..0 and
   * L <= uniter_attn_varlen_max_len()): only the Mp = cu_seqlens[B] valid positions are computed --
   * every GEMM / LayerNorm runs on Mp rows instead of B*L.  hidden_out keeps its padded [B,L,H]
   * layout; padded positions are written as zeros (the reference computes values there that nothing
   * downstream reads). */
  const int32_t* cu_seqlens;     /* [B+1] device: prefix sums of the per-sample valid lengths */
  const int64_t* pack_src;       /* [Mp] device: row of the [B*S] text|image embedding block feeding packed row r
                                    (b*S + gather_index[b,l], or b*S + l without gather_index) */
  const int64_t* pack_dst;       /* [Mp] device: row b*L + l of the padded output that packed row r fills */
  int32_t Mp;
} uniter_batch_t;

/* 0 (default): exact fp32 MFMA GEMMs.  1: bf16 MFMA for the dense GEMMs of the schedule (operands
 * rounded to bf16 in flight; storage, LayerNorm, softmax, attention, loss and optimizer stay fp32).
 * 2: as 1 with bf16-RESIDENT GEMM operands: the encoder weights are read from a bf16 mirror of the
 * flat parameter buffer (uniter_model_set_weight_mirror; the caller refreshes it after every update
 * with uniter_cast_bf16) and activations are handed from kernel to kernel as bf16 copies; the FFN
 * activation and its gradient exist only in bf16.
 * 3 (`fp32x3`): fp32 results from the bf16 matrix pipe -- every dense product of the encoder layers (forward, input
 * gradient, weight gradient: model/layer.py:76-78,112,140,153) runs on x3 operands (three bf16 pieces per value, six
 * MFMA products, fp32 accumulate: uniter_gemm_x3_cfg), no less accurate than the fp32 MFMA kernels of mode 0; attention,
 * LayerNorm, embeddings, loss and optimizer are those of mode 0.  The weight mirror then holds the three pieces of every
 * parameter, piece p at mirror + p * numel (3 * numel bf16 elements; uniter_adam_step_x3 / uniter_split3 write it). */
int  uniter_model_set_weight_mirror(uniter_model_t* m, const float* flat_base, const void* mirror_bf16, size_t numel);
int  uniter_model_set_weight_pairing(uniter_model_t* m, int on);
int  uniter_model_set_precision(uniter_model_t* m, int precision);
/* Number of uniter_model_forward calls on this handle so far.  The handle keeps ONE plan (activations of the latest
 * forward): a caller that wants to backpropagate records the value after its forward and must see the same value at
 * backward time -- the reference's autograd allows several forwards per backward (model/model.py:336-367), this
 * library does not, and the host mirror turns a mismatch into an error instead of wrong gradients. */
uint64_t uniter_model_generation(const uniter_model_t* m);
/* Overlap of the optimizer step with the next forward: `events` = hipEvent_t[num_hidden_layers + 1]
 * (embedding block, layer 0, layer 1, ..), each fired once that block's parameters (and zeroed
 * gradients) are final.  The next uniter_model_forward makes its stream wait for events[0] before
 * the embeddings and events[1 + l] before layer l, then forgets them.  n = 0 clears.
 * n = num_hidden_layers + 2: events[0] covers the embeddings' parameters WITHOUT the word-embedding table and the last
 * event the table alone (22 of the embeddings' 24 M parameters): the forward then runs its image branch (region
 * projection, the three LayerNorms of model/model.py:261-272), which reads no word embedding, behind events[0] and
 * only the text embeddings wait for the table. */
int  uniter_model_set_ready_events(uniter_model_t* m, void* const* events, int n);
size_t uniter_model_ws_bytes(const uniter_model_t* m, int B, int T, int R, int L, int train);
/* hidden_out: [B,L,H] last layer (all_layers = 0) or [nl,B,L,H] (all_layers = 1).
 * train != 0 applies dropout and keeps activations in `ws` for uniter_model_backward. */
int uniter_model_forward(uniter_model_t* m, const uniter_batch_t* batch, float* hidden_out,
                         int all_layers, int train, uint64_t seed, uint32_t offset,
                         void* ws, size_t ws_bytes, void* stream);
/* Hidden-dropout keep flags drawn ahead (round 5; the dropout of model/layer.py:113,154 inside the fused dropout + residual +
 * LayerNorm passes): uniter_hidden_keep_bits_gen draws the flags of `nsites` sites of `elements` values each -- site of even
 * s = 2 k: site_a0 + k * site_step, of odd s: site_b0 + k * site_step (the two sites of encoder layer k) -- in one launch, as
 * one nibble per 4-element group (uniter_hidden_keep_bits_bytes per site); uniter_ln_set_next_keep_bits hands ONE site's flags to
 * the next LayerNorm row pass (forward or backward) launched by this host thread, which then reads them instead of running
 * Philox (same flags bit for bit; NULL / not called: the pass draws them itself).  uniter_model_forward does both by itself with
 * UNITER_HIDDEN_PREGEN=1 and an auxiliary stream (off by default: measured no gain inside the step). */
size_t uniter_hidden_keep_bits_bytes(size_t elements);
int uniter_hidden_keep_bits_gen(void* bits, size_t site_stride_bytes, int nsites, uint32_t site_a0, uint32_t site_b0,
                                uint32_t site_step, size_t elements, float p_drop, uint64_t seed, uint32_t offset, void* stream);
int uniter_ln_set_next_keep_bits(const void* site_bits);
/* Precision 3: the clip norm's share of the encoder layers (train_template.py:104: clip_grad_norm_ over ALL parameters) taken
 * by the layers' own weight-gradient launches.  With `parts` set, uniter_model_backward_layer(l) leaves the sum of squares of
 * EVERY gradient of layer l's bucket (4 weights, 12 vectors) as uniter_model_norm_partials_per_layer() unreduced partial sums
 * at parts + l * stride_doubles (fixed slots, bit-reproducible); uniter_sumsq_combine joins them with the other buckets'.
 * parts = NULL switches it off.  Precisions 2 (bf16) and 3 (fp32x3).  _per_layer (valid after a training-mode forward: it answers
 * for that forward's plan) returns 0 when the current precision / switches / plan have no such riders: the caller
 * then reduces the bucket itself (uniter_grad_sumsq_part). */
int uniter_model_set_norm_partials(uniter_model_t* m, double* parts, size_t stride_doubles);
/* A stream for launches of the forward pass that depend on nothing the step computes -- the dropout keep flags of the attention
 * probabilities (a function of seed and offset alone: 60 us of Philox rounds for twelve layers) -- so that they run BESIDE the head
 * of the forward pass instead of in front of it; the first attention kernel waits for them.  NULL (default): on `stream`. */
int uniter_model_set_aux_stream(uniter_model_t* m, void* aux_stream);
/* Data parallel (replaces nothing in the reference: its nn.DataParallel, train_template.py:58-59, has no overlap to protect): the
 * persistent matrix kernels of this model's forward / backward calls -- one 144-KB workgroup per CU in the fp32x3 mode, the grouped
 * bf16 weight-gradient launch -- leave `cus` CUs free for the kernels of the gradient exchange (RCCL) that runs beside the backward
 * pass.  Without it a collective's workgroups take CUs as persistent workgroups exit, and the launch that counted on all 256 runs its
 * last workgroups -- and their whole static share of the tiles -- behind them.  0 (default) = every CU. */
int uniter_model_set_cu_reserve(uniter_model_t* m, int cus);
int uniter_model_norm_partials_per_layer(const uniter_model_t* m);
/* Gradient accumulation semantics without the clearing pass (optimizer.zero_grad, train_template.py:107): after an optimizer
 * step that did NOT clear the encoder layers' weight gradients (uniter_adam_step*: chunk flag + 4), announce it here and the
 * NEXT backward pass writes query|key|value / attention-output / intermediate / output weight gradients with `=` instead of
 * `+=` (whole-K tiles: a plain store; kernels that add partial tiles clear their output first), then the flag resets:
 * further backward passes (gradient_accumulation > 1) accumulate as before.  Everything else (biases, LayerNorm, embeddings,
 * pooler) always accumulates into buffers the optimizer step cleared. */
int uniter_model_set_wgrad_overwrite(uniter_model_t* m, int on);
/* d_hidden: same layout as hidden_out.  Accumulates into the bound gradient buffers.
 * Stages let the caller interleave gradient all-reduce with backward:
 *   uniter_model_backward_begin, then layer nl-1 .. 0, then uniter_model_backward_embed. */
int uniter_model_backward_begin(uniter_model_t* m, const uniter_batch_t* batch, const float* d_hidden,
                                int all_layers, uint64_t seed, uint32_t offset,
                                void* ws, size_t ws_bytes, void* stream, void* side_stream);
int uniter_model_backward_layer(uniter_model_t* m, int layer);
int uniter_model_backward_embed(uniter_model_t* m);
/* convenience: begin + all layers + embed */
int uniter_model_backward(uniter_model_t* m, const uniter_batch_t* batch, const float* d_hidden,
                          int all_layers, uint64_t seed, uint32_t offset,
                          void* ws, size_t ws_bytes, void* stream, void* side_stream);

/* per-kernel-kind HIP-event timing of the model schedule (bench.py roofline) */
enum { UNITER_K_NONE = 0, UNITER_K_GEMM_FFN_UP_FWD = 1, UNITER_K_GEMM_FFN_DOWN_FWD = 2,
       UNITER_K_GEMM_QKV_FWD = 3, UNITER_K_GEMM_ATTN_OUT_FWD = 4, UNITER_K_ATTN_FWD = 5,
       UNITER_K_GEMM_DGRAD = 6, UNITER_K_GEMM_WGRAD = 7, UNITER_K_ATTN_BWD = 8,
       UNITER_K_LN = 9 /* LayerNorm forward */, UNITER_K_LN_BWD = 10, UNITER_K_COUNT = 11 };
int uniter_prof_enable(uniter_model_t* m, int kind);             /* 0 disables */
int uniter_prof_collect(uniter_model_t* m, int* n_launches, double* total_ms);
/* kind = -1 in uniter_prof_enable times EVERY kind; this returns launches and summed milliseconds per UNITER_K_* kind
 * (arrays of n_kinds >= UNITER_K_COUNT entries). */
int uniter_prof_collect_kinds(uniter_model_t* m, int* n_launches, double* total_ms, int n_kinds);
/* In-kernel launch stamps: with stamps on, every workgroup of every GEMM of the model schedule stores its start and
 * end time (100 MHz real-time clock) into its own words of a device buffer -- per-launch device durations (first start
 * to last end, reduced on the host) with nothing added to the streams, so they can stay on inside a timed region
 * (event pairs cost ~7 us each and serialise the two backward streams).  uniter_prof_collect_stamps synchronises the device and returns launches / summed milliseconds
 * per GEMM kind (UNITER_K_GEMM_*).  Not a library-call path of the reference: measurement only. */
int uniter_prof_enable_stamps(uniter_model_t* m, int on, void* stream);
int uniter_prof_collect_stamps(uniter_model_t* m, int* n_launches, double* total_ms, int n_kinds);  /* synchronises */
/* milliseconds during which at least one stamped launch of a kind in kind_mask (bit k = UNITER_K_* k) was running: the
 * union of the launches' intervals, e.g. what the input- and weight-gradient GEMMs, which share the chip on two streams,
 * took together.  Call before uniter_prof_enable_stamps(m, 0, ..) discards the stamps. */
int uniter_prof_stamps_union(uniter_model_t* m, unsigned kind_mask, double* union_ms);
/* Every stamped launch in launch order: its UNITER_K_* kind and [first workgroup's start, last workgroup's end] in
 * microseconds since the first stamped launch's start -- where on the step's time axis each GEMM ran (synchronises). */
int uniter_prof_stamp_spans(uniter_model_t* m, int* kinds, double* start_us, double* end_us, int cap, int* n);

#ifdef __cplusplus
}
#endif
#endif /* UNITER_HIP_H */
