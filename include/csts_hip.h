/* csts_hip.h -- C ABI of libcsts_hip.so, the MI355X (gfx950) kernel library behind the CSTS hot path.
 *
 * The reference (BolinLai/CSTS) owns no native code: its "FFI" for this path is the set of torch.nn ops
 * its model calls.  Each entry point below replaces one such op (forward and backward),
 * and cites the reference call site it stands in for (paths relative to the reference root).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory unless stated otherwise.
 *   - tensors are token-major: (B, N = T*H*W, C) row-major, C fastest; "dt" arguments are CSTS_F32 / CSTS_BF16.
 *   - every function enqueues work on `stream` and returns immediately: 0 on success, negative on a
 *     rejected call (then csts_last_error() -- thread local -- describes why).  No allocation, no
 *     synchronisation, no global state: workspaces are supplied by the caller (size queries alongside),
 *     calls are safe from any host thread (e.g. the autograd worker) and are hipGraph-capturable.
 */
#ifndef CSTS_HIP_H
#define CSTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

enum { CSTS_F32 = 0, CSTS_BF16 = 1, CSTS_HALF = 1 };   /* CSTS_HALF: the 16-bit type of the build (see csts_half_kind) */
enum { CSTS_GEMM_NT = 0, CSTS_GEMM_NN = 1, CSTS_GEMM_TN = 2 };
enum { CSTS_EPI_NONE = 0, CSTS_EPI_GELU = 1, CSTS_EPI_DGELU = 2 };
enum { CSTS_MASK_NONE = 0, CSTS_MASK_SPATIAL = 1 };

/* Version of this header's struct layouts and call semantics.  Bumped whenever a struct grows or a field changes meaning
 * (2: csts_gemm_args.res_up; 3: compact k|v rows, 16-bit build, loss scaler in csts_opt_args; 4: csts_opt_args.extra_sq, factored AdamW; 5: grouped stencil weight gradients; 6: csts_copy_token_segments, csts_wgrad_grouped8_limited; 7: csts_wgrad_grouped5, csts_gemm algo 500).  csts_abi_version() returns the value the
 * LIBRARY was built with: a caller must compare it with the CSTS_ABI_VERSION it was compiled against and refuse a mismatch
 * (the Python binding does, csts_amd/lib.py::load). */
#define CSTS_ABI_VERSION 7
const char* csts_last_error(void);
int csts_abi_version(void);
int csts_half_kind(void);   /* the 16-bit type behind CSTS_BF16 in THIS library: 0 bfloat16 (libcsts_hip.so), 1 IEEE half (libcsts_hip_f16.so) */

/* ---- GEMM: nn.Linear (attention.py:88-89,130,159; common.py:20-33; custom_multimodal_builder.py:223-224),
 *      the (1,8,8) fusion Conv3d as a skinny GEMM (custom_multimodal_builder.py:227-229) and the patch-embed
 *      Conv3d after im2col (stem_helper.py:27-38); plus their data- and weight-gradients.
 *      NT: C = A[M,K] B[N,K]^T ; NN: C = A[M,K] B[K,N] ; TN: C = A[K,M]^T B[K,N].
 *      epilogue: v = acc + bias[n]; GELU: aux[m,n] = v, v = gelu_erf(v); DGELU: v *= gelu'(aux[m,n]);
 *                v *= row_scale[m / rows_per_scale] (drop-path, common.py:46-59); v += residual[m % res_row_mod, n].
 *      res_up = {Ti, Hi, Wi, To, Ho, Wo} (To > 0): `residual` is the COARSE token grid [B * Ti*Hi*Wi][ldr] of the decoder's
 *                skip and row m of C is fine token (b, to, ho, wo) of [B * To*Ho*Wo]: the epilogue adds
 *                nn.Upsample(mode='trilinear', align_corners=False)(residual)[m, n] (attention.py:463-471), computed with the
 *                arithmetic of csts_trilinear_fwd -- the upsampled skip never goes through HBM.  To, Ho, Wo powers of two,
 *                split_k == 1, res_row_mod == 0; not for the persistent LDS-DMA kernels (the library picks accordingly).
 *      split_k > 1: with a workspace, deterministic partial slabs + finishing pass (full epilogue);
 *                   without, fp32 atomic accumulation into a pre-zeroed f32 C (bias only). */
typedef struct {
  int layout;
  const void* A; int a_dt; int64_t lda;
  const void* B; int b_dt; int64_t ldb;
  void* C; int c_dt; int64_t ldc;
  int64_t M, N, K;
  const float* bias;
  int epilogue;
  void* aux; int aux_dt; int64_t ldaux;
  const void* residual; int r_dt; int64_t ldr; int64_t res_row_mod;
  const float* row_scale; int64_t rows_per_scale;
  int compute;   /* CSTS_BF16: v_mfma_f32_32x32x16_bf16 ; CSTS_F32: v_mfma_f32_32x32x2_f32 (exact fp32) */
  int split_k;
  void* workspace; size_t ws_bytes;   /* optional: makes split_k deterministic (partial slabs + finishing pass) */
  float* colsum;   /* optional, TN + bf16 v2 kernel only: colsum[m] = sum_k A[k,m] (the bias gradient of a Linear) */
  int tile_rows;   /* 0 = library heuristic; 64 / 128 / 256 force the bf16 kernel's row tile (tuning sweeps) */
  int algo;        /* 0 = library heuristic; 2 = register-staged kernel; 1000 * wg_per_cu + 300 + 10 * (tile_rows / 64) + stages = persistent LDS-DMA NT kernel; 1000 * wg_per_cu + 400 + variant = 8-wave LDS-DMA NT kernel, needs K % 64 == 0 (tuning sweeps); 500 = the streaming thin-operand NT kernel (gemm5.hip: K = 96 / 192 / 384, N % 96 == 0, M % 32 == 0; 503 / 506: 96 / 192 columns per workgroup at K = 192) */
  int res_up[6];   /* {Ti, Hi, Wi, To, Ho, Wo}; all 0 = plain residual */
} csts_gemm_args;
int csts_gemm(const csts_gemm_args* args, hipStream_t stream);
size_t csts_gemm_splitk_workspace(int64_t M, int64_t N, int64_t K, int split_k);
int csts_gemm_plan(const csts_gemm_args* args, int* v2, int* tile_rows, int* nsplit);
/* The kernel csts_gemm would launch, by NAME as rocprofv3 prints it (for attributing live timings); *nsplit = k-splits. */
int csts_gemm_kernel_name(const csts_gemm_args* a, char* buf, int buflen, int* nsplit);   /* which kernel / tile / k-split csts_gemm picks: v2 = 0 generic, 1 register-staged bf16 kernel, 30 + stages = persistent LDS-DMA NT kernel */
int csts_gemm_v2_eligible(const csts_gemm_args* args);   /* 1 when the fast bf16 kernel (and fused colsum) applies */

/* Grouped weight gradients: every dW[M,N] = dY[tokens,M]^T X[tokens,N] of a backward pass (nn.Linear weight gradients of
 * attention.py:88-89, common.py:20-21 ...) as (tile, token-chunk) work items of ONE launch; items live in DEVICE memory.
 * A (dY) is bf16 or fp32 for the whole launch (a_f32), B (X) bf16, C fp32 (the gradient itself, or one chunk's partial slab
 * to be summed by csts_reduce_rows_batched); tile_rows (64 | 128, or 256 with bf16 dY) x 128 tiles, one height per launch.
 * colsum (optional): bias-gradient partial sum over the item's tokens for the tile's rows (written by n0 == 0 tiles). */
typedef struct {
  const void* A; const void* B; float* C; float* colsum;
  int64_t lda, ldb, ldc;
  int64_t kbeg, kend;
  int M, N, m0, n0;
} csts_wgrad_item;
int csts_wgrad_grouped(const csts_wgrad_item* device_items, int nitems, int a_f32, int tile_rows, hipStream_t stream);
/* the same items as 192 x 384 tiles on 8-wave workgroups (bf16 dY; M % 192 == 0 and N % 384 == 0 layers): half the operand
 * bytes per FLOP through the L2 -> CU path */
int csts_wgrad_grouped8(const csts_wgrad_item* device_items, int nitems, hipStream_t stream);
/* the same launch on at most max_wgs workgroups (a multiple of 8 below it, >= 8), each walking several items: the kernel's 144 KB of LDS
 * take a whole CU, so a launch of max_wgs < 256 workgroups leaves the other CUs to whatever runs beside it (round 5: the grouped weight
 * gradients of the 384- / 768-channel stages beside the memory-bound end of the backward pass).  max_wgs <= 0: one workgroup per item. */
int csts_wgrad_grouped8_limited(const csts_wgrad_item* device_items, int nitems, int max_wgs, hipStream_t stream);
/* the thin layers (M % 96 == 0, N % 96 == 0, bf16 dY, token ranges in multiples of 16) as 96 x 96 tiles, one item per WAVE: eight
 * consecutive slots of one XCD's list per 8-wave workgroup (nitems = 8 x the longest list, position slot * 8 + xcd, padding A == NULL);
 * every wave streams its token range through a private LDS-DMA ring -- no workgroup barrier (round 5).  Item fields as above except
 * M: the STAGE STEP -- M > 1: the item takes the 16-token stages kbeg, kbeg + 16 M, ... < kend (the M items of a tile interleave) */
int csts_wgrad_grouped5(const csts_wgrad_item* device_items, int nitems, hipStream_t stream);

/* ---- LayerNorm: nn.LayerNorm(C, eps=1e-6) block norms (attention.py:192,214) and nn.LayerNorm(hd, eps=1e-5)
 *      on pooled q/k/v (attention.py:108,112,116).  mean/rstd are fp32 [rows]; dgamma,dbeta one [2*C] buffer. */
int csts_layernorm_fwd(const void* x, int x_dt, const float* gamma, const float* beta, void* y, int y_dt, float* mean,
                       float* rstd, int64_t rows, int C, float eps, hipStream_t stream);
/* same, normalising x + addend (dtype of x) and writing that sum to sum_out: the decoder's skip `feat + en_feat`
 * (custom_multimodal_builder.py:467-479) folded into the next block's norm1 (attention.py:192,238) */
int csts_layernorm_fwd_add(const void* x, const void* addend, void* sum_out, int x_dt, const float* gamma, const float* beta,
                           void* y, int y_dt, float* mean, float* rstd, int64_t rows, int C, float eps, hipStream_t stream);
size_t csts_layernorm_bwd_workspace(int64_t rows, int C);
/* addend (optional, dtype of dx): dx = LN'(dy) + addend -- the residual-branch gradient of x + f(LN(x)) (attention.py:242,247)
 * folded in.  dx_bf16 (optional): a bf16 copy of dx for the GEMMs that consume it next (they round to bf16 while staging
 * anyway, so results are unchanged while they read half the bytes).  dgamma == dbeta == NULL defers the cross-workgroup second stage: workspace then holds
 * csts_layernorm_bwd_workspace(rows, C) / (2*C*4) partial rows of [2*C] for csts_reduce_rows(_batched). */
int csts_layernorm_bwd(const void* dy, int dy_dt, const void* x, int x_dt, const float* gamma, const float* mean,
                       const float* rstd, void* dx, int dx_dt, const void* addend, void* dx_bf16, float* dgamma,
                       float* dbeta, void* workspace, size_t ws_bytes, int64_t rows, int C, hipStream_t stream);
/* same with two extras.  dy2 (optional, dtype and shape of dy): the LayerNorm output had a second consumer (a block that
 * changes the channel count feeds norm2's output to fc1 AND to its skip projection, attention.py:243-246) -- the kernel reads
 * dy + dy2 instead of autograd adding the two first.  copy_row_scale (optional): the bf16 copy is multiplied by a per-sample
 * scale (row r: copy_row_scale[r / rows_per_scale]; dx itself is not scaled) -- the stochastic-depth factor of the residual
 * branch that consumes this gradient next (drop_path, common.py:46-59; x + drop_path(f(x)) at attention.py:242,247), which then
 * needs no pass of its own over dx. */
int csts_layernorm_bwd_ex(const void* dy, const void* dy2, int dy_dt, const void* x, int x_dt, const float* gamma, const float* mean,
                          const float* rstd, void* dx, int dx_dt, const void* addend, void* dx_bf16,
                          const float* copy_row_scale, int64_t rows_per_scale, float* dgamma, float* dbeta,
                          void* workspace, size_t ws_bytes, int64_t rows, int C, hipStream_t stream);
/* two stacked tensors of `rows` rows each (dy, x, dx, mean, rstd contiguous: [2][rows]...) with their own gammas in one
 * launch (norm_k and norm_v of one attention); dgb0/dgb1 = [2*C] dgamma|dbeta of each, or both NULL to defer the second
 * stage: workspace = 2 x csts_layernorm_bwd_workspace(rows, C), tensor i's partial rows at offset i * that size */
int csts_layernorm_bwd2(const void* dy, int dy_dt, const void* x, int x_dt, const float* gamma0, const float* gamma1,
                        const float* mean, const float* rstd, void* dx, int dx_dt, float* dgb0, float* dgb1,
                        void* workspace, size_t ws_bytes, int64_t rows, int C, hipStream_t stream);
int csts_reduce_rows(const float* ws, float* out, int64_t nrows, int64_t ncols, float scale, hipStream_t stream);
/* many deferred second stages (LayerNorm dgamma/dbeta, stencil dweight) in ONE launch; descriptors in DEVICE memory */
typedef struct { const float* ws; float* out; int64_t nrows; int64_t ncols; float scale; int pad_; } csts_reduce_desc;
int csts_reduce_rows_batched(const csts_reduce_desc* device_descs, int n, int64_t max_ncols, hipStream_t stream);
/* same contract for wide, shallow reductions (few rows, many columns): needs ncols % 4 == 0 and 16-byte aligned ws / out */
int csts_reduce_rows_wide(const csts_reduce_desc* device_descs, int n, int64_t max_ncols, hipStream_t stream);

/* ---- depthwise 3x3x3 token stencils: attention_pool's Conv3d (attention.py:11-49,104-116) and
 *      attention_upsample's ConvTranspose3d (attention.py:251-289,344-348).  "fine" is the larger grid.
 *      weight is the reference layout (hd, 1, 3, 3, 3) fp32, shared by all heads (channel c uses c % HD). */
typedef struct {
  int B, C, HD;
  int Tf, Hf, Wf, Tc, Hc, Wc;      /* coarse = floor((fine-1)/stride)+1 */
  int st, sh, sw;
  int64_t fine_batch_stride, fine_token_stride, coarse_batch_stride, coarse_token_stride;  /* elements */
} csts_dwconv_geom;
int csts_dwconv_strided(const csts_dwconv_geom* g, const void* fine, int fine_dt, const float* weight, void* coarse,
                        int coarse_dt, hipStream_t stream);     /* pool fwd ; upsample bwd-data */
int csts_dwconv_transposed(const csts_dwconv_geom* g, const void* coarse, int coarse_dt, const float* weight, void* fine,
                           int fine_dt, hipStream_t stream);    /* upsample fwd ; pool bwd-data */
size_t csts_dwconv_wgrad_workspace(const csts_dwconv_geom* g);
/* dweight NULL: second stage deferred (workspace = rows of [HD*27] partials, rows = workspace bytes / (HD*27*4)) */
int csts_dwconv_wgrad(const csts_dwconv_geom* g, const void* fine, int fine_dt, const void* coarse, int coarse_dt,
                      float* dweight, void* workspace, size_t ws_bytes, hipStream_t stream);

/* two tensors that share one geometry (the k and v pools of one attention) in one launch each; workspace of wgrad2 =
 * 2 x csts_dwconv_wgrad_workspace(g), slot i's partial rows at offset i * that size */
int csts_dwconv_transposed2(const csts_dwconv_geom* g, const void* const coarse[2], int coarse_dt, const float* const weight[2],
                            void* const fine[2], int fine_dt, hipStream_t stream);
int csts_dwconv_wgrad2(const csts_dwconv_geom* g, const void* const fine[2], int fine_dt, const void* const coarse[2],
                       int coarse_dt, float* const dweight[2], void* workspace, size_t ws_bytes, hipStream_t stream);
/* Grouped first stage: every stencil weight gradient of a backward pass (the q / k / v pools of all blocks, attention.py:104-116,
 * and the decoder's transposed convs, :344-348) in ONE launch instead of one ~24 us latency-bound launch each.  Item i is one
 * csts_dwconv_wgrad problem whose partial rows go to its own workspace (csts_dwconv_wgrad_grouped_workspace(&geom) bytes -- the
 * grouped launch cuts a tensor into fewer, longer token chunks than the single one; rows = bytes / (HD*27*4)); the second stage
 * (the row sums) is the caller's, exactly as with dweight NULL above.  _plan runs on the
 * host: it validates the items and writes the DEVICE TABLE IMAGE (nitems * CSTS_DWCONV_WGRAD_TABLE_ENTRY bytes, items
 * re-ordered longest workgroups first) into table_host and the grid size into *nblocks; the caller copies the image to device
 * memory in stream order and passes that address to csts_dwconv_wgrad_grouped.  All items of a call share dtype dt. */
typedef struct {
  csts_dwconv_geom geom;
  const void* fine; const void* coarse;
  void* workspace;
} csts_dwconv_wgrad_item;
#define CSTS_DWCONV_WGRAD_TABLE_ENTRY 128
size_t csts_dwconv_wgrad_grouped_workspace(const csts_dwconv_geom* g);
int csts_dwconv_wgrad_grouped_plan(const csts_dwconv_wgrad_item* items, int nitems, void* table_host, size_t table_bytes, int* nblocks);
int csts_dwconv_wgrad_grouped(const void* table_dev, int nitems, int nblocks, int dt, hipStream_t stream);

/* attention_pool fused (attention.py:11-49): depthwise Conv3d k=3 p=1 stride s of the head-split q/k/v slot + LayerNorm(hd)
 * of the pooled rows, for nslots (1 or 2) tensors sharing the geometry.  conv_out = pre-LN pooled tensor (needed by
 * backward), y = normalised tensor, mean/rstd fp32 [B * N_coarse * heads]; all tensors of dtype dt. */
typedef struct {
  csts_dwconv_geom geom;        /* coarse strides describe conv_out AND y */
  int nslots;
  const void* fine[2];
  const float* weight[2]; const float* gamma[2]; const float* beta[2];
  void* conv_out[2]; void* y[2];
  float* mean[2]; float* rstd[2];
  int dt; float eps;
} csts_pool_ln_args;
int csts_pool_ln_fwd(const csts_pool_ln_args* args, hipStream_t stream);

/* ---- residual-path resampling: MaxPool3d skip (attention.py:193-195,234-236,240), nn.Upsample trilinear skip
 *      (attention.py:463-467,471) and F.interpolate of the patch feature (custom_multimodal_builder.py:479).
 *      maxpool: kernel = stride+1 where stride>1 else 1, padding = kernel/2.  trilinear: out = in*stride,
 *      align_corners=False; optional fused `addend` (same shape as y). */
typedef struct {
  int B, C;
  int Ti, Hi, Wi, To, Ho, Wo;
  int st, sh, sw;
} csts_pool_geom;
int csts_maxpool_fwd(const csts_pool_geom* g, const void* x, int dt, void* y, uint8_t* argmax, hipStream_t stream);
int csts_maxpool_bwd(const csts_pool_geom* g, const void* dy, int dt, const uint8_t* argmax, void* dx, hipStream_t stream);
int csts_trilinear_fwd(const csts_pool_geom* g, const void* x, int x_dt, const void* addend, int addend_dt, void* y,
                       int y_dt, hipStream_t stream);
int csts_trilinear_bwd(const csts_pool_geom* g, const void* dy, int dy_dt, void* dx, int dx_dt, hipStream_t stream);

/* ---- fused attention core softmax(q k^T scale [same-frame mask]) v  (attention.py:154-158,384-388;
 *      av_attention.py:137-141,334-350).  q/k/v/o are (B, N, H, hd) views given by element strides
 *      {batch, token, head}; hd contiguous.  LSE/delta are fp32 (B, H, Nq); LSE is in the log2 domain. */
typedef struct {
  const void* Q; const void* K; const void* V; void* O; float* LSE;
  const void* dO; float* delta; void* dQ; void* dK; void* dV;
  int dtype, B, H, Nq, Nk, head_dim;
  int64_t q_strides[3], k_strides[3], v_strides[3], o_strides[3];
  int64_t do_strides[3], dq_strides[3], dk_strides[3], dv_strides[3];
  float scale;
  int mask_mode, mask_T, mask_HW;
} csts_attn_args;
int csts_attn_fwd(const csts_attn_args* a, hipStream_t stream);
size_t csts_attn_bwd_workspace(const csts_attn_args* a);
int csts_attn_bwd(const csts_attn_args* a, void* workspace, size_t ws_bytes, hipStream_t stream);
int csts_attn_probs(const csts_attn_args* a, float* probs /* (B,H,Nq,Nk) */, hipStream_t stream);

/* ---- patch embedding: Conv3d k(3,7,7) s(2,4,4) p(1,3,3) + flatten/transpose (stem_helper.py:27-38) as
 *      im2col + csts_gemm; separable positional embedding (custom_multimodal_builder.py:362-370). */
typedef struct {
  int B, Cin, T, H, W;
  int kernel[3], stride[3], padding[3];
  int To, Ho, Wo;
  int Kpad;
} csts_im2col_geom;
int csts_im2col(const csts_im2col_geom* g, const void* x, int x_dt, void* col, int col_dt, hipStream_t stream);
int csts_posembed_build(const float* spatial, const float* temporal, float* pos, int T, int HW, int C, hipStream_t stream);

/* ---- layout / reductions */
int csts_transpose_batched(const void* in, int in_dt, void* out, int out_dt, int64_t batch, int R, int Cc,
                           hipStream_t stream);   /* token fold for the (1,8,8) fusion convs */
/* Many bf16 matrices (R x C, both % 8 == 0, 16-byte aligned) transposed in one launch: one 64 x 64 tile per entry of a DEVICE
 * table.  Keeps the [in][out] twins of the Linear weights' bf16 shadows, so that the data gradient dX = dY W runs as an
 * NT GEMM (k-contiguous operands) like the forward (nn.Linear backward: attention.py:130,159; common.py:27-33). */
typedef struct { const void* src; void* dst; int R, C, r0, c0; } csts_transpose_tile;
int csts_transpose_multi(const csts_transpose_tile* device_tiles, int ntiles, hipStream_t stream);
size_t csts_colsum_workspace(int64_t batch, int64_t M, int64_t N);
int csts_colsum(const void* X, int dt, const float* row_weight, float* out, int64_t batch, int64_t M, int64_t N,
                void* workspace, size_t ws_bytes, hipStream_t stream);   /* bias / pos-embed / classifier grads */
int csts_axpby(const void* a, int a_dt, const void* b, int b_dt, void* out, int out_dt, int64_t n, float alpha, float beta,
               hipStream_t stream);

int csts_add2(const void* a, int a_dt, const void* b, int b_dt, float* out, void* out_bf16, int64_t n,
              hipStream_t stream);
/* same; the bf16 copy (only the copy) is multiplied by copy_scale[i / elems_per_scale]: the per-sample stochastic-depth scale of
 * the residual branch that consumes this gradient next (see csts_layernorm_bwd_ex) */
int csts_add2_scaled_copy(const void* a, int a_dt, const void* b, int b_dt, float* out, void* out_bf16, const float* copy_scale,
                          int64_t elems_per_scale, int64_t n, hipStream_t stream);   /* out = a + b (fp32) and optionally its bf16 copy: the two gradients of an encoder
                                      * feature that the decoder skip re-uses (custom_multimodal_builder.py:467-479) */

/* ---- Rows a strided K/V pool reads.  The K and V pools are 3x3x3 depthwise convs with stride (1, s, s), padding 1
 *      (attention.py:11-49,104-116): output cell o of an axis reads fine cells o*s - 1, o*s, o*s + 1, so for s >= 3 only
 *      (3/s)^2 of the token rows of k|v are ever consumed (14 % at s = 8, 3.5 % at s = 16).  Those rows form a COMPACT grid
 *      (T, Hc, Wc): compact cell ch <-> fine cell ((ch + 1) / 3) * s + (ch + 1) % 3 - 1, Hc = 3 * ceil(H / s) - 1 (minus one
 *      more when the last cell would fall outside H), and the pool over it is the SAME conv with stride (1, 3, 3): the k|v
 *      halves of the qkv Linear are computed for these rows only (rows are independent: identical values).
 *      gather: dst[b,t,ch,cw,:] = src[b,t,h,w,:]; scatter_add (data gradient back to all rows): dst[b,t,h,w,:] += src[b,t,ch,cw,:]. */
typedef struct { int B, C, T, H, W, sh, sw, Hc, Wc; } csts_kv_rows_geom;
int csts_rows_gather(const csts_kv_rows_geom* g, const void* src, int dt, void* dst, hipStream_t stream);
int csts_rows_scatter_add(const csts_kv_rows_geom* g, const void* src, int src_dt, void* dst, int dst_dt, hipStream_t stream);

/* ---- Token-axis concatenation / split of (B, N, C) tensors in ONE launch (round 5): the fusion head joins the visual tokens with
 *      the pooled audio tokens (custom_multimodal_builder.py:421-423 torch.cat(dim=1), :448) and cuts the fused sequences apart again
 *      (:432,454-455 x[:, :N], x[:, N:]); through torch those are cat / slice / copy / fill / add nodes in both directions.
 *      Segment i copies n rows of C contiguous elements per batch element: dst[b * dst_bs + dst_off + r * C + c] =
 *      src[b * src_bs + src_off + r * C + c] (offsets and batch strides in elements; all multiples of 16 bytes).  nseg <= 4. */
typedef struct { const void* src; void* dst; int64_t src_bs, dst_bs, src_off, dst_off; int n; int pad_; } csts_token_segment;
int csts_copy_token_segments(const csts_token_segment* segs, int nseg, int B, int C, int dt, hipStream_t stream);

int csts_scale_rows(const void* x, int x_dt, const float* row_scale, int64_t rows_per_scale, void* out, int out_dt,
                    int64_t M, int64_t N, hipStream_t stream);   /* drop-path backward (common.py:46-59) */

int csts_rowdot2(const void* a, int a_dt, const void* b, int b_dt, float* out, int64_t M, int C,
                 hipStream_t stream);   /* out[m] = <a[m,:], b[m,:]>: gradient of a per-row weight */

/* ---- audio -> pixel attention map of the spatial fusion block (MVIT.SPATIAL_AUDIO_ATTN: av_attention.py:356-370
 *      min-max-rescaled probabilities of audio token t over the HW video tokens of frame t, per head; averaged over the
 *      heads into the per-token weight of custom_multimodal_builder.py:438-440).  qkv = the block's (B, T*HW + T, 3C) QKV
 *      projection; audio_attn (optional) = (B, H, T, HW) per-head maps; wmap = (B, T*HW).  The backward writes
 *      d(q of the audio tokens) and d(k) into a ZERO-INITIALISED buffer shaped like qkv. */
int csts_audio_attn_fwd(const void* qkv, int dt, float* audio_attn, float* wmap, int B, int T, int HW, int C, int H,
                        float scale, hipStream_t stream);
int csts_audio_attn_bwd(const void* qkv, int dt, const float* d_wmap, void* dqkv, int B, int T, int HW, int C, int H,
                        float scale, hipStream_t stream);

/* ---- fusion glue (custom_multimodal_builder.py:454-461 re-weighting, :493-494 token mean) */
int csts_reweight_fwd(const float* x, const float* w, float* y, int64_t BT, int HW, int C, hipStream_t stream);
int csts_reweight_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, int64_t BT, int HW, int C,
                      hipStream_t stream);
int csts_token_mean_fwd(const float* x, float* out, int64_t B, int N, int C, hipStream_t stream);
int csts_token_mean_bwd(const float* dout, float* dx, int64_t B, int N, int C, hipStream_t stream);

/* ---- head + losses: classifier Conv3d(96,1,1) (custom_multimodal_builder.py:301,481), frame_softmax
 *      (slowfast/utils/utils.py:5-12), KLDiv (slowfast/models/losses.py:59-82), sim_matrix
 *      (slowfast/utils/utils.py:15-24), EgoNCE (slowfast/models/losses.py:157-170). */
int csts_rowdot_fwd(const void* x, int x_dt, const float* w, const float* bias, float* out, int64_t M, int C,
                    hipStream_t stream);
int csts_rowdot_dx(const float* dout, const float* w, void* dx, int dx_dt, int64_t M, int C, hipStream_t stream);
int csts_softmax_fwd(const float* x, float* p, int64_t rows, int n, float temperature, hipStream_t stream);
int csts_softmax_bwd(const float* x, const float* dp, float* dx, int64_t rows, int n, float temperature, hipStream_t stream);
int csts_kldiv_fwd(const float* p, const float* q, float* loss, float* rowloss_ws, int64_t rows, int n, float scale,
                   hipStream_t stream);
int csts_kldiv_bwd(const float* p, const float* q, const float* grad_out, float* dp, int64_t rows, int n, float scale,
                   hipStream_t stream);
int csts_rownorm_fwd(const float* a, float* a_normed, float* norms, int64_t rows, int D, float eps, hipStream_t stream);
int csts_rownorm_bwd(const float* a, const float* norms, const float* d_normed, float* da, int64_t rows, int D, float eps,
                     hipStream_t stream);
int csts_egonce_fwd(const float* sim, float* loss, float* lse_row, float* lse_col, int n, float temperature,
                    hipStream_t stream);
int csts_egonce_bwd(const float* sim, const float* lse_row, const float* lse_col, const float* grad_out, float* dsim, int n,
                    float temperature, hipStream_t stream);

/* ---- optimizer step ("next" row): clip_grad_norm_(params, max_norm) (tools/train_avgaze_net.py:105-106)
 *      + torch.optim.AdamW (slowfast/models/optimizer.py:85-93; eps 1e-8, decoupled weight decay, per-tensor decay so that
 *      1-D parameters and biases get 0, optimizer.py:48-50,98-104) over the whole parameter set in three launches, and the
 *      refresh of the bf16 shadow weights the GEMMs read.  All pointers are DEVICE memory.  The parameter set is described
 *      as chunks of at most chunk_elems elements that never straddle a tensor (chunk -> tensor id + element offset).
 *      state = {step (incremented by the call), total gradient L2 norm (out), clip coefficient (out)} fp32[3];
 *      lr is read from device memory, so a captured graph follows the schedule.  grads[i] == NULL skips tensor i. */
typedef struct {
  void* p; void* m; void* v;   /* fp32 parameter, exp_avg, exp_avg_sq */
  void* w16;                   /* optional bf16 shadow of p (NULL: none) */
  int64_t n;
  float weight_decay; int pad_;
} csts_opt_tensor;
typedef struct {
  const int32_t* chunk_tensor; const int64_t* chunk_off; int nchunks; int chunk_elems;
  const csts_opt_tensor* tensors; const void* const* grads; int ntensors;
  float* partial;              /* fp32[nchunks] workspace */
  float* state;                /* fp32[4]: step count, total gradient norm, clip coefficient (x 1 / loss scale), skipped (0 / 1) */
  const float* lr;
  float beta1, beta2, eps, max_grad_norm;   /* max_grad_norm <= 0: no clipping */
  int grad_dt;                 /* dtype of EVERY gradient: CSTS_F32, or CSTS_BF16 = the library's 16-bit type (data-parallel buckets that
                                * travelled in 16 bits: read once, accumulated in fp32 here) */
  /* optional dynamic loss scaling, torch.cuda.amp.GradScaler (tools/train_avgaze_net.py:99-109,277): scaler = fp32[2] device
   * {scale, growth tracker}.  Gradients are the gradients of scale * loss: the norm / clip / update use g / scale
   * (scaler.unscale_); a non-finite norm SKIPS the step (no parameter, moment or step-count change; state[3] = 1) and multiplies
   * the scale by backoff; growth_interval consecutive good steps multiply it by growth (scaler.update()). */
  float* scaler; float growth, backoff; int growth_interval;
  /* optional: extra_sq[0 .. n_extra_sq) device floats added to the sum of squared gradients before the norm is taken -- the
   * squared norms of gradients that are never materialised (csts_adamw_factored); they are in the same (loss-scaled) units as
   * the gradients */
  const float* extra_sq; int n_extra_sq;
} csts_opt_args;
int csts_adamw_step(const csts_opt_args* args, hipStream_t stream);

/* Factored AdamW: the weight gradient of a fusion conv (custom_multimodal_builder.py:227-229) is dW[N][K] = dY[T][N]^T A[T][K]
 * with T = B * T' token rows (32 at b = 4, 16 frames) for N x K = 768 x 49152 -- a rank-T update.  Instead of writing dW to
 * memory (151 MB, read twice by the clip norm and the update) the optimizer forms g = dY^T A ON THE FLY, in fp32, inside the
 * update of p / m / v (same AdamW arithmetic, same clip coefficient and loss-scale handling as csts_adamw_step, whose `state`
 * it reads -- call it after csts_adamw_step of the same iteration).  csts_factored_sqnorm gives the squared Frobenius norm of
 * every item's never-materialised gradient for that step's clip norm (csts_opt_args.extra_sq) as
 * sum_{t,t'} (dY dY^T)[t,t'] (A A^T)[t,t'], from T x T Gram matrices.  dy fp32 [T][N]; a [T][K] in a_dt; T <= 64, K % 256 == 0,
 * N % 16 == 0.  workspace: csts_factored_sqnorm_workspace(items) bytes.  Round 5: csts_adamw_factored takes T <= 256 when every
 * item's `a` is in the library's 16-bit type (the MFMA form of the update has no T x K tile in LDS: it loops over T) -- the
 * data-parallel chain's W * B * T' gathered rows at eight ranks; for T > 64 the caller forms the two Gram matrices with csts_gemm
 * (A A^T and dY dY^T, split-K) and their inner product with csts_rowdot2 + csts_reduce_rows (csts_amd/optim.py). */
typedef struct {
  float* p; float* m; float* v; void* w16;
  const float* dy; const void* a; int a_dt;
  int N, K, T; float weight_decay; int pad_;
} csts_opt_factored;
size_t csts_factored_sqnorm_workspace(const csts_opt_factored* items, int nitems);
int csts_factored_sqnorm(const csts_opt_factored* items, int nitems, float* out_sq, void* workspace, size_t ws_bytes, hipStream_t stream);
int csts_adamw_factored(const csts_opt_factored* items, int nitems, const float* state, const float* lr, float beta1, float beta2,
                        float eps, hipStream_t stream);

/* ---- evaluation metric ("next" row): metrics.adaptive_f1 (slowfast/utils/metrics.py:9-74) with the
 *      per-frame min-max rescale of its callers (tools/test_avgaze_net.py:66-68, tools/train_avgaze_net.py:125-127) folded
 *      in (rescale != 0).  preds, labels_hm: fp32 [nframes][hw]; tracked[f] != 0 marks the fixation frames that count
 *      (metrics.py:64-65); thresholds fp32 [nthr] (nthr <= 64).  out = {f1, recall, precision, index of the best threshold}. */
size_t csts_adaptive_f1_workspace(int64_t nframes, int nthr);
int csts_adaptive_f1(const float* preds, const float* labels_hm, const uint8_t* tracked, const float* thresholds, int nthr,
                     int64_t nframes, int hw, int rescale, float* out, void* workspace, size_t ws_bytes, hipStream_t stream);

/* ---- input pipeline on the device ("next" row): what the reference does on the CPU just before the
 *      model is called.  frames_normalize: uint8 (B, T*H*W, C) -> fp32 (B, C, T*H*W), (x/255 - mean)/std
 *      (slowfast/datasets/utils.py:290-307, ego4d_avgaze_forecast.py:294-296).  stft_logpower: fp32 waveform (B, n) ->
 *      log(|STFT|^2 + eps) (B, n_fft/2+1, csts_stft_frames(n, n_fft, hop)), librosa.stft semantics (center, zero padding,
 *      periodic Hann of win samples centred in n_fft; data/preprocess.py:276-290).  audio_windows: (B, 1, T, nbins, width)
 *      windows of the spectrogram centred on centers[b][t] (ego4d_avgaze_forecast.py:214-219).  gaze_heatmaps: labels
 *      (nframes, label_stride) with x, y in [0,1] -> (nframes, H, W) OpenCV-Gaussian maps normalised to sum 1
 *      (ego4d_avgaze_forecast.py:318-326,404-422). */
int csts_frames_normalize(const uint8_t* frames_thwc, float* out_cthw, int B, int64_t thw, int C, const float mean[3],
                          const float std[3], hipStream_t stream);
int csts_stft_frames(int n, int n_fft, int hop);
int csts_stft_logpower(const float* wav, float* spec, int B, int n, int n_fft, int hop, int win, float eps, hipStream_t stream);
int csts_audio_windows(const float* spec, const int* centers, float* out, int B, int T, int nbins, int cols, int width,
                       hipStream_t stream);
int csts_gaze_heatmaps(const float* labels, int label_stride, float* heatmaps, int64_t nframes, int H, int W, int ksize,
                       hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CSTS_HIP_H */
