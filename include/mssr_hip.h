/*
 * mssr_hip.h -- C ABI of libmssr_hip.so: the MI355X (gfx950) kernels behind the MSSRNet
 * transformer training-step hot path.
 *
 * The reference (OldBirdAZ/mssrnet_style_transfer) has no FFI on this path: its five L1
 * files call stock ATen ops.  Each entry point below replaces one such op chain; the
 * comment on it cites the reference lines it stands in for.  The Python host side
 * (mssrnet_style_transfer_amd/_lib.py) binds these with ctypes; INTEGRATION.md shows the
 * binding.  Conventions:
 *   - every pointer is a DEVICE pointer unless it says "host";
 *   - `dtype`: MSSR_F32 or MSSR_BF16 = element type of activations ("void*" tensors);
 *     LayerNorm gains/biases, Linear biases, statistics and parameter gradients are float;
 *   - `stream` is a hipStream_t (NULL = the default stream); calls only enqueue work;
 *   - return value MSSR_OK (0) or an error code; mssr_last_error() gives the message
 *     (thread-local).  Shape / alignment violations are rejected before anything is launched;
 *   - no entry point allocates or frees device memory; scratch is passed in by the caller.
 */
#ifndef MSSR_HIP_H
#define MSSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MSSR_F32 = 0, MSSR_BF16 = 1 };
enum { MSSR_OK = 0, MSSR_ERR_ARG = 1, MSSR_ERR_LAUNCH = 2, MSSR_ERR_UNSUPPORTED = 3 };

const char* mssr_last_error(void);
int mssr_abi_version(void);

/* ---------------------------------------------------------------- LayerNorm (eps inside) --
 * Replaces nn.LayerNorm(d, eps=1e-6): onmt/encoders/transformer.py:34,101,
 * onmt/decoders/transformer.py:69-70,244, onmt/modules/position_ffn.py:20,
 * onmt/style_generator_builder.py:529.  One wave64 per row; rows are contiguous (stride d).
 * d must be a multiple of 4 and <= 4096. */
int mssr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                       float* mean, float* rstd, int64_t rows, int d, float eps, int dtype,
                       void* stream);
/* Number of float rows of `partial` scratch (each of width d) bwd / bwd_bwd need per output. */
int mssr_layernorm_partial_rows(int64_t rows);
/* dx, dgamma[d], dbeta[d] from dy.  partial: float[2 * partial_rows * d]. */
int mssr_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                       const float* rstd, void* dx, float* dgamma, float* dbeta, float* partial,
                       int64_t rows, int d, int dtype, void* stream);
/* Backward of mssr_layernorm_bwd (the WGAN-GP double-backward, onmt/trainer_wgan_wh.py:
 * 1247-1250,1315-1318): given ggx = dL/d(dx), ggg = dL/d(dgamma) and ggb = dL/d(dbeta) (either
 * of the latter two may be NULL) produce g_dy = dL/d(dy), g_x = dL/dx and g_gamma[d] = dL/dgamma.
 * partial: float[partial_rows * d]. */
int mssr_layernorm_bwd_bwd(const void* ggx, const float* ggg, const float* ggb, const void* dy,
                           const void* x, const float* gamma, const float* mean, const float* rstd,
                           void* g_dy, void* g_x, float* g_gamma, float* partial, int64_t rows,
                           int d, int dtype, void* stream);
/* dgamma / dbeta alone, batched: out[z][0][:] = sum_rows dy*xhat, out[z][1][:] = sum_rows dy for z < batch, operand z at
 * base + z*stride (strides in elements; out_bs >= 2d floats).  What the engines need after their backward sweeps (which
 * produce dx themselves): one launch per LayerNorm KIND over all layers (onmt/decoders/transformer.py:69-70,244,
 * onmt/encoders/transformer.py:34,101, onmt/modules/position_ffn.py:20 -- autograd's weight / bias gradients of nn.LayerNorm).
 * d a multiple of 8 (bf16) / 4 (fp32), operands 16-byte aligned; accumulate != 0: added to what `out` holds. */
int mssr_layernorm_param_grads(const void* dy, const void* x, const float* mean, const float* rstd, float* out, int batch,
                               int64_t dy_bs, int64_t x_bs, int64_t stat_bs, int64_t out_bs, int64_t rows, int d,
                               int accumulate, int dtype, void* stream);

/* ------------------------------------------------------------------------- dense GEMM -----
 * C[z] = epilogue(alpha * A[z] . B[z]) for z in [0, batch_outer*batch_inner).
 * Replaces nn.Linear / torch.matmul on the path: the Q/K/V/out projections
 * (onmt/modules/multi_headed_attn.py:60-68,134-136,150-164,215), scores and context products
 * (:189,205), FFN (onmt/modules/position_ffn.py:18-19,35-36), the soft-embedding product
 * (onmt/modules/embeddings.py:280), the generator Linear (onmt/model_builder.py:181) and every
 * backward of them.  MFMA: bf16 -> v_mfma_f32_16x16x32_bf16, f32 -> v_mfma_f32_16x16x4_f32,
 * fp32 accumulate.
 * A is addressed as [m,k] with element strides (a_rs, a_cs); B as [k,n] with (b_rs, b_cs);
 * one of each pair must be 1.  C, residual and gate are row-major with leading dimensions
 * ldc / ldr / ldg.  Batch z = zo*batch_inner + zi is offset by zo*x_bo + zi*x_bi elements.
 * epilogue: v = alpha*acc (+ bias[n]) ; relu ? max(v,0) ; (+ residual) ; gate ? v*(gate>0) ;
 *           (+ beta*C_old).  */
typedef struct {
    const void* a;
    const void* b;
    void* c;
    const float* bias;
    const void* residual;
    const void* gate;
    int64_t m, n, k;
    int64_t a_rs, a_cs, b_rs, b_cs, ldc, ldr, ldg;
    int64_t batch_outer, batch_inner;
    int64_t a_bo, a_bi, b_bo, b_bi, c_bo, c_bi;   /* residual / gate share C's batch strides */
    float alpha, beta;
    int relu;
    int in_dtype;    /* dtype of A, B, residual, gate */
    int out_dtype;   /* dtype of C */
    /* dropout fused into the epilogue (after ReLU, before the residual): drop_p in [0,1), 0 = off;
     * keep(row,col) = hash(drop_seed, z*m*n + row*n + col) >= drop_p * 2^32 -- the same counter-based
     * hash as mssr_dropout_fwd, so a backward pass can regenerate the mask from the seed. */
    float drop_p;
    uint64_t drop_seed;
    /* optional fp32 scratch for the split-K slabs of the K-split kernel (k > 2048):
     * ceil(k/2048) * m * n floats.  NULL: such problems use the general kernel. */
    void* workspace;
    int64_t workspace_bytes;
    int no_skinny;   /* 1 = always use the general tiled kernel */
    /* LayerNorm folded into the product (`LN -> Linear`, onmt/decoders/transformer.py:150-157,
     * onmt/modules/position_ffn.py:35), K-split kernel only (bf16, m <= 8192, k <= 2048): when ln_colsum
     * is non-NULL, A holds the RAW rows x, B holds gamma-scaled weights W'[n,k] = gamma[k] W[n,k],
     * ln_colsum[n] = sum_k W'[n,k], bias[n] = sum_k beta[k] W[n,k] + b[n], and the result is
     *   rstd_m (acc - mean_m ln_colsum[n]) + bias[n]  ( = LN(x) W^T + b ), then the usual epilogue.
     * The row statistics (biased variance, ln_eps inside the sqrt) go to ln_mean / ln_rstd [m] if non-NULL. */
    const float* ln_colsum;
    float* ln_mean;
    float* ln_rstd;
    float ln_eps;
    /* rows below drop_row0 are exempt from the epilogue dropout (kept, not rescaled): eval-mode rows that share a
     * launch with training-mode rows (mssr_decode_plan.eval_rows).  0 = every row is dropped as drop_p says. */
    int64_t drop_row0;
    /* Segmented reduction index, for weight gradients over the rows of a [T][B, .] slab that skip the first sentences of
     * every step block (mssr_decode_plan.bwd_row0): logical k = s * k_seg_len + j (j < k_seg_len) is read at physical
     * k index s * k_seg_stride + j of BOTH operands (their pointers already point at the first wanted row); `k` is
     * the logical length.  Only the weight-gradient form (bf16, A m-contiguous, B n-contiguous, plain epilogue) takes
     * it; 0 / 0 = plain. */
    int64_t k_seg_len, k_seg_stride;
} mssr_gemm_desc;
int mssr_gemm(const mssr_gemm_desc* desc, void* stream);
/* mssr_gemm for callers that CANNOT ALLOCATE -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  No counterpart in the reference (its ops allocate freely through torch).  Same descriptor and
 * results as mssr_gemm, bit for bit; desc->workspace is ignored: the split-K slabs of the K-split kernel (k > 2048)
 * live in a file-scope device array of MSSR_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until
 * the process ends (looked up once per device with hipGetSymbolAddress; nothing is ever allocated or freed).
 * A split-K problem whose slabs ceil(k/1024) * m * n * 4 bytes exceed that size is rejected with MSSR_ERR_ARG and a
 * message (mssr_gemm with a workspace takes it); the limit applies only where mssr_gemm's own dispatch order would
 * pick the K-split kernel: problems that need no slabs (k <= 2048, the pipelined kernel's shapes from 2048 rows on,
 * the general kernel's shapes) run as in mssr_gemm whatever their size.
 * PRECONDITION (hard): at most ONE mssr_gemm_noalloc call may be in flight per GPU at any time.  All calls on one GPU
 * share that GPU's array, so split-K calls that overlap in time (two streams, two threads, two captured graphs
 * replayed concurrently) silently corrupt each other's slabs -- keep every noalloc call of a GPU, and every graph
 * that captured one, on ONE stream, or use mssr_gemm with a workspace per stream.
 * Nothing in the array has to start at zero: every slab element is written before it is read. */
#define MSSR_NOALLOC_SCRATCH_BYTES (16u << 20)
int mssr_gemm_noalloc(const mssr_gemm_desc* desc, void* stream);
int64_t mssr_gemm_noalloc_scratch_bytes(void);
/* Block tile edge (64 or 32) mssr_gemm picks for an m x n x batch problem; lets a profiler name the
 * kernel instance (gemm_kernel<.., 64, 64> / <.., 32, 32>) a launch used.  No reference counterpart. */
int mssr_gemm_tile(int64_t m, int64_t n, int64_t batch);
/* 1 when mssr_gemm would route `desc` to the K-split ("skinny") kernel (m <= 8192, bf16, both
 * operands k-contiguous and 16-byte aligned, k a multiple of 32): C = A[m,k] . W[n,k]^T, one block
 * per 16x16 (m <= 128) or 32x32 output tile, its four waves splitting K and loading straight into
 * MFMA fragments -- a single memory round trip per launch.  No reference counterpart. */
int mssr_gemm_is_skinny(const mssr_gemm_desc* desc);
/* Diagnostics (tools/stamp_bench.py), no reference counterpart: while `buf` (device memory, 8 x uint64 per thread block
 * of the launches that follow) is set, the K-split kernel's blocks record the 100 MHz real-time clock at six points
 * (start, loads issued, loads landed -- this adds a full wait --, tile in LDS, after the barrier, end).  NULL: off. */
void mssr_debug_stamps(void* buf);

/* Launch timing for the benchmark's roofline (no reference counterpart): between mssr_prof_start
 * and mssr_prof_stop every mssr_gemm launch -- direct or from inside the decode engine -- is timed with a HIP
 * event pair handed to hipExtLaunchKernelGGL on the stream it is enqueued on (begin / end of the dispatch
 * itself, the interval rocprofv3's kernel trace reports; ms < 0: nothing was launched).  kernel: 0 = gemm_kernel (tile = 32/64),
 * 1 = gemm_skinny_kernel (tile = 16/32 output tile edge, ksplit = split-K slabs, ln = LayerNorm prologue),
 * 2 = gemm_tn_kernel (both operands k-major: weight gradients), 3 = gemm_lnb_kernel (LayerNorm backward fused with the
 * product behind it: decode engine only).
 * Host-side only. */
typedef struct {
    int kernel, in_dtype, out_dtype, tile, ksplit, ln;
    int64_t batch, m, n, k;
    float ms;
} mssr_prof_record;
int mssr_prof_start(int max_records);
int mssr_prof_stop(void);                       /* waits for the device; returns the record count */
int mssr_prof_get(int i, mssr_prof_record* out);

/* --------------------------------------------------------------------- row softmax --------
 * y = softmax(scale * x) along the last axis, optionally masked, optionally log-softmax.
 * Replaces scores.float().masked_fill(mask,-1e18) -> softmax (multi_headed_attn.py:195-202),
 * F.softmax(logits / T) in the soft decode (onmt/train_and_translate.py:115) and the
 * generator's LogSoftmax (onmt/model_builder.py:179).  Computes in fp32.
 * Rows: r in [0, rows); x row r at x + r*ldx, y row at y + r*ldy.
 * mask (uint8, nonzero = masked -> score replaced by -1e18) may be NULL; the mask row of r is
 * mask + (r / rows_per_batch)*mask_bs + (r % q_len)*mask_qs. */
int mssr_softmax_fwd(const void* x, void* y, const uint8_t* mask, int64_t rows, int n, int64_t ldx,
                     int64_t ldy, int64_t rows_per_batch, int64_t q_len, int64_t mask_bs,
                     int64_t mask_qs, float scale, int log_mode, int in_dtype, int out_dtype,
                     void* stream);
/* dx = scale * y*(dy - sum(y*dy))   (log_mode: scale * (dy - exp(y)*sum(dy))). */
int mssr_softmax_bwd(const void* y, const void* dy, void* dx, int64_t rows, int n, int64_t ldy,
                     int64_t lddy, int64_t lddx, float scale, int log_mode, int y_dtype,
                     int dy_dtype, int dx_dtype, void* stream);
/* Backward of mssr_softmax_bwd w.r.t. y (softmax mode): given g = dL/d(dx):
 *   g_y = scale * ( g*(dy - sum(y*dy)) - dy*sum(g*y) )          (g_dy is softmax_bwd(y, g)). */
int mssr_softmax_bwd_bwd(const void* g, const void* y, const void* dy, void* g_y, int64_t rows, int n,
                         int64_t ld, float scale, int dtype, void* stream);

/* --------------------------------------------------------------------- elementwise --------
 * dropout / residual sites: onmt/encoders/transformer.py:51, onmt/decoders/transformer.py:
 * 162,169, onmt/modules/position_ffn.py:35-37, multi_headed_attn.py:203, embeddings.py:53. */
/* y = keep * x / (1-p) (+ residual); keep ~ Bernoulli(1-p) from a counter-based hash of
 * (seed, element index); keep mask written to mask_out (uint8) for the backward passes. */
int mssr_dropout_fwd(const void* x, const void* residual, void* y, uint8_t* mask_out, int64_t n,
                     float p, uint64_t seed, int dtype, void* stream);
/* y = x * mask * scale (+ residual): re-applies a stored keep mask (dropout backward / replay). */
int mssr_mask_scale(const void* x, const uint8_t* mask, const void* residual, void* y, int64_t n,
                    float scale, int dtype, void* stream);
/* y = x * (gate > 0): ReLU backward as an op (position_ffn.py:35). */
int mssr_gate_gt0(const void* x, const void* gate, void* y, int64_t n, int dtype, void* stream);
/* Critic gain at the label column: out[i] = ratio * sigmoid(x[i, labels[i]]) (sigmoid != 0: the "sigmoid"
 * force_positive_gain_mode, onmt/trainer_wgan_wh.py:402-416) or x[i, labels[i]] ("normal"), i.e. the gain followed by
 * cal_disc_loss's per-row pick (:1192-1200) in one launch; x float [n, C] with row stride ldx, labels int64 [n].
 * _bwd: dx [n, C] contiguous = onehot(labels) * dy * f'(x).  _bwd_bwd: the derivative of _bwd w.r.t. (dy, x) for an
 * incoming ddx [n, C] (row stride ldd): ddy [n], dx2 [n, C] -- the gradient penalty differentiates the critic's input
 * gradient, which runs through _bwd.  A label outside [0, C) gives NaN. */
int mssr_gain_pick_fwd(const float* x, int64_t ldx, const int64_t* labels, float* out, int n, int C, float ratio,
                       int sigmoid, void* stream);
int mssr_gain_pick_bwd(const float* dy, const float* x, int64_t ldx, const int64_t* labels, float* dx, int n, int C,
                       float ratio, int sigmoid, void* stream);
int mssr_gain_pick_bwd_bwd(const float* ddx, int64_t ldd, const float* dy, const float* x, int64_t ldx,
                           const int64_t* labels, float* ddy, float* dx2, int n, int C, float ratio, int sigmoid,
                           void* stream);
/* WGAN-GP penalty of per-token input gradients (penalize_gen_style_grad / penalize_gen_text_grad,
 * onmt/trainer_wgan_wh.py:1253-1261, :1321-1329: gradients.norm(2, dim=1) over the rows the sequence mask selects,
 * ((norm - k) ** 2).mean() * lamb) in one launch.  Row (i0, i1), i0 < n0, i1 < n1, is g + i0*s0 + i1*s1 (elements, d
 * contiguous values) and mask + i0*m0 + i1*m1 (uint8, != 0: selected).  norms [n0*n1] receives every row's norm;
 * acc: float[4] ZEROED by the caller, on return acc[0] = sum of (norm - k)^2, acc[1] = selected rows, acc[3] = the
 * penalty (acc[2] is the kernel's ticket).  No selected row -> NaN, as the reference's mean of an empty tensor. */
int mssr_gp_penalty_fwd(const void* g, int64_t s0, int64_t s1, int n0, int n1, int d, const uint8_t* mask, int64_t m0,
                        int64_t m1, float k, float lamb, float* norms, float* acc, int dtype, void* stream);
/* Its gradient: dg [n0*n1, d] contiguous = dpen[0] * lamb / acc[1] * 2 (norm - k) / norm * g on selected rows (0 at
 * norm = 0: the subgradient torch's norm backward uses), 0 elsewhere; norms / acc as left by the forward. */
int mssr_gp_penalty_bwd(const void* g, int64_t s0, int64_t s1, int n0, int n1, int d, const uint8_t* mask, int64_t m0,
                        int64_t m1, float k, float lamb, const float* norms, const float* acc, const float* dpen,
                        void* dg, int dtype, void* stream);
/* ContextReps2's pooling (onmt/style_generator_builder.py:243-256: the second Linear's scores w_fn[2](h) [B, L, 1], the
 * in-place -inf mask past each sentence's length, softmax over time, weighted sum of the states), one launch:
 *   s_t = h_t . w2 + b2,  a = softmax over t < lengths[b] of s (0 beyond),  pooled[b] = sum_t a_t x[b, t].
 * x [B, L, d] and h [B, L, m] (h = relu(w_fn[0](x)), computed by the caller's GEMM) are addressed as
 * base + b*_bs + t*_ts (elements; features contiguous), so [L, B, .] storage is read in place; w2 float [m], b2 float
 * [1], lengths int64 [B]; pooled [B, d] and a float [B, L] contiguous.  d % 4 == 0, d <= 1024, L <= 1024.
 * _bwd: from dp = d(loss)/d(pooled) [B, d] (may be NULL) and q = d(loss)/d(a) float [B, L] (may be NULL; the
 *   second-order pass produces one): dx [B, L, d] (NULL when dp is), dh [B, L, m], and dw2b2 float [m + 1] (dw2 then
 *   db2) which the caller ZEROED -- sentences add into it.
 * _bwd_bwd: the derivative of _bwd for the cotangents ddx (of dx), ddh (of dh), ddw2 float [m], ddb2 float [1] (each may
 *   be NULL): g_dp [B, d], g_x [B, L, d], g_a float [B, L] (hand it to _bwd as q), g_h [B, L, m] (required with ddw2,
 *   else may be NULL), g_w2 float [m] ZEROED by the caller (required with ddh).  The WGAN-GP penalties differentiate
 *   the critics' input gradient (onmt/trainer_wgan_wh.py:1247-1250, 1315-1318), which runs through _bwd. */
int mssr_ctx_pool_fwd(const void* x, int64_t x_bs, int64_t x_ts, const void* h, int64_t h_bs, int64_t h_ts, const float* w2,
                      const float* b2, const int64_t* lengths, void* pooled, float* a, int B, int L, int d, int m,
                      int dtype, void* stream);
int mssr_ctx_pool_bwd(const void* dp, const float* q, const void* x, int64_t x_bs, int64_t x_ts, const void* h, int64_t h_bs,
                      int64_t h_ts, const float* w2, const float* a, const int64_t* lengths, void* dx, int64_t dx_bs,
                      int64_t dx_ts, void* dh, int64_t dh_bs, int64_t dh_ts, float* dw2b2, int B, int L, int d, int m,
                      int dtype, void* stream);
int mssr_ctx_pool_bwd_bwd(const void* ddx, int64_t ddx_bs, int64_t ddx_ts, const void* ddh, int64_t ddh_bs, int64_t ddh_ts,
                          const float* ddw2, const float* ddb2, const void* dp, const float* q, const void* x, int64_t x_bs,
                          int64_t x_ts, const void* h, int64_t h_bs, int64_t h_ts, const float* w2, const float* a,
                          const int64_t* lengths, void* g_dp, void* g_x, int64_t gx_bs, int64_t gx_ts, void* g_h,
                          int64_t gh_bs, int64_t gh_ts, float* g_w2, float* g_a, int B, int L, int d, int m, int dtype,
                          void* stream);
/* out[n] (+)= sum over rows of x[rows, n] (Linear bias gradient). */
int mssr_colsum(const void* x, float* out, int64_t rows, int n, int64_t ldx, int accumulate,
                int dtype, void* stream);
/* The same for `batch` matrices x_bs elements apart (the bias gradients of all layers at once): out [batch, n];
 * accumulate != 0: added to what `out` holds (a caller that zeroes all its small gradient outputs with one fill). */
int mssr_colsum_batched(const void* x, float* out, int batch, int64_t x_bs, int64_t rows, int n, int64_t ldx,
                        int accumulate, int dtype, void* stream);
/* y = a*x + b*z elementwise (z may be NULL). */
int mssr_axpby(const void* x, const void* z, void* y, int64_t n, float a, float b, int dtype,
               void* stream);

/* ------------------------------------------------------------------ weight packs ----------
 * Every derived copy of the fp32 master parameters that the engines read, refreshed in ONE launch after an
 * optimizer step: compute-dtype copies packed into [layer][rows][k] slabs (Q|K|V rows of
 * onmt/modules/multi_headed_attn.py:63-70 side by side, ...), their transposes (the k-contiguous operand of
 * dX = dY W), LayerNorm folded into the Linear behind it (gamma * W, the row sums of the rounded values, W beta + b:
 * onmt/encoders/transformer.py:47-52, onmt/decoders/transformer.py:110-171, onmt/modules/position_ffn.py:35-37)
 * and fp32 copies of bias / LayerNorm vectors.  The reference has no counterpart: it reads the parameters in place.
 * A job = one parameter tensor src [rows, cols] (row-major fp32) and where its copies go; every output pointer may
 * be NULL.  The job table lives in DEVICE memory (built once per network: the pointers do not change), block0 =
 * first block of the job in the launch, mssr_pack_blocks(rows) blocks per job, n_blocks = their sum. */
typedef struct {
    const float* src;
    int32_t rows, cols;
    void* dst;   int64_t dst_ld;      /* dst[r*dst_ld + c]     = T(src[r][c]) */
    void* dst_t; int64_t dst_t_ld;    /* dst_t[c*dst_t_ld + r] = T(src[r][c]) */
    void* dst_g; int64_t dst_g_ld;    /* dst_g[r*dst_g_ld + c] = T(src[r][c] * gamma[c]) */
    const float* gamma;               /* [cols], with dst_g / colsum */
    const float* beta;                /* [cols], with bias_f */
    const float* bias;                /* [rows] or NULL, with bias_f */
    float* colsum;                    /* [rows]  sum_c float(T(src[r][c] * gamma[c])) */
    float* bias_f;                    /* [rows]  sum_c src[r][c] * beta[c] + bias[r] */
    float* dst_f;                     /* [rows*cols] fp32 copy */
    int32_t block0, pad_;
} mssr_pack_job;
int mssr_pack_blocks(int rows);
int mssr_pack_weights(const mssr_pack_job* jobs_dev, int n_jobs, int n_blocks, int dtype, void* stream);

/* ------------------------------------------------------- gradient norm, clipping, Adam -----
 * One optimizer step of a whole network: torch.nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.Adam.step()
 * as built and called by the reference trainer (onmt/trainer_wgan_wh.py:192-211 Adam(lr, betas=(beta1, beta2)),
 * :623-628 and :973-977 clip then step; onmt/trainer_teacher.py:220,265-267).  `params`, `grads`, `numels`,
 * `state_offsets` are HOST arrays of n entries; params[i] / grads[i] are DEVICE pointers to contiguous fp32 tensors of
 * numels[i] elements, 16-byte aligned; grads[i] == NULL: the parameter received no gradient and is skipped, as torch
 * skips it.  Pointers travel in kernel arguments (96 tensors per launch): nothing is uploaded or allocated. */
/* floats of scratch mssr_grad_norm needs for n tensors */
int64_t mssr_grad_norm_scratch_floats(int n_tensors);
/* out[0] = global L2 norm of all gradients, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)) (1 when
 * max_norm <= 0).  Deterministic two-stage reduction (no atomics).  partial: float[mssr_grad_norm_scratch_floats(n)]. */
int mssr_grad_norm(int n, void* const* grads, const int64_t* numels, float max_norm, float* partial, float* out,
                   void* stream);
/* Adam with bias correction, no weight decay, no amsgrad.  steps[i] (host array) = how many updates tensor i has had
 * INCLUDING this one (torch.optim.Adam counts steps per parameter: one that got no gradient lags).  The moments of tensor i live at
 * exp_avg + state_offsets[i] / exp_avg_sq + state_offsets[i] (offsets multiples of 4).  norm_coef: the `out` of
 * mssr_grad_norm (gradients are multiplied by out[1] on the fly) or NULL; write_clipped != 0 also stores the
 * scaled gradients back, as clip_grad_norm_ does. */
int mssr_adam_step(int n, void* const* params, void* const* grads, const int64_t* numels,
                   const int64_t* state_offsets, float* exp_avg, float* exp_avg_sq, const float* norm_coef, float lr,
                   float beta1, float beta2, float eps, const int32_t* steps, int write_clipped, void* stream);
/* dst[i][:] = scale * src[i][:] for n contiguous fp32 tensors (src[i] == NULL: zeros), pointers in kernel arguments as
 * above: the gather of a gradient bucket into its flat all-reduce buffer -- the "flat buckets" of the reference's dormant
 * all_reduce_and_rescale_tensors (onmt/utils/distributed.py:47-86), one launch per 96 tensors. */
int mssr_multi_copy(int n, void* const* dst, void* const* src, const int64_t* numels, float scale, void* stream);

/* ---------------------------------------------------------------- embeddings + PE ---------
 * Replaces nn.Embedding lookup + PositionalEncoding (onmt/modules/embeddings.py:48-54,282):
 * out[r,:] = W[ids[r],:]*scale + pe[pos(r),:],  pos(r) = step >= 0 ? step : r / batch.
 * W is float [V,d] (the parameter itself), pe is float [max_len, d]. */
int mssr_embedding_fwd(const int64_t* ids, const float* weight, const float* pe, void* out,
                       int64_t rows, int d, int64_t vocab, int64_t batch, int step, float scale,
                       int dtype, void* stream);
/* dW[ids[r],:] += scale * dout[r,:] for ids[r] != pad_idx (float atomics). */
int mssr_embedding_bwd(const void* dout, const int64_t* ids, float* dweight, int64_t rows, int d,
                       int64_t vocab, int64_t pad_idx, float scale, int dtype, void* stream);
/* Soft path epilogue (embeddings.py:48-52 after :280): out = x*scale + pe[pos(r)]. */
int mssr_scale_add_pe(const void* x, const float* pe, void* out, int64_t rows, int d, int64_t batch,
                      int step, float scale, int dtype, void* stream);

/* ------------------------------------------------------------- fused attention ------------
 * ctx = dropout(softmax(mask(q k^T / sqrt(dh)))) v for short sequences, one kernel per direction.
 * Replaces onmt/modules/multi_headed_attn.py:187-205 (query scaling, matmul, masked_fill(mask, -1e18),
 * fp32 softmax, dropout, matmul) on the no-cache branch (:161-166) and its autograd backward.
 * q/k/v are [B, L, H, dh] views (head h at column offset h*dh) addressed by batch and row strides in
 * elements (multiples of 4), so the packed projection [B*L, 3d] is used in place; mask is uint8,
 * nonzero = masked, row of (b, i) at mask + b*mask_bs + i*mask_qs (mask_qs = 0: key padding only).
 * Two kernel families behind one entry point (mssr_attention_fits() tells whether either applies; larger shapes go
 * through mssr_gemm + mssr_softmax_*):
 *   matrix-core kernels (csrc/attention_mfma.hip): bf16, dh = 32 or 64, Lk <= 160 (and Lq <= 160 backward), q/k/v
 *     rows 16-byte aligned -- the sentence lengths of every configuration incl. the 128-word stress case;
 *   scalar kernels (csrc/attention.hip): fp32 or bf16, Lk <= 64, dh <= 64 (and Lq <= 64 backward), staged tiles
 *     within 64 KB of LDS -- the fp32 parity runs and odd head sizes.  MSSR_ATTN_MFMA=0 prefers them where both fit.
 * Dropout: keep(b,h,i,t) = hash16(seed, ((b*H+h)*Lq + i)*256 + t) >= p_drop * 2^16 (one 32-bit hash per key pair),
 * the same decision in both families, regenerated by the backward from the same seed. */
typedef struct {
    const void *q, *k, *v;
    const uint8_t* mask;          /* may be NULL */
    int B, H, Lq, Lk, dh;
    int64_t q_bs, q_ls, k_bs, k_ls, v_bs, v_ls, mask_bs, mask_qs;
    float p_drop;                 /* 0 = no dropout */
    uint64_t seed;
    int dtype;
    int drop_b0;                  /* sentences b < drop_b0 are NOT dropped (eval-mode rows batched with training rows) */
} mssr_attn_desc;
int mssr_attention_fits(int Lq, int Lk, int dh, int dtype, int backward);
/* ctx row (b, i) at ctx + b*ctx_bs + i*ctx_ls, head h at column h*dh; probs (may be NULL) receives the
 * probabilities BEFORE dropout as contiguous [B, H, Lq, Lk] (what the module returns, :223-227). */
int mssr_attention_fwd(const mssr_attn_desc* desc, void* ctx, int64_t ctx_bs, int64_t ctx_ls, void* probs,
                       void* stream);
/* dq, dk, dv from dctx (same layout conventions); probabilities are recomputed from q and k. */
int mssr_attention_bwd(const mssr_attn_desc* desc, const void* dctx, int64_t dc_bs, int64_t dc_ls, void* dq,
                       int64_t dq_bs, int64_t dq_ls, void* dk, int64_t dk_bs, int64_t dk_ls, void* dv,
                       int64_t dv_bs, int64_t dv_ls, void* stream);

/* ------------------------------------------------------- encoder-stack engine -------------
 * Replaces the Python layer loop of TransformerEncoder.forward (onmt/encoders/transformer.py:117-136)
 * -> n_layers x TransformerEncoderLayer.forward (:37-52) -> final LayerNorm (:132), the same loop
 * inside StyleGenerator2.forward (onmt/style_generator.py:40-47), and their autograd backward.
 * The caller embeds the input (rows batch-first, R = B*L) into X[0] and owns every buffer.
 * Weights in the compute dtype, nn.Linear layout [out,in] (w_*) and transposed (wt_*, backward);
 * biases and LayerNorm parameters float.  `layers` is a HOST array. */
typedef struct {
    const void *w_qkv, *wt_qkv;   /* self_attn query|keys|values rows packed [3d,d]; wt [d,3d] */
    const void *w_o, *wt_o;       /* self_attn.final_linear [d,d] */
    const void *w_1, *wt_1;       /* feed_forward.w_1 [ff,d]; wt [d,ff] */
    const void *w_2, *wt_2;       /* feed_forward.w_2 [d,ff]; wt [ff,d] */
    const float *b_qkv, *b_o, *b_1, *b_2;
    const float *ln1_g, *ln1_b;   /* layer.layer_norm */
    const float *lnf_g, *lnf_b;   /* feed_forward.layer_norm */
    /* fuse_ln (forward-only bf16 plans; see mssr_gemm_desc.ln_colsum): gamma-scaled weights, their row
     * sums, W beta + b.  May be NULL when fuse_ln = 0. */
    const void *wg_qkv, *wg_1;
    const float *cs_qkv, *cs_1, *bf_qkv, *bf_1;
} mssr_enc_layer;

typedef struct {
    int dtype, B, L, d, heads, ff, n_layers;
    int save;                     /* 1: keep per-layer activations for mssr_encoder_bwd */
    int training;                 /* 1: dropout active */
    int fuse_ln;                  /* 1: LayerNorms folded into the projections (save = 0, bf16 only) */
    float p_drop, p_attn;
    uint64_t seed;
    const mssr_enc_layer* layers; /* host array [n_layers] */
    const float *lnF_g, *lnF_b;   /* encoder.layer_norm */
    const uint8_t* mask;          /* [B, L] nonzero = padding key (may be NULL) */
    void* X;                      /* [n_layers+1][R, d]: X[0] = input, X[l+1] = output of layer l */
    void* out;                    /* [R, d] = LayerNorm(X[n_layers]) */
    void* qkv;                    /* [Ls][R, 3d]   (Ls = save ? n_layers : 1) */
    void* actD;                   /* [Ls][4][R, d]  n1 | ctx | h | n2 */
    void* actF;                   /* [Ls][R, ff]    FFN inner activation */
    float* stats;                 /* [2*n_layers+1][2][R] LayerNorm mean, rstd */
    void* gemm_ws;                /* split-K scratch of mssr_gemm (may be NULL) */
    int64_t gemm_ws_bytes;
    /* backward only */
    const void* dout;             /* [R, d] dL/d out */
    void* dX;                     /* [R, d] on return: dL/d X[0] */
    void* dqkv;                   /* [n_layers][R, 3d] */
    void* dactD;                  /* [n_layers][4][R, d]  dn1 | d_o | dn2 | d_f2 (gradients of the outputs of
                                   * layer_norm, final_linear, feed_forward.layer_norm, w_2) */
    void* dactF;                  /* [n_layers][R, ff]    gradient of the w_1 output */
    void* dctx;                   /* [R, d] scratch */
    /* Two passes over the same weights in ONE call (no reference counterpart: the reference encodes the critics'
     * inputs in eval mode under no_grad, onmt/trainer_wgan_wh.py:530-535, and the generator phase's in train mode,
     * :898-968, one after the other): sentences [0, eval_rows) run WITHOUT dropout whatever `training` says.  Their
     * rows take part in the backward like any other (the caller feeds zero gradient for them).  0 = plain. */
    int eval_rows;
} mssr_encoder_plan;
int mssr_encoder_fwd(const mssr_encoder_plan* plan, void* stream);
/* Leaves the gradient of every Linear output in the d* slabs; the caller contracts them with the saved
 * activations into parameter gradients (one batched product per kind over all layers). */
int mssr_encoder_bwd(const mssr_encoder_plan* plan, void* stream);

/* ------------------------------------------------- step-wise soft-decode engine -----------
 * Replaces the Python loop of the reference's differentiable decoding,
 * TrainTranslator.decode_batch_ori (onmt/train_and_translate.py:202-231) ->
 * TransformerDecoder.forward(step=s) (onmt/decoders/transformer.py:289-337) -> n_layers x
 * TransformerDecoderLayer._forward (:110-171) with the layer_cache protocol of
 * onmt/modules/multi_headed_attn.py:133-160 -> generator Linear + softmax(/T)
 * (onmt/train_and_translate.py:95-116), and the autograd backward of that loop.
 * One call enqueues one decode step (or all of them) for the whole batch; the caller owns
 * every buffer (nothing is allocated here).  Weights are in the compute dtype, nn.Linear layout
 * [out,in] (w_*) plus a transposed copy [in,out] (wt_*, backward only); biases and LayerNorm
 * parameters are float.  `layers` is a HOST array. */
typedef struct {
    const void *w_qkv, *wt_qkv;   /* self attention query|keys|values rows packed [3d,d]; wt [d,3d] */
    const void *w_so, *wt_so;     /* self attention final_linear [d,d] */
    const void *w_cq, *wt_cq;     /* context attention linear_query [d,d] */
    const void *w_co, *wt_co;     /* context attention final_linear [d,d] */
    const void *w_1, *wt_1;       /* feed_forward.w_1 [ff,d]; wt [d,ff] */
    const void *w_2, *wt_2;       /* feed_forward.w_2 [d,ff]; wt [ff,d] */
    const float *b_qkv, *b_so, *b_cq, *b_co, *b_1, *b_2;
    const float *ln1_g, *ln1_b;   /* layer_norm_1 */
    const float *ln2_g, *ln2_b;   /* layer_norm_2 */
    const float *lnf_g, *lnf_b;   /* feed_forward.layer_norm */
    /* LayerNorm folded into the Linear that follows it (mssr_decode_plan.fuse_ln; see mssr_gemm_desc.ln_colsum):
     * wg_* = gamma-scaled weights (compute dtype), cs_* = their row sums, bf_* = W beta + b (float).
     * qkv <- layer_norm_1, cq <- layer_norm_2, w_1 <- feed_forward.layer_norm.  May be NULL when fuse_ln = 0. */
    const void *wg_qkv, *wg_cq, *wg_1;
    const float *cs_qkv, *cs_cq, *cs_1;
    const float *bf_qkv, *bf_cq, *bf_1;
} mssr_dec_layer;

typedef struct {
    int dtype;                    /* compute dtype of weights / activations */
    int B, d, heads, ff, n_layers;
    int V, Vp;                    /* vocabulary and its padding to a multiple of 32 */
    int Ls, T;                    /* source (memory) length, number of decode steps */
    int save;                     /* 1: keep per-step activations for mssr_decode_step_bwd */
    int training;                 /* 1: dropout active (p_drop residual/FFN/embedding, p_attn attention) */
    int n_chunks;                 /* 1..8: the batch is cut into this many row chunks whose (independent)
                                   * dependency chains run on concurrent internal streams forked from / joined
                                   * into the caller's stream; dropout masks depend on the chunking */
    int fuse_ln;                  /* 1 (bf16 only): the four LayerNorms before a Linear run inside that Linear's
                                   * kernel; the n1/n2/n3 kinds of actD and nf are then NOT written (the caller
                                   * recomputes them from X / h1 / h2 and `stats` for the weight gradients) */
    float p_drop, p_attn, temperature;
    uint64_t seed;
    const mssr_dec_layer* layers; /* host array [n_layers] */
    const void* emb_t;            /* decoder embedding table transposed [d, Vp] (padded columns zero) */
    const void* emb;              /* the table itself [V, d] (backward) */
    const float* pe;              /* positional table [>= T, d] */
    const void* w_gen;            /* generator Linear [V, d] */
    const void* wt_gen;           /* its transpose [d, Vp], padded columns zero (backward) */
    const float* b_gen;
    const float *lnF_g, *lnF_b;   /* decoder.layer_norm */
    const void* wg_gen;           /* fuse_ln: generator weights scaled by lnF_g [V, d], row sums, W lnF_b + b */
    const float *cs_gen, *bf_gen;
    const void* mem_kv;           /* [B, Ls, n_layers, 2d]: context keys|values of the memory bank, all layers */
    const uint8_t* src_mask;      /* [B, Ls], nonzero = source padding */
    void* in_all;                 /* [T+1][B, Vp] step inputs; slot 0 = one-hot <s>; padded columns zero */
    float* probs;                 /* [T][B, V] output probabilities (float) */
    void* X;                      /* [n_layers+1][Ts][B, d]   (Ts = save ? T : 1) */
    void* qkv;                    /* [n_layers][B][T, 3d]     self-attention projections = K/V cache (sentence-major) */
    void* actD;                   /* [n_layers][8][Ts][B, d]  n1|ctx1|h1|n2|qc|ctx2|h2|n3 */
    void* actF;                   /* [n_layers][Ts][B, ff] */
    float* stats;                 /* [3*n_layers+1][2][Ts][B] LayerNorm mean, rstd */
    void* nf;                     /* [Ts][B, d] */
    void* logits;                 /* [B, Vp] */
    void* tmp;                    /* [B, d] */
    void* gemm_ws;                /* split-K scratch of mssr_gemm (see mssr_gemm_desc.workspace) */
    int64_t gemm_ws_bytes;
    /* backward only (NULL for forward-only plans) */
    float* dprobs;                /* [T][B, V] in: dL/dprobs; the feedback path accumulates into it */
    void* dlogits;                /* [T][B, Vp], padded columns zero */
    void* dX;                     /* [2][B, d]: running residual-stream gradient, two buffers used in turn */
    void* dqkv;                   /* [n_layers][T][B, 3d] */
    void* dactD;                  /* [n_layers][7][T][B, d]  dn1|d_so|dn2|d_cq|d_co|dn3|d_f2 */
    void* dactF;                  /* [n_layers][T][B, ff]    d_f1 */
    void* dctx;                   /* [n_layers][2][T][B, d] gradients of the self / context attention outputs */
    float* ds_self;               /* [n_layers][B*heads][T keys][T steps][2]: pairs (score gradient ds_s[t], dropped
                                   * probability pd_s[t]), t <= s, of the self attention, key-major: key / value
                                   * gradients are gathered from these (dK[t] = sum_{s>=t} ds_s[t] q_s, ...), and
                                   * the coefficients of one key over the steps are contiguous */
    float* ds_ctx;                /* [n_layers][2][T][B*heads][Ls]: the same for the context attention; the caller
                                   * contracts them over the steps: dL/d mem K = sum_s ds_s (x) qc_s,
                                   * dL/d mem V = sum_s pd_s (x) dctx_s */
    void* d_e;                    /* [T][B, d] gradient of the soft-embedding product */
    void* dnf;                    /* [T][B, d] */
    /* greedy search (beam 1): onmt/translate/translator_wh.py:709-746 + onmt/translate/greedy_search.py:135-165.
     * greedy = 1 (forward-only plans): each step writes only the argmax token ids[step][b] (int32; no probability
     * slab), and steps > 0 embed ids[step-1] by a lookup in `emb` (step 0 reads the one-hot <s> of in_all[0]).
     * Sentences are independent: the host drops finished ones between steps by building a smaller plan from the
     * survivors' rows (decode_engine.greedy_decode_ids; reference: translator_wh.py:744-746). */
    int32_t* ids;                 /* [T][B] or NULL */
    int greedy;
    /* Two decodes of the same weights in ONE sweep (no reference counterpart: the reference runs them one after the
     * other, onmt/trainer_wgan_wh.py:530-535 in eval mode for the critics' fake text and :917-920 in train mode for the
     * generator step; a sweep costs the same ~100 dependent launches per step whether it carries 80 rows or 240):
     * sentences [0, eval_rows) run WITHOUT dropout whatever `training` says, and the backward sweep covers sentences
     * [bwd_row0, B) only (the others carry no gradient).  Both 0 = the plain behaviour.  Need n_chunks = 1 and
     * eval_rows <= bwd_row0. */
    int eval_rows;
    int bwd_row0;
    /* Teacher-forced sentences in the same sweep (no reference counterpart as one pass: the reference runs the
     * teacher-forced reconstruction, TransformerDecoder.forward with step = None, onmt/trainer_wgan_wh.py:632-664, and
     * the soft decodes one after the other): sentences [0, forced_rows) read their inputs of steps 1 .. forced_steps-1
     * from in_all as the CALLER preset them (one-hot target words; every sentence's step-0 input is preset anyway) --
     * a step's output does not replace them -- and the backward sends no gradient from those inputs into the previous
     * step's probabilities.  self_mask (uint8 [B, T], nonzero = this step's key is padding; may be NULL) is the target
     * padding mask the full-sequence decoder applies to its self attention (onmt/decoders/transformer.py:301-302; the
     * step-wise reference path has none).  Not with greedy or eval_rows. */
    int forced_rows, forced_steps;
    const uint8_t* self_mask;
    /* Sentences that need fewer steps than the sweep has (a batched sweep serves decodes of different lengths): the
     * BACKWARD steps >= late_step cover sentences [late_row0, B) only; the caller guarantees that the others carry
     * no gradient at those steps (dprobs zero) and has ZEROED their rows of steps >= late_step in dqkv, dactD, dactF,
     * dctx, d_e, dnf, ds_self and ds_ctx (the slabs the gathers and the parameter-gradient contractions read over all
     * steps).  late_row0 = 0: every backward step covers [bwd_row0, B).  Needs n_chunks = 1.
     * late_fwd = 1: the FORWARD steps >= late_step skip sentences [0, late_row0) as well (their decodes have ended:
     * probs, activations and cache rows of those steps are not written; a saving caller zeroes those rows of X, qkv,
     * actD, actF, stats and nf, which the parameter-gradient contractions read).  Set it the same for both sweeps. */
    int late_step, late_row0, late_fwd;
} mssr_decode_plan;
/* One decode step: embeds in_all[step], runs the layers with the cache rows 0..step, writes
 * probs[step] and in_all[step+1]. */
int mssr_decode_step_fwd(const mssr_decode_plan* plan, int step, void* stream);
/* Its backward; call for step = T-1 .. 0.  Leaves the gradient of every GEMM output in the d*
 * slabs (the caller contracts them with the saved activations into parameter gradients, one
 * product per parameter over all steps) and the context-attention score gradients in ds_ctx. */
int mssr_decode_step_bwd(const mssr_decode_plan* plan, int step, void* stream);
/* All steps in one call (the T-iteration loop of onmt/train_and_translate.py:202-231 and its reverse).
 * Replay: a sweep's ~2 800 launches depend only on the plan's contents; the second time a plan with the same
 * contents (every field but `seed`, and the layer table) arrives on a stream, its launches are recorded as a hipGraph
 * and from then on the call is one seed store + one hipGraphLaunch (0.9 ms of host time instead of ~9 ms) with the
 * same launches and bit-identical results; the per-call dropout seed is read by the recorded kernels from a device
 * word.  Plans whose buffers move from call to call never match and run eagerly, as do calls made while the caller
 * captures the stream itself or records profiles (mssr_prof_start).  MSSR_DECODE_GRAPH=0 disables the mechanism.
 * At most 16 recordings are kept (least recently used is dropped).  No device memory is allocated. */
int mssr_decode_fwd(const mssr_decode_plan* plan, void* stream);
int mssr_decode_bwd(const mssr_decode_plan* plan, void* stream);
/* Number of recorded sweeps currently kept (diagnostics, tests).  No reference counterpart. */
int mssr_decode_graphs(void);
/* Replay on (default) / off for the whole process; returns the previous setting.  Off: every sweep is launched eagerly.
 * A process that also drives a second stream (data-parallel ranks: the gradient all-reduce, parallel.GradSync) should
 * turn it off -- measured on MI355X / ROCm 7.2: once a side stream has carried work, each replayed sweep runs ~2 ms
 * longer on the GPU (43.2 -> 48.2 ms per WGAN step), while eagerly launched sweeps are unaffected (43.2 -> 43.7). */
int mssr_decode_graphs_enable(int on);
/* Recordings made since the process started (never decreases: a full cache evicts its least recently used sweep). */
long mssr_decode_graphs_recorded(void);
/* Gradient of the memory keys | values of a decode sweep, all layers in one launch: contracts the per-step slabs the
 * backward sweep left in plan.ds_ctx [n_layers][2 = ds|pd][T][B*heads][Ls] (float) with the saved context queries
 * (qc: layer l at qc + l*qc_layer_stride elements, [T][B, d]) and context gradients (dctx, likewise) over the steps:
 *   out[b, j, l, 0, h, :] = sum_s ds_s[b,h,j] qc_s[b,h,:],  out[b, j, l, 1, h, :] = sum_s pd_s[b,h,j] dctx_s[b,h,:]
 * with out [B, Ls, n_layers, 2, d] in the compute dtype: the layout of mssr_decode_plan.mem_kv, so its rows feed the
 * K|V projections' backward directly.  Sentences below b0 are skipped (rows of a mixed sweep without gradient; their
 * rows of out are not written).  Replaces the autograd backward of the cached context attention's
 * matmul(attn, value) / matmul(query, key^T) over the step loop (onmt/modules/multi_headed_attn.py:187-205 under
 * onmt/train_and_translate.py:202-231).  Head width <= 64. */
int mssr_ctx_kv_grad(const float* ds_ctx, const void* qc, int64_t qc_layer_stride, const void* dctx,
                     int64_t dctx_layer_stride, void* out, int n_layers, int T, int B, int b0, int heads, int dh, int Ls,
                     int dtype, void* stream);

/* ------------------------------------------------ teacher-forced decoder-stack engine ------
 * Replaces the Python layer loop of TransformerDecoder.forward with step = None
 * (onmt/decoders/transformer.py:289-337) -> n_layers x TransformerDecoderLayer._forward (:110-171) -> final
 * LayerNorm (:331), and its autograd backward: the whole target sequence at once (the reconstruction phase's decoder
 * pass, onmt/trainer_wgan_wh.py:632-664).  Rows are batch-first, R = B*T; the caller embeds the target into X[0],
 * projects the memory bank onto every layer's context keys | values with ONE product (mem_kv [B*Ls, n_layers*2d],
 * layer l at columns [l*2d, (l+1)*2d)) and owns every buffer.  `layers` (a HOST array) is the step-wise engine's
 * mssr_dec_layer: both engines read the same weight pack.  Dropout as in the encoder engine (regenerated from the
 * seed by the backward). */
typedef struct {
    int dtype;
    int B, T, Ls, d, heads, ff, n_layers;
    int save;                     /* 1: keep per-layer activations for mssr_decstack_bwd */
    int training;                 /* 1: dropout active */
    float p_drop, p_attn;
    uint64_t seed;
    const mssr_dec_layer* layers;
    const float *lnF_g, *lnF_b;   /* TransformerDecoder.layer_norm */
    const uint8_t* self_mask;     /* [B, T, T] nonzero = masked: causal | target padding (:132-149) */
    const uint8_t* src_mask;      /* [B, Ls]  nonzero = padding key of the memory bank, or NULL */
    void* X;                      /* [n_layers+1][R, d]; X[0] = embedded target rows (input) */
    void* out;                    /* [R, d] final LayerNorm output */
    const void* mem_kv;           /* [B*Ls, n_layers*2d] */
    void* probs;                  /* [B, heads, T, Ls] context-attention probabilities of the LAST layer, or NULL */
    void* qkv;                    /* [n_layers][R, 3d] (one slot when save = 0) */
    void* actD;                   /* [n_layers][8][R, d]: n1 | ctx_self | h1 | n2 | ctx_query | ctx_ctx | h2 | n3 */
    void* actF;                   /* [n_layers][R, ff] */
    float* stats;                 /* [3*n_layers+1][2][R]  mean | rstd of every LayerNorm */
    void* gemm_ws;
    int64_t gemm_ws_bytes;
    /* backward */
    const void* dout;             /* [R, d] gradient of `out` */
    void* dX;                     /* [R, d] gradient of X[0] (output) */
    void* dqkv;                   /* [n_layers][R, 3d] */
    void* dactD;                  /* [n_layers][7][R, d]: dn1 | d_so | dn2 | d_cq | d_co | dn3 | d_f2 (gradients of the outputs
                                   * of layer_norm_1, self final_linear, layer_norm_2, context linear_query, context
                                   * final_linear, feed_forward.layer_norm, w_2) */
    void* dactF;                  /* [n_layers][R, ff] gradient of the w_1 output */
    void* dctx;                   /* [R, d] scratch */
    void* dmem_kv;                /* [B*Ls, n_layers*2d] gradient of mem_kv (every column block written once) */
} mssr_decstack_plan;
int mssr_decstack_fwd(const mssr_decstack_plan* plan, void* stream);
int mssr_decstack_bwd(const mssr_decstack_plan* plan, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSSR_HIP_H */
