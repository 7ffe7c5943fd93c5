/*
 * lpm_hip.h -- C ABI of liblpm_hip.so: the MI355X (gfx950) hot path behind the
 * NetVladV1 / NetVladV2 model-registry API of pomonam/LearnablePoolingMethods.
 *
 * The reference is 100 % Python on TensorFlow 1.x and has no FFI of its own; each
 * entry point below replaces the TF op sub-graph cited next to it (file:line under
 * the reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless noted;
 *   - the caller owns every buffer (the library never allocates or frees);
 *   - tensors are contiguous row-major fp32 unless a leading dimension is passed;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and is
 *     asynchronous; nothing synchronises internally;
 *   - returns LPM_OK (0) or a negative LPM_ERR_*; lpm_last_error() (thread-local
 *     string) says why;
 *   - re-entrant.  Mutable state outside the caller's buffers: the thread-local error string, and ONE process-global
 *     measurement switch -- lpm_kernel_timing_enable(1) makes the K1 / K2 / assignment-tile / finalize launches of EVERY
 *     thread record start / stop events into the library's own table until lpm_kernel_timing_enable(0); enable it from one
 *     thread at a time and read the table (lpm_kernel_timing_read) after synchronising the streams (bench.py, tools/).
 *     Environment switches (LPM_*) are read once, at the first call that consults them.
 */
#ifndef LPM_HIP_H
#define LPM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPM_VERSION 100 /* major*10000 + minor*100 + patch */

typedef void* lpm_stream_t; /* hipStream_t */

enum {
    LPM_OK = 0,
    LPM_ERR_BADARG = -1,
    LPM_ERR_UNSUPPORTED_SHAPE = -2,
    LPM_ERR_WORKSPACE = -3,
    LPM_ERR_LAUNCH = -4
};

/* flags of lpm_vlad_aggregate_{fwd,bwd} */
enum {
    LPM_VLAD_SOFTMAX = 1,      /* `assign` holds logits; apply affine + softmax over K (NetVLAD)      */
    LPM_VLAD_RESIDUAL = 2,     /* subtract (sum_t a) * centres (NetVLAD, NetVladAttenCluster)           */
    LPM_VLAD_OUT_KMAJOR = 4,   /* lpm_vlad_finalize_*: descriptor laid out [B,K,D] instead of [B,D*K]   */
    LPM_VLAD_NRM_RAW = 8,      /* lpm_vlad_finalize2_fwd: leave `nrm` as the un-normalised sums U (no in-place write of the
                                  intra-normalised copy); lpm_vlad_aggregate_bwd_tiles: `nrm` holds U and the normalised
                                  descriptor is rebuilt as U * rsqrt(max(colsq, eps)) where it is read                     */
    LPM_VLAD_OUT_BF16 = 16,    /* lpm_vlad_finalize2_fwd: `out` is bf16 storage (d-major layout only)                   */
    LPM_VLAD_TILES_BF16 = 32,  /* lpm_vlad_aggregate_bwd_tiles: bf16 storage -- xr and the dU tiles are plain bf16 tiles, `assign`
                                  holds bf16 logits                                                                         */
    LPM_VLAD_NRM_BF16 = 64,    /* lpm_vlad_finalize2_fwd: `nrm` holds the un-normalised sums as bf16 (what lpm_vlad_aggregate_tiles3_fwd_bf16
                                  writes; with LPM_VLAD_NRM_RAW); implied by LPM_VLAD_TILES_BF16 in lpm_vlad_aggregate_bwd_tiles   */
    LPM_VLAD_RAW_KMAJOR = 128, /* lpm_vlad_aggregate_bwd_tiles: `dout` and `nrm` are k-major [B, K, D] and `nrm` holds the un-normalised
                                  sums lpm_vlad_aggregate_raw_kmajor_fwd stored (split-bf16, no-input-gradient form only)        */
    LPM_VLAD_DEBUG_FALLBACK = 256, /* lpm_vlad_aggregate_fused_fwd, tests only: one workgroup of every clip behaves as if its wait
                                  for the clip had timed out, so the follow-up finalize pass runs for every clip             */
    LPM_VLAD_WIDE_ALL = 512,   /* lpm_vlad_aggregate_kmajor_scaled_fwd (K = 256), tests / A-B: every clip runs as wide items ...    */
    LPM_VLAD_WIDE_NONE = 1024  /* ... / no clip does (default: whole rounds of clips wide, the rest as 128 x 128 items)          */
};

int lpm_version(void);
const char* lpm_last_error(void);

/* Kernel timing for bench.py: while enabled, the K1 (video stream, tag 1) and K2 (tag 2) forward launches are issued with
 * hipExtLaunchKernelGGL and a start/stop HIP event pair on their own stream, i.e. the elapsed time is the kernel's duration
 * as rocprofv3 --kernel-trace reports it.  lpm_kernel_timing_read synchronises, returns up to `max` durations (ms) of `tag`
 * in launch order and releases their events. */
void lpm_kernel_timing_enable(int on);
/* K1's hand-scheduled forward forms (csrc/assign_flat.hip) can be switched off for the process: bit 0 the flat 96-row forms, bit 1 the
 * 160 x 512 plain-bf16 form; lpm_assign_gemm_tiles_fwd[_bf16] then take the tile-GEMM form.  Returns the previous mask.  The Python host
 * checks every form against the tile-GEMM form once per process (ops._k1_selfcheck) and uses this if they ever disagree. */
int lpm_k1_forms_disable(int mask);
int lpm_kernel_timing_read(int tag, float* ms, int max);

/* Measurement only: the shader clock over time.  lpm_clock_sampler launches ONE wave that stores, every `period_ticks` ticks of the
 * constant 100 MHz counter, the pair (constant counter, shader-clock counter) into out[2 i], out[2 i + 1] (i < n; n * period_ticks / 100
 * microseconds in all) -- on a stream of its own, before the work to be observed.  lpm_clock_marker stores one such pair into slot
 * `slot` of `out` from the stream it is launched on (a time stamp in the same time base).  tools/clock_trace.py. */
int lpm_clock_sampler(uint64_t* out, int n, int period_ticks, lpm_stream_t stream);
int lpm_clock_marker(uint64_t* out, int slot, lpm_stream_t stream);

/* Input normalisation of the training step (tf.nn.l2_normalize(model_input_raw, 2), train.py:262-264):
 * y[r,:] = x[r,:] * rsqrt(max(sum x[r,:]^2, 1e-12)) for `rows` rows of F floats (F %% 4 == 0, F <= 2048).  y may alias x. */
int lpm_l2_normalize_rows(const float* x, int64_t rows, int F, float* y, lpm_stream_t stream);

/* The frame reader's tail folded into the same pass (readers.py:176-193, utils.py:28-43, train.py:262-264): quantised uint8
 * frames q [B, max_frames, F] -> Dequantize (q * range/255 + range/512 + min) -> frames >= num_frames[b] set to 0 (the
 * reader pads after dequantising) -> per-frame L2 normalisation -> y fp32 [B, max_frames, F].  F %% 4 == 0, F <= 2048. */
int lpm_dequantize_l2_normalize(const unsigned char* q, const int32_t* num_frames, int B, int max_frames, int F,
                                float max_quantized_value, float min_quantized_value, float* y, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a2 + a3: SampleUniformFrames + input_bn
 *   replaces model_utils.py:101-122 (gather_nd) + frame_level_models.py:2265-2271 (slim.batch_norm).
 * raw [B, max_frames, F]; num_frames [B] int32; S sampled frames per clip; rows = B*S.
 * Training: lpm_frame_stats writes per-block column partials (sum, sum of squares) of the gathered
 * rows into `partial` (lpm_frame_stats_workspace_bytes); lpm_bn_fold turns them into the folded
 * affine and updates the moving statistics; lpm_frame_apply writes y = gather(raw)*scale + shift.
 * ------------------------------------------------------------------------------------------- */
size_t lpm_frame_stats_workspace_bytes(int B, int S, int F);
int lpm_frame_stats(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                    float* partial, lpm_stream_t stream);
int lpm_frame_apply(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                    const float* scale, const float* shift, float* y, lpm_stream_t stream);
/* lpm_frame_apply that also emits the split-bf16 tile copies K2 reads (see lpm_split_frames): columns [0,Dv) -> xt_video,
 * [Dv, Dv+Da) -> xt_audio (either may be NULL); Dv + Da == F. */
int lpm_frame_apply_tiles(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                          const float* scale, const float* shift, float* y, void* xt_video, int Dv, void* xt_audio, int Da,
                          lpm_stream_t stream);
/* ... with the two column blocks as TWO contiguous matrices, y_video [B*S, Dv] and y_audio [B*S, Da] (round 6: NetVladV2, whose frame
 * encoders and aggregations want whole rows -- a column slice of one [B*S, F] matrix cost a contiguous copy per stream, a second
 * lpm_split_frames per stream and a concatenation of the two gradients), and lpm_frame_bn_bwd taking the gradient the same way. */
int lpm_frame_apply_tiles_split(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S, const float* scale,
                                const float* shift, float* y_video, float* y_audio, void* xt_video, int Dv, void* xt_audio, int Da,
                                lpm_stream_t stream);
int lpm_frame_bn_bwd_split(const float* dy_video, int64_t ldv, const float* dy_audio, int64_t lda, int Dv, const float* raw,
                           const int32_t* num_frames, int B, int max_frames, int F, int S, const float* mean, const float* var, float eps,
                           float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, lpm_stream_t stream);
/* ... and the split-bf16 ROW tiles K1 reads (the layout of lpm_split_rows_tiles, lpm_row_tiles_bytes(B, S, D) each; xr_* may be
 * NULL), so that no pass over the fp32 matrix is needed between a3 and K1. */
int lpm_frame_apply_tiles2(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                           const float* scale, const float* shift, float* y, void* xt_video, void* xr_video, int Dv,
                           void* xt_audio, void* xr_audio, int Da, lpm_stream_t stream);
int lpm_frame_stats_nblk(int B, int S);   /* rows of `partial` lpm_frame_stats writes (for lpm_bn_fold) */
/* backward of input_bn's affine parameters only (the frames are data, never a trainable tensor, so no
 * gradient w.r.t. raw is produced): dgamma = sum dy*xhat, dbeta = sum dy over the gathered rows.
 * dy [B*S, F] with row stride lddy; mean/var = the batch statistics lpm_bn_fold returned.
 * workspace: lpm_frame_stats_workspace_bytes(B,S,F). */
int lpm_frame_bn_bwd(const float* dy, int64_t lddy, const float* raw, const int32_t* num_frames, int B,
                     int max_frames, int F, int S, const float* mean, const float* var, float eps, float* dgamma,
                     float* dbeta, void* workspace, size_t workspace_bytes, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training-mode batch-norm statistics -> folded affine
 *   replaces the statistics half of slim.batch_norm (frame_level_models.py:2266,2784).
 * partial [nblk, 2, C] (sum, sumsq over `rows` rows in total).  Outputs: mean, var (biased),
 * scale = gamma*rsqrt(var+eps), shift = beta - mean*scale.  If moving_mean != NULL:
 * moving = moving*decay + batch*(1-decay) with the unbiased variance (fused-BN semantics).
 * ------------------------------------------------------------------------------------------- */
int lpm_bn_fold(const float* partial, int nblk, int C, int64_t rows, const float* gamma, const float* beta,
                float eps, float decay, float* mean, float* var, float* scale, float* shift,
                float* moving_mean, float* moving_var, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K1: soft-assignment GEMM  logits[M,K] = x[M,D] (row stride ldx) . w[D,K]
 *   replaces tf.matmul at frame_level_models.py:2781 plus the batch-statistics reduction of the
 *   cluster_bn that follows (:2783-2789): the epilogue writes per-block column partials
 *   (sum, sumsq) into `partial` [lpm_assign_gemm_nblk(M), 2, K] for lpm_bn_fold.
 * precision: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 1 = split-bf16 (3 bf16 MFMAs, ~fp32).
 * ------------------------------------------------------------------------------------------- */
int lpm_assign_gemm_nblk(int M);
int lpm_assign_gemm_fwd(const float* x, int64_t ldx, const float* w, int M, int D, int K, int precision,
                        float* logits, float* partial, lpm_stream_t stream);

/* K1 and its backward on the bf16 matrix pipe (csrc/tile_gemm.hip, csrc/tile_gemm.h): split-bf16 operands (a = hi + lo
 * bf16 planes; ah*bh + ah*bl + al*bh accumulated in fp32, ~5e-6 relative error) stored as MFMA-fragment tiles:
 *   tile[plane][lane][e] = M[outer = 32*tile + (lane&31)][reduction = 16*step + 8*(lane>>5) + e],  16 bytes per lane.
 *   row tiles    [b][mt][cs]  of a [B*T, C] matrix, per clip, mt < 2*ceil(T/64) (rows >= T zero)     lpm_split_rows_tiles
 *   weight tiles [rs][nt]     of M[R, N] (reduction R); `transposed`: the source is stored [N, R]     lpm_split_weight_tiles
 *   frame tiles  [b][s][ct]   reduction along the frame axis                                          lpm_split_frames
 * Needs D %% 32 == 0, K %% 32 == 0, K <= 512 (lpm_assign_gemm_tiles_supported).
 *   fwd:    logits[B*T,K] = x . w  + per-workgroup column statistics `partial` [lpm_assign_gemm_tiles_nblk(B,T), 2, K]
 *           (replaces tf.matmul + cluster_bn statistics, frame_level_models.py:2781-2789)
 *   bwd_dx: dx[B*T, D] (row stride lddx) += dlogits . w^T     dlr = row tiles of dlogits, wtt = weight tiles of w^T
 *   bwd_dw: dw[D,K] = x^T . dlogits                           xt / dlt = frame tiles of x and dlogits */
size_t lpm_row_tiles_bytes(int B, int T, int C);
size_t lpm_weight_tiles_bytes(int R, int N);
int lpm_assign_gemm_tiles_supported(int T, int D, int K);
int lpm_assign_gemm_tiles_nblk(int B, int T);
int lpm_split_rows_tiles(const float* x, int64_t ldx, int B, int T, int C, void* out, lpm_stream_t stream);
int lpm_split_weight_tiles(const float* w, int R, int N, int transposed, void* wt, lpm_stream_t stream);
int lpm_assign_gemm_tiles_fwd(const void* xr, const void* wt, int B, int T, int D, int K, float* logits, float* partial,
                              lpm_stream_t stream);
int lpm_assign_gemm_tiles_bwd_dx(const void* dlr, const void* wtt, int B, int T, int D, int K, float* dx, int64_t lddx,
                                 lpm_stream_t stream);
/* y[M,N] (row stride ldo) = x . w for an encoder dense layer (transformer_utils.py:559-561,583,701-711) on the same tile GEMM:
 * xr = row tiles of x [M,Kd] (lpm_split_rows_tiles with B = 1, T = M), wt = weight tiles of w [Kd,N]; Kd %% 16 == 0, N %% 32 == 0.
 * form 0 / 2: 128-row workgroups where the shape allows, 1: 64-row workgroups. */
int lpm_dense_tiles_fwd(const void* xr, const void* wt, int M, int Kd, int N, float* y, int64_t ldo, int form, lpm_stream_t stream);
/* FeedForwardNetwork's first dense layer (transformer_utils.py:701-711) and its backward on the 256-row tile GEMM with operand-image
 * epilogues: the [M, 4F] tensor between the two dense layers exists only as the split-bf16 image the GEMMs downstream read.
 *   lpm_dense_tiles_supported    : M %% 256 == 0, Kd %% 16 == 0, Kd >= 256, N %% 256 == 0
 *   lpm_dense_tiles_act_image_fwd: out3 [M,3N] bf16 = [hi|lo|hi] of relu(x . w + bias)   (xr row tiles of x [M,Kd], wt weight tiles of w)
 *   lpm_dense_tiles_relu_bwd_image: g = (dy . w^T) * [act > 0] -> out3 [M,3N] = [hi|hi|lo] of g, dbias [N] = column sums of g
 *                                  (dyr row tiles of dy [M,Kd]; wtt = lpm_split_weight_tiles(w [N,Kd], transposed = 1); act3 = the forward's image)
 *   lpm_image_row_tiles          : an operand image [M,3K] (order 0 [hi|lo|hi], 1 [hi|hi|lo]) -> row tiles (lpm_row_tiles_bytes(1, M, K)) */
int lpm_dense_tiles_supported(int M, int Kd, int N);
int lpm_dense_tiles_act_image_fwd(const void* xr, const void* wt, const float* bias, int M, int Kd, int N, void* out3, lpm_stream_t stream);
size_t lpm_dense_tiles_relu_bwd_workspace_bytes(int M, int N);
int lpm_dense_tiles_relu_bwd_image(const void* dyr, const void* wtt, const void* act3, int M, int Kd, int N, void* out3, float* dbias,
                                   void* workspace, size_t workspace_bytes, lpm_stream_t stream);
int lpm_image_row_tiles(const void* x3, int M, int K, int order, void* out, lpm_stream_t stream);
/* dw[N1,N2] = x^T . dy for a skinny batch R (multiple of 16): xt / dyt = weight tiles of x [R,N1] and dy [R,N2]
 * (lpm_split_weight_tiles, not transposed).  The hidden1 weight gradient (frame_level_models.py:2314-2319 backward). */
int lpm_skinny_weight_grad_tiles(const void* xt, const void* dyt, int R, int N1, int N2, float* dw, lpm_stream_t stream);
size_t lpm_assign_gemm_tiles_bwd_dw_workspace_bytes(int B, int T, int D, int K);
int lpm_assign_gemm_tiles_bwd_dw(const void* xt, const void* dlt, int B, int T, int D, int K, float* dw, void* workspace,
                                 size_t workspace_bytes, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2: fused [BN-affine -> softmax] -> residual aggregation -> intra-normalisation
 *   replaces frame_level_models.py:2798-2819 (NetVLAD), :2856-2872 (LightVLAD, no RESIDUAL flag),
 *   video_pooling_modules.py:1646-1655 (NetVladAttenCluster, no SOFTMAX flag).
 * assign [B*T, K]: logits (SOFTMAX: a = softmax(assign*scale + shift), scale/shift [K], either may be
 *   NULL = 1 / 0) or the similarities themselves.  x [B*T, D] with row stride ldx.  centres [D,K].
 * Outputs: nrm [B, D, K] (d-major, each cluster column L2-normalised over D),
 *          asum [B,K] = sum_t a, colsq [B,K] = column square norm before normalisation,
 *          csq [B,K] = column square norm after it.
 * lpm_vlad_finalize_fwd applies the global L2 (frame_level_models.py:2821-2822):
 *   gsq[b] = sum_k csq[b,k]; out = nrm * rsqrt(max(gsq,1e-12)) laid out [B, D*K] (d-major, the
 *   reference layout) or [B,K,D] with LPM_VLAD_OUT_KMAJOR (App. C5 view for the V1 encoders).
 * ------------------------------------------------------------------------------------------- */
int lpm_vlad_aggregate_fwd(const float* assign, const float* scale, const float* shift, const float* x,
                           int64_t ldx, const float* centres, int B, int T, int D, int K, int flags,
                           float* nrm, float* asum, float* colsq, float* csq, lpm_stream_t stream);
int lpm_vlad_finalize_fwd(const float* nrm, const float* csq, int B, int D, int K, int flags, float* out,
                          float* gsq, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2 on the bf16 matrix pipe (split-bf16, "bf16x3": each fp32 operand is carried as hi/lo bf16 planes and a
 * product is accumulated as ah*xh + ah*xl + al*xh in fp32 -- ~1e-5 relative error, inside the 1e-3 parity bar,
 * at 5.3x less matrix-pipe time than the exact-fp32 MFMA).  Same reference lines and outputs as
 * lpm_vlad_aggregate_fwd; operands arrive in MFMA-fragment ("tile") order produced by:
 *   lpm_split_frames : x [B*T, D] (row stride ldx) -> xt (lpm_xt_bytes(B,T,D) bytes)
 *   lpm_assign_tiles : assign [B*T, K] (logits with LPM_VLAD_SOFTMAX: softmax(assign*scale+shift); else the
 *                      similarities themselves) -> at (lpm_at_bytes(B,T,K) bytes)
 * Layouts (16-byte units): xt[b][s][d/32][plane][lane], at[b][k/32][s][plane][lane]; lane l holds the 8 frames
 * t = 16 s + 8 (l>>5) + 0..7 of column d (or cluster k) = 32*tile + (l&31); plane 0 = hi, 1 = lo.
 * ------------------------------------------------------------------------------------------- */
size_t lpm_xt_bytes(int B, int T, int D);
size_t lpm_at_bytes(int B, int T, int K);
int lpm_split_frames(const float* x, int64_t ldx, int B, int T, int D, void* xt, lpm_stream_t stream);
int lpm_assign_tiles(const float* assign, const float* scale, const float* shift, int B, int T, int K, int flags,
                     void* at, lpm_stream_t stream);
int lpm_vlad_aggregate_tiles_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K,
                                 int flags, float* nrm, float* asum, float* colsq, float* csq, lpm_stream_t stream);

/* K2, LDS-shared form (csrc/vlad_tiles3.hip): 512-thread workgroups own 128 clusters x 128 columns of a clip, tiles
 * arrive by LDS-DMA and are shared by 8 waves (2.8x less L2 -> CU traffic than the streaming form).  Needs
 * D %% 128 == 0 and K %% 128 == 0 (lpm_vlad_tiles3_supported).  It writes the UN-normalised residual sums into nrm and
 * per-column-slab partial square norms colsq_part [B, D/128, K]; lpm_vlad_finalize2_fwd then applies BOTH normalisations:
 * nrm <- intra-normalised (in place, what lpm_vlad_aggregate_bwd reads), out = globally normalised in the flagged
 * layout, plus colsq, csq [B,K] and gsq [B].  P = number of partial slabs (D/128). */
int lpm_vlad_tiles3_supported(int D, int K);
int lpm_vlad_aggregate_tiles3_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K,
                                  int flags, float* nrm, float* asum, float* colsq_part, lpm_stream_t stream);
int lpm_vlad_finalize2_fwd(float* nrm, const float* colsq_part, int P, int B, int D, int K, int flags, float* out,
                           float* colsq, float* csq, float* gsq, lpm_stream_t stream);
/* ... with the clips' descriptors out_batch_stride elements apart in `out` (>= D * K, a multiple of 4; fp32, d-major or the scalar
 * k-major form): a column slot of the streams' joined [B, total] buffer -- tf.concat(..., 1) at frame_level_models.py:2309 without a copy */
int lpm_vlad_finalize2_fwd_ld(float* nrm, const float* colsq_part, int P, int B, int D, int K, int flags, float* out,
                              int64_t out_batch_stride, float* colsq, float* csq, float* gsq, lpm_stream_t stream);

/* K2 for a consumer that applies the normalisation itself (the NetVladV1 cluster encoders, App. C5: tokens = clusters):
 * lpm_vlad_aggregate_raw_kmajor_fwd stores the UN-normalised residual sums k-major [B, K, D] -- once, straight from the accumulators
 * -- plus asum and the partial norms colsq_part [B, D/128, K]; lpm_vlad_row_scales turns those into scale [B, K] = 1 / (n_k sqrt(g))
 * and colsq, csq [B, K], gsq [B], so that descriptor[b, k, :] = raw[b, k, :] * scale[b, k] (frame_level_models.py:2819-2822 as one
 * factor per row).  No finalize pass: the [B, D, K]-sized tensor is written once and never re-read by the pooling.  Consumers:
 * lpm_split_rows_scaled (operand image of the q/k/v GEMM), lpm_layer_norm_act_fwd_rs (the encoder's residual); the backward is
 * lpm_vlad_aggregate_bwd_tiles with LPM_VLAD_RAW_KMAJOR (dout and the sums both k-major: no transposes). */
int lpm_vlad_aggregate_raw_kmajor_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags,
                                      float* raw_kmajor, float* asum, float* colsq_part, lpm_stream_t stream);
int lpm_vlad_row_scales(const float* colsq_part, int P, int B, int K, float* scale, float* colsq, float* csq, float* gsq,
                        lpm_stream_t stream);
/* The same function on CLIP-WIDE items (vlad_clip.hip, round 4; K = 256): a workgroup owns all 256 clusters x a third of a clip's
 * columns (11 / 11 / 10 column tiles at D = 1024), so the frame tiles pass through the LDS-DMA path once and the assignment tiles
 * three times -- 168 MB per launch at cfg-2 instead of 389 MB, on the path that bounds the 128 x 128 form.  Outputs as
 * lpm_vlad_aggregate_raw_kmajor_fwd, except colsq_part [B, P, K] with P = lpm_vlad_clip_slabs(D, K) (0: shape not supported);
 * lpm_vlad_row_scales(colsq_part, P, ...) follows.  Reference: frame_level_models.py:2803-2817. */
int lpm_vlad_clip_slabs(int D, int K);
int lpm_vlad_aggregate_clip_kmajor_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags,
                                       float* raw_kmajor, float* asum, float* colsq_part, lpm_stream_t stream);
/* ... the same kernel leaving the un-normalised sums d-major [B, D, K] (NetVladV2's lazily normalised descriptor; lpm_vlad_row_scales follows) */
int lpm_vlad_aggregate_clip_dmajor_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags,
                                       float* raw_dmajor, float* asum, float* colsq_part, lpm_stream_t stream);
/* K2 + row scales in ONE launch (vlad_kmajor.hip; frame_level_models.py:2803-2822 for the lazily normalised k-major descriptor):
 * raw_kmajor [B,K,D] un-normalised residual sums, scale [B,K] with descriptor[b,k,:] = raw[b,k,:] * scale[b,k], and asum / colsq /
 * csq [B,K], gsq [B] for the backward.  K = 256: "wide" workgroups (all clusters x 128 columns) for whole rounds of clips, 128 x 128
 * items for the rest; other K (multiples of 128, <= 1024): 128 x 128 items.  workspace: lpm_vlad_kmajor_workspace_bytes(B, D, K). */
size_t lpm_vlad_kmajor_workspace_bytes(int B, int D, int K);
int lpm_vlad_aggregate_kmajor_scaled_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags,
                                         float* raw_kmajor, float* scale, float* asum, float* colsq, float* csq, float* gsq,
                                         void* workspace, size_t workspace_bytes, lpm_stream_t stream);
/* ... and with the SOFTMAX inside the aggregation kernel: frame_level_models.py:2798-2822 as one launch (+ a small row-statistics launch
 * before it and lpm_vlad_row_scales after it).  logits [B*T, K] fp32 = K1's output, scale / shift [K] = cluster_bn folded (either may
 * be NULL; shift = the bias without batch norm); per 16-frame step the workgroup's raw logits arrive by LDS-DMA next to the frame tiles
 * and are turned into split-bf16 A fragments in LDS -- no assignment tensor or tile copy exists anywhere.  flags: LPM_VLAD_SOFTMAX
 * (required) | LPM_VLAD_RESIDUAL.  stats: lpm_vlad_smx_stats_bytes(B, T) bytes of scratch.  K in {128, 256, 512}, 33 <= T <= 4096. */
size_t lpm_vlad_smx_stats_bytes(int B, int T);
int lpm_vlad_smx_supported(int T, int D, int K);
int lpm_vlad_aggregate_raw_kmajor_smx_fwd(const float* logits, const float* scale, const float* shift, const void* xt,
                                          const float* centres, int B, int T, int D, int K, int flags, float* raw_kmajor,
                                          float* asum, float* colsq_part, float* stats, lpm_stream_t stream);
int lpm_split_rows_scaled(const float* x, int64_t ldx, int64_t M, int K, const float* row_scale, void* out3, lpm_stream_t stream);
int lpm_layer_norm_act_fwd_rs(const float* a, const float* bias, int relu, const float* r, const float* r_scale, const float* gamma,
                              const float* beta, int B, int L, int F, float eps, float* y, int64_t y_batch_stride, float* z,
                              float* stats, void* workspace, size_t workspace_bytes, lpm_stream_t stream);
/* The same forward with y ALSO written as the split-bf16 activation image y3 [B*L, 3F] bf16 = [hi | lo | hi] (lpm_split_rows order 0) of
 * the dense layer that reads it next (FeedForwardNetwork's first layer after the attention's layer norm, transformer_utils.py:405-411,
 * 701-711): the split pass over y is not run.  r_scale may be NULL. */
int lpm_layer_norm_act_image_fwd(const float* a, const float* bias, int relu, const float* r, const float* r_scale, const float* gamma,
                                 const float* beta, int B, int L, int F, float eps, float* y, int64_t y_batch_stride, void* y3, float* z,
                                 float* stats, void* workspace, size_t workspace_bytes, lpm_stream_t stream);

/* K2 with the finalize pass fused in (frame_level_models.py:2803-2822 as ONE kernel): the same workgroups, but each waits for
 * the other workgroups of its clip (per-clip arrival counter, bounded), forms 1/n_k and the clip's 1/sqrt(g) from the partial
 * norms all of them published, and stores its tile of the NORMALISED descriptor straight from the accumulators into `out`
 * ([B, D*K], or [B,K,D] with LPM_VLAD_OUT_KMAJOR); colsq, csq [B,K] and gsq [B] as from lpm_vlad_finalize2_fwd.  `nrm` receives
 * the un-normalised sums only with LPM_VLAD_NRM_RAW (the form lpm_vlad_aggregate_bwd_tiles reads); it must be a valid
 * [B, D, K] buffer either way (scratch of the follow-up pass that finishes a clip whose workgroups were not resident
 * together -- results never depend on dispatch order).  workspace: lpm_vlad_fused_workspace_bytes(B, D, K), cleared by the
 * call itself.  Needs lpm_vlad_fused_supported(D, K): the tiles3 shapes with K <= 512. */
int lpm_vlad_fused_supported(int D, int K);
size_t lpm_vlad_fused_workspace_bytes(int B, int D, int K);
int lpm_vlad_aggregate_fused_fwd(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags,
                                 float* nrm, float* out, float* asum, float* colsq, float* csq, float* gsq, void* workspace,
                                 size_t workspace_bytes, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a9: the VLAD -> hidden projection (frame_level_models.py:2314-2319, tf.matmul(vlad, hidden1_weights)) as weight-stream kernels
 * (csrc/proj_gemm.hip).  M <= 128 clips against the [Kd, N] fp32 weight (0.55 GB at cfg-2, 2.2 GB at cfg-5), read ONCE per pass from
 * its fp32 master copy and split into bf16 hi / lo planes in registers (3 MFMAs per product, fp32 accumulation, ~5e-6):
 *   lpm_proj_fwd : y[M, N]  = x[M, Kd] . W            (split-K partial sums in `workspace` + a fixed-order reduce)
 *   lpm_proj_dx  : dx[M, Kd] = dy[M, N] . W^T          (dyt = lpm_split_rows_tiles(dy, N, 1, M, N): row tiles of the gradient)
 * ldx / lddx: row strides of x / dx in floats (>= Kd).  A stride that is a multiple of a large power of two -- cfg-2's Kd = 270 336 =
 * 2^13 x 33 floats -- puts the M row pieces a workgroup needs per step into the same few L2 / memory channels; callers that own the
 * buffer pad it (ops.DescriptorSlots).  Needs lpm_proj_supported(M, Kd, N): M <= 128, N %% 512 == 0, Kd %% 16 == 0.  The weight
 * gradient dW = x^T dy is lpm_skinny_weight_grad_tiles. */
int lpm_proj_supported(int M, int64_t Kd, int N);
size_t lpm_proj_fwd_workspace_bytes(int M, int64_t Kd, int N);
int lpm_proj_fwd(const float* x, int64_t ldx, const float* W, int M, int64_t Kd, int N, float* y, void* workspace,
                 size_t workspace_bytes, lpm_stream_t stream);
int lpm_proj_dx(const void* dyt, const float* W, int M, int64_t Kd, int N, float* dx, int64_t lddx, lpm_stream_t stream);
/* The projection of a LAZILY NORMALISED d-major descriptor (NetVladV2, frame_level_models.py:2437-2448; the model without the cluster
 * encoders): y = [x1 * scale | x2] . W, where x1 [M, n1a] (row stride ldx1) holds the un-normalised residual sums of the video stream
 * as K2 wrote them ([D, K] per clip), scale [M, ks] = lpm_vlad_row_scales' 1 / (n_k sqrt g) is applied to column c as scale[row][c %% ks]
 * where the operand is read (frame_level_models.py:2819-2822 / video_pooling_modules.py:1655-1658 as a factor per (clip, cluster)), and
 * x2 [M, Kd - n1a] (row stride ldx2; NULL when n1a == Kd) is the audio stream's descriptor as it is: neither the finalize pass of the
 * pooling nor tf.concat (:2445) runs.  n1a a multiple of 32 and of ks, ks a multiple of 4; otherwise lpm_proj_fwd's conditions.
 * x1_bf16: x1 is stored as bf16 (the bf16-storage configuration's un-normalised sums; ldx1 in elements, a multiple of 4).
 * lpm_split_weight_tiles_parts writes the weight tiles (lpm_weight_tiles_bytes(R, N)) of the same virtual matrix: the X factor of the
 * hidden1 weight gradient (lpm_factored_clip_adam*, lpm_skinny_weight_grad_tiles). */
int lpm_proj_fwd_parts(const void* x1, int64_t ldx1, int64_t n1a, int x1_bf16, const float* scale, int ks, const float* x2, int64_t ldx2,
                       const float* W, int M, int64_t Kd, int N, float* y, void* workspace, size_t workspace_bytes, lpm_stream_t stream);
int lpm_split_weight_tiles_parts(const void* x1, int64_t ld1, int64_t n1a, int x1_bf16, const float* scale, int ks, const float* x2,
                                 int64_t ld2, int R, int64_t N, void* wt, lpm_stream_t stream);
/* bf16 storage (BASELINE configs[4]) with a bf16 COMPUTE COPY of hidden1_weights (frame_level_models.py:2309-2319; SURVEY section 7: "keep
 * master fp32 + bf16 compute copy"): W16 [Kd, N] = bf16(W), kept beside the fp32 master by lpm_factored_clip_adam_copy.  The forward (x1 the
 * bf16-stored sums, as lpm_proj_fwd_parts with x1_bf16 = 1) and the input gradient (dyt as for lpm_proj_dx; N a multiple of 64) stream
 * 2 bytes per weight instead of 4 and run one bf16 MFMA per product: x = bf16(x1 * scale), dy = its hi plane -- the arithmetic of this
 * configuration's other operands. */
int lpm_proj_fwd_parts_w16(const void* x1, int64_t ldx1, int64_t n1a, const float* scale, int ks, const float* x2, int64_t ldx2,
                           const void* W16, int M, int64_t Kd, int N, float* y, void* workspace, size_t workspace_bytes, lpm_stream_t stream);
int lpm_proj_dx_w16(const void* dyt, const void* W16, int M, int64_t Kd, int N, float* dx, int64_t lddx, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * bf16 storage (BASELINE configs[4]: "Gated NetVLAD K=512 + MoE-4, 300x1152 bf16"): the tensors SURVEY 8(d) counts in the
 * algorithmic bytes of K1 / K2 -- frames, logits / assignment, descriptor -- live in HBM as bf16; every product is ONE bf16 MFMA
 * with fp32 accumulation; batch statistics, norms and all gradients stay fp32.  Operand tiles are plain bf16 tiles: the layouts of
 * the split form with ONE 1 KB plane per (tile, step), and every clip padded with zero frames to whole 64-frame blocks
 * (lpm_frame_steps_bf16(T) = 4 ceil(T / 64) frame steps, 2 ceil(T / 64) row tiles).  The reference computes in fp32 throughout
 * (frame_level_models.py:2765-2824); bf16 is the build's choice for this configuration and carries a bf16 tolerance.
 *   lpm_frame_apply_tiles_bf16 : a2 + a3 -> frame tiles xt_* (K2's operand) AND row tiles xr_* (K1's operand) of both streams in one
 *                                pass; y (fp32 [B*S, F]) optional (NULL: not written).  Each buffer: lpm_frame_tiles_bf16_bytes.
 *   lpm_split_weight_tiles_bf16 / lpm_split_frames_bf16 : fp32 matrix -> plain bf16 weight / frame tiles (half the split form's size)
 *   lpm_assign_gemm_tiles_fwd_bf16 : K1; logits stored as bf16 [B*T, K], BN partial statistics from the fp32 accumulators
 *   lpm_assign_tiles_bf16          : bf16 logits -> [affine -> softmax] -> plain bf16 assignment tiles
 *   lpm_vlad_aggregate_tiles3_fwd_bf16 : K2 on those tiles; `nrm` receives the un-normalised sums AS bf16 [B, D, K], the partial
 *                                norms come from the fp32 accumulators; lpm_vlad_finalize2_fwd with LPM_VLAD_NRM_RAW |
 *                                LPM_VLAD_NRM_BF16 | LPM_VLAD_OUT_BF16 then writes the bf16 descriptor [B, D*K]
 *   lpm_assign_gemm_tiles_bwd_dw_bf16, lpm_vlad_aggregate_bwd_tiles with LPM_VLAD_TILES_BF16 : the backward on plain bf16 tiles */
size_t lpm_frame_tiles_bf16_bytes(int B, int S, int D);
int lpm_frame_steps_bf16(int T);
int lpm_frame_apply_tiles_bf16(const float* raw, const int32_t* num_frames, int B, int max_frames, int F, int S,
                               const float* scale, const float* shift, float* y, void* xt_video, void* xr_video, int Dv,
                               void* xt_audio, void* xr_audio, int Da, lpm_stream_t stream);
int lpm_split_weight_tiles_bf16(const float* w, int R, int N, int transposed, void* wt, lpm_stream_t stream);
int lpm_split_frames_bf16(const float* x, int64_t ldx, int B, int T, int D, void* xt, lpm_stream_t stream);
int lpm_assign_gemm_tiles_fwd_bf16(const void* xr, const void* wt, int B, int T, int D, int K, void* logits_bf16, float* partial,
                                   lpm_stream_t stream);
int lpm_assign_gemm_tiles_bwd_dw_bf16(const void* xt, const void* dlt, int B, int T, int D, int K, float* dW, void* workspace,
                                      size_t workspace_bytes, lpm_stream_t stream);
int lpm_assign_tiles_bf16(const void* assign_bf16, const float* scale, const float* shift, int B, int T, int K, int flags, void* at,
                          lpm_stream_t stream);
int lpm_vlad_aggregate_tiles3_fwd_bf16(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags,
                                       float* nrm, float* asum, float* colsq_part, lpm_stream_t stream);
/* K2 for bf16 storage on CLIP-WIDE items (csrc/vlad_clip16.hip, round 6): the contract of lpm_vlad_aggregate_tiles3_fwd_bf16, but a
 * 512-thread workgroup owns 256 clusters x a third of a clip's columns (6 workgroups per clip at K = 512, D = 1024), a wave a 2 x 6
 * register tile, and colsq_part is [B, P, K] with P = lpm_vlad_clip16_slabs(D, K) column slabs (0: shape not supported -- K %% 256 == 0,
 * D %% 32 == 0, D >= 384).  frame_level_models.py:2803-2817. */
int lpm_vlad_clip16_slabs(int D, int K);
int lpm_vlad_aggregate_clip_fwd_bf16(const void* at, const void* xt, const float* centres, int B, int T, int D, int K, int flags,
                                     void* nrm_bf16, float* asum, float* colsq_part, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3: backward of K2 (TF autodiff of the same lines; formulas SURVEY.md App. F.1-F.3).
 * dout: gradient w.r.t. `out` in the layout given by flags.  Saved from forward: nrm, asum, colsq, csq, gsq.
 * Outputs: dassign [B*T,K] = gradient w.r.t. the affine-transformed logits (SOFTMAX) or w.r.t. the
 *   similarities; dx [B*T, D] row stride lddx, ACCUMULATED (+=) when accumulate_dx != 0;
 *   dcentres [D,K] (overwritten; NULL without RESIDUAL).
 * workspace: lpm_vlad_bwd_workspace_bytes(B,D,K) bytes of scratch.
 * ------------------------------------------------------------------------------------------- */
size_t lpm_vlad_bwd_workspace_bytes(int B, int D, int K);
int lpm_vlad_aggregate_bwd(const float* dout, const float* nrm, const float* asum, const float* colsq,
                           const float* csq, const float* gsq, const float* assign, const float* scale,
                           const float* shift, const float* x, int64_t ldx, const float* centres, int B, int T,
                           int D, int K, int flags, float* dassign, float* dx, int64_t lddx, int accumulate_dx,
                           float* dcentres, void* workspace, size_t workspace_bytes, lpm_stream_t stream);

/* K3 on the bf16 matrix pipe (tile form; needs D %% 32 == 0, K %% 32 == 0, K <= 512).  dU = u dO - v N is written once per
 * clip as split-bf16 fragment tiles and both GEMMs of the backward run through the tile GEMM (csrc/tile_gemm.h):
 *   lpm_vlad_aggregate_bwd_tiles:    dassign (softmax backward fused in the epilogue) and dcentres; xr = row tiles of x
 *                                    (lpm_split_rows_tiles).  Leaves the dU and assignment tiles in `workspace`.
 *                                    g0 (optional, [B, D]): "no input gradient" mode for frames that come straight out of
 *                                    input_bn (frame_level_models.py:2265-2277) -- writes g0[b][d] = sum_k dU_b[d,k] U_b[d,k]
 *                                    and dcentres = -sum_b asum_b dU_b (also without RESIDUAL), from which the caller forms
 *                                    input_bn's gamma / beta gradients in closed form; the dx operands are NOT produced
 *                                    (lpm_vlad_aggregate_bwd_tiles_dx must not follow).
 *   lpm_vlad_aggregate_bwd_tiles_dx: dx[B*T, D] (=, or += when accumulate_dx) sum_k a dU  [+ dl . w^T when dlr / wtt, the row
 *                                    tiles of the assignment GEMM's dlogits and the weight tiles of w^T, are given: the
 *                                    soft-assignment GEMM's input gradient rides in the same pass].  Same workspace. */
size_t lpm_vlad_bwd_tiles_workspace_bytes(int B, int T, int D, int K);
int lpm_vlad_aggregate_bwd_tiles(const float* dout, const float* nrm, const float* asum, const float* colsq,
                                 const float* csq, const float* gsq, const float* assign, const float* scale,
                                 const float* shift, const void* xr, const float* centres, int B, int T, int D, int K,
                                 int flags, float* dassign, float* dcentres, float* g0, void* workspace, size_t workspace_bytes,
                                 lpm_stream_t stream);
/* ... with the clips' gradients dout_batch_stride elements apart (>= D * K, a multiple of 4; d-major only): a column slice of the
 * gradient of the joined descriptors, read in place */
int lpm_vlad_aggregate_bwd_tiles_ld(const float* dout, int64_t dout_batch_stride, const float* nrm, const float* asum,
                                    const float* colsq, const float* csq, const float* gsq, const float* assign, const float* scale,
                                    const float* shift, const void* xr, const float* centres, int B, int T, int D, int K, int flags,
                                    float* dassign, float* dcentres, float* g0, void* workspace, size_t workspace_bytes,
                                    lpm_stream_t stream);
/* input_bn's gradients in the "no input gradient" mode (frames x = gamma xhat + beta straight out of input_bn,
 * frame_level_models.py:2265-2277, columns [0, D) of one stream), from lpm_vlad_aggregate_bwd_tiles' g0 [B, D] and dcentres [D, K],
 * the soft-assignment GEMM's W [D, K] and weight gradient dW [D, K], centres [D, K] (NULL without a residual term) and
 * colsum_dl [K] = column sums of the logit gradient (NULL when zero, i.e. after a training-mode batch norm):
 *   dbeta = -rowsum(dcentres) + W colsum_dl;   dgamma = (sum_b g0 + rowsum(W o dW) - rowsum(dcentres o centres) - beta dbeta) / gamma */
int lpm_input_bn_grads(const float* dcentres, const float* centres, const float* W, const float* dW, const float* g0,
                       const float* colsum_dl, const float* gamma, const float* beta, int B, int D, int K, float* dgamma,
                       float* dbeta, lpm_stream_t stream);
int lpm_vlad_aggregate_bwd_tiles_dx(const void* workspace, size_t workspace_bytes, const void* dlr, const void* wtt, int B,
                                    int T, int D, int K, float* dx, int64_t lddx, int accumulate_dx, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Backward through the training-mode batch-norm on the logits:
 *   in : dlt [M,K] (grad wrt gamma*Lhat+beta), logits L [M,K], mean, var [K], gamma [K]
 *   out: dl [M,K] (may alias dlt), dgamma [K], dbeta [K].   workspace: lpm_bn_bwd_workspace_bytes(M,K).
 * ------------------------------------------------------------------------------------------- */
/* Channel-last training batch norm of a [M, C] matrix (slim.batch_norm on the V2 encoder's [B, L, C] tensors,
 * transformer_utils.py:666,747,760, seen as M = B*L rows): y = (x - mean) * rsqrt(var + eps) * gamma + beta with the batch
 * statistics written to mean / var and folded into the moving averages (biased_moving_variance != 0: TF's non-fused path,
 * which rank-3 inputs take; 0: the fused path's unbiased estimate).  The backward is lpm_bn_bwd with logits := x. */
size_t lpm_bn_rows_workspace_bytes(int M, int C);
int lpm_bn_rows_fwd(const float* x, int M, int C, const float* gamma, const float* beta, float eps, float decay,
                    int biased_moving_variance, float* y, float* mean, float* var, float* moving_mean, float* moving_var,
                    void* workspace, size_t workspace_bytes, lpm_stream_t stream);
/* ... of act(x + pre_bias): the bias add (+ ReLU) of the dense layer in front (tf.layers.dense(use_bias=True, activation=relu) ->
 * slim.batch_norm, transformer_utils.py:741-760) rides in the statistics and apply passes; the activation is never stored. */
int lpm_bn_rows_act_fwd(const float* x, const float* pre_bias, int pre_relu, int M, int C, const float* gamma, const float* beta,
                        float eps, float decay, int biased_moving_variance, float* y, float* mean, float* var, float* moving_mean,
                        float* moving_var, void* workspace, size_t workspace_bytes, lpm_stream_t stream);
/* its backward: x = the raw dense output; dl = the gradient of x (ReLU mask applied), dbias = its column sums */
int lpm_bn_act_bwd_supported(int M, int K);
size_t lpm_bn_act_bwd_workspace_bytes(int M, int K);
int lpm_bn_act_bwd(const float* dlt, const float* x, const float* pre_bias, int pre_relu, const float* mean, const float* var,
                   const float* gamma, float eps, int M, int K, float* dl, float* dgamma, float* dbeta, float* dbias, void* workspace,
                   size_t workspace_bytes, lpm_stream_t stream);
/* FeedForwardNetworkMod (transformer_utils.py:741-756: dense -> relu -> batch_norm -> dense) without the fp32 round trip of the [M, 4F]
 * tensor between its dense layers: the batch norm writes its result ONLY as the split-bf16 activation image out3 [M, 3C] = [hi|lo|hi]
 * the second dense layer's GEMM reads, and its backward writes the gradient of the first dense layer's raw output ONLY as the gradient
 * image dl3 [M, 3K] = [hi|hi|lo].  Arguments otherwise as lpm_bn_rows_act_fwd / lpm_bn_act_bwd; C, K multiples of 8. */
int lpm_bn_rows_act_image_fwd(const float* x, const float* pre_bias, int pre_relu, int M, int C, const float* gamma, const float* beta,
                              float eps, float decay, int biased_moving_variance, void* out3, float* mean, float* var, float* moving_mean,
                              float* moving_var, void* workspace, size_t workspace_bytes, lpm_stream_t stream);
int lpm_bn_act_bwd_image(const float* dlt, const float* x, const float* pre_bias, int pre_relu, const float* mean, const float* var,
                         const float* gamma, float eps, int M, int K, void* dl3, float* dgamma, float* dbeta, float* dbias, void* workspace,
                         size_t workspace_bytes, lpm_stream_t stream);
size_t lpm_bn_bwd_workspace_bytes(int M, int K);
int lpm_bn_bwd(const float* dlt, const float* logits, const float* mean, const float* var, const float* gamma,
               float eps, int M, int K, float* dl, float* dgamma, float* dbeta, void* workspace,
               size_t workspace_bytes, lpm_stream_t stream);
/* ... where the normalised tensor is stored as bf16 [M, K] (BASELINE configs[4]'s logits; 16-byte aligned): read in place -- the same
 * numbers as lpm_bn_bwd on its fp32 copy, without the copy (round 6: 62 us per cfg-5 step) */
int lpm_bn_bwd_x16(const float* dlt, const void* logits_bf16, const float* mean, const float* var, const float* gamma, float eps, int M, int K,
                   float* dl, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Split-bf16 operand preparation for the encoder's dense layers (tf.layers.dense at
 * transformer_utils.py:559-561,583,701,708).  The GEMMs stay library GEMMs (hipBLASLt); these kernels produce the
 * operand format that runs them on the bf16 matrix pipe at fp32-grade accuracy:
 *   x W ~= xh Wh + xl Wh + xh Wl = [xh | xl | xh] . [Wh ; Wh ; Wl]   (one bf16 GEMM, fp32 accumulation)
 * lpm_split_rows:   x [M,K] fp32 (row stride ldx; optional fused relu(x + bias)) -> out3 [M,3K] bf16, plane order
 *                   [hi|lo|hi] (order 0: activations) or [hi|hi|lo] (order 1: gradients; what lpm_split_rows_relu_bwd emits)
 * lpm_split_weight: W [K,N] fp32 -> w3n [N,3K] bf16, row n = [Wh[:,n] | Wh[:,n] | Wl[:,n]], and
 *                                     w3k [K,3N] bf16, row k = [Wh[k,:] | Wl[k,:] | Wh[k,:]]  (w3k may be NULL)
 *   forward  y = X3 . w3n^T;   dx = DY3 . w3k^T (DY3 in gradient order): "B stored transposed", the layout hipBLASLt runs
 *   fastest;   dW = X3[3M,K]^T . DY3[3M,N] -- the two plane orders pair up row by row, so the weight gradient is a GEMM
 *   over the [3M, .] views of the two images (split over the reduction as a batched GEMM by the host code).
 * ------------------------------------------------------------------------------------------- */
int lpm_split_rows(const float* x, int64_t ldx, int64_t M, int K, const float* bias, int relu, int order, void* out3,
                   lpm_stream_t stream);
int lpm_split_weight(const float* W, int K, int N, void* w3n, void* w3k, lpm_stream_t stream);
/* The split-K partial sums [Z, K, N] of a weight-gradient product (a batched library GEMM over Z slices of the token reduction) added in
 * slice order and written straight into up to three [K, N / nouts] destinations -- the gradient slots of the column blocks of a
 * concatenated weight (q | k | v): one launch instead of one strided reduction per destination (TF autodiff of tf.layers.dense,
 * transformer_utils.py:559-561,583,701-711).  N / nouts a multiple of 4; unused destinations NULL. */
int lpm_sum_splits(const float* part, int Z, int K, int N, float* out0, float* out1, float* out2, int nouts, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Round 5: the operand FORMAT of the dense GEMMs' 16-bit images and fragment tiles (csrc/operand_format.h).
 *   LPM_OPERAND_BF16X3: x = xh + xl as bf16 planes, three products per a . b (~5e-6 per GEMM) -- every entry point above, and what a
 *                       NULL format means below.  Image rows: [hi | lo | hi] (activations), [hi | hi | lo] (gradients).
 *   LPM_OPERAND_FP16X3: ACTIVATIONS as fp16 (hi, lo) planes, image rows [hi | lo | hi]: the forward product keeps all three terms
 *                       against a weight split into fp16 planes as well ([Wh ; Wh ; Wl]) -- ~1e-6 per GEMM.
 *   LPM_OPERAND_FP16X2: GRADIENTS as fp16 (hi, lo) planes, image rows [hi | lo].  The input gradient is a TWO-term product against the
 *                       weight rounded once to fp16 ([Wh ; Wh]) -- 2.1e-4 relative L2 per GEMM (the 2^-12 rounding of the one-plane
 *                       operand).  The weight gradient, as the Python host runs it by default (ops.DW_TERMS = 1), is a ONE-term
 *                       product xh^T dyh of the two images' hi planes -- both operands rounded once, 2.9e-4 per GEMM; with
 *                       LPM_DW_TERMS=2 it is xh^T [dyh | dyl] (the gradient exact, 1.4e-4).
 *   fp16 kinds: every value is multiplied by `scale` (a power of two chosen by the caller so that the tensor sits inside fp16's range;
 *   values beyond it saturate at +-65504) before it is split; the consumer of the GEMM multiplies by 1 / scale.
 *   Weight forms of the fp16 kinds (lpm_split_weight_fmt / lpm_split_weight_tiles_fmt / lpm_weight_pack with kind != BF16X3):
 *   wn [N, 3K] fp16 rows [Wh^T | Wh^T | Wl^T] (forward), wk [K, 2N] fp16 rows [Wh | Wh] (input gradient), weight tiles of W with
 *   (hi, lo) fp16 planes (forward tile GEMM), weight tiles of W^T with the hi plane only (input-gradient tile GEMM).
 * amax (optional, any kind): the producer records max |x| of what it wrote, BEFORE scaling, by atomic maxima into the site's
 * LPM_OPERAND_AMAX_SUB sub-slots amax[LPM_OPERAND_AMAX_STRIDE * i] (one cache line apart, chosen by workgroup: same-address atomics
 * serialise; the caller zeroes all of them and takes the maximum over them) -- the measurement a delayed scale is chosen from
 * (ops.OperandScales).
 * NetVladV1's encoder GEMMs (transformer_utils.py:559-561,583,701-711 and TF autodiff of them) run on the fp16 kinds when the
 * trainer's scales are calibrated; NetVladV2 stays on LPM_OPERAND_BF16X3.
 * ------------------------------------------------------------------------------------------- */
enum { LPM_OPERAND_BF16X3 = 0, LPM_OPERAND_FP16X2 = 1, LPM_OPERAND_FP16X3 = 2 };
#define LPM_OPERAND_AMAX_SUB 32
#define LPM_OPERAND_AMAX_STRIDE 16
typedef struct LpmOperandFormat {
    int kind;       /* LPM_OPERAND_* */
    float scale;    /* > 0, a power of two; 1 for LPM_OPERAND_BF16X3 */
    float* amax;    /* device, may be NULL: LPM_OPERAND_AMAX_SUB * LPM_OPERAND_AMAX_STRIDE floats */
} LpmOperandFormat;
/* lpm_split_rows / lpm_split_rows_scaled in either format: x [M,K] (row stride ldx), optionally * row_scale[m], + bias, ReLU ->
 * image [M, planes K] (bf16x3: `order` 0 = [hi|lo|hi], 1 = [hi|hi|lo]; fp16x2: [hi|lo]). */
int lpm_split_rows_fmt(const float* x, int64_t ldx, int64_t M, int K, const float* bias, int relu, int order, const float* row_scale,
                       void* out, const LpmOperandFormat* fmt, lpm_stream_t stream);
/* lpm_split_rows_relu_bwd in either format: g = alpha * df * [act > 0] (alpha = 1 / the scale of the GEMM operand df came from), the
 * mask from the hi plane of the forward's activation image in format act_kind, dbias = column sums of g (un-scaled). */
int lpm_split_rows_relu_bwd_fmt(const float* df, int64_t M, int K, float alpha, const void* act_img, int act_kind, void* out_img, float* dbias,
                                void* workspace, size_t workspace_bytes, const LpmOperandFormat* fmt, lpm_stream_t stream);
/* lpm_split_weight in either format.  fp16 kinds: wn [N, 3K] fp16 rows [Wh^T | Wh^T | Wl^T], wk [K, 2N] fp16 rows [Wh | Wh] (wk may be NULL). */
int lpm_split_weight_fmt(const float* W, int K, int N, void* wn, void* wk, int kind, lpm_stream_t stream);
/* lpm_split_weight_tiles in either format.  LPM_OPERAND_FP16X3: fp16 (hi, lo) planes (the forward's weight operand);
 * LPM_OPERAND_FP16X2: the fp16 hi plane only, lpm_weight_tiles_bytes(R, N) / 2 bytes (the weight operand of a two-term product). */
int lpm_split_weight_tiles_fmt(const float* w, int R, int N, int transposed, void* wt, int kind, lpm_stream_t stream);
/* lpm_split_rows_tiles / lpm_image_row_tiles in either format (fp16x2 row tiles: fp16 (hi, lo) planes, same geometry). */
int lpm_split_rows_tiles_fmt(const float* x, int64_t ldx, int B, int T, int C, void* out, const LpmOperandFormat* fmt, lpm_stream_t stream);
int lpm_image_row_tiles_fmt(const void* img, int M, int K, int order, void* out, int kind, lpm_stream_t stream);
/* lpm_dense_tiles_act_image_fwd / lpm_dense_tiles_relu_bwd_image in either format.  The data operand: row tiles (x_image_kind /
 * dy_image_kind < 0, lpm_split_rows_tiles_fmt / lpm_image_row_tiles_fmt) or -- no extra pass -- the operand IMAGE itself of that kind
 * ([M][planes Kd]: an activation image for the forward, a gradient image for the backward), gathered row by row by the kernel's
 * LDS-DMA loads.  in_inv_scale: 1 / the scale the data operand was written with (the accumulators are multiplied by it); `fmt`: the
 * format of the image that leaves (and, for the backward, `act_kind`: the format of the forward's activation image whose hi plane
 * is the ReLU mask).  dbias: column sums of the UN-scaled masked gradient. */
int lpm_dense_tiles_act_image_fwd_fmt(const void* xr, int x_image_kind, const void* wt, const float* bias, int M, int Kd, int N,
                                      float in_inv_scale, void* out_img, const LpmOperandFormat* fmt, lpm_stream_t stream);
int lpm_dense_tiles_relu_bwd_image_fmt(const void* dyr, int dy_image_kind, const void* wtt, const void* act_img, int act_kind, int M, int Kd,
                                       int N, float in_inv_scale, void* out_img, float* dbias, void* workspace, size_t workspace_bytes,
                                       const LpmOperandFormat* fmt, lpm_stream_t stream);
/* lpm_layer_norm_act_image_fwd / lpm_layer_norm_act_bwd with the image (y_img / da_image) in either format. */
int lpm_layer_norm_act_image_fwd_fmt(const float* a, const float* bias, int relu, const float* r, const float* r_scale, const float* gamma,
                                     const float* beta, int B, int L, int F, float eps, float* y, int64_t y_batch_stride, void* y_img,
                                     float* z, float* stats, void* workspace, size_t workspace_bytes, const LpmOperandFormat* fmt,
                                     lpm_stream_t stream);
int lpm_layer_norm_act_mask_image_fwd_fmt(const float* a, const float* bias, int relu, const unsigned char* mask, float mask_scale, const float* r,
                                          const float* gamma, const float* beta, int B, int L, int F, float eps, float* y,
                                          int64_t y_batch_stride, void* y_img, float* z, float* stats, void* workspace,
                                          size_t workspace_bytes, const LpmOperandFormat* fmt, lpm_stream_t stream);
int lpm_layer_norm_act_bwd_fmt(const float* dy, int64_t dy_batch_stride, const float* z, const float* stats, const float* gamma,
                               const float* a, const float* bias, int relu, int B, int L, int F, float* dz, float* da, float* dgamma,
                               float* dbeta, float* dbias, const float* dr_extra, void* da_image, void* workspace, size_t workspace_bytes,
                               const LpmOperandFormat* fmt, lpm_stream_t stream);
/* lpm_bn_rows_act_image_fwd / lpm_bn_act_bwd_image (NetVladV2's dense -> batch norm -> dense chains, transformer_utils.py:666-677,741-756)
 * with the image in any operand format. */
int lpm_bn_rows_act_image_fwd_fmt(const float* x, const float* pre_bias, int pre_relu, int M, int C, const float* gamma, const float* beta,
                                  float eps, float decay, int biased_moving_variance, void* out_img, float* mean, float* var,
                                  float* moving_mean, float* moving_var, void* workspace, size_t workspace_bytes,
                                  const LpmOperandFormat* fmt, lpm_stream_t stream);
int lpm_bn_act_bwd_image_fmt(const float* dlt, const float* x, const float* pre_bias, int pre_relu, const float* mean, const float* var,
                             const float* gamma, float eps, int M, int K, void* dl_img, float* dgamma, float* dbeta, float* dbias,
                             void* workspace, size_t workspace_bytes, const LpmOperandFormat* fmt, lpm_stream_t stream);
/* lpm_mha_fwd_x3_image / lpm_mha_bwd_x3_image with the images in either format: o_fmt = the format the attention result's image is
 * written with (forward) / was written with (backward: kind and scale are read, amax ignored), g_fmt = the format of the
 * [dq | dk | dv] gradient image (fp16x2: row = [hi(3N) | lo(3N)]). */
int lpm_mha_fwd_x3_image_fmt(const float* q, const float* k, const float* v, int64_t ld, int B, int L, int h, int d, float scale,
                             void* o_img, float* lse, const LpmOperandFormat* o_fmt, lpm_stream_t stream);
int lpm_mha_bwd_x3_image_fmt(const float* q, const float* k, const float* v, int64_t ld, const void* o_img, const LpmOperandFormat* o_fmt,
                             const float* dout, int64_t ldo, const float* lse, int B, int L, int h, int d, float scale, void* dqkv_img,
                             const LpmOperandFormat* g_fmt, lpm_stream_t stream);
/* lpm_sum_splits for the fp16 weight-gradient product.  halves = 2 (the two-term form dW = xh^T [dyh | dyl]): part [Z, K, 2 N] (the lo half's products in columns
 * [N, 2N)), out_j[k, n] = alpha * sum_z (part[z, k, j Nj + n] + part[z, k, N + j Nj + n]), Nj = N / nouts; alpha = 1 / (the two
 * operands' scales).  halves = 1 (the one-term form dW = xh^T dyh, the host's default): part [Z, K, N], the plain sum times alpha. */
int lpm_sum_splits_scaled(const float* part, int Z, int K, int N, int halves, float alpha, float* out0, float* out1, float* out2, int nouts,
                          lpm_stream_t stream);

/* Every operand form of every dense-layer weight of a step in ONE launch (weight_pack.hip, round 4): a job names a weight W [K, N]
 * (row stride ldw) -- or the column block [n_off, n_off + N) of a concatenated weight with Ntot columns (q | k | v) -- and the forms
 * wanted of it, each optional (NULL): w3n / w3k as lpm_split_weight writes them (of the concatenated weight: w3n [Ntot, 3K],
 * w3k [K, 3 Ntot]), wt = lpm_split_weight_tiles(W, K, Ntot, transposed = 0) and wtt = lpm_split_weight_tiles of the same storage
 * with transposed = 1 (the B operand of the input-gradient GEMM; whole weights only).  Bit for bit the single-weight entry points'
 * outputs.  K, N, n_off, Ntot multiples of 32; at most LPM_WEIGHT_PACK_MAX_JOBS jobs per call.  The job array is HOST memory, read
 * during the call.  Reference: transformer_utils.py:559-561,583,701-711; frame_level_models.py:2781-2789 (the weights of tf.layers.dense
 * / tf.matmul, constant within a step). */
#define LPM_WEIGHT_PACK_MAX_JOBS 24
typedef struct LpmWeightPackJob {
    const float* w;     /* device, [K, N] fp32, row stride ldw */
    int K, N, ldw;
    int Ntot, n_off;    /* the weight's place in a concatenation along N (whole weight: Ntot = N, n_off = 0) */
    void* w3n;          /* device outputs, NULL = not wanted */
    void* w3k;
    void* wt;
    void* wtt;
    int kind;           /* LPM_OPERAND_BF16X3, or an fp16 kind: w3n [Ntot, 3K] fp16 = [Wh^T | Wh^T | Wl^T], w3k [K, 2 Ntot] fp16 = [Wh | Wh], wt with
                           fp16 (hi, lo) planes, wtt with the fp16 hi plane only (lpm_split_weight_fmt, lpm_split_weight_tiles_fmt) */
} LpmWeightPackJob;
int lpm_weight_pack(const LpmWeightPackJob* jobs, int njobs, lpm_stream_t stream);
/* backward of the fused relu(x + bias) split (FeedForwardNetwork, transformer_utils.py:701-711): g = df * [act > 0]
 * (act3 = the forward's [M,3K] split image of the activation), out3 = split image of g, dbias [K] = column sums of g.
 * workspace: lpm_split_rows_relu_bwd_workspace_bytes(M, K). */
size_t lpm_split_rows_relu_bwd_workspace_bytes(int64_t M, int K);
int lpm_split_rows_relu_bwd(const float* df, int64_t M, int K, const void* act3, void* out3, float* dbias, void* workspace,
                            size_t workspace_bytes, lpm_stream_t stream);
/* bias (+ ReLU) of a dense layer whose output does not go straight into the next GEMM's operand split (tf.layers.dense(use_bias=True,
 * activation=relu) in front of a batch norm: transformer_utils.py:741-760).  forward: y <- act(y + bias) IN PLACE;  backward: dx = dy *
 * [y > 0] with y the forward's output (dx may alias dy; without relu dx is not written), dbias = column sums of the masked gradient. */
int lpm_bias_act_fwd(float* y, const float* bias, int relu, int64_t M, int C, lpm_stream_t stream);
size_t lpm_bias_act_bwd_workspace_bytes(int64_t M, int C);
int lpm_bias_act_bwd(const float* dy, const float* y, int relu, int64_t M, int C, float* dx, float* dbias, void* workspace,
                     size_t workspace_bytes, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Residual add + tf.contrib.layers.layer_norm with TF1 defaults (transformer_utils.py:405-411,451-454,712-713):
 * one mean/variance per example over all L*F non-batch elements, gamma/beta [F], eps inside the sqrt.
 *   fwd: z = a + r (r may be NULL: then z is not written and a plays its role); y = (z-mean)*rstd*gamma + beta;
 *        stats [B,2] = (mean, rstd).   bwd: dz (gradient w.r.t. z, i.e. w.r.t. both a and r), dgamma, dbeta.
 * a, r, y, z, dy, dz: [B, L, F] contiguous.  workspace: lpm_layer_norm_workspace_bytes(B, F).
 * ------------------------------------------------------------------------------------------- */
size_t lpm_layer_norm_workspace_bytes(int B, int F);
int lpm_layer_norm_fwd(const float* a, const float* r, const float* gamma, const float* beta, int B, int L, int F,
                       float eps, float* y, float* z, float* stats, void* workspace, size_t workspace_bytes,
                       lpm_stream_t stream);
int lpm_layer_norm_bwd(const float* dy, const float* z, const float* stats, const float* gamma, int B, int L, int F,
                       float* dz, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                       lpm_stream_t stream);
/* The same with the producing dense layer's bias add (and ReLU) fused: z = act(a + bias) (+ r), act = relu when `relu`
 * (tf.layers.dense(use_bias=True[, activation=relu]) at transformer_utils.py:583, :708-711 followed by the residual
 * layer_norm).  Backward: dz as above (= the residual's gradient), da = dz * [a + bias > 0] (written to `da` only when
 * relu; otherwise da = dz), dbias = column sums of da.  bias == NULL reduces to lpm_layer_norm_fwd / _bwd.
 * dr_extra (optional, [B,L,F]): added to the residual's gradient on its way out (dz = dz + dr_extra; da is then written
 * separately, `da` required) -- a second consumer's gradient of the residual tensor, folded in without an add pass.
 * da_image (optional, [B*L, 3F] bf16): da written as the split-bf16 gradient image [hi | hi | lo] the next GEMMs read
 * (lpm_split_rows order 1) instead of / besides fp32 `da` (which may then be NULL).
 * y_batch_stride / dy_batch_stride (floats, 0 = L*F): y may be one clip-slot of a wider [B, total] buffer (the pooled
 * descriptors of both streams side by side, tf.concat at frame_level_models.py:2309 without the copy) and dy the matching
 * column slice of that buffer's gradient. */
int lpm_layer_norm_act_fwd(const float* a, const float* bias, int relu, const float* r, const float* gamma, const float* beta,
                           int B, int L, int F, float eps, float* y, int64_t y_batch_stride, float* z, float* stats,
                           void* workspace, size_t workspace_bytes, lpm_stream_t stream);
/* layer_norm(layer_norm(act(a + bias) + r; gamma1, beta1) + r; gamma2, beta2) -- the tail of the V1 encoder
 * (transformer_utils.py:708-713 followed by :409-411; both layer norms add the same residual r) in three passes: the first
 * layer norm's apply pass adds r again and reduces the second one's statistics.  Saves z1/stats1 and z2/stats2, which
 * lpm_layer_norm_act_bwd takes unchanged for either layer norm. */
int lpm_layer_norm_pair_fwd(const float* a, const float* bias, int relu, const float* r, const float* gamma1, const float* beta1,
                            const float* gamma2, const float* beta2, int B, int L, int F, float eps, float* y,
                            int64_t y_batch_stride, float* z1, float* stats1, float* z2, float* stats2, void* workspace,
                            size_t workspace_bytes, lpm_stream_t stream);
int lpm_layer_norm_act_bwd(const float* dy, int64_t dy_batch_stride, const float* z, const float* stats, const float* gamma,
                           const float* a,
                           const float* bias, int relu, int B, int L, int F, float* dz, float* da, float* dgamma, float* dbeta,
                           float* dbias, const float* dr_extra, void* da_image, void* workspace, size_t workspace_bytes,
                           lpm_stream_t stream);
/* ... with tf.layers.dropout between the dense layer and the layer norm (NetVladV2's TransformerEncoderMod, transformer_utils.py:450-454):
 * y = layer_norm(act(a + bias) * keep * mask_scale + r).  mask [B, L, F]: one byte per element, non-zero = kept, 4-byte aligned;
 * mask_scale = 1 / keep probability; y3 (optional) as in lpm_layer_norm_act_image_fwd.  The backward returns da (or da_image) =
 * dz * [ReLU mask] * keep * mask_scale -- the gradient of the dense layer's RAW output -- and dbias = its column sums. */
int lpm_layer_norm_act_mask_image_fwd(const float* a, const float* bias, int relu, const unsigned char* mask, float mask_scale,
                                      const float* r, const float* gamma, const float* beta, int B, int L, int F, float eps, float* y,
                                      int64_t y_batch_stride, void* y3, float* z, float* stats, void* workspace, size_t workspace_bytes,
                                      lpm_stream_t stream);
int lpm_layer_norm_act_mask_bwd(const float* dy, int64_t dy_batch_stride, const float* z, const float* stats, const float* gamma,
                                const float* a, const float* bias, int relu, const unsigned char* mask, float mask_scale, int B, int L,
                                int F, float* dz, float* da, float* dgamma, float* dbeta, float* dbias, const float* dr_extra,
                                void* da_image, void* workspace, size_t workspace_bytes, lpm_stream_t stream);
/* ... with da_image in either operand format (fmt NULL: split-bf16 [hi | hi | lo]; fp16 two-product: [hi | lo] of value * scale, max |value|
 * recorded) -- lpm_layer_norm_act_bwd_fmt's image for the masked form (round 6). */
int lpm_layer_norm_act_mask_bwd_fmt(const float* dy, int64_t dy_batch_stride, const float* z, const float* stats, const float* gamma,
                                    const float* a, const float* bias, int relu, const unsigned char* mask, float mask_scale, int B, int L,
                                    int F, float* dz, float* da, float* dgamma, float* dbeta, float* dbias, const float* dr_extra,
                                    void* da_image, void* workspace, size_t workspace_bytes, const LpmOperandFormat* fmt, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4: multi-head attention core  o = softmax(scale * q k^T) v   per (batch, head)
 *   replaces transformer_utils.py:564-581 (split_heads, q scaling, matmul, softmax, matmul,
 *   combine_heads).  q,k,v,o are the [B, L, h*d] outputs of the dense projections (heads
 *   interleaved on the last axis, row stride ld); no [B,h,L,L] tensor is ever materialised.
 *   lse [B,h,L] (log-sum-exp per query row) is saved for the backward.
 *   lpm_mha_bwd: dz_partial (optional, used with key_scale) [B*h, 2, L] receives per-(batch,head)
 *   column sums over the query axis of dz and dz*s (s = raw logit) for the logits_bn backward.
 *   The batch-statistics term of that backward enters as two per-key vectors (both [L], optional):
 *   ds = key_scale*dz - corr_a[key] - s*corr_b[key].  With dq = dk = dv = NULL only dz_partial is
 *   produced (first pass of the two-pass logits_bn backward).
 * With key_scale/key_shift != NULL (both [L]) the logits are first mapped
 *   z[q,j] = (q.k_j)*key_scale[j] + key_shift[j] -- the folded logits_bn of MultiHeadAttentionBN
 *   (transformer_utils.py:652-659); lpm_mha_logit_stats produces the column partials for it.
 * Supported: d in {8,16}, L <= 512.
 * ------------------------------------------------------------------------------------------- */
int lpm_mha_fwd(const float* q, const float* k, const float* v, int64_t ld, int B, int L, int h, int d,
                float scale, const float* key_scale, const float* key_shift, float* o, int64_t ldo, float* lse,
                lpm_stream_t stream);
/* The same contract on the bf16 matrix pipe (csrc/mha_x3.hip): split-bf16 operands (3 bf16 MFMAs per product, fp32
 * accumulation, ~1e-5 relative error on the logits), v_mfma_f32_16x16x32_bf16.  Pointers 16-byte aligned. */
int lpm_mha_fwd_x3(const float* q, const float* k, const float* v, int64_t ld, int B, int L, int h, int d,
                float scale, const float* key_scale, const float* key_shift, float* o, int64_t ldo, float* lse,
                lpm_stream_t stream);
int lpm_mha_bwd(const float* q, const float* k, const float* v, int64_t ld, const float* o, const float* dout,
                int64_t ldo, const float* lse, int B, int L, int h, int d, float scale, const float* key_scale,
                const float* key_shift, float* dq, float* dk, float* dv, int64_t ldd, const float* corr_a,
                const float* corr_b, float* dz_partial, lpm_stream_t stream);
/* lpm_mha_bwd on the bf16 matrix pipe (split-bf16 operands, csrc/mha_x3.hip); same contract, 16-byte aligned pointers. */
int lpm_mha_bwd_x3(const float* q, const float* k, const float* v, int64_t ld, const float* o, const float* dout,
                int64_t ldo, const float* lse, int B, int L, int h, int d, float scale, const float* key_scale,
                const float* key_shift, float* dq, float* dk, float* dv, int64_t ldd, const float* corr_a,
                const float* corr_b, float* dz_partial, lpm_stream_t stream);
/* Arithmetic of lpm_mha_bwd_x3 / lpm_mha_bwd_x3_image[_fmt] for the process: 2 (opt-in: LPM_MHA_BWD_TERMS=2) = the three products behind
 * dS (dV = dO^T P, dK = Q^T dS, dQ = K^T dS) on two fp16 terms -- the operand re-read from LDS exact as fp16 (hi, lo), P and dS rounded
 * once to fp16 (2^-12: the arithmetic of the dense layers' input gradients), dO under a power-of-two scale the kernels take from
 * max |dO| themselves; the scores and dP = dO V^T stay split-bf16 x3.  Launches WITH correction vectors (logits_bn's dq pass) keep
 * three terms whatever the setting.  3 (the default) = split-bf16, three-term products throughout.  Any other value only queries.
 * Returns the previous setting. */
int lpm_mha_bwd_set_terms(int terms);
/* logits_bn backward in one pass over the scores: after lpm_mha_bwd_x3(dq = NULL, dk, dv, corr_a = corr_b = NULL, dz_partial) -- the
 * key / value gradients WITHOUT the batch statistics' share, and the statistics --, lpm_mha_bn_corrections, and
 * lpm_mha_bwd_x3(dq, dk = dv = NULL, corr_a, corr_b) for the query gradient, this subtracts the share from dk in place:
 *   dk[b, j, head] -= corr_a[j] Sq + corr_b[j] Qm k[b, j, head],   Sq = sum_q scale q,  Qm = sum_q (scale q)(scale q)^T per (b, head)
 * (the two correction terms of ds are affine in the raw score; transformer_utils.py:652-658, backward).  d in {8, 16}.
 * moments: NULL, or the forward's lpm_mha_logit_stats_moments output (Qm, Sq of the unscaled q) -- then q is not read. */
int lpm_mha_bn_dk_correct(const float* q, const float* k, int64_t ld, int B, int L, int h, int d, float scale, const float* corr_a,
                          const float* corr_b, float* dk, int64_t ldd, const float* moments, lpm_stream_t stream);
/* The same backward with the q / k / v gradients leaving as the [dq | dk | dv] GRADIENT IMAGE that the q/k/v layer's input- and weight-
 * gradient GEMMs read (lpm_mha_bwd_x3_image_fmt's image; g_fmt NULL: split-bf16 [hi | hi | lo] planes of 3 h d columns, fp16 two-product:
 * [hi | lo]) -- no fp32 [B*L, 3 h d] gradient and no lpm_split_rows pass over it (round 6).  Three launches, each writing its own columns:
 *   lpm_mha_bwd_x3_bn_image_fmt(dk_plain != NULL, corr_a = corr_b = NULL, dz_partial): the key / value sweep -- dv into the image, dk
 *       WITHOUT the batch statistics' share into dk_plain [B*L, h*d] fp32, the statistics into dz_partial;
 *   lpm_mha_bn_corrections;
 *   lpm_mha_bwd_x3_bn_image_fmt(dk_plain = NULL, corr_a, corr_b, dz_partial = NULL): the query sweep -- dq into the image;
 *   lpm_mha_bn_dk_correct_image: dk_plain minus the share (lpm_mha_bn_dk_correct's formula) into dk's columns of the image.
 * o and dout are plain fp32 (the attention result of this variant is batch-normalised before its GEMM).  All pointers 16-byte aligned.
 * Replaces the tail of /root/reference/transformer_utils.py:652-661's backward in front of :559-561's. */
int lpm_mha_bwd_x3_bn_image_fmt(const float* q, const float* k, const float* v, int64_t ld, const float* o, const float* dout, int64_t ldo,
                                const float* lse, int B, int L, int h, int d, float scale, const float* key_scale, const float* key_shift,
                                float* dk_plain, const float* corr_a, const float* corr_b, float* dz_partial, void* dqkv_img,
                                const LpmOperandFormat* g_fmt, lpm_stream_t stream);
int lpm_mha_bn_dk_correct_image(const float* q, const float* k, int64_t ld, int B, int L, int h, int d, float scale, const float* corr_a,
                                const float* corr_b, const float* dk_plain, const float* moments, void* dqkv_img,
                                const LpmOperandFormat* g_fmt, lpm_stream_t stream);
/* logits_bn's per-(batch, head) statistics (what lpm_mha_logit_stats computes: partial [B*h][2][L] = sum_q q.k_j and sum_q (q.k_j)^2)
 * from the d x d moments of q, which it also hands out: moments [B*h][d*d + d] = (Qm = sum_q q q^T, Sq = sum_q q), or NULL.  Given to
 * lpm_mha_bn_dk_correct (`moments`, of the UNSCALED q; q may then be NULL) the backward does not form them again.  d in {8, 16}. */
int lpm_mha_logit_stats_moments(const float* q, const float* k, int64_t ld, int B, int L, int h, int d, float* partial, float* moments,
                                lpm_stream_t stream);
/* The same backward (no logits_bn) writing the q/k/v gradients ONLY as the split-bf16 gradient image the projection GEMMs read:
 * dqkv3 [B*L, 9*h*d] bf16, row = [hi | hi | lo] planes of the concatenated columns [dq | dk | dv] (what lpm_split_rows with
 * order = 1 would produce from the fp32 gradients) -- the fp32 dq/dk/dv and the split pass over them never exist. */
int lpm_mha_bwd_x3_image(const float* q, const float* k, const float* v, int64_t ld, const void* o, int o_is_image,
                         const float* dout, int64_t ldo, const float* lse, int B, int L, int h, int d, float scale, void* dqkv3,
                         lpm_stream_t stream);
/* The forward writing its result ONLY as the split-bf16 activation image the output projection GEMM reads: o3 [B*L, 3*h*d]
 * bf16, row = [hi | lo | hi] planes (lpm_split_rows order 0).  lpm_mha_bwd_x3_image takes it back with o_is_image = 1 (it
 * rebuilds o = hi + lo for D_q = <dO_q, O_q>); with o_is_image = 0 `o` is the fp32 [B, L, ldo] tensor (ldo shared with dout). */
int lpm_mha_fwd_x3_image(const float* q, const float* k, const float* v, int64_t ld, int B, int L, int h, int d, float scale,
                         void* o3, float* lse, lpm_stream_t stream);
size_t lpm_mha_logit_stats_workspace_bytes(int B, int L, int h);
int lpm_mha_logit_stats(const float* q, const float* k, int64_t ld, int B, int L, int h, int d, float* partial,
                        lpm_stream_t stream);
/* slim.batch_norm (training mode) of a SMALL [M, C] matrix, M <= 256 rows, with what follows it, in ONE launch each way -- the clip-level
 * tail of the model (frame_level_models.py:2321-2327 hidden1_bn + relu6 :2337; :2354-2368 gating_bn + the context gate):
 *   act 0: y = bn(x);  act 1: y = relu6(bn(x));  act 2: y = mul * sigmoid(bn(x)).
 * mean / rstd [C]: the batch statistics, kept for the backward; moving_mean / moving_var (or both NULL) are updated in place with
 * `decay` and the unbiased variance (TF's fused rank-2 batch norm).  Backward: dx, dgamma, dbeta and (act 2) dmul [M, C]. */
int lpm_bn_small_fwd(const float* x, int M, int C, const float* gamma, const float* beta, float eps, float decay, int act,
                     const float* mul, float* y, float* mean, float* rstd, float* moving_mean, float* moving_var, lpm_stream_t stream);
int lpm_bn_small_bwd(const float* dy, const float* x, int M, int C, const float* gamma, const float* beta, const float* mean,
                     const float* rstd, int act, const float* mul, float* dx, float* dgamma, float* dbeta, float* dmul,
                     lpm_stream_t stream);
/* logits_bn backward bookkeeping in one launch: partial [nblk][2][L] = lpm_mha_bwd's statistics pass (column sums of dz and dz * s per
 * (batch, head)) -> dgamma, dbeta and, in training (corr_a / corr_b given), the correction vectors of the main backward pass;
 * n = B * h * L logits per key position (transformer_utils.py:652-658, backward) */
int lpm_mha_bn_corrections(const float* partial, int nblk, int L, const float* mean, const float* var, const float* kscale, float eps,
                           int64_t n, float* dgamma, float* dbeta, float* corr_a, float* corr_b, lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * MoeModel tail + CrossEntropyLoss (video_level_models.py:116-126, losses.py:41-51), fused:
 *   g = softmax over the m+1 gate activations of a (clip, class), e = sigmoid of its m expert activations,
 *   predictions[b,c] = sum_{i<m} g_i e_i;   loss = mean_b sum_c -[y log(p+eps) + (1-y) log(1-p+eps)].
 * gate_act [B, V*(m+1)], expert_act [B, V*m] (bias added), labels [B,V] float 0/1 (NULL: predictions only, then loss and
 * loss_partial [lpm_moe_ce_nblk(B,V)] must be NULL too).  Backward: dgate_act / dexpert_act from dloss (scalar on the
 * device) and/or dpredictions [B,V] (either may be NULL).
 * ------------------------------------------------------------------------------------------- */
int lpm_moe_ce_nblk(int B, int V);
int lpm_moe_ce_fwd(const float* gate_act, const float* expert_act, const float* labels, int B, int V, int num_mixtures, float eps,
                   float* predictions, float* loss, float* loss_partial, lpm_stream_t stream);
int lpm_moe_ce_bwd(const float* gate_act, const float* expert_act, const float* labels, const float* dloss,
                   const float* dpredictions, int B, int V, int num_mixtures, float eps, float* dgate_act, float* dexpert_act,
                   lpm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a14 + a15: per-variable clip_by_norm + TF-style Adam over a flat parameter arena
 *   replaces utils.clip_gradient_norms (utils.py:170-189) + tf.train.AdamOptimizer.apply_gradients
 *   (train.py:336).  `offsets` [ntensors+1] (int64, DEVICE) delimits each variable inside the
 *   flat fp32 arenas param/grad/m/v; every offset is a multiple of LPM_ARENA_ALIGN floats (padding
 *   holds zeros).  scratch: lpm_clip_adam_scratch_bytes(total, ntensors).  step is 1-based.  Per variable:
 *   g *= clip / max(||g||, clip);  m,v update;  p -= lr_t * m / (sqrt(v) + eps),
 *   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)   (epsilon outside the bias correction, as TF1).
 * ------------------------------------------------------------------------------------------- */
#define LPM_ARENA_ALIGN 4096
size_t lpm_clip_adam_scratch_bytes(int64_t total, int ntensors);
int lpm_multi_tensor_clip_adam(float* param, const float* grad, float* m, float* v, const int64_t* offsets,
                               int ntensors, int64_t total, float clip_norm, float lr, float beta1, float beta2,
                               float eps, int64_t step, float* scratch, lpm_stream_t stream);
/* ... with the analytic gradient of each variable's L2 penalty (slim.l2_regularizer on the MoE weights, video_level_models.py:84-100:
 * part of the loss whose gradient utils.py:170-189 clips), l2coef[t] * param, added on the fly in the norm pass and in the update pass
 * (round 6).  l2coef: device array [ntensors], 0 = none; NULL = lpm_multi_tensor_clip_adam.  Equals `grad += l2coef[t] * param` in front of
 * the plain entry point without that pass over the arena. */
int lpm_multi_tensor_clip_adam_l2(float* param, const float* grad, float* m, float* v, const int64_t* offsets, const float* l2coef,
                                  int ntensors, int64_t total, float clip_norm, float lr, float beta1, float beta2, float eps,
                                  int64_t step, float* scratch, lpm_stream_t stream);

/* a14 + a15 for a variable whose gradient is dW [N1, N2] = X^T DY, X [R, N1], DY [R, N2], R = the batch over ALL towers (the SUM of
 * utils.combine_gradients, utils.py:192-213, is the product over the concatenated rows): the hidden projection's weight
 * (frame_level_models.py:2314-2319), 75 % (cfg-2) ... 94 % (cfg-5) of the model's parameters.  The gradient is never written:
 * a first tile-GEMM pass yields ||dW|| (fixed summation order), a second one applies clip + TF-Adam to param / m / v in place from
 * its accumulators.  xt, dyt: lpm_split_weight_tiles of X and DY (not transposed; towers' tile buffers concatenate along R).
 * Same per-element arithmetic as lpm_skinny_weight_grad_tiles followed by lpm_multi_tensor_clip_adam.  scratch:
 * lpm_factored_clip_adam_scratch_bytes(N1, N2); on return scratch[bytes/4 - 4] holds the clip factor, [.. - 3] the norm. */
size_t lpm_factored_clip_adam_scratch_bytes(int N1, int N2);
int lpm_factored_clip_adam(const void* xt, const void* dyt, int R, int N1, int N2, float* param, float* m, float* v,
                           float clip_norm, float lr, float beta1, float beta2, float eps, int64_t step, float* scratch,
                           size_t scratch_bytes, lpm_stream_t stream);
/* ... with the norm from quadratic forms instead of a first tile-GEMM pass: ||X^T DY||^2 = sum over columns n1 of x_n1^T G x_n1,
 * G = DY DY^T [R, R] computed by the caller (fp32) and handed over as gdt = lpm_split_rows_tiles(G, R, 1, R, R); x = the fp32 matrix
 * [R, N1] (row stride ldx) that xt was split from.  2 R^2 N1 flops for the norm instead of 2 R N1 N2.  R <= 128. */
int lpm_factored_clip_adam_q(const void* xt, const void* dyt, const float* x, int64_t ldx, const void* gdt, int R, int N1, int N2,
                             float* param, float* m, float* v, float clip_norm, float lr, float beta1, float beta2, float eps,
                             int64_t step, float* scratch, size_t scratch_bytes, lpm_stream_t stream);
/* ... keeping the bf16 compute copy param_bf16 [N1, N2] = bf16(param) beside the master: written by the update pass's epilogue.  x and gdt
 * both NULL: the norm from a GEMM pass (lpm_factored_clip_adam); both given: from the quadratic forms (lpm_factored_clip_adam_q). */
int lpm_factored_clip_adam_copy(const void* xt, const void* dyt, const float* x, int64_t ldx, const void* gdt, int R, int N1, int N2,
                                float* param, float* m, float* v, void* param_bf16, float clip_norm, float lr, float beta1, float beta2,
                                float eps, int64_t step, float* scratch, size_t scratch_bytes, lpm_stream_t stream);
/* ... and returning the projection's INPUT gradient with it (round 6): dx [R, N1] (row stride ld_dx) = DY W_old^T, formed inside the update
 * pass from the weights it streams anyway instead of by lpm_proj_dx_w16 from 2 more bytes per weight.  dy_bf16 = DY [R, N2] rounded once to
 * bf16, row-major, 16-byte aligned (dx too; ld_dx a multiple of 4); W_old enters by its bf16 rounding -- the compute copy's values, i.e. what the forward multiplied by:
 * lpm_proj_dx_w16's arithmetic, one bf16 MFMA per product.  A workgroup owns 64 rows of the variable and ALL its columns, so the partial
 * sums of a dx element meet in registers in a fixed order; the update itself is lpm_factored_clip_adam_copy's bit for bit.
 * lpm_factored_fold_supported: R % 16 == 0, R <= 128, N1 % 64 == 0, N2 % 128 == 0; elsewhere LPM_ERR_UNSUPPORTED_SHAPE.
 * The caller must not have read dx's consumers' input before this returns on `stream`: the call sits INSIDE the backward pass, at the
 * projection (frame_level_models.py:2309-2319 backward + utils.py:170-189 + train.py:332-336 for this one variable). */
int lpm_factored_fold_supported(int R, int N1, int N2);
int lpm_factored_clip_adam_copy_dx(const void* xt, const void* dyt, const float* x, int64_t ldx, const void* gdt, int R, int N1, int N2,
                                   float* param, float* m, float* v, void* param_bf16, const void* dy_bf16, float* dx, int64_t ld_dx,
                                   float clip_norm, float lr, float beta1, float beta2, float eps, int64_t step, float* scratch,
                                   size_t scratch_bytes, lpm_stream_t stream);

/* The keep mask of tf.layers.dropout (transformer_utils.py:450): mask[i] = 1 with probability keep_prob (16-bit resolution), 0 otherwise, from
 * a counter-based hash of (seed, i) -- one store stream instead of torch's bernoulli_ (round 6).  n bytes, a multiple of 16; 16-byte aligned.
 * The same (seed, n, keep_prob) gives the same mask.  The consumers are lpm_layer_norm_act_mask_image_fwd[_fmt] / lpm_layer_norm_act_mask_bwd[_fmt]. */
int lpm_dropout_keep_mask(void* mask, int64_t n, float keep_prob, uint64_t seed, lpm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LPM_HIP_H */
