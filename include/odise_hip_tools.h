/*
 * odise_hip_tools.h — developer / measurement hooks of libodise_hip.so.  NOT part of the drop-in boundary (include/odise_hip.h):
 * nothing on the product path calls these; tools/ (tile calibration, A/B runs of kernel generations, the MFMA and LDS rate probes)
 * and the ABI self-check of tests/test_lib_abi.py do.
 */
#ifndef ODISE_HIP_TOOLS_H
#define ODISE_HIP_TOOLS_H

#include "odise_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* sizeof of the descriptor structs as the library was compiled (tests validate the ctypes mirrors against them) */
int odise_hip_sizeof_gemm_desc(void);
int odise_hip_sizeof_conv_desc(void);
int odise_hip_sizeof_attn_desc(void);
int odise_hip_sizeof_post_desc(void);
int odise_hip_sizeof_infer_desc(void);

/* odise_hip_gemm / odise_hip_conv2d with the tile shape and the split-K factor forced instead of chosen by the cost model
 * (tile ids: gemm.hip kTileBM / kTileBN; -1 / 0 = automatic) */
int odise_hip_gemm_forced(odise_hip_ctx* ctx, const odise_gemm_desc* d, int tile, int splitk);
int odise_hip_conv2d_forced(odise_hip_ctx* ctx, const odise_conv_desc* d, int tile, int splitk);
/* tile id | split-K factor << 8 that the calling thread's last odise_hip_gemm / odise_hip_conv2d launch ran with (-1: none yet) */
int odise_hip_last_tile(void);
/* a GEMM with a LayerNorm folded into its epilogue, as the CLIP towers chain them (csrc/common.h LnEpi; any pointer may be NULL):
 *   producer  stats_out [M][N/64][2]: partial (sum, sum of squares) of every output row, per 64 columns
 *   consumer  part [M][parts][2] + colsum [N]: C = act(rstd_m (A W'^T - mean_m colsum) + bias_n) (+ residual), the row statistics finished from
 *             the partials with 1/inv_c channels and eps; final_out [M][2] receives (-mean rstd, rstd)
 *   swapped   fin [N][2] + rowsum [M]: the normalised operand is W (its rows are the tokens), statistics per output column */
int odise_hip_gemm_ln(odise_hip_ctx* ctx, const odise_gemm_desc* d, const float* part, int parts, float inv_c, float eps, const float* colsum,
                      float* final_out, const float* fin, const float* rowsum, float* stats_out);
/* the conv -> GroupNorm pair of the ResBlocks with the conv's tile forced: the conv epilogue reduces the GroupNorm statistics (per channel and
 * row block) into stats_scratch [N * ceil(OH*OW/64) * Cout * 2] and the GroupNorm only finalises + applies; y_norm = act(gn(conv(x))) (f16).
 * *stats_blocks = row blocks per image (0: this kernel declined the fusion, the stand-alone GroupNorm ran) */
int odise_hip_conv2d_gn_forced(odise_hip_ctx* ctx, const odise_conv_desc* d, int tile, int splitk, const float* gamma, const float* beta,
                               int groups, float eps, int act, void* y_norm, float* stats_scratch, int* stats_blocks);
/* process-wide kernel-selection switches for A/B measurements (bits: gemm.hip launch_gemm) */
int odise_hip_gemm_debug(int flags);

/* 1: the post-processing kernels never take their exact-x4-upsampling specialisations (tests assert both forms are bit-identical);
 * 2: the specialisations, but the per-pixel pass in its thread-per-cell-column form instead of the tiled one */
int odise_hip_post_generic(int on);
/* force the tile of the semantic GEMM [K, pixels] = P^T S^T (A/B of the 256-row rule in odise_hip_postprocess_batch); -1 = the rule */
int odise_hip_sem_tile(int tile);
/* GroupNorm (csrc/norm.hip): chunks of the statistics pass per image = compute units * chunk_factor / images (default 2; 0 keeps the current value; a NEGATIVE value sets instead the row count from which LayerNorm takes 8 rows per wavefront to its magnitude);
 * fold_in_apply != 0 (default): with <= 64 chunks per image the apply kernel folds the partials itself and gn_finalize_kernel does not run */
int odise_hip_gn_tuning(int chunk_factor, int fold_in_apply);
/* 1: the pixel decoder's MSDeformAttn layers as msda_prepare_kernel + the native-op kernel (the round 1-5 form) instead of the fused gather (A/B, bit-compare) */
int odise_hip_msda_unfused(int on);
/* the pixel decoder's MSDeformAttn on raw projections: value f16 [B, Lq, M, 32], off f32 [B*Lq, M*12*2], aw f32 [B*Lq, M*12] (3 levels hs3 x ws3, 4 points),
 * out f16 [B*Lq, M*32].  fused = 2 (or any other non-zero value): msda_fused_kernel as the pixel decoder runs it, 1: its 8-lanes-per-pair variant; 0: msda_prepare_kernel (into loc_scratch [B*Lq*M*24] / w_scratch [B*Lq*M*12]) + the native-op kernel */
int odise_hip_msda_fused_forward(odise_hip_ctx* ctx, const void* value, const float* off, const float* aw, const int* hs3, const int* ws3, int B, int M, int fused,
                                 void* out, float* loc_scratch, float* w_scratch);

/* stage boundaries of the model calls made while the timeline is on: at each boundary (csrc: stage_mark) an event on the stream the stage
 * enqueues to and the host clock.  _read synchronises the device and writes, relative to the first mark: gpu_ms[i] = when the device reached mark
 * i, host_ms[i] = when the host had enqueued everything before it; names = '\n'-separated.  tools/stage_timeline.py prints both columns. */
int odise_hip_stage_timeline(odise_hip_ctx* ctx, int on);
int odise_hip_stage_timeline_read(odise_hip_ctx* ctx, char* names, int names_cap, float* gpu_ms, double* host_ms, int cap, int* n);

/* encoder prefetch (odise_hip_infer_prefetch), what happened so far on this context: encoders enqueued ahead of their batch, prefetched results the
 * next odise_hip_infer consumed (hits), prepared results that were dropped because another batch came next, registrations that could not be
 * enqueued (the call in progress is unaffected).  Any pointer may be NULL.  tests/test_gpu_fullsize_batch.py asserts hit / drop per call. */
int odise_hip_prefetch_stats(odise_hip_ctx* ctx, int* enqueued, int* hits, int* dropped, int* failed);

/* per-context log of every GEMM / convolution launch the cost model decided (tests print which choices differ between two batch sizes):
 * odise_hip_launch_log(ctx, 1) starts / clears it, (ctx, 0) drops it; _read copies records of 6 ints (conv, M, N, K, tile id, split-K factor) */
int odise_hip_launch_log(odise_hip_ctx* ctx, int on);
int odise_hip_launch_log_read(odise_hip_ctx* ctx, int* out6, int cap, int* n);

/* 1: the feature extractor enqueues everything on one stream; 2 (default): its CLIP -> UNet branch runs on a second stream beside the VAE */
int odise_hip_set_lanes(odise_hip_ctx* ctx, int lanes);

/* the s2..s5 backbone maps still resident in the context's arena after odise_hip_backbone_forward / odise_hip_infer, converted to fp32 NCHW
 * [B,C,h,w] device arrays out4[i] (NULL entries are skipped); shape_bchw4x4 (optional, host) receives the four (B, C, h, w).  What the
 * parity tests feed to the fp32 oracle head to attribute a re-decided query (tests/fullsize.py ideal_on_device_features) */
int odise_hip_backbone_maps(odise_hip_ctx* ctx, float** out4, int* shape_bchw4x4);

/* probe (probe.hip): MFMA output layout (tests/test_gpu_probe.py).  The rate probes and yardstick kernels live in odise_hip_lab.h and only
 * in the measurement build of the library. */
int odise_hip_mfma_probe(odise_hip_ctx* ctx, float* host_out);

/* MaskCLIP's visibility rows (clip.py:288-318: bilinear resize of the mask probabilities to the CLIP input, max over each patch, >= 0.5):
 * out [B][T + Q][ldm] u8 (1 = hidden; rows < T are the image tokens' all-visible rows) from logits [B,Q,h,w] f16; T = (S / patch)^2 + 1.
 * plain = 1 runs the form that interpolates both source rows of every sample row anew (the two forms must agree to the bit). */
int odise_hip_maskclip_token_mask(odise_hip_ctx* ctx, const void* logits_f16, void* out_u8, int B, int Q, int h, int w, int S, int patch, int T,
                                  int64_t ldm, int plain);
#ifdef __cplusplus
}
#endif
#endif /* ODISE_HIP_TOOLS_H */
