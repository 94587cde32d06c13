/*
 * odise_hip.h — C ABI of libodise_hip.so, the MI355X (gfx950) device side of the
 * ODISE panoptic-inference hot path.
 *
 * Every entry point takes plain pointers and sizes (no torch types).  Device pointers
 * are raw HIP device addresses (hipMalloc'ed by odise_hip_malloc or borrowed from a
 * torch-ROCm tensor's data_ptr()).  All functions return 0 on success and a negative
 * code on failure; odise_hip_last_error() returns a human readable message.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   - odise_hip_ms_deform_attn_forward  <->  MSDA.ms_deform_attn_forward
 *       third_party/Mask2Former/mask2former/modeling/pixel_decoder/ops/src/ms_deform_attn.h:25-44
 *       third_party/Mask2Former/mask2former/modeling/pixel_decoder/ops/src/cuda/ms_deform_attn_cuda.cu:25-85
 *   - odise_hip_unet_features           <->  LdmExtractor.unet_forward
 *       odise/modeling/meta_arch/ldm.py:469-491 (taps = concat-inputs of output blocks 2,5,8,11)
 *   - odise_hip_mask_pooling            <->  MaskPooling.forward  odise/modeling/meta_arch/odise.py:937-963
 *   - op-level entry points (gemm / conv / group-norm / layer-norm / attention) are the
 *     building blocks the stage-level calls are made of; they are exported so every
 *     stage can be parity-tested in isolation (SURVEY.md §8b last row).
 */
#ifndef ODISE_HIP_H
#define ODISE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct odise_hip_ctx odise_hip_ctx;

enum { ODISE_F16 = 0, ODISE_F32 = 1 };
enum { ODISE_ACT_NONE = 0, ODISE_ACT_SILU = 1, ODISE_ACT_RELU = 2, ODISE_ACT_GELU = 3, ODISE_ACT_QUICKGELU = 4 };

/* error codes */
enum {
    ODISE_OK = 0,
    ODISE_ERR_ARG = -1,      /* bad argument (shape/alignment/dtype) */
    ODISE_ERR_HIP = -2,      /* a HIP runtime call failed           */
    ODISE_ERR_STATE = -3,    /* missing weights / wrong call order  */
    ODISE_ERR_NOMEM = -4,    /* workspace / arena exhausted         */
    ODISE_ERR_UNSUPPORTED = -5 /* a valid input this library does not handle (e.g. an arithmetic-coded JPEG) */
};

/* ---- context / plumbing ------------------------------------------------------------ */
int odise_hip_create(int device, odise_hip_ctx** out);
int odise_hip_destroy(odise_hip_ctx* ctx);
const char* odise_hip_last_error(void);
int odise_hip_version(void);
/* adopt an external hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int odise_hip_set_stream(odise_hip_ctx* ctx, void* hip_stream);
int odise_hip_malloc(odise_hip_ctx* ctx, size_t bytes, void** dptr);
int odise_hip_free(odise_hip_ctx* ctx, void* dptr);
int odise_hip_memcpy_h2d(odise_hip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int odise_hip_memcpy_d2h(odise_hip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int odise_hip_memset(odise_hip_ctx* ctx, void* dst_dev, int value, size_t bytes);
int odise_hip_sync(odise_hip_ctx* ctx);
/* HIP-event timing on the context's stream (what bench.py uses around the timed region) */
int odise_hip_timer_start(odise_hip_ctx* ctx);
int odise_hip_timer_stop(odise_hip_ctx* ctx, float* elapsed_ms);
int odise_hip_device_info(odise_hip_ctx* ctx, char* name_buf, int buf_len, int* cu_count, size_t* hbm_bytes);
/* Per-context execution options.  They choose between arithmetic-equivalent forms of a stage, never between results and no results:
 *   ODISE_OPT_CLIP_LN_FOLD     how the CLIP towers (clip.py:177-206, 252-323) take their LayerNorms: 0 = folded into the neighbouring GEMMs from
 *                              8192 token rows per call (default; below that as kernels), 1 = always folded, 2 = never.  The two forms differ by
 *                              fp16 rounding, so a caller that needs results independent of how many crops share a call pins 1 or 2.
 *   ODISE_OPT_VAE_CHUNK_BYTES  > 0: the AutoencoderKL levels (ldm.py:493-533, 585-606) run over as many crops per launch as keep one activation
 *                              tensor below this many bytes (a smaller working set and arena); 0 (default) = all crops of a call at once, which
 *                              measured faster on MI355X (profiles/r04_vae_chunking_experiment.txt).  Per-crop arithmetic is unchanged.
 *   ODISE_OPT_PREFETCH_CU_EIGHTHS  1..7: the encoder-prefetch stream (odise_hip_infer_prefetch) is created with a CU mask of that many of every 8
 *                              compute units, so the batch in progress always finds free CUs for its small dependent launches - a CU-masked stream
 *                              has NORMAL priority and default flags (HIP offers neither with a mask); 0 / 8 (default 0) = no mask, lowest
 *                              priority, non-blocking.  Read when the stream is first created.
 *   ODISE_OPT_PREFETCH_START   where the next batch's encoder is enqueued: 0 = behind the current batch's VAE lane, 1 (default) = behind its
 *                              backbone, i.e. beside the serial tail of small launches (pixel decoder .. post-processing)
 *   ODISE_OPT_ATTN_KV_RESIDENT 0 (default) = attention with d_head 64 and at most 608 keys runs the K / V^T-resident kernel where (head, image)
 *                              pairs fill the chip in whole rounds (the CLIP tower of 16 / 32 crops), unmasked self-attention over whole
 *                              128-query / 64-key tiles (the SD UNet's 64^2 and 32^2 levels) the software-pipelined kernel, the tiled kernel
 *                              elsewhere; 2 = never the K / V^T-resident kernel, 4 = never the pipelined one, 6 = always the tiled kernel.  The
 *                              resident form steps the running softmax maximum per 32 instead of 64 keys (results agree to fp32 rounding);
 *                              the pipelined form is bit-identical to the tiled one.
 *   ODISE_OPT_MASKCLIP_PASSES  MaskCLIP (clip.py:252-323) hides the mask tokens from every query (clip.py:314-315), so the 577 image tokens of a
 *                              picture run the plain tower whatever the masks are and the Q mask tokens only read its keys and values.
 *                              0 (default) = two passes: the image tokens leave q|k and V^T of every block, the mask tokens follow as B x Q
 *                              rows; in odise_hip_infer the pictures' image tokens RIDE IN THE CROPS' CLIP TOWER of the implicit captioner (the
 *                              same frozen ViT-L/14@336: pictures + crops are one batch of token rows on the second lane), so only the
 *                              mask-token pass is left on the serial tail behind the mask head.  1 = two passes, both where the reference runs
 *                              the tower (after the mask head).  2 = one pass over [577 image | Q mask] token rows (the reference's layout).
 *                              3 = as 0 with the first pass as a tower of its own on the second lane behind the UNet.  The forms run the same
 *                              arithmetic per row on different GEMM tiles and LayerNorm forms (CLIP_LN_FOLD counts the rows of the tower the
 *                              tokens are in): results agree to fp16 rounding.  odise_hip_classify / odise_hip_maskclip_embed called on their
 *                              own run 0 and 3 as 1. */
enum { ODISE_OPT_CLIP_LN_FOLD = 1, ODISE_OPT_VAE_CHUNK_BYTES = 2, ODISE_OPT_ATTN_KV_RESIDENT = 3, ODISE_OPT_PREFETCH_CU_EIGHTHS = 4, ODISE_OPT_PREFETCH_START = 5,
       ODISE_OPT_MASKCLIP_PASSES = 6 };
int odise_hip_set_option(odise_hip_ctx* ctx, int option, int64_t value);
int odise_hip_get_option(odise_hip_ctx* ctx, int option, int64_t* value);
/* Launch probe (measurement, bench.py's `roofline`): HIP events around every launch of ONE shape - conv != 0: the implicit GEMM of a convolution
 * with M = N*OH*OW output pixels, N = Cout, K = KH*KW*Cin (convolutions with the fused 2x upsample are not matched); conv = 0: odise_hip_gemm's M, N, K - recorded on whichever stream the library
 * launches it on, for the next max_launches matching launches.  odise_hip_probe_read waits for the recorded launches, writes their durations
 * in microseconds (at most cap) and the number recorded, and disarms the probe. */
int odise_hip_probe_arm(odise_hip_ctx* ctx, int conv, int M, int N, int K, int max_launches);
int odise_hip_probe_read(odise_hip_ctx* ctx, float* us_out, int cap, int* n_launches);

/* ---- MSDeformAttn forward ------------------------------ */
/* value        [B, S, M, D]      dtype value_dtype (device)
 * spatial_shapes [L, 2] int64 (HOST)   (H_l, W_l)
 * level_start_index [L] int64 (HOST)
 * sampling_loc [B, Lq, M, L, P, 2] f32 (device), normalised to [0,1]
 * attn_weight  [B, Lq, M, L, P]   f32 (device)
 * out          [B, Lq, M*D]       dtype value_dtype (device)
 * im2col_step is accepted for signature parity and validated like the reference
 * (B % min(B, im2col_step) == 0, ms_deform_attn_cuda.cu:52) but the kernel is one launch. */
int odise_hip_ms_deform_attn_forward(odise_hip_ctx* ctx, const void* value, const int64_t* spatial_shapes,
                                     const int64_t* level_start_index, const float* sampling_loc,
                                     const float* attn_weight, int B, int S, int M, int D, int Lq, int L, int P,
                                     int im2col_step, int value_dtype, void* out);

/* ---- GEMM: C[M,N] = epilogue(alpha * A[M,K] * W[N,K]^T) -----------------------------
 * Replaces every nn.Linear / projection of the path (W in PyTorch's [out, in] layout, bias_n = bias): the q/k/v/out projections and
 * FFNs of M2F/modeling/transformer_decoder/mask2former_transformer_decoder.py:17-204 and of MSDeformAttn (ops/modules/ms_deform_attn.py:98-125),
 * PooledMaskEmbed's pool_proj / mask_embed MLPs (odise/modeling/meta_arch/odise.py:975-1009), PositionalLinear (ldm.py:624-635), the
 * CLIP and SD-UNet transformer blocks driven from clip.py:252-280 / ldm.py:469-491 (GEGLU, per-row terms and row-group adds are their
 * fused tails), text_proj and the cosine logits of odise.py:181-207. */
typedef struct {
    int M, N, K;              /* K % 8 == 0 */
    const void* A; int64_t lda;   /* f16, row-major, lda % 8 == 0, 16-byte aligned */
    const void* W; int64_t ldw;   /* f16, row-major [N,K] ("B^T" form) */
    void* C; int64_t ldc;
    int c_dtype;              /* ODISE_F16 | ODISE_F32 */
    const float* bias_n;      /* [N] or NULL */
    const float* bias_m;      /* [M] or NULL */
    const float* scale_m;     /* [M] or NULL: per-row multiplier applied to the accumulator first */
    const void* residual;     /* f16 [M, ldr] or NULL (added after activation) */
    int64_t ldr;
    const float* rowgroup_add;/* [ceil(M/rows_per_group), N] f32 or NULL (added before activation) */
    int rows_per_group;
    int64_t ldg;              /* row stride of rowgroup_add (0 -> N) */
    int act;                  /* ODISE_ACT_* */
    int geglu;                /* 1: columns are interleaved (a,gate) pairs -> out[M,N/2] = a*gelu(gate) */
    float alpha;
    int batch;                /* >=1; batched over grid.z with element strides below */
    int64_t strideA, strideW, strideC, strideR;
} odise_gemm_desc;
int odise_hip_gemm(odise_hip_ctx* ctx, const odise_gemm_desc* d);

/* ---- implicit-GEMM convolution, NHWC f16 --------------------------------------------
 * Replaces nn.Conv2d / detectron2.layers.Conv2d wherever the path convolves: the SD UNet and VAE blocks walked by ldm.py:424-533
 * (incl. Upsample = nearest 2x + 3x3 conv -> upsample2x, the time-embedding broadcast -> per_image_add), the BottleneckBlock projections
 * of feature_extractor.py:53-66, the pixel decoder's input_proj / lateral / output / mask_features convs
 * (M2F/modeling/pixel_decoder/msdeformattn.py:206-286) and CLIP's patch embedding (clip.py:253). */
typedef struct {
    int N, H, W, Cin;         /* input  X [N,H,W,Cin] f16, Cin % 8 == 0 */
    int Cout, KH, KW, stride, pad_t, pad_l, OH, OW;
    int upsample2x;           /* 1: conv runs on nearest-2x upsampled X (fused gather) */
    const void* X;
    const void* Wt;           /* [Cout, KH, KW, Cin] f16 */
    void* Y; int y_dtype;     /* [N,OH,OW,Cout] */
    const float* bias;        /* [Cout] or NULL */
    const void* residual;     /* f16 [N,OH,OW,Cout] or NULL */
    const float* per_image_add;/* [N, Cout] f32 or NULL (time-embedding broadcast) */
    int64_t per_image_add_ld; /* row stride of per_image_add (0 -> Cout) */
    int act;
} odise_conv_desc;
int odise_hip_conv2d(odise_hip_ctx* ctx, const odise_conv_desc* d);

/* ---- normalisation -------------------------------------------------------------------
 * GroupNorm: nn.GroupNorm(32, C) of msdeformattn.py:218-225 / get_norm("GN") of the detectron2 Conv2d wrappers and BottleneckBlocks,
 * ldm's Normalize in every ResnetBlock / ResBlock (+ fused SiLU).  LayerNorm: decoder norms (mask2former_transformer_decoder.py:26, 84,
 * 149), decoder_norm, PooledMaskEmbed (odise.py:975-977), CLIP ln_pre / ln_1 / ln_2 / ln_post (clip.py:264-276). */
/* GroupNorm over NHWC f16 x[N,HW,C]; stats in fp32; y = act(gn(x)*gamma+beta) (f16) */
int odise_hip_group_norm(odise_hip_ctx* ctx, const void* x, void* y, const float* gamma, const float* beta,
                         int N, int HW, int C, int groups, float eps, int act);
/* y = act(GroupNorm(x) + residual) + accum; residual / accum optional f16 tensors shaped like x
 * (detectron2 BottleneckBlock tail and the per-stride sum of feature_extractor.py:171-176) */
int odise_hip_group_norm_ex(odise_hip_ctx* ctx, const void* x, void* y, const float* gamma, const float* beta,
                            int N, int HW, int C, int groups, float eps, int act, const void* residual, const void* accum);
/* LayerNorm over the last dim of x[rows, C] f16 -> y f16 */
int odise_hip_layer_norm(odise_hip_ctx* ctx, const void* x, void* y, const float* gamma, const float* beta,
                         int rows, int C, float eps);

/* ---- fused attention -------------------------------------------------------------------
 * Replaces nn.MultiheadAttention of the masked decoder (mask2former_transformer_decoder.py:22, 80: self-attention and cross-attention
 * with the boolean attn_mask of odise.py:763-775), CLIP's transformer with the mask-token attention mask (clip.py:271, 300-323) and the
 * SD UNet's CrossAttention (self and text-conditioned, d_head 40..160) reached through ldm.py:469-491. */
/* O[b,q,h*D+d] = softmax_k(scale * Q[b,q,h*D+:] . K[b,k,h*D+:] + mask) V
 * Q  [B, Lq, ldq] f16, K [B, Lk, ldk] f16, Vt [B, H*D, ldvt] f16 (V TRANSPOSED: row h*D+d, col key),
 * O  [B, Lq, ldo] f16.  mask: optional u8 [B, Lq, ldmask] (1 = key not visible), shared by heads.
 * D in {32,40,64,80,160}; ldvt % 8 == 0, ldmask % 4 == 0. */
typedef struct {
    int B, H, Lq, Lk, D;
    const void* Q; int64_t ldq, strideQ;
    const void* K; int64_t ldk, strideK;
    const void* Vt; int64_t ldvt, strideVt;
    void* O; int64_t ldo, strideO;
    const uint8_t* mask; int64_t ldmask, strideMask;
    float scale;
} odise_attn_desc;
int odise_hip_attention(odise_hip_ctx* ctx, const odise_attn_desc* d);

/* ---- layout / elementwise helpers ---------------------------------------------------- */
int odise_hip_nchw_f32_to_nhwc_f16(odise_hip_ctx* ctx, const float* x, void* y, int N, int C, int H, int W, int Cpad);
int odise_hip_nhwc_f16_to_nchw_f32(odise_hip_ctx* ctx, const void* x, float* y, int N, int C, int H, int W);
int odise_hip_cast_f32_to_f16(odise_hip_ctx* ctx, const float* x, void* y, size_t n);
int odise_hip_cast_f16_to_f32(odise_hip_ctx* ctx, const void* x, float* y, size_t n);
/* y[n,p,:] = cat(a[n,p,:Ca], b[n,p,:Cb]) (f16, channels-last) */
int odise_hip_concat_channels(odise_hip_ctx* ctx, const void* a, const void* b, void* y, size_t pixels, int Ca, int Cb);

/* ---- MaskPooling (odise.py:937-963) --------------------------------------------------- */
/* x [B,C,H,W] f32 NCHW, mask logits [B,Q,H,W] f32 -> pooled [B,Q,C] f32
 * pooled = einsum(x, sigmoid(mask)>0.5) / (sum(mask>0) + 1e-8) */
int odise_hip_mask_pooling(odise_hip_ctx* ctx, const float* x, const float* mask, float* pooled,
                           int B, int C, int Q, int HW);

/* ---- SD v1 UNet single-step feature extraction (ldm.py:469-491) ----------------------- */
/* Weights are registered by their checkpoint key (model.diffusion_model.* stripped of that prefix),
 * host fp32, any rank<=4; the library converts to its packed fp16 layouts.  */
int odise_hip_load_weight(odise_hip_ctx* ctx, const char* name, const float* host_data, const int64_t* shape, int ndim);
int odise_hip_unet_build(odise_hip_ctx* ctx);     /* after all weights are loaded; packs + uploads */
int odise_hip_clear_host_weights(odise_hip_ctx* ctx);  /* drop the host fp32 staging copies after build */
/* x_t [B,4,h,w] f32 NCHW (device), context [B,77,768] f32 (device), cond_emb [B,1280] f32 or NULL (device).
 * taps: u2 [B,2560,h/8,w/8], u5 [B,1920,h/4,w/4], u8 [B,960,h/2,w/2], u11 [B,640,h,w] f32 NCHW (device), any may be NULL.
 * timestep t (the reference uses t=0).  */
int odise_hip_unet_features(odise_hip_ctx* ctx, const float* x_t, const float* context, const float* cond_emb,
                            int B, int h, int w, int t, float* tap_u2, float* tap_u5, float* tap_u8, float* tap_u11);
/* same, but keeps taps as fp16 NHWC inside the context (bench / fused pipeline path); returns device pointers */
int odise_hip_unet_features_nhwc(odise_hip_ctx* ctx, const float* x_t, const float* context, const float* cond_emb,
                                 int B, int h, int w, int t, void** taps4);
/* analytic MACs of the last unet call (per the layer shapes actually launched) */
int odise_hip_unet_last_macs(odise_hip_ctx* ctx, double* macs);
/* capture the unet forward for (B,h,w) into a hipGraph and replay it on subsequent calls (0 disables) */
int odise_hip_unet_use_graph(odise_hip_ctx* ctx, int enable);

/* ---- LdmImplicitCaptionerExtractor.forward (ldm.py:697-718 -> 543-621) ------------------------------ */
/* Weights (host fp32, checkpoint keys): first_stage_model.* and model.diffusion_model.* (SD v1 ckpt "state_dict"),
 * clip.visual.* (OpenAI ViT-L-14-336px archive, prefixed with "clip."), backbone.feature_extractor.{clip_project.*,
 * alpha_cond, time_embed_project.*, alpha_cond_time_embed} (ODISE ckpt "model"), plus the two frozen buffers
 * backbone.feature_extractor.ldm_extractor.{ldm.uncond_inputs [1,77,768], shared_noise [1,4,64,64]}. */
int odise_hip_extractor_build(odise_hip_ctx* ctx);
/* image [B,3,H,W] f32 device in [0,1] (H,W multiples of 64; reference crops are 512x512).  taps8: 8 device pointers
 * (fp32 NCHW, any may be NULL) in the reference's order enc5, enc7, u2, u5, u8, u11, dec2, dec5 (ldm.py:608). */
int odise_hip_extractor_forward(odise_hip_ctx* ctx, const float* image, int B, int H, int W, float** taps8);
/* hot-path variant: taps stay fp16 NHWC inside the library arena; returns pointers and [n,c,h,w] per tap */
int odise_hip_extractor_forward_nhwc(odise_hip_ctx* ctx, const float* image, int B, int H, int W, void** taps8, int* shapes8x4);
int odise_hip_extractor_last_macs(odise_hip_ctx* ctx, double* macs);

/* ---- FeatureExtractorBackbone (feature_extractor.py:139-250): slide-window crops -> extractor -> projections -> stitch ---- */
/* extra weights: backbone.feature_projections.{0..7}.0.{conv1,conv2,conv3[,shortcut]}.{weight,norm.weight,norm.bias} */
int odise_hip_backbone_build(odise_hip_ctx* ctx);
/* image [B,3,H,W] f32 device in [0,1], H,W >= 512 and multiples of 64.  out4: s2,s3,s4,s5 fp32 NCHW [B,512,H/4..H/32,..]
 * device pointers (array or entries may be NULL); fp16 NHWC copies stay inside the library for odise_hip_head_forward. */
int odise_hip_backbone_forward(odise_hip_ctx* ctx, const float* image, int B, int H, int W, float** out4);

/* ---- MaskFormerHead: MSDeformAttn pixel decoder + ODISE masked transformer decoder (msdeformattn.py:314-358, odise.py:642-776) */
/* weights: sem_seg_head.pixel_decoder.*, sem_seg_head.predictor.* */
int odise_hip_head_build(odise_hip_ctx* ctx);
/* feats4: s2..s5 fp32 NCHW device pointers [B,Cin,H4>>i,W4>>i], or NULL to consume the last backbone_forward.
 * outputs (device fp32, any may be NULL): pred_masks [B,Q,H4,W4] logits, mask_embed [B,Q,C], mask_pooled_features [B,Q,C];
 * logit_scale (host) = clamp(exp(logit_scale), max=100). */
int odise_hip_head_forward(odise_hip_ctx* ctx, const float* const* feats4, int B, int Cin, int H4, int W4, float* pred_masks,
                           float* mask_embed, float* mask_pooled, float* logit_scale);
int odise_hip_maskgen_info(odise_hip_ctx* ctx, int* num_queries, int* hidden_dim, double* last_macs);
/* the two halves of the head on their own (the reference's modules are callable one by one):
 * MSDeformAttnPixelDecoder.forward_features (msdeformattn.py:314-358): feats4 = s2..s5 fp32 NCHW -> mask_features [B,C,H4,W4] and the three
 * multi-scale maps [B,C,H4/8,W4/8], [B,C,H4/4,W4/4], [B,C,H4/2,W4/2] (fp32 NCHW device, any may be NULL) */
int odise_hip_pixel_decoder_forward(odise_hip_ctx* ctx, const float* const* feats4, int B, int Cin, int H4, int W4, float* mask_features,
                                    float* const* multi_scale3);
/* ODISEMultiScaleMaskedTransformerDecoder.forward (odise.py:642-727): multi_scale3 = 3 fp32 NCHW maps of sizes hw3 [3][2] (HOST),
 * mask_features [B,C,H4,W4]; outputs as odise_hip_head_forward (and kept inside for odise_hip_classify / postprocess) */
int odise_hip_predictor_forward(odise_hip_ctx* ctx, const float* const* multi_scale3, const int* hw3, const float* mask_features, int B, int H4, int W4,
                                float* pred_masks, float* mask_embed, float* mask_pooled, float* logit_scale);

/* ---- open-vocabulary classification (odise.py:285-323; clip.py:252-361; helper.py:79-109) ---------------------------- */
/* weights: category_head.text_proj.{weight,bias}, category_head.null_embed (ODISE ckpt) + the CLIP tower of the extractor */
int odise_hip_classify_build(odise_hip_ctx* ctx);
/* vocabulary = CLIP text embeddings (HOST fp32) of the two prompt sets the reference builds per label tuple
 * (category_head: plain labels; clip_head: "a photo of a {}."), K synonym groups of group_sizes[k] strings each,
 * overlap[k] = 1 if the class overlaps a training class (category_overlapping_mask, odise.py:1479-1491), alpha/beta of clip_head */
int odise_hip_set_vocabulary(odise_hip_ctx* ctx, const float* cat_text, const float* clip_text, int K_tot, int dim,
                             const int* group_sizes, const int* overlap, int K, float alpha, float beta);
/* image [B,3,H,W] fp32 device in [0,1]; uses the last head_forward; mask_cls [B,Q,K+1] fp32 device (log-probabilities);
 * clip_embed (optional) [B,Q,dim] fp32 device = MaskCLIP.get_mask_embed */
int odise_hip_classify(odise_hip_ctx* ctx, const float* image, int B, int H, int W, float* mask_cls, float* clip_embed);
/* MaskCLIP.get_mask_embed stand-alone (clip.py:325-338; the arithmetic of PoolingCLIPHead.forward, odise.py:1469-1542): image [B,3,H,W] fp32 in
 * [0,1], pred_masks [B,Q,h,w] fp32 logits (device) -> clip_embed [B,Q,dim] fp32.  Needs the CLIP tower (odise_hip_extractor_build). */
int odise_hip_maskclip_embed(odise_hip_ctx* ctx, const float* image, int B, int H, int W, const float* pred_masks, int Q, int h, int w, float* clip_embed);

/* ---- post-processing (odise.py:326-370; maskformer_model.py:280-380) ------------------------------------------------- */
/* fused mask upsample (x4 bilinear to pad_h x pad_w, crop img_h x img_w, bilinear to out_h x out_w) + sigmoid +
 * semantic einsum + panoptic argmax and area counters for image b of the last head_forward.
 *   kscore [Q] fp32 device: score of kept queries (label != null, score > object_mask_threshold), < 0 for dropped queries
 *   semT   [K,Q] fp32 device = softmax(mask_cls)[:, :-1]^T, or NULL
 *   sem_seg [K,out_h,out_w] fp32 / ids [out_h*out_w] int32 (kept-query index | 1<<16 if inside the mask, -1 if none) /
 *   counts [3*Q] int32 (mask_area, original_area, intersection) / inst_stats [2*Qpad] fp32 (sum of prob over positive pixels,
 *   positive pixel count; Qpad = Q rounded up to 8): device pointers, each optional */
int odise_hip_postprocess_pixels(odise_hip_ctx* ctx, int b, const float* kscore, const float* semT, int K, int pad_h, int pad_w,
                                 int img_h, int img_w, int out_h, int out_w, float* sem_seg, int* ids, int* counts, float* inst_stats);
/* seg[p] = map[q] for pixels whose argmax query q is inside its own mask, else 0 (maskformer_model.py:321-333) */
int odise_hip_panoptic_write(odise_hip_ctx* ctx, const int* ids, const int* map, int* seg, int npix);
/* out [n,out_h,out_w] fp32 = (upsampled mask logit of query idx[i] > 0)  (maskformer_model.py:371) */
int odise_hip_instance_masks(odise_hip_ctx* ctx, int b, const int* idx, int n, int pad_h, int pad_w, int img_h, int img_w, int out_h,
                             int out_w, float* out);

/* ---- the three inference heads for a whole batch, decisions on the device (odise.py:336-370; maskformer_model.py:280-380) --------
 * Consumes the mask logits of the last odise_hip_head_forward and mask_cls of odise_hip_classify.  Everything the reference decides
 * on the host from device tensors - kept queries, the per-segment loop with its `.item()` round trips (maskformer_model.py:312-340),
 * the top-k of the instance head (:349-369) - is decided by kernels, so a call enqueues work and never synchronises; the caller reads
 * the small tables back when it needs them.  Per image b the outputs are (HOST arrays of B device pointers; an array or an entry may
 * be NULL to skip that output):
 *   sem_seg[b]     fp32 [K, oh, ow]                    semantic_inference (:280-284)
 *   sem_argmax[b]  int32 [oh*ow]                       argmax over classes of the same scores WITHOUT materialising [K, oh, ow]
 *                                                      (what detectron2's SemSegEvaluator keeps; A-847 at 1280x1280 is 5.5 GB otherwise)
 *   panoptic[b]    int32 record [oh*ow | 1 | 3*ODISE_MAX_SEGMENTS]: panoptic ids, n_segments, (id, isthing, category_id) rows -
 *                  the per-image record of the multi-GPU exchange (odise_hip_allgather_predictions)
 *   inst_masks[b]  fp32 [topk, oh, ow] (first n valid) ; inst_table (device, [B][1 + 2*topk] int32: n | query index | class) ;
 *                  inst_scores (device, [B][topk] fp32), sorted by class score descending (instance_inference, :344-380) */
#define ODISE_MAX_SEGMENTS 100
typedef struct {
    int B;                        /* batch of the last head_forward */
    int pad_h, pad_w;             /* padded network input size (ImageList.from_tensors(images, size_divisibility), odise.py:240) */
    const int* img_hw;            /* HOST [B][2] true image sizes (the crop of sem_seg_postprocess) */
    const int* out_hw;            /* HOST [B][2] requested output sizes ("height" / "width" of the input dicts) */
    const float* mask_cls;        /* device [B,Q,K+1] log-probabilities (output of odise_hip_classify) */
    const uint8_t* isthing;       /* HOST [K]: 1 for "thing" classes (metadata.thing_dataset_id_to_contiguous_id) */
    int semantic_on, panoptic_on, instance_on;
    float object_mask_threshold;  /* maskformer_model.py:290 */
    double overlap_threshold;     /* :318 (double: compared against an integer ratio exactly like the reference's Python float) */
    int topk;                     /* test_topk_per_image */
    float* const* sem_seg;
    int32_t* const* sem_argmax;
    int32_t* const* panoptic;
    float* const* inst_masks;
    int32_t* inst_table;
    float* inst_scores;
} odise_post_desc;
int odise_hip_postprocess_batch(odise_hip_ctx* ctx, const odise_post_desc* d);

/* ---- CategoryODISE.forward / CaptionODISE.forward, eval branch, as ONE call (odise.py:236-246, 282-372; the call
 * `self.model(batched_inputs)` of OpenPanopticInference.forward, odise/modeling/wrapper/pano_wrapper.py:64) ------------------------
 * images[b]: device pointer to image b, uint8 [h_b, w_b, 3] (image_layout 0: HWC, the DatasetMapper's decoded picture),
 * uint8 [3, h_b, w_b] (1: CHW, the "image" tensor of the reference's input dicts) or fp32 [3, h_b, w_b] with values 0..255 (2).
 * The library normalises ((x - 0) / 255), pads to the batch maximum rounded up to 64 (backbone input) and to the batch maximum
 * (MaskCLIP input, odise.py:242-244), runs backbone -> head -> classification -> the three heads.  post.B / pad_h / pad_w / img_hw /
 * mask_cls are filled in by the library (post.out_hw NULL = image sizes).  mask_cls_out (optional): device [B,Q,K+1] fp32. */
typedef struct {
    int B;
    const void* const* images;    /* HOST [B] device pointers */
    int image_layout;
    const int* img_hw;            /* HOST [B][2] */
    float* mask_cls_out;
    odise_post_desc post;
} odise_infer_desc;
int odise_hip_infer(odise_hip_ctx* ctx, const odise_infer_desc* d);
/* Encoder prefetch for a stream of batches (the reference's evaluation loop consumes a prefetching loader: odise/evaluation/evaluator.py:
 * 87-126, odise/data/build.py:138-151).  Registers the NEXT batch (only B / images / image_layout / img_hw of `next` are read; the image
 * buffers must stay valid and unchanged until that batch's own odise_hip_infer returns): the following odise_hip_infer call, once the VAE
 * lane of ITS batch is done, enqueues the next batch's normalise / pad, window extraction, VAE encoder and latent on a lowest-priority
 * stream into a side arena, and the odise_hip_infer of exactly that batch (same pointers, layout and sizes) starts from the stored latent
 * and encoder taps.  Same kernels on the same shapes: every output is bit-identical to the call without prefetch.  A prefetched batch that
 * is not the next one inferred is dropped.  next == NULL cancels a registration. */
int odise_hip_infer_prefetch(odise_hip_ctx* ctx, const odise_infer_desc* next);

/* ---- input resize and evaluator reductions of the eval loop --------------------------------------------
 * Replaces, on device buffers: detectron2 T.ResizeShortestEdge -> PIL.Image.resize(BILINEAR) of the DatasetMapper
 * (configs/common/data/pano_open_d2_eval.py:74-107) and the per-pixel parts of the evaluators configured there
 * (odise/evaluation/d2_evaluator.py:49 COCOPanopticEvaluator -> panopticapi pq_compute_single_core, :63 SemSegEvaluator.process). */
/* src uint8 [H,W,C] -> dst uint8 [OH,OW,C], bit-identical to Pillow's 8-bit bilinear resampler (horizontal pass, then vertical) */
int odise_hip_resize_bilinear_u8(odise_hip_ctx* ctx, const void* src, int H, int W, int C, void* dst, int OH, int OW);
/* dst fp32 [C,H,W] = scale * src uint8 [H,W,C] */
int odise_hip_u8_hwc_to_f32_chw(odise_hip_ctx* ctx, const void* src, float* dst, int H, int W, int C, float scale);
/* dst fp32 [C,Hp,Wp] = the same, zero padded at the bottom / right (detectron2 ImageList.from_tensors(images, size_divisibility),
 * odise.py:240) */
int odise_hip_u8_hwc_to_f32_chw_padded(odise_hip_ctx* ctx, const void* src, float* dst, int H, int W, int C, int Hp, int Wp, float scale);
/* conf int64 [(K+1)*(K+1)] += count of (argmax_k sem_seg[k,p], gt[p]); gt outside [0,K] (ignore label) counts in column K.
 * The caller zeroes conf before the first image and sums it across ranks (all-gather / all-reduce of (K+1)^2 int64). */
int odise_hip_semantic_confusion(odise_hip_ctx* ctx, const float* sem_seg, const int* gt, int K, int npix, int64_t* conf);
/* hist int32 [na*nb] += count of (a[p], b[p]) pairs with 0 <= a < na, 0 <= b < nb (segment-index co-occurrence of PQ matching) */
int odise_hip_pair_histogram(odise_hip_ctx* ctx, const int* a, const int* b, int npix, int na, int nb, int* hist);

/* ---- JPEG input -------------------------------------------------------------------------------------------
 * Replaces detectron2 `read_image(file, "RGB")` = PIL.Image.open -> EXIF transpose -> convert("RGB") of the DatasetMapper
 * (configs/common/data/pano_open_d2_eval.py:74-107; demo/demo.py:399) for baseline JPEG files, bit-identical to Pillow /
 * libjpeg-turbo defaults (islow IDCT, fancy upsampling).  Huffman decoding runs on the host, everything per-sample on the device.
 * 8-bit Huffman files: baseline / extended sequential (SOF0 / SOF1) and progressive (SOF2), grey or YCbCr with luma sampling
 * 1x1 / 2x1 / 2x2; anything else (arithmetic coding, lossless, CMYK, RGB-coded): ODISE_ERR_UNSUPPORTED. */
typedef struct odise_jpeg_info {
    int32_t width, height;        /* coded size (before the EXIF orientation is applied) */
    int32_t components;           /* 1 (grey) or 3 (YCbCr) */
    int32_t h_samp, v_samp;       /* luma sampling factors */
    int32_t orientation;          /* EXIF tag 0x0112, 1..8 (1 when absent) */
    int32_t restart_interval;     /* MCUs, 0 = none */
    int32_t blocks_x[3], blocks_y[3]; /* 8x8-block grid of every component (padded to whole MCUs) */
    int64_t coef_count;           /* int16 coefficients of all components */
} odise_jpeg_info;
/* host only: parse the headers of a JPEG byte stream */
int odise_hip_jpeg_info(const void* data, int64_t len, odise_jpeg_info* info);
/* host only: entropy-decode into coefs int16 [coef_count] (component after component, [blocks_y][blocks_x][64], natural order,
 * not dequantised); qtables (optional) uint16 [components*64] receives every component's quantisation table in natural order */
int odise_hip_jpeg_entropy_decode(const void* data, int64_t len, int16_t* coefs, int64_t capacity, uint16_t* qtables);
/* data: host bytes of the file.  dst_rgb: device uint8 [out_h,out_w,3] (capacity in bytes); the decoded size is returned in
 * out_h / out_w (height and width swap for EXIF orientations 5..8 when apply_orientation != 0).  Asynchronous on the context's stream. */
int odise_hip_jpeg_decode(odise_hip_ctx* ctx, const void* data, int64_t len, void* dst_rgb, int64_t dst_capacity, int apply_orientation,
                          int* out_h, int* out_w);
/* the device half on its own: coefficients and tables as produced by odise_hip_jpeg_entropy_decode (which is thread-safe and can run
 * in loader threads while this context is busy with the previous image) */
int odise_hip_jpeg_decode_coefs(odise_hip_ctx* ctx, const odise_jpeg_info* info, const int16_t* coefs, const uint16_t* qtables, void* dst_rgb,
                                int64_t dst_capacity, int apply_orientation, int* out_h, int* out_w);

/* ---- multi-GPU exchange step ---------------------------------------------------------------------------------------
 * One process per GPU; images are independent units sharded across ranks, there is no collective inside the model forward, and
 * exactly one exchange: an RCCL all-gather (xGMI) of fixed-size int32 prediction records, replacing detectron2's pickled
 * `comm.gather` of per-image predictions reached from odise/evaluation/evaluator.py:144 (inference_on_dataset -> evaluator.evaluate)
 * and, for semantic evaluation, the gather behind SemSegEvaluator.evaluate (odise/evaluation/d2_evaluator.py:63) -> one all-reduce of
 * the (K+1)^2 int64 confusion counters.  The communicator belongs to the context; collectives run on a second HIP stream ordered after
 * the work queued on the compute stream so far, so the next batch's kernels overlap the exchange.  A world of one rank is valid. */
#define ODISE_COMM_ID_BYTES 128
/* rank 0: fill id128 (ODISE_COMM_ID_BYTES host bytes) - the launcher broadcasts it to the other ranks over its CPU rendezvous */
int odise_hip_comm_unique_id(void* id128);
int odise_hip_comm_init(odise_hip_ctx* ctx, const void* id128, int rank, int world);
int odise_hip_comm_destroy(odise_hip_ctx* ctx);
int odise_hip_comm_info(odise_hip_ctx* ctx, int* rank, int* world);   /* world = 0 when no communicator exists */
/* all [world * count] = concatenation over ranks of local [count] (device int32; count identical on every rank).  Record layout per
 * image (odise_amd/distributed.py): panoptic_seg [H*W] | n_segments | segments [100][3] = (id, isthing, category_id).  Asynchronous. */
int odise_hip_allgather_predictions(odise_hip_ctx* ctx, const int32_t* local, int64_t count, int32_t* all);
/* The same exchange for UNEVEN shards: this rank contributes n_records (0 .. max_records) records of record_len int32 each; max_records is
 * identical on every rank (the largest shard: ceil(images / world) per batch, known to every rank without communication).  all
 * [world * max_records * record_len]: rank r's records at row r * max_records, its missing rows filled with -1 (a valid record never starts
 * with -1: its first element is a panoptic id >= 0).  local may alias this rank's slice of `all`.  Asynchronous. */
int odise_hip_allgather_records(odise_hip_ctx* ctx, const int32_t* local, int n_records, int max_records, int64_t record_len, int32_t* all);
/* data [count] int64 device, summed in place over ranks (confusion matrices of odise_hip_semantic_confusion).  Asynchronous. */
int odise_hip_allreduce_sum_i64(odise_hip_ctx* ctx, int64_t* data, int64_t count);
/* join the last collective: block_host != 0 waits on the host, otherwise makes the compute stream wait for it */
int odise_hip_comm_wait(odise_hip_ctx* ctx, int block_host);

#ifdef __cplusplus
}
#endif
#endif /* ODISE_HIP_H */
