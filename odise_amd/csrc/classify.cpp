// classify.cpp — open-vocabulary classification and post-processing of CategoryODISE's eval branch
// (odise/modeling/meta_arch/odise.py:282-372):
//   CategoryEmbed.forward (eval) + cal_pred_logits          odise.py:1290-1307, 181-207
//   PoolingCLIPHead.forward -> MaskCLIP                     odise.py:1469-1542, clip.py:252-361
//   merge with the null probability                         odise.py:300-323
//   mask upsample + sem_seg_postprocess + semantic / panoptic / instance inference
//                                                           odise.py:326-370, M2F/maskformer_model.py:280-380
// The vocabulary arrives as precomputed CLIP text embeddings (the reference caches them per label tuple, odise.py:1281-1288);
// the text tower itself is a later row of SURVEY.md §8f.  Data-dependent host loops of the reference (per-segment `.item()`
// round trips, maskformer_model.py:312-340) are replaced by one fused per-pixel kernel that produces integer area counters,
// so the host decides segments from 3*Q integers per image.
#include <math.h>
#include <string.h>

#include "engine.h"

namespace odise {

struct ClassifyModel {
    bool built = false;
    bool caption = false;  // CaptionODISE: word_head.text_proj, no null embedding, learned binary class head (odise.py:545-569)
    LinW text_proj;
    const HostTensor* null_host = nullptr;
    std::vector<float> null_embed;
    // vocabulary
    bool has_vocab = false;
    int K = 0, Ktot = 0, dim = 0, pdim = 0;
    f16* T1 = nullptr;   // [Ktot+1, pdim]  normalised text_proj(text bank) + null row
    f16* T2 = nullptr;   // [Ktot, dim]     normalised MaskCLIP text bank
    int* seg = nullptr;  // [K+1]
    int* ovl = nullptr;  // [K]
    float alpha = 0.3f, beta = 0.7f;
};

void classify_destroy(ModelStore* ms) {
    delete ms->classify;
    ms->classify = nullptr;
}

static int dev_upload(odise_hip_ctx* ctx, ModelStore* ms, const void* host, size_t bytes, void** dev) {
    ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ODISE_CHECK_HIP(hipMalloc(dev, bytes ? bytes : 16));
    ms->dev_allocs.push_back(*dev);
    ODISE_CHECK_HIP(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return ODISE_OK;
}

static int classify_build(odise_hip_ctx* ctx) {
    ModelStore* ms = store_of(ctx);
    classify_destroy(ms);
    ClassifyModel* c = new ClassifyModel();
    ms->classify = c;
    // CategoryODISE: category_head.{text_proj, null_embed} (odise.py:1236-1241); CaptionODISE: word_head.text_proj only (odise.py:1040-1060)
    Packer pcat{ctx, ms, "category_head.", ""};
    Packer pword{ctx, ms, "word_head.", ""};
    c->caption = pcat.find("text_proj.weight") == nullptr && pword.find("text_proj.weight") != nullptr;
    Packer& pk = c->caption ? pword : pcat;
    ODISE_TRY(pk.linear("text_proj", c->text_proj));
    if (c->caption) {
        c->null_embed.assign((size_t)c->text_proj.in, 0.f);  // the null row of the bank is unused in this variant
    } else {
        const HostTensor* ne = pk.find("null_embed");
        if (!ne || ne->numel() != c->text_proj.in) {
            set_error("classify: bad or missing category_head.null_embed");
            return ODISE_ERR_STATE;
        }
        c->null_embed = ne->data;
    }
    c->dim = c->text_proj.in;
    c->pdim = c->text_proj.out;
    c->built = true;
    return ODISE_OK;
}

}  // namespace odise

using namespace odise;

extern "C" int odise_hip_classify_build(odise_hip_ctx* ctx) {
    ODISE_REQUIRE(ctx, "classify_build: null context");
    return classify_build(ctx);
}

extern "C" int odise_hip_set_vocabulary(odise_hip_ctx* ctx, const float* cat_text, const float* clip_text, int K_tot, int dim,
                                        const int* group_sizes, const int* overlap, int K, float alpha, float beta) {
    ODISE_REQUIRE(ctx && cat_text && clip_text && group_sizes && overlap, "set_vocabulary: null argument");
    ModelStore* ms = store_of(ctx);
    ClassifyModel* c = ms->classify;
    if (!c || !c->built) {
        set_error("set_vocabulary: call odise_hip_classify_build first");
        return ODISE_ERR_STATE;
    }
    ODISE_REQUIRE(dim == c->dim && dim % 8 == 0, "set_vocabulary: text embedding dim %d != %d", dim, c->dim);
    ODISE_REQUIRE(K >= 1 && K <= 4096, "set_vocabulary: K=%d out of range", K);
    std::vector<int> seg(K + 1, 0);
    for (int k = 0; k < K; ++k) {
        ODISE_REQUIRE(group_sizes[k] >= 1, "set_vocabulary: empty synonym group %d", k);
        seg[k + 1] = seg[k] + group_sizes[k];
    }
    ODISE_REQUIRE(seg[K] == K_tot, "set_vocabulary: group sizes sum to %d, expected %d", seg[K], K_tot);
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));
    // T1 = normalize(text_proj([cat_text; null_embed])) ; T2 = normalize(clip_text)
    std::vector<f16> in16((size_t)(K_tot + 1) * dim);
    for (size_t i = 0; i < (size_t)K_tot * dim; ++i) in16[i] = (f16)cat_text[i];
    for (int i = 0; i < dim; ++i) in16[(size_t)K_tot * dim + i] = (f16)c->null_embed[i];
    f16* d_in = nullptr;
    float* d_proj = nullptr;
    float* d_clip = nullptr;
    ODISE_TRY(dev_upload(ctx, ms, in16.data(), in16.size() * 2, (void**)&d_in));
    ODISE_CHECK_HIP(hipMalloc((void**)&d_proj, (size_t)(K_tot + 1) * c->pdim * 4));
    ms->dev_allocs.push_back(d_proj);
    ODISE_TRY(dev_upload(ctx, ms, clip_text, (size_t)K_tot * dim * 4, (void**)&d_clip));
    ODISE_CHECK_HIP(hipMalloc((void**)&c->T1, (size_t)(K_tot + 1) * c->pdim * 2));
    ms->dev_allocs.push_back(c->T1);
    ODISE_CHECK_HIP(hipMalloc((void**)&c->T2, (size_t)K_tot * dim * 2));
    ms->dev_allocs.push_back(c->T2);
    odise_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.M = K_tot + 1; d.N = c->pdim; d.K = dim;
    d.A = d_in; d.lda = dim; d.W = c->text_proj.w; d.ldw = dim;
    d.C = d_proj; d.ldc = c->pdim; d.c_dtype = ODISE_F32; d.bias_n = c->text_proj.b; d.alpha = 1.f; d.batch = 1;
    ODISE_TRY(odise_hip_gemm(ctx, &d));
    ODISE_TRY(launch_l2_normalize_f32(ctx, d_proj, c->T1, K_tot + 1, c->pdim));
    ODISE_TRY(launch_l2_normalize_f32(ctx, d_clip, c->T2, K_tot, dim));
    ODISE_TRY(dev_upload(ctx, ms, seg.data(), seg.size() * sizeof(int), (void**)&c->seg));
    ODISE_TRY(dev_upload(ctx, ms, overlap, (size_t)K * sizeof(int), (void**)&c->ovl));
    c->K = K; c->Ktot = K_tot; c->alpha = alpha; c->beta = beta;
    c->has_vocab = true;
    ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return ODISE_OK;
}

// image [B,3,H,W] f32 device in [0,1] (the de-normalised images of the batch, odise.py:240-242); consumes the outputs of the last
// odise_hip_head_forward; mask_cls [B,Q,K+1] f32 device (log-probabilities, odise.py:323); clip_embed (optional) [B,Q,dim] f32.
extern "C" int odise_hip_classify(odise_hip_ctx* ctx, const float* image, int B, int H, int W, float* mask_cls, float* clip_embed_out) {
    ODISE_REQUIRE(ctx && image && mask_cls, "classify: null argument");
    ModelStore* ms = store_of(ctx);
    ClassifyModel* c = ms->classify;
    if (!c || !c->has_vocab) {
        set_error("classify: call odise_hip_classify_build and odise_hip_set_vocabulary first");
        return ODISE_ERR_STATE;
    }
    HeadOutputs ho;
    ODISE_TRY(head_outputs(ms, &ho));
    ODISE_REQUIRE(ho.B == B, "classify: batch %d differs from the last head_forward (%d)", B, ho.B);
    int S = 0, patch = 0, T = 0, cdim = 0;
    if (clip_dims(ms, &S, &patch, &T, &cdim) != ODISE_OK) {
        set_error("classify: the CLIP tower is not built (odise_hip_extractor_build)");
        return ODISE_ERR_STATE;
    }
    ODISE_REQUIRE(cdim == c->dim, "classify: CLIP output dim %d != text dim %d", cdim, c->dim);
    Exec ex{ctx, ms};
    const size_t mk = ms->arena.mark();
    const int Q = ho.Q;
    const int64_t MQ = (int64_t)B * Q;
    // ---- category logits: cos(mask_embed, text_proj(text bank)) --------------------------------------------------------
    f16* me_n = (f16*)ex.alloc_bytes((size_t)MQ * ho.C * 2);
    float* L1 = (float*)ex.alloc_bytes((size_t)MQ * (c->Ktot + 1) * 4);
    float* L2 = (float*)ex.alloc_bytes((size_t)MQ * c->Ktot * 4);
    f16* ce = (f16*)ex.alloc_bytes((size_t)MQ * c->dim * 2);
    f16* ce_n = (f16*)ex.alloc_bytes((size_t)MQ * c->dim * 2);
    const int64_t ldm = round_up(T, 8);
    uint8_t* tmask = (uint8_t*)ex.alloc_bytes((size_t)B * (T + Q) * ldm);
    if (!me_n || !L1 || !L2 || !ce || !ce_n || !tmask) return ODISE_ERR_NOMEM;
    ODISE_REQUIRE(ho.C == c->pdim, "classify: mask_embed dim %d != text_proj out %d", ho.C, c->pdim);
    ODISE_TRY(launch_l2_normalize_f16(ctx, ho.mask_embed, me_n, MQ, ho.C));
    odise_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.M = (int)MQ; d.N = c->Ktot + 1; d.K = ho.C;
    d.A = me_n; d.lda = ho.C; d.W = c->T1; d.ldw = ho.C;
    d.C = L1; d.ldc = c->Ktot + 1; d.c_dtype = ODISE_F32; d.alpha = 1.f; d.batch = 1;
    ODISE_TRY(ex.gemm(d));
    // ---- MaskCLIP (clip.py:325-338): image and masks bilinearly resized to 336^2 -----------------------------------------
    Act img;
    ODISE_TRY(ex.alloc(img, B, S, S, 8));
    ODISE_TRY(launch_resize_bilinear_norm(ctx, image, img.p, B, H, W, S));
    ODISE_TRY(launch_maskclip_token_mask(ctx, ho.pred_masks, tmask, B, Q, ho.h4, ho.w4, S, patch, T, ldm));
    ODISE_TRY(clip_tower(ex, img, Q, tmask, ldm, ce));
    if (clip_embed_out) ODISE_TRY(odise_hip_cast_f16_to_f32(ctx, ce, clip_embed_out, (size_t)MQ * c->dim));
    ODISE_TRY(launch_l2_normalize_f16(ctx, ce, ce_n, MQ, c->dim));
    memset(&d, 0, sizeof(d));
    d.M = (int)MQ; d.N = c->Ktot; d.K = c->dim;
    d.A = ce_n; d.lda = c->dim; d.W = c->T2; d.ldw = c->dim;
    d.C = L2; d.ldc = c->Ktot; d.c_dtype = ODISE_F32; d.alpha = 1.f; d.batch = 1;
    ODISE_TRY(ex.gemm(d));
    // ---- ensemble + null merge -----------------------------------------------------------------------------------------------
    if (c->caption && !ho.class_logits) {
        set_error("classify: word_head weights were loaded but the decoder has no class_embed (not a CaptionODISE checkpoint)");
        return ODISE_ERR_STATE;
    }
    ODISE_TRY(launch_classify_rows(ctx, L1, L2, c->seg, c->ovl, c->caption ? ho.class_logits : nullptr, mask_cls, MQ, c->K, c->Ktot,
                                   ho.logit_scale, 100.0f, c->alpha, c->beta));
    ms->arena.release(mk);
    return ODISE_OK;
}

// Per-image post-processing of image b of the last head_forward.
//   kscore [Q] f32 device: panoptic score of kept queries, < 0 for dropped ones (host computes softmax/max of mask_cls)
//   semT   [K, Qpad] f16-as-f32? -> given as fp32 device [K, Q] = softmax(mask_cls)[:, :-1]^T (or NULL: no semantic output)
//   outputs (device, optional): sem_seg [K, oh, ow] f32, ids [oh*ow] int32, counts [3*Q] int32 (zeroed here), inst [2*Qpad] f32
extern "C" int odise_hip_postprocess_pixels(odise_hip_ctx* ctx, int b, const float* kscore, const float* semT, int K, int pad_h, int pad_w,
                                            int img_h, int img_w, int out_h, int out_w, float* sem_seg, int* ids, int* counts,
                                            float* inst_stats) {
    ODISE_REQUIRE(ctx && kscore, "postprocess_pixels: null argument");
    ModelStore* ms = store_of(ctx);
    HeadOutputs ho;
    ODISE_TRY(head_outputs(ms, &ho));
    ODISE_REQUIRE(b >= 0 && b < ho.B, "postprocess_pixels: image index %d out of range", b);
    ODISE_REQUIRE(img_h <= pad_h && img_w <= pad_w && out_h > 0 && out_w > 0, "postprocess_pixels: bad geometry");
    PostGeom g;
    g.h4 = ho.h4; g.w4 = ho.w4; g.ph = pad_h; g.pw = pad_w; g.ih = img_h; g.iw = img_w; g.oh = out_h; g.ow = out_w;
    g.Q = ho.Q; g.Qpad = (int)round_up(ho.Q, 8);
    Exec ex{ctx, ms};
    const size_t mk = ms->arena.mark();
    const int npix = out_h * out_w;
    const bool need_S = (sem_seg && semT) || inst_stats;
    f16* S = need_S ? (f16*)ex.alloc_bytes((size_t)npix * g.Qpad * 2) : nullptr;
    if (need_S && !S) return ODISE_ERR_NOMEM;
    if (counts) ODISE_CHECK_HIP(hipMemsetAsync(counts, 0, 3 * (size_t)ho.Q * sizeof(int), ctx->stream));
    int* cnt = counts;
    if (!cnt) {
        cnt = (int*)ex.alloc_bytes(3 * (size_t)ho.Q * sizeof(int));
        if (!cnt) return ODISE_ERR_NOMEM;
        ODISE_CHECK_HIP(hipMemsetAsync(cnt, 0, 3 * (size_t)ho.Q * sizeof(int), ctx->stream));
    }
    const f16* logits = ho.pred_masks + (size_t)b * ho.Q * ho.h4 * ho.w4;
    ODISE_TRY(launch_postprocess_pixels(ctx, logits, kscore, S, ids, cnt, g));
    if (sem_seg && semT) {
        // sem_seg[c, p] = sum_q softmax(mask_cls)[q, c] * sigmoid(mask)[q, p]   (maskformer_model.py:280-284) as an MFMA GEMM
        f16* A = (f16*)ex.alloc_bytes((size_t)K * g.Qpad * 2);
        if (!A) return ODISE_ERR_NOMEM;
        ODISE_CHECK_HIP(hipMemsetAsync(A, 0, (size_t)K * g.Qpad * 2, ctx->stream));
        // semT fp32 [K, Q] -> fp16 [K, Qpad] (row pitch change via 2D copy is not possible with a dtype change: cast row-wise)
        for (int k0 = 0; k0 < K; k0 += 4096) (void)k0;
        f16* tmp = (f16*)ex.alloc_bytes((size_t)K * ho.Q * 2);
        if (!tmp) return ODISE_ERR_NOMEM;
        ODISE_TRY(odise_hip_cast_f32_to_f16(ctx, semT, tmp, (size_t)K * ho.Q));
        ODISE_CHECK_HIP(hipMemcpy2DAsync(A, (size_t)g.Qpad * 2, tmp, (size_t)ho.Q * 2, (size_t)ho.Q * 2, K, hipMemcpyDeviceToDevice, ctx->stream));
        odise_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.M = K; d.N = npix; d.K = g.Qpad;
        d.A = A; d.lda = g.Qpad; d.W = S; d.ldw = g.Qpad;
        d.C = sem_seg; d.ldc = npix; d.c_dtype = ODISE_F32; d.alpha = 1.f; d.batch = 1;
        ODISE_TRY(ex.gemm(d));
    }
    if (inst_stats) {
        float* partial = (float*)ex.alloc_bytes((size_t)512 * 2 * g.Qpad * 4);
        if (!partial) return ODISE_ERR_NOMEM;
        ODISE_TRY(launch_column_stats(ctx, S, partial, inst_stats, npix, g.Qpad));
    }
    ms->arena.release(mk);
    return ODISE_OK;
}

extern "C" int odise_hip_panoptic_write(odise_hip_ctx* ctx, const int* ids, const int* map, int* seg, int npix) {
    ODISE_REQUIRE(ctx && ids && map && seg, "panoptic_write: null argument");
    return launch_panoptic_write(ctx, ids, map, seg, npix);
}

extern "C" int odise_hip_instance_masks(odise_hip_ctx* ctx, int b, const int* idx, int n, int pad_h, int pad_w, int img_h, int img_w, int out_h,
                                        int out_w, float* out) {
    ODISE_REQUIRE(ctx && (n == 0 || (idx && out)), "instance_masks: null argument");
    ModelStore* ms = store_of(ctx);
    HeadOutputs ho;
    ODISE_TRY(head_outputs(ms, &ho));
    ODISE_REQUIRE(b >= 0 && b < ho.B, "instance_masks: image index %d out of range", b);
    PostGeom g;
    g.h4 = ho.h4; g.w4 = ho.w4; g.ph = pad_h; g.pw = pad_w; g.ih = img_h; g.iw = img_w; g.oh = out_h; g.ow = out_w;
    g.Q = ho.Q; g.Qpad = (int)round_up(ho.Q, 8);
    return launch_instance_masks(ctx, ho.pred_masks + (size_t)b * ho.Q * ho.h4 * ho.w4, idx, out, n, g);
}
