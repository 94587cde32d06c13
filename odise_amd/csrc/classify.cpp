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
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "engine.h"

namespace odise {

struct ClassifyModel {
    std::vector<void*> owned;   // device weights of this stage (AllocScope)
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
    // growable device scratch of the post-processing stage (pixel-major sigmoid matrix, argmax ids, decision state) and of odise_hip_infer
    // (converted / padded inputs, mask_cls): outside the arena, whose size is fixed by the network input, because the requested OUTPUT size
    // ("height" / "width") is unrelated to it
    // device banks of recently used vocabularies (the reference caches text embeddings per label tuple and keeps <= 4, odise.py:1092-1102,
    // 1281-1288): OpenPanopticInference swaps the vocabulary in and out around EVERY call (pano_wrapper.py:62-66), which must not cost
    // allocations, uploads or a synchronisation
    struct Bank {
        uint64_t key = 0, key2 = 0;
        int K = 0, Ktot = 0;
        f16* T1 = nullptr;
        f16* T2 = nullptr;
        int* seg = nullptr;
        int* ovl = nullptr;
        float alpha = 0.f, beta = 0.f;
        uint64_t stamp = 0;
    };
    std::vector<Bank> banks;
    uint64_t bank_clock = 0;
    void* post_buf = nullptr;
    size_t post_cap = 0;
    void* in_buf = nullptr;
    size_t in_cap = 0;
    // postprocess_batch: image b's pixel pass writes set b % kPostSets of (S, ids); ev_set[i] = the decision chain that read set i is done
    static constexpr int kPostSets = 4;
    hipEvent_t ev_set[kPostSets] = {nullptr, nullptr, nullptr, nullptr};
};

static int g_sem_tile = -1;   // tools hook (odise_hip_sem_tile): force the tile of the semantic GEMM for A/B runs; -1 = the rule in postprocess_batch

static int scratch_reserve(odise_hip_ctx* ctx, void** buf, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return ODISE_OK;
    (void)ctx;
    ODISE_CHECK_HIP(hipDeviceSynchronize());   // the second lane reads these buffers too (postprocess_batch)
    if (*buf) ODISE_CHECK_HIP(hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    const size_t want = bytes + bytes / 8;
    ODISE_CHECK_HIP(hipMalloc(buf, want));
    *cap = want;
    return ODISE_OK;
}

// releases the arena mark on every exit path
struct ArenaScope {
    Arena& a;
    size_t mk;
    explicit ArenaScope(Arena& arena) : a(arena), mk(arena.mark()) {}
    ~ArenaScope() { a.release(mk); }
};

void classify_destroy(ModelStore* ms) {
    if (ms->classify) {
        for (auto& bk : ms->classify->banks) {
            (void)hipFree(bk.T1); (void)hipFree(bk.T2); (void)hipFree(bk.seg); (void)hipFree(bk.ovl);
        }
        if (ms->classify->post_buf) (void)hipFree(ms->classify->post_buf);
        if (ms->classify->in_buf) (void)hipFree(ms->classify->in_buf);
        for (hipEvent_t& ev : ms->classify->ev_set)
            if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
        free_allocs(ms->classify->owned);
    }
    delete ms->classify;
    ms->classify = nullptr;
}

static int classify_build(odise_hip_ctx* ctx) {
    ModelStore* ms = store_of(ctx);
    classify_destroy(ms);
    ClassifyModel* c = new ClassifyModel();
    ms->classify = c;
    AllocScope scope(ms, c->owned);
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
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));
    ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));   // a REbuild frees the previous head's banks and scratch: queued kernels may still read them
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
    // ---- cached?  The key is 128 bits - two independent multiply-xorshift hashes over the inputs, eight bytes per step (~1 ms for the
    // 9 MB of a 1.5k-string vocabulary; the byte-wise FNV used before took ~10 ms per call) - so that a collision activating the wrong
    // bank is out of reach; the Python side short-circuits repeated label tuples before getting here.
    uint64_t key = 1469598103934665603ull, key2 = 0x9E3779B97F4A7C15ull;
    auto mix = [&](const void* p, size_t n) {
        const unsigned char* b = (const unsigned char*)p;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t w;
            memcpy(&w, b + i, 8);
            key = (key ^ w) * 1099511628211ull; key ^= key >> 29;
            key2 = (key2 + w) * 0xD6E8FEB86659FD93ull; key2 ^= key2 >> 32;
        }
        uint64_t w = 0;
        if (i < n) memcpy(&w, b + i, n - i);
        w ^= (uint64_t)n << 56;
        key = (key ^ w) * 1099511628211ull; key ^= key >> 29;
        key2 = (key2 + w) * 0xD6E8FEB86659FD93ull; key2 ^= key2 >> 32;
    };
    mix(cat_text, (size_t)K_tot * dim * 4); mix(clip_text, (size_t)K_tot * dim * 4); mix(group_sizes, (size_t)K * 4); mix(overlap, (size_t)K * 4);
    mix(&alpha, 4); mix(&beta, 4); mix(&K_tot, 4);
    auto activate = [&](ClassifyModel::Bank& bk) {
        c->T1 = bk.T1; c->T2 = bk.T2; c->seg = bk.seg; c->ovl = bk.ovl;
        c->K = bk.K; c->Ktot = bk.Ktot; c->alpha = bk.alpha; c->beta = bk.beta;
        bk.stamp = ++c->bank_clock;
        c->has_vocab = true;
    };
    for (auto& bk : c->banks)
        if (bk.key == key && bk.key2 == key2 && bk.K == K && bk.Ktot == K_tot) { activate(bk); return ODISE_OK; }
    ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));   // a new bank: queued work may still read the one about to be evicted
    if (c->banks.size() >= 8) {
        size_t lru = 0;
        for (size_t i = 1; i < c->banks.size(); ++i) if (c->banks[i].stamp < c->banks[lru].stamp) lru = i;
        ClassifyModel::Bank& ev = c->banks[lru];
        (void)hipFree(ev.T1); (void)hipFree(ev.T2); (void)hipFree(ev.seg); (void)hipFree(ev.ovl);
        c->banks.erase(c->banks.begin() + lru);
    }
    // T1 = normalize(text_proj([cat_text; null_embed])) ; T2 = normalize(clip_text)
    std::vector<f16> in16((size_t)(K_tot + 1) * dim);
    for (size_t i = 0; i < (size_t)K_tot * dim; ++i) in16[i] = (f16)cat_text[i];
    for (int i = 0; i < dim; ++i) in16[(size_t)K_tot * dim + i] = (f16)c->null_embed[i];
    ClassifyModel::Bank bk;
    bk.key = key; bk.key2 = key2; bk.K = K; bk.Ktot = K_tot; bk.alpha = alpha; bk.beta = beta;
    f16* d_in = nullptr;
    float* d_proj = nullptr;
    float* d_clip = nullptr;
    auto up = [&](const void* host, size_t bytes, void** dev) -> int {
        ODISE_CHECK_HIP(hipMalloc(dev, bytes ? bytes : 16));
        ODISE_CHECK_HIP(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
        return ODISE_OK;
    };
    int rc = up(in16.data(), in16.size() * 2, (void**)&d_in);
    if (rc == ODISE_OK) rc = up(clip_text, (size_t)K_tot * dim * 4, (void**)&d_clip);
    if (rc == ODISE_OK && hipMalloc((void**)&d_proj, (size_t)(K_tot + 1) * c->pdim * 4) != hipSuccess) rc = ODISE_ERR_NOMEM;
    if (rc == ODISE_OK && hipMalloc((void**)&bk.T1, (size_t)(K_tot + 1) * c->pdim * 2) != hipSuccess) rc = ODISE_ERR_NOMEM;
    if (rc == ODISE_OK && hipMalloc((void**)&bk.T2, (size_t)K_tot * dim * 2) != hipSuccess) rc = ODISE_ERR_NOMEM;
    if (rc == ODISE_OK) rc = up(seg.data(), seg.size() * sizeof(int), (void**)&bk.seg);
    if (rc == ODISE_OK) rc = up(overlap, (size_t)K * sizeof(int), (void**)&bk.ovl);
    if (rc == ODISE_OK) {
        odise_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.M = K_tot + 1; d.N = c->pdim; d.K = dim;
        d.A = d_in; d.lda = dim; d.W = c->text_proj.w; d.ldw = dim;
        d.C = d_proj; d.ldc = c->pdim; d.c_dtype = ODISE_F32; d.bias_n = c->text_proj.b; d.alpha = 1.f; d.batch = 1;
        rc = odise_hip_gemm(ctx, &d);
    }
    if (rc == ODISE_OK) rc = launch_l2_normalize_f32(ctx, d_proj, bk.T1, K_tot + 1, c->pdim);
    if (rc == ODISE_OK) rc = launch_l2_normalize_f32(ctx, d_clip, bk.T2, K_tot, dim);
    const hipError_t se = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in); (void)hipFree(d_proj); (void)hipFree(d_clip);
    if (rc != ODISE_OK || se != hipSuccess) {
        (void)hipFree(bk.T1); (void)hipFree(bk.T2); (void)hipFree(bk.seg); (void)hipFree(bk.ovl);
        if (rc == ODISE_ERR_NOMEM) set_error("set_vocabulary: out of device memory");
        if (rc == ODISE_OK) { set_error("set_vocabulary: %s", hipGetErrorString(se)); rc = ODISE_ERR_HIP; }
        return rc;
    }
    c->banks.push_back(bk);
    activate(c->banks.back());
    return ODISE_OK;
}

// image [B,3,H,W] f32 device in [0,1] (the de-normalised images of the batch, odise.py:240-242); consumes the outputs of the last
// odise_hip_head_forward; mask_cls [B,Q,K+1] f32 device (log-probabilities, odise.py:323); clip_embed (optional) [B,Q,dim] f32.
// MaskCLIP's tower over B pictures and their Q mask tokens (tmask: launch_maskclip_token_mask's [B][T + Q][ldm] rows), in the form
// ODISE_OPT_MASKCLIP_PASSES names.  kv_ready: the image-token pass of THESE pictures has been enqueued already (odise_hip_infer).
static int maskclip_tower(Exec& ex, const float* image01, int B, int H, int W, int S, int T, int Q, const uint8_t* tmask, int64_t ldm, f16* ce, bool kv_ready) {
    odise_hip_ctx* ctx = ex.ctx;
    const bool have_pass1 = kv_ready && ex.ms->mclip.ready && ex.ms->mclip.B == B;
    // (the two-pass forms need their key / value store, ~0.1 GB per picture: when it cannot be reserved the reference's one-pass layout runs instead)
    if (ctx->maskclip_passes == 2 || (!have_pass1 && !maskclip_kv_available(ctx, ex.ms, B))) {
        Act img;
        ODISE_TRY(ex.alloc(img, B, S, S, 8));
        ODISE_TRY(launch_resize_bilinear_norm(ctx, image01, img.p, B, H, W, S));
        return clip_tower(ex, img, Q, tmask, ldm, ce);
    }
    ClipKV& kv = ex.ms->mclip;
    if (kv_ready && kv.ready && kv.B == B) {
        if (kv.on_lane2) ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_mclip, 0));
    } else {
        ODISE_TRY(maskclip_image_pass(ex, image01, B, H, W));
    }
    kv.ready = false;   // consumed
    return maskclip_mask_pass(ex, Q, tmask + (size_t)T * ldm, ldm, (int64_t)(T + Q) * ldm, ce);
}

static int classify_impl(odise_hip_ctx* ctx, const float* image, int B, int H, int W, float* mask_cls, float* clip_embed_out, bool kv_ready);
extern "C" int odise_hip_classify(odise_hip_ctx* ctx, const float* image, int B, int H, int W, float* mask_cls, float* clip_embed_out) {
    return classify_impl(ctx, image, B, H, W, mask_cls, clip_embed_out, false);
}
static int classify_impl(odise_hip_ctx* ctx, const float* image, int B, int H, int W, float* mask_cls, float* clip_embed_out, bool kv_ready) {
    ODISE_REQUIRE(ctx && image && mask_cls, "classify: null argument");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
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
    ArenaScope scope(ms->arena);
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
    ODISE_TRY(launch_maskclip_token_mask(ctx, ho.pred_masks, tmask, B, Q, ho.h4, ho.w4, S, patch, T, ldm));
    stage_mark(ctx, "classify: text logits + MaskCLIP inputs");
    ODISE_TRY(maskclip_tower(ex, image, B, H, W, S, T, Q, tmask, ldm, ce, kv_ready));
    stage_mark(ctx, "classify: MaskCLIP tower done");
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
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    ModelStore* ms = store_of(ctx);
    HeadOutputs ho;
    ODISE_TRY(head_outputs(ms, &ho));
    ODISE_REQUIRE(b >= 0 && b < ho.B, "postprocess_pixels: image index %d out of range", b);
    ODISE_REQUIRE(img_h <= pad_h && img_w <= pad_w && out_h > 0 && out_w > 0, "postprocess_pixels: bad geometry");
    PostGeom g;
    g.h4 = ho.h4; g.w4 = ho.w4; g.ph = pad_h; g.pw = pad_w; g.ih = img_h; g.iw = img_w; g.oh = out_h; g.ow = out_w;
    g.Q = ho.Q; g.Qpad = (int)round_up(ho.Q, 8);
    Exec ex{ctx, ms};
    ArenaScope scope(ms->arena);
    const int npix = out_h * out_w;
    const bool need_S = (sem_seg && semT) || inst_stats;
    f16* S = need_S ? (f16*)ex.alloc_bytes((size_t)npix * g.Qpad * 2) : nullptr;
    if (need_S && !S) {   // outputs much larger than the network input: the arena (sized by the input) is too small for S
        ClassifyModel* cm = ms->classify;
        if (!cm) { set_error("postprocess_pixels: call odise_hip_classify_build first"); return ODISE_ERR_STATE; }
        ODISE_TRY(scratch_reserve(ctx, &cm->post_buf, &cm->post_cap, (size_t)npix * g.Qpad * 2));
        S = (f16*)cm->post_buf;
    }
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
        unsigned int* partial = (unsigned int*)ex.alloc_bytes((size_t)512 * 2 * g.Qpad * 4);
        if (!partial) return ODISE_ERR_NOMEM;
        ODISE_TRY(launch_column_stats(ctx, S, partial, inst_stats, npix, g.Qpad));
    }
    return ODISE_OK;
}

extern "C" int odise_hip_panoptic_write(odise_hip_ctx* ctx, const int* ids, const int* map, int* seg, int npix) {
    ODISE_REQUIRE(ctx && ids && map && seg, "panoptic_write: null argument");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    return launch_panoptic_write(ctx, ids, map, seg, npix);
}

extern "C" int odise_hip_instance_masks(odise_hip_ctx* ctx, int b, const int* idx, int n, int pad_h, int pad_w, int img_h, int img_w, int out_h,
                                        int out_w, float* out) {
    ODISE_REQUIRE(ctx && (n == 0 || (idx && out)), "instance_masks: null argument");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    ModelStore* ms = store_of(ctx);
    HeadOutputs ho;
    ODISE_TRY(head_outputs(ms, &ho));
    ODISE_REQUIRE(b >= 0 && b < ho.B, "instance_masks: image index %d out of range", b);
    PostGeom g;
    g.h4 = ho.h4; g.w4 = ho.w4; g.ph = pad_h; g.pw = pad_w; g.ih = img_h; g.iw = img_w; g.oh = out_h; g.ow = out_w;
    g.Q = ho.Q; g.Qpad = (int)round_up(ho.Q, 8);
    return launch_instance_masks(ctx, ho.pred_masks + (size_t)b * ho.Q * ho.h4 * ho.w4, idx, out, n, g);
}

extern "C" int odise_hip_maskclip_embed(odise_hip_ctx* ctx, const float* image, int B, int H, int W, const float* pred_masks, int Q, int h, int w,
                                        float* clip_embed) {
    ODISE_REQUIRE(ctx && image && pred_masks && clip_embed && B >= 1 && Q >= 1 && h >= 1 && w >= 1, "maskclip_embed: bad argument");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    ModelStore* ms = store_of(ctx);
    int S = 0, patch = 0, T = 0, cdim = 0;
    if (clip_dims(ms, &S, &patch, &T, &cdim) != ODISE_OK) {
        set_error("maskclip_embed: the CLIP tower is not built (odise_hip_extractor_build)");
        return ODISE_ERR_STATE;
    }
    {
        // this path allocates ABOVE whatever an earlier head call left in the arena (its outputs stay valid for a later classify /
        // post-processing call), so the requirement counts from the current mark; if the arena has to grow it is reallocated, and the
        // head outputs that pointed into the old one are invalidated instead of dangling
        const char* base0 = ms->arena.base;
        ODISE_TRY(ensure_arena(ctx, ms, ms->arena.off + ((size_t)1 << 30) + (size_t)B * (T + Q) * 1024 * 2 * 64));
        if (ms->arena.base != base0) maskgen_invalidate_outputs(ms);
    }
    Exec ex{ctx, ms};
    ArenaScope scope(ms->arena);
    const int64_t MQ = (int64_t)B * Q, ldm = round_up(T, 8);
    f16* logits16 = (f16*)ex.alloc_bytes((size_t)MQ * h * w * 2);
    f16* ce = (f16*)ex.alloc_bytes((size_t)MQ * cdim * 2);
    uint8_t* tmask = (uint8_t*)ex.alloc_bytes((size_t)B * (T + Q) * ldm);
    if (!logits16 || !ce || !tmask) return ODISE_ERR_NOMEM;
    ODISE_TRY(odise_hip_cast_f32_to_f16(ctx, pred_masks, logits16, (size_t)MQ * h * w));
    ODISE_TRY(launch_maskclip_token_mask(ctx, logits16, tmask, B, Q, h, w, S, patch, T, ldm));
    ODISE_TRY(maskclip_tower(ex, image, B, H, W, S, T, Q, tmask, ldm, ce, false));
    return odise_hip_cast_f16_to_f32(ctx, ce, clip_embed, (size_t)MQ * cdim);
}

// ---- the three heads for a batch, decisions on the device -------------------------------------------------------------------------------
extern "C" int odise_hip_postprocess_batch(odise_hip_ctx* ctx, const odise_post_desc* d) {
    ODISE_REQUIRE(ctx && d, "postprocess_batch: null argument");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    ModelStore* ms = store_of(ctx);
    ClassifyModel* c = ms->classify;
    if (!c || !c->has_vocab) {
        set_error("postprocess_batch: call odise_hip_classify_build and odise_hip_set_vocabulary first");
        return ODISE_ERR_STATE;
    }
    HeadOutputs ho;
    ODISE_TRY(head_outputs(ms, &ho));
    const int B = d->B, Q = ho.Q, K = c->K, Qpad = (int)round_up(Q, 8);
    ODISE_REQUIRE(B == ho.B, "postprocess_batch: batch %d differs from the last head_forward (%d)", B, ho.B);
    ODISE_REQUIRE(d->mask_cls && d->img_hw, "postprocess_batch: mask_cls / img_hw missing");
    ODISE_REQUIRE(!(d->panoptic_on || d->instance_on) || d->isthing, "postprocess_batch: isthing[K] is required by the panoptic and instance heads");
    ODISE_REQUIRE(d->pad_h == 4 * ho.h4 && d->pad_w == 4 * ho.w4, "postprocess_batch: padded size %dx%d does not match the mask logits (%dx%d x 4)", d->pad_h,
                  d->pad_w, ho.h4, ho.w4);
    const bool want_pan = d->panoptic_on && d->panoptic, want_inst = d->instance_on && d->inst_table && d->inst_scores;
    const bool want_sem = d->semantic_on && (d->sem_seg || d->sem_argmax);
    const int topk = d->topk > 0 ? d->topk : 100;
    int max_pix = 1;
    for (int b = 0; b < B; ++b) {
        const int ih = d->img_hw[2 * b], iw = d->img_hw[2 * b + 1];
        const int oh = d->out_hw ? d->out_hw[2 * b] : ih, ow = d->out_hw ? d->out_hw[2 * b + 1] : iw;
        ODISE_REQUIRE(ih >= 1 && iw >= 1 && ih <= d->pad_h && iw <= d->pad_w && oh >= 1 && ow >= 1 && (int64_t)oh * ow < (1ll << 30),
                      "postprocess_batch: bad geometry of image %d", b);
        max_pix = std::max(max_pix, oh * ow);
    }
    // ---- scratch layout (256-byte aligned pieces)
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_kscore = take((size_t)B * Q * 4), o_label = take((size_t)B * Q * 4), o_map = take((size_t)B * Q * 4);
    const size_t o_counts = take((size_t)B * 3 * Q * 4), o_stats = take((size_t)B * 2 * Qpad * 4), o_stuff = take((size_t)K * 4);
    const size_t o_thing = take((size_t)K), o_probs = take(want_inst ? (size_t)B * Q * K * 4 : 0), o_semT = take((size_t)B * K * Qpad * 2);
    int max_stat_blocks = 512;   // block partials of the instance statistics: column_stats_kernel's 512, or one per tile of the tiled pixel pass
    for (int b = 0; b < B; ++b) {
        PostGeom gb;
        gb.h4 = ho.h4; gb.w4 = ho.w4; gb.ph = d->pad_h; gb.pw = d->pad_w; gb.ih = d->img_hw[2 * b]; gb.iw = d->img_hw[2 * b + 1];
        gb.oh = d->out_hw ? d->out_hw[2 * b] : gb.ih; gb.ow = d->out_hw ? d->out_hw[2 * b + 1] : gb.iw;
        gb.Q = Q; gb.Qpad = Qpad;
        if (postprocess_pixels_tiled(gb)) max_stat_blocks = std::max(max_stat_blocks, postprocess_pixels_stat_blocks(gb));
    }
    const size_t partial_set = ((size_t)max_stat_blocks * 2 * Qpad * 4 + 255) & ~(size_t)255;
    // (S, ids) of up to kPostSets images in flight: the next image's pixel pass does not wait for the decision chain that still reads this one's
    const int nsets = std::min(B, (int)ClassifyModel::kPostSets);
    const size_t ids_set = ((size_t)max_pix * 4 + 255) & ~(size_t)255, S_set = ((size_t)max_pix * Qpad * 2 + 255) & ~(size_t)255;
    const size_t o_ids = take(ids_set * nsets), o_S = take(S_set * nsets), o_partial = take(partial_set * nsets);
    ODISE_TRY(scratch_reserve(ctx, &c->post_buf, &c->post_cap, off));
    char* base = (char*)c->post_buf;
    float* kscore = (float*)(base + o_kscore);
    int* label = (int*)(base + o_label);
    int* map = (int*)(base + o_map);
    int* counts = (int*)(base + o_counts);
    float* stats = (float*)(base + o_stats);
    int* stuff = (int*)(base + o_stuff);
    uint8_t* thing = (uint8_t*)(base + o_thing);
    float* probs = want_inst ? (float*)(base + o_probs) : nullptr;
    f16* semT = (f16*)(base + o_semT);
    if (d->isthing) ODISE_CHECK_HIP(hipMemcpyAsync(thing, d->isthing, (size_t)K, hipMemcpyHostToDevice, ctx->stream));
    ODISE_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)B * 3 * Q * 4, ctx->stream));
    // the panoptic records may be the source buffer of the previous batch's all-gather (still running on the exchange stream while the
    // backbone / head / classification of this batch executed): order the first write after it - a stream-side wait, the host never blocks
    if (want_pan && ctx->comm) ODISE_TRY(odise_hip_comm_wait(ctx, 0));
    ODISE_TRY(launch_post_decide(ctx, d->mask_cls, kscore, label, semT, probs, B, Q, Qpad, K, d->object_mask_threshold));
    Exec ex{ctx, ms};
    auto geometry = [&](int b) {
        PostGeom g;
        g.h4 = ho.h4; g.w4 = ho.w4; g.ph = d->pad_h; g.pw = d->pad_w; g.ih = d->img_hw[2 * b]; g.iw = d->img_hw[2 * b + 1];
        g.oh = d->out_hw ? d->out_hw[2 * b] : g.ih; g.ow = d->out_hw ? d->out_hw[2 * b + 1] : g.iw;
        g.Q = Q; g.Qpad = Qpad;
        return g;
    };
    // The per-image chain after the pixel pass - mask statistics for the instance scores, the sequential segment walk, the record write, the
    // instance head's top-k and its binary masks - is a handful of small kernels (one of them a single thread per image) plus one write-bound
    // one.  With the second lane present they run there, beside the NEXT image's pixel pass on the main stream: every image in flight has its
    // own (S, ids), so the main stream is four pixel passes back to back and only the last image's chain is exposed (round 5: the next pixel
    // pass waited for the chain, and the top-k + masks of all images followed the loop: 2.5 ms for four pictures).
#ifdef ODISE_TOOLS
    static const bool serial_post = getenv("ODISE_POST_SERIAL") != nullptr;   // A/B: the chain on the main stream
#else
    const bool serial_post = false;
#endif
    const bool side = !serial_post && ctx->lanes == 2 && ctx->stream2 && ctx->ev_fork && ctx->ev_join;
    if (side)
        for (int i = 0; i < nsets; ++i)
            if (!c->ev_set[i]) ODISE_CHECK_HIP(hipEventCreateWithFlags(&c->ev_set[i], hipEventDisableTiming));
    bool side_pending = false;
    bool set_busy[ClassifyModel::kPostSets] = {false, false, false, false};
    int n_img = 0;   // images processed so far (skipped ones do not take a set)
    for (int b = 0; b < B; ++b) {
        const PostGeom g = geometry(b);
        const int npix = g.oh * g.ow;
        const f16* logits = ho.pred_masks + (size_t)b * Q * ho.h4 * ho.w4;
        float* sem = (want_sem && d->sem_seg) ? d->sem_seg[b] : nullptr;
        int* amax = (want_sem && d->sem_argmax) ? d->sem_argmax[b] : nullptr;
        int* pan = want_pan ? d->panoptic[b] : nullptr;
        const bool inst = want_inst;
        // the semantic scores come out of the pixel pass itself where its tiled form applies (an exact 4x upsampling, <= 112 queries): no
        // pixel-major matrix S is written for them; S is still produced for the instance statistics, the fused arg-max and the GEMM fallback
        const bool fused_sem = sem && g_sem_tile < 0 && postprocess_pixels_fuses_semantic(g);
        // ... and the tiled form leaves the instance statistics as block partials, so the instance head alone does not need S either
        const bool fused_stats = inst && g_sem_tile < 0 && postprocess_pixels_tiled(g);
        const bool need_S = (sem && !fused_sem) || amax || (inst && !fused_stats);
        if (!need_S && !pan && !sem && !inst) continue;
        const int set = n_img++ % nsets;
        int* ids = (int*)(base + o_ids + ids_set * set);
        f16* S = (f16*)(base + o_S + S_set * set);
        unsigned int* partial = (unsigned int*)(base + o_partial + partial_set * set);
        if (set_busy[set]) ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, c->ev_set[set], 0));   // the chain of the image that used this set before
        set_busy[set] = false;
        if (fused_sem) ms->macs += (double)K * npix * Qpad;
        ODISE_TRY(launch_postprocess_pixels(ctx, logits, kscore + (size_t)b * Q, need_S ? S : nullptr, pan ? ids : nullptr, counts + (size_t)b * 3 * Q, g,
                                            fused_sem ? semT + (size_t)b * K * Qpad : nullptr, fused_sem ? sem : nullptr, fused_sem ? K : 0,
                                            fused_stats ? partial : nullptr));
        auto decisions = [&]() -> int {
            if (inst && fused_stats) ODISE_TRY(launch_column_fold(ctx, partial, stats + (size_t)b * 2 * Qpad, postprocess_pixels_stat_blocks(g), Qpad));
            else if (inst) ODISE_TRY(launch_column_stats(ctx, S, partial, stats + (size_t)b * 2 * Qpad, npix, Qpad));
            if (pan) {
                ODISE_TRY(launch_panoptic_decide(ctx, counts + (size_t)b * 3 * Q, kscore + (size_t)b * Q, label + (size_t)b * Q, thing, map + (size_t)b * Q,
                                                 pan + npix, Q, K, d->overlap_threshold, ODISE_MAX_SEGMENTS, stuff));
                ODISE_TRY(launch_panoptic_write(ctx, ids, map + (size_t)b * Q, pan, npix));
            }
            if (inst) {   // the image's top-k (one block) and the selected masks
                int* tb = d->inst_table + (size_t)b * (1 + 2 * topk);
                ODISE_TRY(launch_instance_topk(ctx, probs + (size_t)b * Q * K, stats + (size_t)b * 2 * Qpad, thing, tb, d->inst_scores + (size_t)b * topk, 1, Q, Qpad,
                                               K, topk, d->panoptic_on ? 1 : 0));
                if (d->inst_masks && d->inst_masks[b]) ODISE_TRY(launch_instance_masks(ctx, logits, tb + 1, d->inst_masks[b], std::min(topk, Q * K), g, tb));
            }
            return ODISE_OK;
        };
        if (side && (inst || pan)) {
            ODISE_CHECK_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
            {
                Lane2 lane(ctx, ms);
                ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_fork, 0));
                ODISE_TRY(decisions());
                ODISE_CHECK_HIP(hipEventRecord(c->ev_set[set], ctx->stream));
                ODISE_CHECK_HIP(hipEventRecord(ctx->ev_join, ctx->stream));
            }
            set_busy[set] = true;
            side_pending = true;
        }
        if (sem && !fused_sem) {   // sem_seg[c, p] = sum_q softmax(mask_cls)[q, c] * sigmoid(mask)[q, p]  (maskformer_model.py:280-284) as an MFMA GEMM
            odise_gemm_desc gd;
            memset(&gd, 0, sizeof(gd));
            gd.M = K; gd.N = npix; gd.K = Qpad;
            gd.A = semT + (size_t)b * K * Qpad; gd.lda = Qpad; gd.W = S; gd.ldw = Qpad;
            gd.C = sem; gd.ldc = npix; gd.c_dtype = ODISE_F32; gd.alpha = 1.f; gd.batch = 1;
            // HBM-bound (K = Qpad ~ 104): per pixel 208 B of S in, 4 B x K out.  The cost model, fitted on MFMA-bound shapes, picks 64-row tiles
            // and every row tile re-reads the 218 MB matrix S (3x at K = 133); a 256-row tile covers up to 256 classes per pass over S - its
            // idle MFMA rows cost nothing here (measured: tools/post_bench.py).
            ms->macs += (double)gd.M * gd.N * gd.K;
            ODISE_TRY(gemm_forced(ctx, &gd, g_sem_tile >= 0 ? g_sem_tile : (K <= 64 ? -1 : 5), 0));
        }
        if (amax) ODISE_TRY(launch_semantic_argmax(ctx, S, semT + (size_t)b * K * Qpad, amax, npix, Qpad, K));
        if (!(side && (inst || pan))) ODISE_TRY(decisions());
    }
    if (want_inst && n_img < B) {
        // images that produced nothing else still owe their (empty) instance tables - cannot happen with want_inst (it makes need_S true); kept as a guard
        set_error("postprocess_batch: an image was skipped although the instance head is on");
        return ODISE_ERR_STATE;
    }
    if (side_pending) ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));   // everything the call produced is ordered on the context's stream
    return ODISE_OK;
}

extern "C" int odise_hip_infer(odise_hip_ctx* ctx, const odise_infer_desc* d) {
    ODISE_REQUIRE(ctx && d && d->images && d->img_hw && d->B >= 1, "infer: null argument");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the calling thread may be new (one host thread per context is the supported way to keep batches in flight)
    ModelStore* ms = store_of(ctx);
    ClassifyModel* c = ms->classify;
    if (!c || !c->has_vocab) {
        set_error("infer: call odise_hip_classify_build and odise_hip_set_vocabulary first");
        return ODISE_ERR_STATE;
    }
    ODISE_REQUIRE(d->image_layout >= 0 && d->image_layout <= 2, "infer: image_layout %d", d->image_layout);
    const int B = d->B;
    int H = 0, W = 0;
    for (int b = 0; b < B; ++b) {
        ODISE_REQUIRE(d->images[b] && d->img_hw[2 * b] >= 1 && d->img_hw[2 * b + 1] >= 1, "infer: bad image %d", b);
        H = std::max(H, d->img_hw[2 * b]);
        W = std::max(W, d->img_hw[2 * b + 1]);
    }
    const int Hp = (int)round_up(H, 64), Wp = (int)round_up(W, 64);   // size_divisibility = 64 (feature_extractor.py:126-128)
    int Q = 0;
    ODISE_TRY(odise_hip_maskgen_info(ctx, &Q, nullptr, nullptr));
    const size_t n_pad = (size_t)B * 3 * Hp * Wp, n_img = (size_t)B * 3 * H * W, n_cls = (size_t)B * Q * (c->K + 1);
    const bool same = Hp == H && Wp == W;
    ODISE_TRY(scratch_reserve(ctx, &c->in_buf, &c->in_cap, (n_pad + (same ? 0 : n_img) + n_cls) * 4 + 1024));
    float* padded = (float*)c->in_buf;
    float* img01 = same ? padded : padded + n_pad;
    float* mask_cls = (same ? padded + n_pad : img01 + n_img);
    for (int b = 0; b < B; ++b) {
        const int h = d->img_hw[2 * b], w = d->img_hw[2 * b + 1];
        ODISE_TRY(launch_image_pad(ctx, d->images[b], d->image_layout, h, w, padded + (size_t)b * 3 * Hp * Wp, Hp, Wp));
        if (!same) ODISE_TRY(launch_image_pad(ctx, d->images[b], d->image_layout, h, w, img01 + (size_t)b * 3 * H * W, H, W));
    }
    stage_mark(ctx, "infer: inputs padded");
    {   // encoder prefetch: is this the batch the previous call prepared?  (anything else that was prepared is dropped: its stream is drained first)
        Prefetch& pf = ms->pf;
        PrefetchKey key;
        key.B = B; key.layout = d->image_layout;
        key.images.assign(d->images, d->images + B);
        key.hw.assign(d->img_hw, d->img_hw + 2 * B);
        pf.use_now = pf.has_ready && pf.ready == key;
        if (pf.has_ready) ++(pf.use_now ? pf.n_hits : pf.n_dropped);
        if (pf.has_ready && !pf.use_now && ctx->stream3) ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream3));
        pf.has_ready = false;   // consumed (or dropped) by this call
    }
    {
        // MaskCLIP's image tokens do not depend on the mask head (engine.h ClipKV): the backbone stage enqueues their pass beside its own lanes
        ClipKV& kv = ms->mclip;
        kv.ready = false;
        kv.planned = ctx->maskclip_passes == 0 || ctx->maskclip_passes == 3;
        kv.plan_image = img01; kv.plan_B = B; kv.plan_H = H; kv.plan_W = W;
        const int rc_bb = odise_hip_backbone_forward(ctx, padded, B, Hp, Wp, nullptr);
        ms->pf.use_now = false;   // on every exit path: a stored latent must never be consumed by a later, unrelated backbone call
        kv.planned = false;       // (likewise: a later backbone call of another caller must not run this call's pass)
        if (rc_bb != ODISE_OK) { kv.ready = false; return rc_bb; }
    }
    stage_mark(ctx, "backbone done (taps projected + stitched)");
    {
        int rc_h = odise_hip_head_forward(ctx, nullptr, B, 0, Hp / 4, Wp / 4, nullptr, nullptr, nullptr, nullptr);
        if (rc_h == ODISE_OK) rc_h = classify_impl(ctx, img01, B, H, W, mask_cls, nullptr, true);
        ms->mclip.ready = false;
        if (rc_h != ODISE_OK) return rc_h;
    }
    stage_mark(ctx, "classification done (MaskCLIP + logits)");
    if (d->mask_cls_out) ODISE_CHECK_HIP(hipMemcpyAsync(d->mask_cls_out, mask_cls, n_cls * 4, hipMemcpyDeviceToDevice, ctx->stream));
    odise_post_desc p = d->post;
    p.B = B; p.pad_h = Hp; p.pad_w = Wp; p.img_hw = d->img_hw; p.mask_cls = mask_cls;
    const int rc = odise_hip_postprocess_batch(ctx, &p);
    stage_mark(ctx, "post-processing done");
    return rc;
}

extern "C" int odise_hip_infer_prefetch(odise_hip_ctx* ctx, const odise_infer_desc* next) {
    ODISE_REQUIRE(ctx, "infer_prefetch: null context");
    ModelStore* ms = store_of(ctx);
    Prefetch& pf = ms->pf;
    if (!next) { pf.has_pending = false; return ODISE_OK; }   // cancel
    ODISE_REQUIRE(next->images && next->img_hw && next->B >= 1 && next->image_layout >= 0 && next->image_layout <= 2, "infer_prefetch: bad descriptor");
    pf.pending.B = next->B;
    pf.pending.layout = next->image_layout;
    pf.pending.images.assign(next->images, next->images + next->B);
    pf.pending.hw.assign(next->img_hw, next->img_hw + 2 * next->B);
    pf.has_pending = true;
    return ODISE_OK;
}

extern "C" int odise_hip_prefetch_stats(odise_hip_ctx* ctx, int* enqueued, int* hits, int* dropped, int* failed) {
    ODISE_REQUIRE(ctx, "prefetch_stats: null context");
    const Prefetch& pf = store_of(ctx)->pf;
    if (enqueued) *enqueued = pf.n_enqueued;
    if (hits) *hits = pf.n_hits;
    if (dropped) *dropped = pf.n_dropped;
    if (failed) *failed = pf.n_failed;
    return ODISE_OK;
}

extern "C" int odise_hip_sem_tile(int tile) { odise::g_sem_tile = tile; return 0; }
