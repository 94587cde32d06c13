// maskgen.cpp — the mask generator of ODISE on the device:
//   FeatureExtractorBackbone.slide_forward / single_forward / forward_features   (odise/modeling/backbone/feature_extractor.py:139-250)
//   MSDeformAttnPixelDecoder.forward_features                                     (M2F/modeling/pixel_decoder/msdeformattn.py:314-358)
//   ODISEMultiScaleMaskedTransformerDecoder.forward / forward_prediction_heads    (odise/modeling/meta_arch/odise.py:642-776)
//   PooledMaskEmbed / MaskPooling                                                  (odise/modeling/meta_arch/odise.py:937-1015)
//
// All crops of all images of a call go through the extractor as one batch (n = image*K + crop); the 8 taps are projected by
// detectron2 BottleneckBlocks (1x1 -> 3x3 -> 1x1, GroupNorm 32), summed per stride group and
// overlap-averaged into s2..s5.  The deformable-attention encoder runs on [B, 21504, 256] token matrices; the decoder's masked
// cross-attention uses attn.hip with a u8 visibility mask computed straight from the mask logits.  PooledMaskEmbed is evaluated
// only for the final prediction head: the nine intermediate evaluations of the reference feed `aux_outputs`, which inference
// never reads (odise.py:713-727).
#include <math.h>
#include <string.h>

#include "engine.h"

namespace odise {

struct BottleneckW {
    ConvW c1, c2, c3, sc;
    NormW n1, n2, n3, nsc;
    bool has_sc = false;
};
struct MsdaLayerW {
    LinW off, aw, value, out, lin1, lin2;
    LinW offaw;   // sampling_offsets | attention_weights stacked: both read the same query, one GEMM writes [rows, 2 MLP + MLP] (ms_deform_attn.py:98-107)
    NormW norm1, norm2;
};
struct MhaW {
    LinW q, k, v, qk, out;  // q [C,C]+bq, k, v (bias applied along M of the swapped GEMM), stacked q|k for self-attention
    float* v_bias = nullptr;
    NormW norm;
};
struct DecLayerW {
    MhaW cross, self;
    LinW lin1, lin2;
    NormW ffn_norm;
};

// Device weights of the two separately (re)buildable halves.  Kept as base structs of the model so that a rebuild can assemble a complete new
// set on the side and put it in place only when every key was found and every upload succeeded: a failed reload (a checkpoint with a missing
// or misshapen tensor) leaves the previous weights in place and usable.
struct BackboneW {
    BottleneckW proj[8];
    int proj_dim = 512;
};
struct HeadW {
    // pixel decoder
    ConvW in_proj[3];
    NormW in_proj_gn[3];
    float* enc_level_embed = nullptr;  // [3, C]
    std::vector<MsdaLayerW> enc_layers;
    ConvW adapter, layer1, mask_feat;
    NormW adapter_gn, layer1_gn;
    f16* mask_feat_wT = nullptr;  // mask_features weight as the A operand of the swapped GEMM (same [C,C] matrix)
    int C = 256, enc_heads = 8, enc_points = 4;
    // transformer decoder
    std::vector<DecLayerW> dec_layers;
    NormW decoder_norm;
    float *query_feat = nullptr, *query_embed = nullptr, *dec_level_embed = nullptr;
    f16* query_feat16 = nullptr;
    LinW mask_mlp[3], pool_proj, post_mlp[3];
    NormW pool_ln, post_ln;
    float logit_scale = 0.f;
    int Q = 100, dec_heads = 8;
    bool has_class_embed = false;  // CaptionODISE: learned Linear(C, 2) on the decoder output (mask2former_transformer_decoder.py:333)
    LinW class_embed;
};

struct MaskGenModel : BackboneW, HeadW {
    std::vector<void*> owned_backbone, owned_head;   // what the two halves allocated on the device (AllocScope)
    bool backbone_built = false, head_built = false;
    // Per-size device tables, kept in small LRU caches keyed by the EXACT shapes: a dataset evaluation sees many aspect ratios, and every
    // size used to cost a fresh hipMalloc that was never freed (40 MB of positional tables per size change at 1024^2).  A hit costs nothing;
    // a miss uploads one set and, beyond kSizeCache entries, frees the least recently used one (after a stream sync: queued kernels may
    // still read it).  `pe` / `enc_pos_all` / `boxes_dev` point into the active entries.
    static const int kSizeCache = 8;
    struct PeEntry {
        int hs[3] = {0, 0, 0}, ws[3] = {0, 0, 0};
        float* pe[3] = {nullptr, nullptr, nullptr};
        float* all = nullptr;
        uint64_t stamp = 0;
    };
    struct BoxEntry {
        int H = 0, W = 0, K = 0;
        int* dev = nullptr;
        uint64_t stamp = 0;
    };
    std::vector<PeEntry> pe_cache;
    std::vector<BoxEntry> box_cache;
    uint64_t size_clock = 0;
    float* pe[4] = {nullptr, nullptr, nullptr, nullptr};  // sine PE [h*w, C] for the 3 transformer levels (+ spare) of the active entry
    float* enc_pos_all = nullptr;                         // [Lq, C] = PE + encoder level_embed
    int* boxes_dev = nullptr;  // [4 groups + image][K][2]
    int boxes_K = 0;
    void drop_size_caches() {   // the tables derive from the weights (level_embed): a rebuilt model starts empty
        for (auto& e : pe_cache) { for (float* p : e.pe) (void)hipFree(p); (void)hipFree(e.all); }
        for (auto& e : box_cache) (void)hipFree(e.dev);
        pe_cache.clear(); box_cache.clear();
        pe[0] = pe[1] = pe[2] = pe[3] = nullptr; enc_pos_all = nullptr; boxes_dev = nullptr; boxes_K = 0;
    }
    ~MaskGenModel() { drop_size_caches(); }
    // outputs of the last calls (arena)
    Act feats[4];              // s2..s5 NHWC fp16
    f16* pred_masks = nullptr;   // [B, Q, H4*W4] logits
    f16* mask_embed = nullptr;   // [B, Q, C]
    f16* mask_pooled = nullptr;  // [B, Q, C]
    float* class_logits = nullptr;  // [B, Q, 2] of the final prediction head
    int out_B = 0, out_h = 0, out_w = 0;
    double last_macs = 0.0;
};

void maskgen_destroy(ModelStore* ms) {
    if (ms->maskgen) {
        free_allocs(ms->maskgen->owned_backbone);
        free_allocs(ms->maskgen->owned_head);
    }
    delete ms->maskgen;
    ms->maskgen = nullptr;
}

static MaskGenModel* maskgen_of(ModelStore* ms) {
    if (!ms->maskgen) ms->maskgen = new MaskGenModel();
    return ms->maskgen;
}

// ---------------------------------------------------------------------------------------------------------------
static int build_convnorm(Packer& pk, const std::string& key, ConvW& c, NormW& n) {
    ODISE_TRY(pk.conv(key, c, false));
    ODISE_TRY(pk.norm(key + ".norm", n));
    return ODISE_OK;
}

static int build_mha(Packer& pk, const std::string& key, MhaW& m, int C, bool stack_qk) {
    const HostTensor* w = pk.find(key + ".in_proj_weight");
    const HostTensor* b = pk.find(key + ".in_proj_bias");
    if (!w || !b || w->numel() != (int64_t)3 * C * C || b->numel() != 3 * C) {
        set_error("decoder: bad or missing '%s.in_proj_*'", key.c_str());
        return ODISE_ERR_STATE;
    }
    const size_t cc = (size_t)C * C;
    std::vector<f16> t(3 * cc);
    for (size_t i = 0; i < 3 * cc; ++i) t[i] = (f16)w->data[i];
    if (stack_qk) {
        m.qk.in = C; m.qk.out = 2 * C;
        ODISE_TRY(pk.upload(t.data(), 2 * cc * sizeof(f16), (void**)&m.qk.w));
        ODISE_TRY(pk.upload(b->data.data(), (size_t)2 * C * sizeof(float), (void**)&m.qk.b));
    } else {
        m.q.in = C; m.q.out = C;
        ODISE_TRY(pk.upload(t.data(), cc * sizeof(f16), (void**)&m.q.w));
        ODISE_TRY(pk.upload(b->data.data(), (size_t)C * sizeof(float), (void**)&m.q.b));
        m.k.in = C; m.k.out = C;
        ODISE_TRY(pk.upload(t.data() + cc, cc * sizeof(f16), (void**)&m.k.w));
        ODISE_TRY(pk.upload(b->data.data() + C, (size_t)C * sizeof(float), (void**)&m.k.b));
    }
    m.v.in = C; m.v.out = C; m.v.b = nullptr;
    ODISE_TRY(pk.upload(t.data() + 2 * cc, cc * sizeof(f16), (void**)&m.v.w));
    ODISE_TRY(pk.upload(b->data.data() + 2 * C, (size_t)C * sizeof(float), (void**)&m.v_bias));
    ODISE_TRY(pk.linear(key + ".out_proj", m.out));
    return ODISE_OK;
}

static int build_backbone_weights(odise_hip_ctx* ctx, ModelStore* ms, BackboneW* g, std::vector<void*>& owned) {
    AllocScope scope(ms, owned);
    Packer pk{ctx, ms, "backbone.feature_projections.", ""};
    for (int i = 0; i < 8; ++i) {
        const std::string k = std::to_string(i) + ".0";
        BottleneckW& b = g->proj[i];
        ODISE_TRY(build_convnorm(pk, k + ".conv1", b.c1, b.n1));
        ODISE_TRY(build_convnorm(pk, k + ".conv2", b.c2, b.n2));
        ODISE_TRY(build_convnorm(pk, k + ".conv3", b.c3, b.n3));
        b.has_sc = pk.find(k + ".shortcut.weight") != nullptr;
        if (b.has_sc) ODISE_TRY(build_convnorm(pk, k + ".shortcut", b.sc, b.nsc));
    }
    g->proj_dim = g->proj[0].c3.cout;
    return ODISE_OK;
}

static int maskgen_build_backbone(odise_hip_ctx* ctx) {
    ModelStore* ms = store_of(ctx);
    MaskGenModel* g = maskgen_of(ms);
    BackboneW fresh;
    std::vector<void*> owned;
    const int rc = build_backbone_weights(ctx, ms, &fresh, owned);
    if (rc != ODISE_OK) {   // the previous projections (if any) stay in place
        free_allocs(owned);
        return rc;
    }
    free_allocs(g->owned_backbone);   // device-synchronising: nothing queued still reads the weights this rebuild replaces
    g->owned_backbone.swap(owned);
    static_cast<BackboneW&>(*g) = fresh;
    g->backbone_built = true;
    return ODISE_OK;
}

static int build_head_weights(odise_hip_ctx* ctx, ModelStore* ms, HeadW* g, std::vector<void*>& owned) {
    AllocScope scope(ms, owned);
    Packer pk{ctx, ms, "sem_seg_head.pixel_decoder.", ""};
    for (int i = 0; i < 3; ++i) {
        ODISE_TRY(pk.conv("input_proj." + std::to_string(i) + ".0", g->in_proj[i]));
        ODISE_TRY(pk.norm("input_proj." + std::to_string(i) + ".1", g->in_proj_gn[i]));
    }
    g->C = g->in_proj[0].cout;
    const int C = g->C;
    ODISE_TRY(pk.vec_f32("transformer.level_embed", &g->enc_level_embed, 3 * C));
    g->enc_layers.clear();
    for (int i = 0;; ++i) {
        const std::string k = "transformer.encoder.layers." + std::to_string(i);
        if (!pk.find(k + ".norm1.weight")) break;
        MsdaLayerW L;
        ODISE_TRY(pk.linear(k + ".self_attn.sampling_offsets", L.off));
        ODISE_TRY(pk.linear(k + ".self_attn.attention_weights", L.aw));
        ODISE_TRY(pk.linear(k + ".self_attn.value_proj", L.value));
        ODISE_TRY(pk.linear(k + ".self_attn.output_proj", L.out));
        ODISE_TRY(pk.norm(k + ".norm1", L.norm1));
        ODISE_TRY(pk.linear(k + ".linear1", L.lin1));
        ODISE_TRY(pk.linear(k + ".linear2", L.lin2));
        ODISE_TRY(pk.norm(k + ".norm2", L.norm2));
        if (L.off.in == L.aw.in && L.off.b && L.aw.b) {
            const size_t wo = (size_t)L.off.out * L.off.in * sizeof(f16), wa = (size_t)L.aw.out * L.aw.in * sizeof(f16);
            char* w = nullptr;
            char* bb = nullptr;
            ODISE_CHECK_HIP(hipMalloc((void**)&w, wo + wa));
            ms->track(w);
            ODISE_CHECK_HIP(hipMalloc((void**)&bb, (size_t)(L.off.out + L.aw.out) * sizeof(float)));
            ms->track(bb);
            ODISE_CHECK_HIP(hipMemcpy(w, L.off.w, wo, hipMemcpyDeviceToDevice));
            ODISE_CHECK_HIP(hipMemcpy(w + wo, L.aw.w, wa, hipMemcpyDeviceToDevice));
            ODISE_CHECK_HIP(hipMemcpy(bb, L.off.b, (size_t)L.off.out * sizeof(float), hipMemcpyDeviceToDevice));
            ODISE_CHECK_HIP(hipMemcpy(bb + (size_t)L.off.out * sizeof(float), L.aw.b, (size_t)L.aw.out * sizeof(float), hipMemcpyDeviceToDevice));
            L.offaw.w = (f16*)w; L.offaw.b = (float*)bb; L.offaw.in = L.off.in; L.offaw.out = L.off.out + L.aw.out;
        }
        g->enc_layers.push_back(L);
    }
    if (g->enc_layers.empty()) {
        set_error("pixel decoder: no transformer.encoder.layers found");
        return ODISE_ERR_STATE;
    }
    g->enc_points = g->enc_layers[0].aw.out / (g->enc_heads * 3);
    ODISE_TRY(build_convnorm(pk, "adapter_1", g->adapter, g->adapter_gn));
    ODISE_TRY(build_convnorm(pk, "layer_1", g->layer1, g->layer1_gn));
    ODISE_TRY(pk.conv("mask_features", g->mask_feat));
    g->mask_feat_wT = g->mask_feat.w;  // a 1x1 conv weight [Cout][1][1][Cin] is already the [Cout, Cin] A operand
    // ---- transformer decoder -----------------------------------------------------------------------------------------
    Packer pd{ctx, ms, "sem_seg_head.predictor.", ""};
    g->dec_layers.clear();
    for (int i = 0;; ++i) {
        const std::string si = std::to_string(i);
        if (!pd.find("transformer_ffn_layers." + si + ".norm.weight")) break;
        DecLayerW L;
        ODISE_TRY(build_mha(pd, "transformer_cross_attention_layers." + si + ".multihead_attn", L.cross, C, false));
        ODISE_TRY(pd.norm("transformer_cross_attention_layers." + si + ".norm", L.cross.norm));
        ODISE_TRY(build_mha(pd, "transformer_self_attention_layers." + si + ".self_attn", L.self, C, true));
        ODISE_TRY(pd.norm("transformer_self_attention_layers." + si + ".norm", L.self.norm));
        ODISE_TRY(pd.linear("transformer_ffn_layers." + si + ".linear1", L.lin1));
        ODISE_TRY(pd.linear("transformer_ffn_layers." + si + ".linear2", L.lin2));
        ODISE_TRY(pd.norm("transformer_ffn_layers." + si + ".norm", L.ffn_norm));
        g->dec_layers.push_back(L);
    }
    if (g->dec_layers.empty()) {
        set_error("decoder: no transformer layers found");
        return ODISE_ERR_STATE;
    }
    ODISE_TRY(pd.norm("decoder_norm", g->decoder_norm));
    g->has_class_embed = pd.find("class_embed.weight") != nullptr;
    if (g->has_class_embed) {
        ODISE_TRY(pd.linear("class_embed", g->class_embed));
        if (g->class_embed.out != 2) {
            set_error("decoder: class_embed must have 2 outputs (object / no-object), found %d", g->class_embed.out);
            return ODISE_ERR_STATE;
        }
    }
    const HostTensor* qf = pd.find("query_feat.weight");
    if (!qf || qf->shape.size() != 2 || qf->shape[1] != C) {
        set_error("decoder: bad or missing query_feat.weight");
        return ODISE_ERR_STATE;
    }
    g->Q = (int)qf->shape[0];
    ODISE_TRY(pd.vec_f32("query_feat.weight", &g->query_feat, (int64_t)g->Q * C));
    ODISE_TRY(pd.vec_f32("query_embed.weight", &g->query_embed, (int64_t)g->Q * C));
    ODISE_TRY(pd.vec_f32("level_embed.weight", &g->dec_level_embed, 3 * C));
    {
        std::vector<f16> q16((size_t)g->Q * C);
        for (size_t i = 0; i < q16.size(); ++i) q16[i] = (f16)qf->data[i];
        ODISE_TRY(pd.upload(q16.data(), q16.size() * sizeof(f16), (void**)&g->query_feat16));
    }
    for (int i = 0; i < 3; ++i) ODISE_TRY(pd.linear("mask_embed.layers." + std::to_string(i), g->mask_mlp[i]));
    ODISE_TRY(pd.norm("post_mask_embed.pool_proj.0", g->pool_ln));
    ODISE_TRY(pd.linear("post_mask_embed.pool_proj.1", g->pool_proj));
    ODISE_TRY(pd.norm("post_mask_embed.mask_embed.0", g->post_ln));
    for (int i = 0; i < 3; ++i) ODISE_TRY(pd.linear("post_mask_embed.mask_embed.1.layers." + std::to_string(i), g->post_mlp[i]));
    const HostTensor* ls = pd.find("post_mask_embed.logit_scale");
    if (!ls || ls->numel() != 1) {
        set_error("decoder: bad or missing post_mask_embed.logit_scale");
        return ODISE_ERR_STATE;
    }
    g->logit_scale = std::min(expf(ls->data[0]), 100.0f);  // torch.clamp(logit_scale.exp(), max=100)  (odise.py:1004)
    return ODISE_OK;
}

static int maskgen_build_head(odise_hip_ctx* ctx) {
    ModelStore* ms = store_of(ctx);
    MaskGenModel* g = maskgen_of(ms);
    HeadW fresh;
    std::vector<void*> owned;
    const int rc = build_head_weights(ctx, ms, &fresh, owned);
    if (rc != ODISE_OK) {   // reload_head() with an incomplete checkpoint: the previous head stays in place and usable
        free_allocs(owned);
        return rc;
    }
    free_allocs(g->owned_head);   // device-synchronising; the previous head's ~56 MB of weights are released, not kept until the context dies
    g->owned_head.swap(owned);
    g->drop_size_caches();        // PE + level_embed tables are derived from these weights: a rebuilt head must not reuse the previous model's
    static_cast<HeadW&>(*g) = std::move(fresh);
    g->head_built = true;
    return ODISE_OK;
}

// ---- host-side tables --------------------------------------------------------------------------------------------------
// PositionEmbeddingSine(normalize=True, scale=2*pi, temperature=10000), position_encoding.py:29-52 -> [h*w, 2*npf] (y half | x half)
static void sine_pe(int h, int w, int npf, std::vector<float>& out) {
    out.assign((size_t)h * w * 2 * npf, 0.f);
    const float scale = 2.0f * (float)M_PI, eps = 1e-6f;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float ye = (float)(y + 1) / ((float)h + eps) * scale, xe = (float)(x + 1) / ((float)w + eps) * scale;
            float* o = out.data() + ((size_t)y * w + x) * 2 * npf;
            for (int i = 0; i < npf; ++i) {
                const float dim_t = powf(10000.0f, (float)(2 * (i / 2)) / (float)npf);
                const float py = ye / dim_t, px = xe / dim_t;
                o[i] = (i % 2 == 0) ? sinf(py) : cosf(py);
                o[npf + i] = (i % 2 == 0) ? sinf(px) : cosf(px);
            }
        }
}

// ---- BottleneckBlock: 1x1+GN+ReLU -> 3x3+GN+ReLU -> 1x1+GN ; ReLU(out + shortcut) (+ accum of the stride group) ----------
static int run_bottleneck(Exec& ex, const BottleneckW& w, const Act& x, const Act* accum, Act& out) {
    ODISE_TRY(ex.alloc(out, x.n, x.h, x.w, w.c3.cout));
    const size_t mk = ex.ms->arena.mark();
    Act a, an, b, bn, c, s, sn;
    ODISE_TRY(ex.conv(x, w.c1, a, 1, 0));
    ODISE_TRY(ex.group_norm(a, w.n1, an, 1e-5f, ODISE_ACT_RELU));
    ODISE_TRY(ex.conv(an, w.c2, b, 1, 1));
    ODISE_TRY(ex.group_norm(b, w.n2, bn, 1e-5f, ODISE_ACT_RELU));
    ODISE_TRY(ex.conv(bn, w.c3, c, 1, 0));
    const f16* resid = x.p;
    if (w.has_sc) {
        ODISE_TRY(ex.conv(x, w.sc, s, 1, 0));
        ODISE_TRY(ex.group_norm(s, w.nsc, sn, 1e-5f, ODISE_ACT_NONE));
        resid = sn.p;
    }
    ODISE_TRY(odise_hip_group_norm_ex(ex.ctx, c.p, out.p, w.n3.g, w.n3.b, c.n, c.h * c.w, c.c, 32, 1e-5f, ODISE_ACT_RELU, resid,
                                      accum ? accum->p : nullptr));
    ex.ms->arena.release(mk);
    return ODISE_OK;
}

static const int kFeatStride[8] = {4, 8, 32, 32, 16, 8, 8, 4};  // LdmExtractor strides clamped to [4,32] (feature_extractor.py:91-93)
static const int kGroups[4][3] = {{0, 7, -1}, {1, 5, 6}, {4, -1, -1}, {2, 3, -1}};  // s2, s3, s4, s5 in summation order
static const int kGroupStride[4] = {4, 8, 16, 32};

// slide-window boxes of an H x W image (feature_extractor.py:197-222): stride = window, the last row / column shifted inwards
static void window_boxes(int H, int W, int cs, std::vector<int>& boxes) {
    const int hg = (std::max(H - cs + cs - 1, 0)) / cs + 1, wg = (std::max(W - cs + cs - 1, 0)) / cs + 1;
    for (int hi = 0; hi < hg; ++hi)
        for (int wi = 0; wi < wg; ++wi) {
            const int y2 = std::min(hi * cs + cs, H), x2 = std::min(wi * cs + cs, W);
            boxes.push_back(std::max(y2 - cs, 0));
            boxes.push_back(std::max(x2 - cs, 0));
        }
}

// Encoder prefetch (engine.h Prefetch): enqueue the registered next batch's input padding, window extraction, VAE encoder and latent on the
// prefetch stream, behind everything the main stream holds so far (= the current batch's VAE lane).  Only batches of the current padded size
// are prefetched (the window table of that size is resident); anything else is dropped and simply computed by its own call.
static int prefetch_enqueue(odise_hip_ctx* ctx, ModelStore* ms, MaskGenModel* g) {
    Prefetch& pf = ms->pf;
    const PrefetchKey key = pf.pending;
    pf.has_pending = false;
    const int B = key.B;
    int H = 0, W = 0;
    for (int b = 0; b < B; ++b) { H = std::max(H, key.hw[2 * b]); W = std::max(W, key.hw[2 * b + 1]); }
    const int Hp = (int)round_up(H, 64), Wp = (int)round_up(W, 64);
    const int S = 512, cs = std::min(S, std::min(Hp, Wp));
    MaskGenModel::BoxEntry* hit = nullptr;
    for (auto& e : g->box_cache)
        if (e.H == Hp && e.W == Wp) hit = &e;
    if (!hit || cs != S) return ODISE_OK;   // not this size class: no prefetch
    const int K = hit->K, n = B * K;
    const int slot = 1 - pf.ready_slot;   // strictly alternating: the batch in progress may have come from the other slot and still reads it
    ODISE_TRY(ensure_prefetch_lane(ctx, ms, slot, encoder_arena_bytes(n, S, S) + (size_t)B * 3 * Hp * Wp * 4 + (size_t)n * 3 * S * S * 4));
    ODISE_CHECK_HIP(hipEventRecord(ctx->ev_pf_go, ctx->stream));
    const double macs0 = ms->macs;
    {
        PrefetchLane lane(ctx, ms, slot);
        ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_pf_go, 0));
        ms->arena.reset();
        Exec ex{ctx, ms};
        float* padded = (float*)ex.alloc_bytes((size_t)B * 3 * Hp * Wp * 4);
        float* crops = (float*)ex.alloc_bytes((size_t)n * 3 * S * S * 4);
        if (!padded || !crops) return ODISE_ERR_NOMEM;
        for (int b = 0; b < B; ++b)
            ODISE_TRY(launch_image_pad(ctx, key.images[b], key.layout, key.hw[2 * b], key.hw[2 * b + 1], padded + (size_t)b * 3 * Hp * Wp, Hp, Wp));
        ODISE_TRY(launch_crop_extract(ctx, padded, crops, B, 3, Hp, Wp, S, K, hit->dev));
        EncoderOut out;
        ODISE_TRY(extractor_encoder_only(ctx, ms, crops, n, S, S, out));
        ODISE_CHECK_HIP(hipEventRecord(ctx->ev_pf_done, ctx->stream));
        pf.out = out;
    }
    ms->macs = macs0;   // the MAC counter describes the call in progress
    pf.ready = key;
    pf.ready_slot = slot;
    pf.crops = n;
    pf.has_ready = true;
    ++pf.n_enqueued;
    return ODISE_OK;
}

// The prefetch is an optimisation of the NEXT call: when it cannot be enqueued (a side arena that does not fit, a failed launch) the call in
// progress must not fail with it - the registration is dropped, whatever reached the prefetch stream is never consumed (has_ready stays
// false), and the next batch is simply computed by its own call.
static void prefetch_try(odise_hip_ctx* ctx, ModelStore* ms, MaskGenModel* g) {
    if (prefetch_enqueue(ctx, ms, g) != ODISE_OK) {
        ms->pf.has_pending = ms->pf.has_ready = false;
        ++ms->pf.n_failed;
    }
}

static int backbone_forward(odise_hip_ctx* ctx, const float* image, int B, int H, int W, float** out4) {
    ModelStore* ms = store_of(ctx);
    MaskGenModel* g = ms->maskgen;
    if (!g || !g->backbone_built || !extractor_ready(ms)) {
        set_error("backbone_forward: call odise_hip_extractor_build and odise_hip_backbone_build first");
        return ODISE_ERR_STATE;
    }
    ODISE_REQUIRE(image && B >= 1 && H % 64 == 0 && W % 64 == 0 && H >= 64 && W >= 64,
                  "backbone_forward: image %dx%d must be a multiple of 64 (the caller pads, odise.py:238-242)", H, W);
    const int S = 512;                                      // backbone_in_size: what the extractor sees
    const int cs = std::min(S, std::min(H, W));             // slide window = min(backbone_in_size, short side)  (feature_extractor.py:199-203)
    // slide-window boxes (feature_extractor.py:197-222): stride = window, the last row / column shifted inwards
    std::vector<int> boxes;
    const int hg = (std::max(H - cs + cs - 1, 0)) / cs + 1, wg = (std::max(W - cs + cs - 1, 0)) / cs + 1;
    for (int hi = 0; hi < hg; ++hi)
        for (int wi = 0; wi < wg; ++wi) {
            const int y2 = std::min(hi * cs + cs, H), x2 = std::min(wi * cs + cs, W);
            boxes.push_back(std::max(y2 - cs, 0));
            boxes.push_back(std::max(x2 - cs, 0));
        }
    const int K = (int)boxes.size() / 2;
    {
        MaskGenModel::BoxEntry* hit = nullptr;
        for (auto& e : g->box_cache)
            if (e.H == H && e.W == W) hit = &e;
        if (!hit) {
            std::vector<int> all;  // [5][K][2]: image pixels, then s2..s5 feature pixels
            for (int k = 0; k < 2 * K; ++k) all.push_back(boxes[k]);
            for (int gi = 0; gi < 4; ++gi)
                for (int k = 0; k < 2 * K; ++k) all.push_back(boxes[k] / kGroupStride[gi]);
            ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));   // queued kernels may still read the entry about to be evicted
            if ((int)g->box_cache.size() >= MaskGenModel::kSizeCache) {
                size_t lru = 0;
                for (size_t i = 1; i < g->box_cache.size(); ++i) if (g->box_cache[i].stamp < g->box_cache[lru].stamp) lru = i;
                (void)hipFree(g->box_cache[lru].dev);
                g->box_cache.erase(g->box_cache.begin() + lru);
            }
            MaskGenModel::BoxEntry e;
            e.H = H; e.W = W; e.K = K;
            ODISE_CHECK_HIP(hipMalloc((void**)&e.dev, all.size() * sizeof(int)));
            if (hipMemcpy(e.dev, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipFree(e.dev);
                set_error("backbone_forward: uploading the window table failed");
                return ODISE_ERR_HIP;
            }
            g->box_cache.push_back(e);
            hit = &g->box_cache.back();
        }
        hit->stamp = ++g->size_clock;
        g->boxes_dev = hit->dev;
        g->boxes_K = hit->K;
    }
    ODISE_TRY(unet_prepare_timestep(ctx, ms, ms->unet, B * K, 0));
    size_t need = extractor_arena_bytes(B * K, S, S) + (size_t)B * K * 3 * S * S * 4;
    need += (size_t)B * K * (128 * 128 + 64 * 64 + 32 * 32 + 16 * 16) * 512 * 2 * 4;   // projections + temporaries
    need += (size_t)B * (H / 4) * (W / 4) * 512 * 2 * 2;                                 // stitched outputs
    need += (size_t)B * (H / 4) * (W / 4) * 256 * 2 * 24 + ((size_t)512 << 20);          // head working set (conservative)
    ODISE_TRY(ensure_arena(ctx, ms, need));
    Exec ex{ctx, ms};
    ms->arena.reset();
    ms->macs = 0.0;
    // outputs first (they must outlive everything else of this call)
    for (int gi = 0; gi < 4; ++gi) ODISE_TRY(ex.alloc(g->feats[gi], B, H / kGroupStride[gi], W / kGroupStride[gi], g->proj_dim));
    const size_t mk = ms->arena.mark();
    float* crops = (float*)ex.alloc_bytes((size_t)B * K * 3 * S * S * 4);
    if (!crops) return ODISE_ERR_NOMEM;
    if (cs == S) ODISE_TRY(launch_crop_extract(ctx, image, crops, B, 3, H, W, S, K, g->boxes_dev));
    else ODISE_TRY(launch_crop_resize_bicubic(ctx, image, crops, B, 3, H, W, cs, S, K, g->boxes_dev));   // windows below 512: bicubic to 512 x 512
    // The extractor leaves its second lane (CLIP -> UNet) un-joined: the projections of the VAE taps - the whole s2 group (enc5, dec5: the
    // largest maps) and enc7 - are enqueued behind the VAE decoder and run while the UNet is still busy on the other stream; the main
    // stream waits for the UNet right before the first projection that reads one of its taps.  Summation order inside a group unchanged.
#ifdef ODISE_TOOLS
    static const bool defer_join = getenv("ODISE_NO_DEFER_JOIN") == nullptr;   // A/B
#else
    const bool defer_join = true;
#endif
    stage_mark(ctx, "backbone: crops extracted");
    ODISE_TRY(extractor_launch(ctx, ms, crops, B * K, S, S, false, /*join=*/!defer_join));
    stage_mark(ctx, "extractor: VAE lane done (main stream; the CLIP -> UNet lane is joined later)");
    ODISE_TRY(maskclip_planned_pass(ctx, ms));   // ODISE_OPT_MASKCLIP_PASSES 3: MaskCLIP's image tokens behind the UNet on the second lane (engine.h ClipKV)
    ms->pf.use_now = false;
    if (ms->pf.has_pending && ctx->prefetch_start == 0) prefetch_try(ctx, ms, g);   // the NEXT batch's encoder behind this batch's VAE lane
    const Act* taps = extractor_taps(ms);
    bool joined = false;
    for (int gi = 0; gi < 4; ++gi) {
        const int stride = kGroupStride[gi], fs = cs / stride;   // features are restored to window / stride (feature_extractor.py:165-168)
        Act acc;
        bool have = false;
        for (int j = 0; j < 3 && kGroups[gi][j] >= 0; ++j) {
            const int idx = kGroups[gi][j];
            if (!joined && idx >= 2 && idx <= 5) {   // u2, u5, u8, u11
                stage_mark(ctx, "backbone: VAE-tap projections done");
                ODISE_TRY(extractor_join(ctx));
                stage_mark(ctx, "backbone: UNet lane joined");
                joined = true;
            }
            Act x = taps[idx];
            if (x.h != fs || x.w != fs) {  // restore to crop/stride (nearest; only u2: 8x8 -> 16x16)
                Act up;
                ODISE_TRY(ex.alloc(up, x.n, fs, fs, x.c));
                ODISE_TRY(launch_upsample_nearest(ctx, x.p, up.p, x.n, x.h, x.w, fs, fs, x.c));
                x = up;
            }
            Act o;
            ODISE_TRY(run_bottleneck(ex, g->proj[idx], x, have ? &acc : nullptr, o));
            acc = o;
            have = true;
        }
        ODISE_TRY(launch_stitch(ctx, acc.p, g->feats[gi].p, out4 ? out4[gi] : nullptr, B, K, g->boxes_dev + (size_t)(1 + gi) * 2 * K, fs, fs,
                                H / stride, W / stride, g->proj_dim));
    }
    if (!joined) ODISE_TRY(extractor_join(ctx));
    // ... or (default) behind the whole backbone: beside the serial tail of small launches (head, MaskCLIP, post-processing)
    if (ms->pf.has_pending) prefetch_try(ctx, ms, g);
    ms->arena.release(mk);
    g->last_macs = ms->macs;
    (void)kFeatStride;
    return ODISE_OK;
}

// ---- sem_seg_head ------------------------------------------------------------------------------------------------------
static int copy_rows_2d(odise_hip_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height) {
    ODISE_CHECK_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, ctx->stream));
    return ODISE_OK;
}

static int gemm_f32out(Exec& ex, const f16* x, int64_t M, const LinW& w, float* y) {
    odise_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.M = (int)M; d.N = w.out; d.K = w.in;
    d.A = x; d.lda = w.in; d.W = w.w; d.ldw = w.in;
    d.C = y; d.ldc = w.out; d.c_dtype = ODISE_F32; d.bias_n = w.b; d.alpha = 1.f; d.batch = 1;
    return ex.gemm(d);
}

// V^T[b] = Wv x[b]^T + bv  -> [B, C, ldvt]
static int gemm_vt(Exec& ex, const LinW& v, const float* v_bias, const f16* x, int B, int64_t L, int64_t ldvt, f16* vt) {
    odise_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.M = v.out; d.N = (int)L; d.K = v.in;
    d.A = v.w; d.lda = v.in; d.W = x; d.ldw = v.in; d.strideW = L * v.in;
    d.C = vt; d.ldc = ldvt; d.strideC = (int64_t)v.out * ldvt; d.c_dtype = ODISE_F16;
    d.bias_m = v_bias; d.alpha = 1.f; d.batch = B;
    return ex.gemm(d);
}

static int mlp3(Exec& ex, const LinW* w, const f16* x, int64_t M, f16* t1, f16* t2, f16* y) {
    ODISE_TRY(ex.linear(x, M, w[0], t1, ODISE_ACT_RELU));
    ODISE_TRY(ex.linear(t1, M, w[1], t2, ODISE_ACT_RELU));
    ODISE_TRY(ex.linear(t2, M, w[2], y));
    return ODISE_OK;
}

// Outputs of the pixel decoder handed to the predictor (all in the arena)
struct PixDec {
    Act ms_feat[3];   // multi_scale_features low -> high resolution (s5, s4, s3 grids), NHWC
    Act mf;           // mask_features, pixel-major [B, HW4, C]  (W operand of the mask-logit GEMM)
    f16* mfT = nullptr;  // mask_features, channel-major [B, C, HW4] (W operand of the pooling GEMM)
    int h4 = 0, w4 = 0;
    // what the masked decoder reads of the encoder output alone (decoder_memory_projections): K and V^T of every layer's cross-attention
    static constexpr int kMaxDecLayers = 16;
    f16* kproj[kMaxDecLayers] = {};
    f16* vtproj[kMaxDecLayers] = {};
    bool kv_done = false, kv_on_lane2 = false;
};

static int ensure_pe_tables(odise_hip_ctx* ctx, ModelStore* ms, MaskGenModel* g, const int hs[3], const int ws[3]);

// The cross-attention keys and values of ALL decoder layers (odise.py:676-690: src + level_embed, pos; mask2former_transformer_decoder.py:95-118)
// depend on the encoder output only: 2 x layers GEMMs that sat inside the layers' dependent chains (20-52 us of each) run here, once - on the
// second lane beside the FPN half of the pixel decoder when the caller says so (side = true; the predictor waits for ev_join before its first
// attention).
static int decoder_memory_projections(odise_hip_ctx* ctx, MaskGenModel* g, PixDec& pd, bool side) {
    ModelStore* ms = store_of(ctx);
    Exec ex{ctx, ms};
    const int C = g->C, B = pd.ms_feat[0].n, nl = (int)g->dec_layers.size();
    ODISE_REQUIRE(nl <= PixDec::kMaxDecLayers, "masked decoder: %d layers", nl);
    int hs[3], ws[3];
    for (int l = 0; l < 3; ++l) { hs[l] = pd.ms_feat[l].h; ws[l] = pd.ms_feat[l].w; }
    ODISE_TRY(ensure_pe_tables(ctx, ms, g, hs, ws));
    f16* valin[3]; f16* keyin[3];
    for (int l = 0; l < 3; ++l) {
        const int64_t P = (int64_t)hs[l] * ws[l];
        valin[l] = (f16*)ex.alloc_bytes((size_t)B * P * C * 2);
        keyin[l] = (f16*)ex.alloc_bytes((size_t)B * P * C * 2);
        if (!valin[l] || !keyin[l]) return ODISE_ERR_NOMEM;
    }
    for (int i = 0; i < nl; ++i) {
        const int l = i % 3;
        const int64_t P = (int64_t)hs[l] * ws[l];
        pd.kproj[i] = (f16*)ex.alloc_bytes((size_t)B * P * C * 2);
        pd.vtproj[i] = (f16*)ex.alloc_bytes((size_t)B * C * round_up(P, 8) * 2);
        if (!pd.kproj[i] || !pd.vtproj[i]) return ODISE_ERR_NOMEM;
    }
    auto run = [&]() -> int {
        for (int l = 0; l < 3; ++l) {
            const int64_t P = (int64_t)hs[l] * ws[l];
            ODISE_TRY(launch_add_vec_table(ctx, pd.ms_feat[l].p, g->dec_level_embed + (size_t)l * C, nullptr, valin[l], B, (int)P, C));
            ODISE_TRY(launch_add_vec_table(ctx, valin[l], nullptr, g->pe[l], keyin[l], B, (int)P, C));
        }
        for (int i = 0; i < nl; ++i) {
            const DecLayerW& L = g->dec_layers[i];
            const int l = i % 3;
            const int64_t P = (int64_t)hs[l] * ws[l];
            ODISE_TRY(ex.linear(keyin[l], B * P, L.cross.k, pd.kproj[i]));
            ODISE_TRY(gemm_vt(ex, L.cross.v, L.cross.v_bias, valin[l], B, P, round_up(P, 8), pd.vtproj[i]));
        }
        return ODISE_OK;
    };
    const bool lane2 = side && ctx->lanes == 2 && ctx->stream2 && ctx->ev_fork && ctx->ev_join;
    if (lane2) {
        ODISE_CHECK_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        Lane2 lane(ctx, ms);
        ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_fork, 0));
        ODISE_TRY(run());
        ODISE_CHECK_HIP(hipEventRecord(ctx->ev_join, ctx->stream));
    } else {
        ODISE_TRY(run());
    }
    pd.kv_done = true;
    pd.kv_on_lane2 = lane2;
    return ODISE_OK;
}

// sine positional tables of the three transformer levels (shared by the pixel decoder's encoder and the predictor)
static int ensure_pe_tables(odise_hip_ctx* ctx, ModelStore* ms, MaskGenModel* g, const int hs[3], const int ws[3]) {
    (void)ms;
    const int C = g->C;
    MaskGenModel::PeEntry* hit = nullptr;
    for (auto& e : g->pe_cache) {
        bool same = true;
        for (int l = 0; l < 3; ++l) same = same && e.hs[l] == hs[l] && e.ws[l] == ws[l];
        if (same) hit = &e;
    }
    if (!hit) {
        int starts[3], Lq = 0;
        for (int l = 0; l < 3; ++l) { starts[l] = Lq; Lq += hs[l] * ws[l]; }
        std::vector<float> all((size_t)Lq * C), lvl, emb((size_t)3 * C, 0.f);
        ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));   // also: queued kernels may still read the entry about to be evicted
        if (g->enc_level_embed) ODISE_CHECK_HIP(hipMemcpy(emb.data(), g->enc_level_embed, emb.size() * 4, hipMemcpyDeviceToHost));
        if ((int)g->pe_cache.size() >= MaskGenModel::kSizeCache) {
            size_t lru = 0;
            for (size_t i = 1; i < g->pe_cache.size(); ++i) if (g->pe_cache[i].stamp < g->pe_cache[lru].stamp) lru = i;
            for (float* p : g->pe_cache[lru].pe) (void)hipFree(p);
            (void)hipFree(g->pe_cache[lru].all);
            g->pe_cache.erase(g->pe_cache.begin() + lru);
        }
        MaskGenModel::PeEntry e;
        auto fail = [&](const char* what) {
            for (float* p : e.pe) (void)hipFree(p);
            (void)hipFree(e.all);
            set_error("positional tables: %s failed", what);
            return ODISE_ERR_HIP;
        };
        for (int l = 0; l < 3; ++l) {
            e.hs[l] = hs[l]; e.ws[l] = ws[l];
            sine_pe(hs[l], ws[l], C / 2, lvl);
            if (hipMalloc((void**)&e.pe[l], lvl.size() * 4) != hipSuccess) return fail("hipMalloc");
            if (hipMemcpy(e.pe[l], lvl.data(), lvl.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy");
            for (size_t p = 0; p < (size_t)hs[l] * ws[l]; ++p)
                for (int c = 0; c < C; ++c) all[((size_t)starts[l] + p) * C + c] = lvl[p * C + c] + emb[(size_t)l * C + c];
        }
        if (hipMalloc((void**)&e.all, all.size() * 4) != hipSuccess) return fail("hipMalloc");
        if (hipMemcpy(e.all, all.data(), all.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy");
        g->pe_cache.push_back(e);
        hit = &g->pe_cache.back();
    }
    hit->stamp = ++g->size_clock;
    for (int l = 0; l < 3; ++l) g->pe[l] = hit->pe[l];
    g->enc_pos_all = hit->all;
    return ODISE_OK;
}

static int g_msda_unfused = 0;   // tools hook (odise_hip_msda_unfused): 1 = prepare kernel + the native-op kernel, for bit-compare and A/B runs

// MSDeformAttnPixelDecoder.forward_features (msdeformattn.py:314-358)
static int pixel_decoder_forward(odise_hip_ctx* ctx, const Act feats[4], PixDec& pd, bool decoder_follows = false) {
    ModelStore* ms = store_of(ctx);
    MaskGenModel* g = ms->maskgen;
    Exec ex{ctx, ms};
    const int C = g->C, B = feats[0].n, Q = g->Q;
    const int hs[3] = {feats[3].h, feats[2].h, feats[1].h}, ws[3] = {feats[3].w, feats[2].w, feats[1].w};  // s5, s4, s3
    int starts[3], Lq = 0;
    for (int l = 0; l < 3; ++l) { starts[l] = Lq; Lq += hs[l] * ws[l]; }
    ODISE_TRY(ensure_pe_tables(ctx, ms, g, hs, ws));
    const int64_t MT = (int64_t)B * Lq;
    // ---- pixel decoder: input_proj (1x1 + GN) into one [B, Lq, C] token matrix ----------------------------------------------
    f16* src = (f16*)ex.alloc_bytes((size_t)MT * C * 2);
    f16* qin = (f16*)ex.alloc_bytes((size_t)MT * C * 2);
    f16* x1 = (f16*)ex.alloc_bytes((size_t)MT * C * 2);
    f16* val = (f16*)ex.alloc_bytes((size_t)MT * C * 2);
    f16* samp = (f16*)ex.alloc_bytes((size_t)MT * C * 2);
    f16* hid = (f16*)ex.alloc_bytes((size_t)MT * g->enc_layers[0].lin1.out * 2);
    const int M8 = g->enc_heads, LP = 3 * g->enc_points;
    float* offaw = (float*)ex.alloc_bytes((size_t)MT * M8 * LP * 3 * 4);   // [MT, 2 MLP | MLP] of the stacked projection; the separate forms use its two parts
    float* off = offaw;
    float* aw = offaw ? offaw + (size_t)MT * M8 * LP * 2 : nullptr;
    float* loc = (float*)ex.alloc_bytes((size_t)MT * M8 * LP * 2 * 4);
    float* wts = (float*)ex.alloc_bytes((size_t)MT * M8 * LP * 4);
    if (!src || !qin || !x1 || !val || !samp || !hid || !off || !aw || !loc || !wts) return ODISE_ERR_NOMEM;
    for (int l = 0; l < 3; ++l) {
        const Act& f = feats[3 - l];
        const size_t mk = ms->arena.mark();
        Act t, tn;
        ODISE_TRY(ex.conv(f, g->in_proj[l], t, 1, 0));
        ODISE_TRY(ex.group_norm(t, g->in_proj_gn[l], tn, 1e-5f, ODISE_ACT_NONE));
        const size_t rowb = (size_t)hs[l] * ws[l] * C * 2;
        ODISE_TRY(copy_rows_2d(ctx, src + (size_t)starts[l] * C, (size_t)Lq * C * 2, tn.p, rowb, rowb, B));
        ms->arena.release(mk);
    }
    int64_t ss[6], ls[3];
    for (int l = 0; l < 3; ++l) { ss[2 * l] = hs[l]; ss[2 * l + 1] = ws[l]; ls[l] = starts[l]; }
    // query = src + pos (msdeformattn.py:108-110): for the first layer a pass of its own, afterwards the second output of the LayerNorm that
    // writes src (the same bits); sampling_offsets | attention_weights of a layer are ONE GEMM over that query
    ODISE_TRY(launch_add_vec_table(ctx, src, nullptr, g->enc_pos_all, qin, B, Lq, C));
    const bool fused_msda = msda_fused_ok(M8, C / M8, 3, g->enc_points) && !g_msda_unfused;
    for (size_t li = 0; li < g->enc_layers.size(); ++li) {
        const MsdaLayerW& L = g->enc_layers[li];
        const bool last = li + 1 == g->enc_layers.size();
        ODISE_TRY(ex.linear(src, MT, L.value, val));
        if (fused_msda && L.offaw.w) {
            const int nof = L.off.out, ld = L.offaw.out;
            ODISE_TRY(gemm_f32out(ex, qin, MT, L.offaw, offaw));
            ODISE_TRY(launch_msda_fused(ctx, val, offaw, offaw + nof, samp, hs, ws, starts, B, Lq, M8, Lq, ld, ld));
        } else {
            ODISE_TRY(gemm_f32out(ex, qin, MT, L.off, off));
            ODISE_TRY(gemm_f32out(ex, qin, MT, L.aw, aw));
            if (fused_msda) {
                ODISE_TRY(launch_msda_fused(ctx, val, off, aw, samp, hs, ws, starts, B, Lq, M8, Lq));
            } else {
                ODISE_TRY(launch_msda_prepare(ctx, off, aw, loc, wts, B, Lq, M8, 3, g->enc_points, hs, ws, starts));
                ODISE_TRY(odise_hip_ms_deform_attn_forward(ctx, val, ss, ls, loc, wts, B, Lq, M8, C / M8, Lq, 3, g->enc_points, 128, ODISE_F16, samp));
            }
        }
        ms->macs += (double)MT * M8 * LP * 4 * (C / M8);                                           // bilinear taps
        ODISE_TRY(ex.linear(samp, MT, L.out, x1, ODISE_ACT_NONE, src));
        ODISE_TRY(ex.layer_norm(x1, src, MT, L.norm1, 1e-5f));
        ODISE_TRY(ex.linear(src, MT, L.lin1, hid, ODISE_ACT_RELU));
        ODISE_TRY(ex.linear(hid, MT, L.lin2, x1, ODISE_ACT_NONE, src));
        ODISE_TRY(layer_norm_add_table(ctx, x1, src, L.norm2.g, L.norm2.b, (int)MT, L.norm2.c, 1e-5f, last ? nullptr : qin, g->enc_pos_all, Lq));
    }
    // multi-scale features (contiguous per level) = split of the encoder output
    Act (&ms_feat)[3] = pd.ms_feat;
    for (int l = 0; l < 3; ++l) {
        ODISE_TRY(ex.alloc(ms_feat[l], B, hs[l], ws[l], C));
        const size_t rowb = (size_t)hs[l] * ws[l] * C * 2;
        ODISE_TRY(copy_rows_2d(ctx, ms_feat[l].p, rowb, src + (size_t)starts[l] * C, (size_t)Lq * C * 2, rowb, B));
    }
    if (decoder_follows) ODISE_TRY(decoder_memory_projections(ctx, g, pd, true));   // beside the FPN half below
    // FPN level on s2 + mask features (both layouts)
    const Act& s2 = feats[0];
    const int64_t HW4 = (int64_t)s2.h * s2.w;
    Act lat, latn, fsum, o3, o3n;
    Act& mf = pd.mf;
    ODISE_TRY(ex.conv(s2, g->adapter, lat, 1, 0));
    ODISE_TRY(ex.group_norm(lat, g->adapter_gn, latn, 1e-5f, ODISE_ACT_NONE));
    ODISE_TRY(ex.alloc(fsum, B, s2.h, s2.w, C));
    ODISE_TRY(launch_bilinear_add(ctx, latn.p, ms_feat[2].p, fsum.p, B, hs[2], ws[2], s2.h, s2.w, C));
    ODISE_TRY(ex.conv(fsum, g->layer1, o3, 1, 1));
    ODISE_TRY(ex.group_norm(o3, g->layer1_gn, o3n, 1e-5f, ODISE_ACT_RELU));
    ODISE_TRY(ex.conv(o3n, g->mask_feat, mf, 1, 0));                     // [B, HW4, C]  (pixel-major: W operand of the mask-logit GEMM)
    f16* mfT = (f16*)ex.alloc_bytes((size_t)B * C * HW4 * 2);            // [B, C, HW4] (channel-major: W operand of the pooling GEMM)
    if (!mfT) return ODISE_ERR_NOMEM;
    {
        LinW v; v.w = g->mask_feat_wT; v.in = C; v.out = g->mask_feat.cout; v.b = nullptr;
        ODISE_TRY(gemm_vt(ex, v, g->mask_feat.b, o3n.p, B, HW4, HW4, mfT));
    }
    pd.mfT = mfT; pd.h4 = s2.h; pd.w4 = s2.w;
    (void)Q;
    return ODISE_OK;
}

// ODISEMultiScaleMaskedTransformerDecoder.forward (odise.py:642-727) + PooledMaskEmbed of the final prediction (odise.py:984-1015)
static int predictor_forward(odise_hip_ctx* ctx, PixDec& pd) {
    ModelStore* ms = store_of(ctx);
    MaskGenModel* g = ms->maskgen;
    Exec ex{ctx, ms};
    const int C = g->C, B = pd.ms_feat[0].n, Q = g->Q;
    const int hs[3] = {pd.ms_feat[0].h, pd.ms_feat[1].h, pd.ms_feat[2].h}, ws[3] = {pd.ms_feat[0].w, pd.ms_feat[1].w, pd.ms_feat[2].w};
    ODISE_TRY(ensure_pe_tables(ctx, ms, g, hs, ws));
    const Act (&ms_feat)[3] = pd.ms_feat;
    const Act& mf = pd.mf;
    f16* mfT = pd.mfT;
    struct { int h, w; } s2{pd.h4, pd.w4};
    const int64_t HW4 = (int64_t)s2.h * s2.w;
    // ---- masked transformer decoder ----------------------------------------------------------------------------------------
    const int64_t MQ = (int64_t)B * Q;
    if (!pd.kv_done) ODISE_TRY(decoder_memory_projections(ctx, g, pd, false));   // (the stand-alone predictor: nothing ran ahead)
    // The attention mask of layer i+1 is the prediction resized to that layer's level and thresholded (odise.py:756-765: F.interpolate(...,
    // mode="bilinear") of the [Q, H/4, W/4] logits, then sigmoid < 0.5).  A mask logit is linear in the mask features, and bilinear resizing is
    // linear too: resize(me . mf) = me . resize(mf).  The three resized copies of mask_features are made once (mfl), and the NINE intermediate
    // predictions are Q x P_l products (P_l = 1/4 .. 1/64 of the pixels) instead of the full Q x HW4 mask tensor each (50 us + the resize
    // kernel per layer on the serial tail); only the final prediction is computed at H/4 x W/4.
    f16* mfl[3];
    for (int l = 0; l < 3; ++l) {
        mfl[l] = (f16*)ex.alloc_bytes((size_t)B * hs[l] * ws[l] * C * 2);
        if (!mfl[l]) return ODISE_ERR_NOMEM;
        ODISE_TRY(launch_bilinear_add(ctx, nullptr, mf.p, mfl[l], B, s2.h, s2.w, hs[l], ws[l], C));
    }
    const int64_t maxP = (int64_t)hs[2] * ws[2];
    const int64_t ldm = round_up(maxP, 8);
    f16* out = (f16*)ex.alloc_bytes((size_t)MQ * C * 2);
    f16* tq = (f16*)ex.alloc_bytes((size_t)MQ * C * 2);
    f16* tqe = (f16*)ex.alloc_bytes((size_t)MQ * C * 2);
    f16* t1 = (f16*)ex.alloc_bytes((size_t)MQ * 2048 * 2);
    f16* t2 = (f16*)ex.alloc_bytes((size_t)MQ * 2 * C * 2);
    f16* dn = (f16*)ex.alloc_bytes((size_t)MQ * C * 2);
    f16* me = (f16*)ex.alloc_bytes((size_t)MQ * C * 2);
    f16* att = (f16*)ex.alloc_bytes((size_t)MQ * C * 2);
    f16* qb = (f16*)ex.alloc_bytes((size_t)MQ * 2 * C * 2);
    f16* vtb = (f16*)ex.alloc_bytes((size_t)B * C * round_up(Q, 8) * 2);   // V^T of the queries' self-attention
    float* lgl = (float*)ex.alloc_bytes((size_t)MQ * maxP * 4);              // prediction at the next layer's level: the GEMM's fp32 accumulators (its sign is the decision)
    f16* masks = (f16*)ex.alloc_bytes((size_t)MQ * HW4 * 2);
    uint8_t* amask = (uint8_t*)ex.alloc_bytes((size_t)MQ * ldm);
    f16* m01 = (f16*)ex.alloc_bytes((size_t)MQ * HW4 * 2);
    float* inv = (float*)ex.alloc_bytes((size_t)MQ * 4);
    f16* pooled = (f16*)ex.alloc_bytes((size_t)MQ * C * 2);
    f16* pooled_x = (f16*)ex.alloc_bytes((size_t)MQ * C * 2);
    f16* mask_embed = (f16*)ex.alloc_bytes((size_t)MQ * C * 2);
    if (!out || !tq || !tqe || !t1 || !t2 || !dn || !me || !att || !qb || !lgl || !vtb || !masks || !amask || !m01 || !inv || !pooled || !pooled_x ||
        !mask_embed)
        return ODISE_ERR_NOMEM;
    // output = query_feat broadcast over the batch
    ODISE_TRY(launch_broadcast_rows(ctx, g->query_feat16, out, (int64_t)Q * C, B));

    auto prediction_heads = [&](int target_level) -> int {  // forward_prediction_heads (odise.py:729-776), mask branch only
        ODISE_TRY(ex.layer_norm(out, dn, MQ, g->decoder_norm, 1e-5f));
        ODISE_TRY(mlp3(ex, g->mask_mlp, dn, MQ, t1, t2, me));
        odise_gemm_desc d;
        memset(&d, 0, sizeof(d));  // outputs_mask[b] = mask_embed[b] (Q x C) . mask_features[b]^T (HW4 x C)
        d.M = Q; d.K = C;
        d.A = me; d.lda = C; d.strideA = (int64_t)Q * C;
        d.ldw = C; d.c_dtype = ODISE_F16; d.alpha = 1.f; d.batch = B;
        if (target_level >= 0) {   // ... at the level the next layer attends to
            const int64_t P = (int64_t)hs[target_level] * ws[target_level];
            d.N = (int)P; d.W = mfl[target_level]; d.strideW = P * C;
            d.C = lgl; d.ldc = P; d.strideC = (int64_t)Q * P; d.c_dtype = ODISE_F32;
            ODISE_TRY(ex.gemm(d));
            ODISE_TRY(launch_attn_mask_f32(ctx, lgl, amask, MQ, hs[target_level], ws[target_level], hs[target_level], ws[target_level], ldm));
        } else {                   // the final prediction
            d.N = (int)HW4; d.W = mf.p; d.strideW = HW4 * C;
            d.C = masks; d.ldc = HW4; d.strideC = (int64_t)Q * HW4;
            ODISE_TRY(ex.gemm(d));
        }
        return ODISE_OK;
    };
    ODISE_TRY(prediction_heads(0));
    const int heads = g->dec_heads, D = C / heads;
    const int nl = (int)g->dec_layers.size();
    for (int i = 0; i < nl; ++i) {
        const DecLayerW& L = g->dec_layers[i];
        const int l = i % 3;
        const int64_t P = (int64_t)hs[l] * ws[l];
        const int64_t ldv = round_up(P, 8);
        // masked cross-attention (keys = level l); query + query_embed left the LayerNorm that wrote `out` as its second output (tqe)
        if (i == 0) ODISE_TRY(launch_add_vec_table(ctx, out, nullptr, g->query_embed, tqe, B, Q, C));
        ODISE_TRY(ex.linear(tqe, MQ, L.cross.q, qb));
        if (i == 0 && pd.kv_on_lane2) ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));   // the keys / values of every layer (decoder_memory_projections)
        odise_attn_desc a;
        memset(&a, 0, sizeof(a));
        a.B = B; a.H = heads; a.Lq = Q; a.Lk = (int)P; a.D = D;
        a.Q = qb; a.ldq = C; a.strideQ = (int64_t)Q * C;
        a.K = pd.kproj[i]; a.ldk = C; a.strideK = P * C;
        a.Vt = pd.vtproj[i]; a.ldvt = ldv; a.strideVt = (int64_t)C * ldv;
        a.O = att; a.ldo = C; a.strideO = (int64_t)Q * C;
        a.mask = amask; a.ldmask = ldm; a.strideMask = (int64_t)Q * ldm;
        a.scale = 1.0f / sqrtf((float)D);
        ODISE_TRY(ex.attention(a));
        ODISE_TRY(ex.linear(att, MQ, L.cross.out, tq, ODISE_ACT_NONE, out));
        ODISE_TRY(layer_norm_add_table(ctx, tq, out, L.cross.norm.g, L.cross.norm.b, (int)MQ, C, 1e-5f, tqe, g->query_embed, Q));
        // self-attention over the queries
        ODISE_TRY(ex.linear(tqe, MQ, L.self.qk, qb));
        const int64_t ldq = round_up(Q, 8);
        ODISE_TRY(gemm_vt(ex, L.self.v, L.self.v_bias, out, B, Q, ldq, vtb));
        memset(&a, 0, sizeof(a));
        a.B = B; a.H = heads; a.Lq = Q; a.Lk = Q; a.D = D;
        a.Q = qb; a.ldq = 2 * C; a.strideQ = (int64_t)Q * 2 * C;
        a.K = qb + C; a.ldk = 2 * C; a.strideK = (int64_t)Q * 2 * C;
        a.Vt = vtb; a.ldvt = ldq; a.strideVt = (int64_t)C * ldq;
        a.O = att; a.ldo = C; a.strideO = (int64_t)Q * C;
        a.scale = 1.0f / sqrtf((float)D);
        ODISE_TRY(ex.attention(a));
        ODISE_TRY(ex.linear(att, MQ, L.self.out, tq, ODISE_ACT_NONE, out));
        ODISE_TRY(ex.layer_norm(tq, out, MQ, L.self.norm, 1e-5f));
        // FFN
        ODISE_TRY(ex.linear(out, MQ, L.lin1, t1, ODISE_ACT_RELU));
        ODISE_TRY(ex.linear(t1, MQ, L.lin2, tq, ODISE_ACT_NONE, out));
        ODISE_TRY(layer_norm_add_table(ctx, tq, out, L.ffn_norm.g, L.ffn_norm.b, (int)MQ, C, 1e-5f, i + 1 < nl ? tqe : nullptr, g->query_embed, Q));
        ODISE_TRY(prediction_heads(i + 1 < nl ? (i + 1) % 3 : -1));
    }
    // ---- learned (object, no-object) logits of the final prediction head: outputs_class = class_embed(decoder_output), odise.py:734
    g->class_logits = nullptr;
    if (g->has_class_embed) {
        float* cl = (float*)ex.alloc_bytes((size_t)MQ * 2 * 4);
        if (!cl) return ODISE_ERR_NOMEM;
        odise_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.M = (int)MQ; d.N = 2; d.K = C;
        d.A = dn; d.lda = C; d.W = g->class_embed.w; d.ldw = C;
        d.C = cl; d.ldc = 2; d.c_dtype = ODISE_F32; d.bias_n = g->class_embed.b; d.alpha = 1.f; d.batch = 1;
        ODISE_TRY(ex.gemm(d));
        g->class_logits = cl;
    }
    // ---- PooledMaskEmbed on the final prediction (odise.py:984-1015) ------------------------------------------------------------
    ODISE_TRY(launch_mask_binarize_f16(ctx, masks, m01, inv, MQ, (int)HW4));
    for (int b = 0; b < B; ++b) {
        odise_gemm_desc d;
        memset(&d, 0, sizeof(d));  // pooled[b] = (m01[b] (Q x HW4) . x[b]^T (C x HW4)) / count
        d.M = Q; d.N = C; d.K = (int)HW4;
        d.A = m01 + (size_t)b * Q * HW4; d.lda = HW4;
        d.W = mfT + (size_t)b * C * HW4; d.ldw = HW4;
        d.C = pooled + (size_t)b * Q * C; d.ldc = C; d.c_dtype = ODISE_F16;
        d.scale_m = inv + (size_t)b * Q; d.alpha = 1.f; d.batch = 1;
        ODISE_TRY(ex.gemm(d));
    }
    ODISE_TRY(ex.layer_norm(pooled, tq, MQ, g->pool_ln, 1e-5f));
    ODISE_TRY(ex.linear(tq, MQ, g->pool_proj, pooled_x, ODISE_ACT_NONE, dn));   // mask_pooled_x = pool_proj(pooled) + decoder_output
    ODISE_TRY(ex.layer_norm(pooled_x, tq, MQ, g->post_ln, 1e-5f));
    ODISE_TRY(mlp3(ex, g->post_mlp, tq, MQ, t1, t2, mask_embed));
    g->pred_masks = masks; g->mask_embed = mask_embed; g->mask_pooled = pooled_x;
    g->out_B = B; g->out_h = s2.h; g->out_w = s2.w;
    g->last_macs = ms->macs;
    return ODISE_OK;
}

static int head_forward(odise_hip_ctx* ctx, const Act feats[4]) {
    PixDec pd;
    ODISE_TRY(pixel_decoder_forward(ctx, feats, pd, true));
    stage_mark(ctx, "head: pixel decoder done");
    const int rc = predictor_forward(ctx, pd);
    stage_mark(ctx, "head: masked decoder done");
    return rc;
}

int head_outputs(ModelStore* ms, HeadOutputs* out) {
    MaskGenModel* g = ms->maskgen;
    if (!g || !g->head_built || !g->pred_masks) {
        set_error("no head outputs available: call odise_hip_head_forward first");
        return ODISE_ERR_STATE;
    }
    out->pred_masks = g->pred_masks; out->mask_embed = g->mask_embed;
    out->B = g->out_B; out->Q = g->Q; out->C = g->C; out->h4 = g->out_h; out->w4 = g->out_w;
    out->logit_scale = g->logit_scale;
    out->class_logits = g->class_logits;
    return ODISE_OK;
}

void maskgen_invalidate_outputs(ModelStore* ms) {
    MaskGenModel* g = ms->maskgen;
    if (!g) return;
    g->pred_masks = nullptr; g->mask_embed = nullptr; g->mask_pooled = nullptr; g->class_logits = nullptr;
    for (Act& f : g->feats) f.p = nullptr;
}

}  // namespace odise

using namespace odise;

extern "C" int odise_hip_backbone_build(odise_hip_ctx* ctx) {
    ODISE_REQUIRE(ctx, "backbone_build: null context");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    return maskgen_build_backbone(ctx);
}
extern "C" int odise_hip_head_build(odise_hip_ctx* ctx) {
    ODISE_REQUIRE(ctx, "head_build: null context");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    return maskgen_build_head(ctx);
}

extern "C" int odise_hip_backbone_forward(odise_hip_ctx* ctx, const float* image, int B, int H, int W, float** out4) {
    ODISE_REQUIRE(ctx, "backbone_forward: null context");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    return backbone_forward(ctx, image, B, H, W, out4);
}

// test / attribution hook (include/odise_hip_tools.h): the s2..s5 maps of the last backbone pass that are still resident in the arena
// (odise_hip_backbone_forward or odise_hip_infer; gone after the next call that resets it), converted to fp32 NCHW [B,C,h,w] device arrays
extern "C" int odise_hip_backbone_maps(odise_hip_ctx* ctx, float** out4, int* shape_bchw4x4) {
    ODISE_REQUIRE(ctx && out4, "backbone_maps: null argument");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));
    ModelStore* ms = store_of(ctx);
    MaskGenModel* g = ms->maskgen;
    ODISE_REQUIRE(g && g->feats[0].p != nullptr, "backbone_maps: no backbone features resident (run the backbone or odise_hip_infer first)");
    for (int i = 0; i < 4; ++i) {
        const Act& a = g->feats[i];
        if (shape_bchw4x4) { shape_bchw4x4[4 * i] = a.n; shape_bchw4x4[4 * i + 1] = a.c; shape_bchw4x4[4 * i + 2] = a.h; shape_bchw4x4[4 * i + 3] = a.w; }
        if (out4[i]) ODISE_TRY(odise_hip_nhwc_f16_to_nchw_f32(ctx, a.p, out4[i], a.n, a.c, a.h, a.w));
    }
    return ODISE_OK;
}

// feats4: s2,s3,s4,s5 fp32 NCHW [B,Cin,H/4..H/32,W/4..W/32] on the device, or NULL to use the maps of the last backbone_forward.
// outputs (device, any may be NULL): pred_masks [B,Q,H/4,W/4] f32, mask_embed [B,Q,C] f32, mask_pooled [B,Q,C] f32; logit_scale (host)
extern "C" int odise_hip_head_forward(odise_hip_ctx* ctx, const float* const* feats4, int B, int Cin, int H4, int W4, float* pred_masks,
                                      float* mask_embed, float* mask_pooled, float* logit_scale) {
    ODISE_REQUIRE(ctx, "head_forward: null context");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    ModelStore* ms = store_of(ctx);
    MaskGenModel* g = ms->maskgen;
    if (!g || !g->head_built) {
        set_error("head_forward: call odise_hip_head_build first");
        return ODISE_ERR_STATE;
    }
    Act feats[4];
    Exec ex{ctx, ms};
    if (feats4) {
        ODISE_REQUIRE(B >= 1 && Cin % 8 == 0 && H4 % 8 == 0 && W4 % 8 == 0, "head_forward: bad feature shapes");
        size_t need = (size_t)B * H4 * W4 * 256 * 2 * 40 + ((size_t)512 << 20);
        ODISE_TRY(ensure_arena(ctx, ms, need));
        ms->arena.reset();
        maskgen_invalidate_outputs(ms);   // backbone maps / head outputs of earlier calls lived in the arena just recycled; head_forward refills its own
        ms->macs = 0.0;
        for (int i = 0; i < 4; ++i) {
            const int h = H4 >> i, w = W4 >> i;
            ODISE_TRY(ex.alloc(feats[i], B, h, w, Cin));
            ODISE_TRY(odise_hip_nchw_f32_to_nhwc_f16(ctx, feats4[i], feats[i].p, B, Cin, h, w, Cin));
        }
    } else {
        ODISE_REQUIRE(g->feats[0].p != nullptr, "head_forward: no backbone features available");
        for (int i = 0; i < 4; ++i) feats[i] = g->feats[i];
    }
    ODISE_TRY(head_forward(ctx, feats));
    const int64_t MQ = (int64_t)g->out_B * g->Q;
    if (pred_masks) ODISE_TRY(odise_hip_cast_f16_to_f32(ctx, g->pred_masks, pred_masks, (size_t)MQ * g->out_h * g->out_w));
    if (mask_embed) ODISE_TRY(odise_hip_cast_f16_to_f32(ctx, g->mask_embed, mask_embed, (size_t)MQ * g->C));
    if (mask_pooled) ODISE_TRY(odise_hip_cast_f16_to_f32(ctx, g->mask_pooled, mask_pooled, (size_t)MQ * g->C));
    if (logit_scale) *logit_scale = g->logit_scale;
    return ODISE_OK;
}

// MSDeformAttnPixelDecoder.forward_features stand-alone (msdeformattn.py:314-358): feats4 = s2..s5 fp32 NCHW device pointers.
// Outputs (device fp32 NCHW, any may be NULL): mask_features [B,C,H4,W4]; multi_scale[0..2] = [B,C,H4/8,W4/8], [B,C,H4/4,W4/4],
// [B,C,H4/2,W4/2] (low -> high resolution; the reference's `transformer_encoder_features` is multi_scale[0]).
extern "C" int odise_hip_pixel_decoder_forward(odise_hip_ctx* ctx, const float* const* feats4, int B, int Cin, int H4, int W4, float* mask_features,
                                               float* const* multi_scale3) {
    ODISE_REQUIRE(ctx && feats4, "pixel_decoder_forward: null argument");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    ModelStore* ms = store_of(ctx);
    MaskGenModel* g = ms->maskgen;
    if (!g || !g->head_built) {
        set_error("pixel_decoder_forward: call odise_hip_head_build first");
        return ODISE_ERR_STATE;
    }
    ODISE_REQUIRE(B >= 1 && Cin % 8 == 0 && H4 % 8 == 0 && W4 % 8 == 0, "pixel_decoder_forward: bad feature shapes");
    Exec ex{ctx, ms};
    ODISE_TRY(ensure_arena(ctx, ms, (size_t)B * H4 * W4 * 256 * 2 * 40 + ((size_t)512 << 20)));
    ms->arena.reset();
    maskgen_invalidate_outputs(ms);   // nothing of an earlier backbone / head call survives: a later classify / postprocess fails with ODISE_ERR_STATE
    ms->macs = 0.0;
    Act feats[4];
    for (int i = 0; i < 4; ++i) {
        const int h = H4 >> i, w = W4 >> i;
        ODISE_TRY(ex.alloc(feats[i], B, h, w, Cin));
        ODISE_TRY(odise_hip_nchw_f32_to_nhwc_f16(ctx, feats4[i], feats[i].p, B, Cin, h, w, Cin));
    }
    PixDec pd;
    ODISE_TRY(pixel_decoder_forward(ctx, feats, pd));
    if (mask_features) ODISE_TRY(odise_hip_cast_f16_to_f32(ctx, pd.mfT, mask_features, (size_t)B * g->C * H4 * W4));   // channel-major = NCHW
    for (int l = 0; l < 3 && multi_scale3; ++l)
        if (multi_scale3[l]) ODISE_TRY(odise_hip_nhwc_f16_to_nchw_f32(ctx, pd.ms_feat[l].p, multi_scale3[l], B, g->C, pd.ms_feat[l].h, pd.ms_feat[l].w));
    return ODISE_OK;
}

// ODISEMultiScaleMaskedTransformerDecoder.forward stand-alone (odise.py:642-727): multi_scale3 = three fp32 NCHW maps [B,C,h_l,w_l]
// (low -> high resolution), mask_features [B,C,H4,W4] fp32 NCHW; outputs as odise_hip_head_forward.
extern "C" int odise_hip_predictor_forward(odise_hip_ctx* ctx, const float* const* multi_scale3, const int* hw3, const float* mask_features, int B, int H4,
                                           int W4, float* pred_masks, float* mask_embed, float* mask_pooled, float* logit_scale) {
    ODISE_REQUIRE(ctx && multi_scale3 && hw3 && mask_features, "predictor_forward: null argument");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    ModelStore* ms = store_of(ctx);
    MaskGenModel* g = ms->maskgen;
    if (!g || !g->head_built) {
        set_error("predictor_forward: call odise_hip_head_build first");
        return ODISE_ERR_STATE;
    }
    ODISE_REQUIRE(B >= 1 && H4 >= 1 && W4 >= 1 && ((int64_t)H4 * W4) % 8 == 0, "predictor_forward: bad mask feature shape");
    Exec ex{ctx, ms};
    ODISE_TRY(ensure_arena(ctx, ms, (size_t)B * H4 * W4 * 256 * 2 * 40 + ((size_t)512 << 20)));
    ms->arena.reset();
    maskgen_invalidate_outputs(ms);   // predictor_forward refills the head outputs; the backbone maps are gone
    ms->macs = 0.0;
    const int C = g->C;
    PixDec pd;
    for (int l = 0; l < 3; ++l) {
        ODISE_REQUIRE(hw3[2 * l] >= 1 && hw3[2 * l + 1] >= 1, "predictor_forward: bad level %d", l);
        ODISE_TRY(ex.alloc(pd.ms_feat[l], B, hw3[2 * l], hw3[2 * l + 1], C));
        ODISE_TRY(odise_hip_nchw_f32_to_nhwc_f16(ctx, multi_scale3[l], pd.ms_feat[l].p, B, C, hw3[2 * l], hw3[2 * l + 1], C));
    }
    ODISE_TRY(ex.alloc(pd.mf, B, H4, W4, C));
    ODISE_TRY(odise_hip_nchw_f32_to_nhwc_f16(ctx, mask_features, pd.mf.p, B, C, H4, W4, C));
    pd.mfT = (f16*)ex.alloc_bytes((size_t)B * C * H4 * W4 * 2);
    if (!pd.mfT) return ODISE_ERR_NOMEM;
    ODISE_TRY(odise_hip_cast_f32_to_f16(ctx, mask_features, pd.mfT, (size_t)B * C * H4 * W4));
    pd.h4 = H4; pd.w4 = W4;
    ODISE_TRY(predictor_forward(ctx, pd));
    const int64_t MQ = (int64_t)g->out_B * g->Q;
    if (pred_masks) ODISE_TRY(odise_hip_cast_f16_to_f32(ctx, g->pred_masks, pred_masks, (size_t)MQ * g->out_h * g->out_w));
    if (mask_embed) ODISE_TRY(odise_hip_cast_f16_to_f32(ctx, g->mask_embed, mask_embed, (size_t)MQ * g->C));
    if (mask_pooled) ODISE_TRY(odise_hip_cast_f16_to_f32(ctx, g->mask_pooled, mask_pooled, (size_t)MQ * g->C));
    if (logit_scale) *logit_scale = g->logit_scale;
    return ODISE_OK;
}

extern "C" int odise_hip_maskgen_info(odise_hip_ctx* ctx, int* num_queries, int* hidden_dim, double* last_macs) {
    ODISE_REQUIRE(ctx, "maskgen_info: null context");
    MaskGenModel* g = store_of(ctx)->maskgen;
    if (num_queries) *num_queries = g ? g->Q : 0;
    if (hidden_dim) *hidden_dim = g ? g->C : 0;
    if (last_macs) *last_macs = g ? g->last_macs : 0.0;
    return ODISE_OK;
}

extern "C" int odise_hip_msda_unfused(int on) { odise::g_msda_unfused = on; return 0; }
