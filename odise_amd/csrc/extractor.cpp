// extractor.cpp — LdmImplicitCaptionerExtractor.forward on the device (odise/modeling/meta_arch/ldm.py:697-718 -> 543-621):
//
//   image [B,3,H,W] in [0,1]
//     |- CLIP ViT-L/14@336 image embed (clip.py:177-231)  -> prefix [B,768]
//     |     cond_inputs = uncond + tanh(alpha_cond) * (Linear(prefix)[:,None] + pos)       (ldm.py:705-709)
//     |     cond_emb    = tanh(alpha_t) * (Linear(prefix) + pos_t)                          (ldm.py:711-714)
//     |- (image-0.5)/0.5 -> AutoencoderKL.encoder (taps = inputs of down blocks 5, 7; ldm.py:424-457) -> quant_conv mean
//     |     latent = 0.18215 * mean;  x_t = q_sample(latent, t=0, shared noise seed 42)      (ldm.py:459-467, 577-598)
//     |- UNet single step (unet.cpp; taps = concat inputs of output blocks 2,5,8,11)
//     `- post_quant_conv(latent/0.18215) -> AutoencoderKL.decoder up to the INPUT of up-block 5 (taps 2, 5; ldm.py:493-533).
//        Everything after the last tap only feeds the discarded reconstruction (ldm.py:606) and is not executed.
//
// Architectures restated from SURVEY.md Appendix A.2 / A.3 (ldm AutoencoderKL, OpenAI CLIP ViT); weights addressed by
// checkpoint keys: first_stage_model.*, model.diffusion_model.*, clip.visual.*, backbone.feature_extractor.*.
// All crops of a call run as ONE batch through every stage (the reference loops crops sequentially,
// feature_extractor.py:216-227).
#include <math.h>
#include <string.h>

#include <functional>
#include "engine.h"

namespace odise {

struct VaeRes {
    NormW n1, n2;
    ConvW c1, c2, nin;
    bool has_nin = false;
};
struct VaeAttn {
    NormW norm;
    LinW qk, v, proj;  // q|k stacked [2C,C] with bias, v [C,C] (+bias applied along M of the swapped GEMM), proj_out
    float* v_bias = nullptr;
    int c = 0;
};
struct ClipBlock {
    NormW ln1, ln2;
    LinW qk, v, out, fc, proj;
    float* v_bias = nullptr;
    // ln_1 / ln_2 folded into the GEMMs that read them (engine.h LnEpi): W' = fp16(W diag(gamma)), b' = b + W beta, cs = row sums of W'
    LinW qk_f, v_f, fc_f;
    float *qk_cs = nullptr, *v_cs = nullptr, *fc_cs = nullptr, *v_f_bias = nullptr;
};

struct ExtractorModel {
    std::vector<void*> owned;   // device weights of this stage (AllocScope); the UNet owns its own
    bool built = false;
    // VAE encoder
    ConvW enc_conv_in, enc_conv_out;
    VaeRes enc_blocks[4][2];
    ConvW enc_down[3];
    VaeRes enc_mid1, enc_mid2;
    VaeAttn enc_attn;
    NormW enc_norm_out;
    // VAE decoder (live part)
    ConvW dec_conv_in;
    VaeRes dec_mid1, dec_mid2;
    VaeAttn dec_attn;
    VaeRes dec_l3[3], dec_l2[2];
    ConvW dec_up3;
    LatentW lat;
    float* noise = nullptr;  // [4, 64*64]
    int noise_hw = 0;
    // CLIP
    ConvW clip_conv1;
    float *clip_cls = nullptr, *clip_pos = nullptr;
    NormW clip_ln_pre, clip_ln_post;
    std::vector<ClipBlock> clip_blocks;
    LinW clip_proj;
    int clip_width = 0, clip_heads = 16, clip_tokens = 0, clip_image = 336, clip_patch = 14, clip_out = 768;
    // captioner
    LinW cap_proj;   // clip_project.linear
    LinW cap_time;   // time_embed_project folded with tanh(alpha) and positional embedding
    float *cap_A1 = nullptr, *cap_A2 = nullptr;
    int ctx_dim = 768, ted = 1280;
    // outputs of the last forward
    Act taps[8];
    double last_macs = 0.0;
};

void extractor_destroy(ModelStore* ms) {
    if (ms->extractor) free_allocs(ms->extractor->owned);
    delete ms->extractor;
    ms->extractor = nullptr;
}

// ---------------------------------------------------------------------------------------------------------------
static int build_vae_res(Packer& pk, const std::string& key, VaeRes& r) {
    ODISE_TRY(pk.norm(key + ".norm1", r.n1));
    ODISE_TRY(pk.conv(key + ".conv1", r.c1));
    ODISE_TRY(pk.norm(key + ".norm2", r.n2));
    ODISE_TRY(pk.conv(key + ".conv2", r.c2));
    r.has_nin = pk.find(key + ".nin_shortcut.weight") != nullptr;
    if (r.has_nin) ODISE_TRY(pk.conv(key + ".nin_shortcut", r.nin));
    return ODISE_OK;
}

static int build_vae_attn(Packer& pk, const std::string& key, VaeAttn& a) {
    ODISE_TRY(pk.norm(key + ".norm", a.norm));
    a.c = a.norm.c;
    const HostTensor *wq = pk.find(key + ".q.weight"), *wk = pk.find(key + ".k.weight"), *bq = pk.find(key + ".q.bias"),
                     *bk = pk.find(key + ".k.bias");
    const size_t cc = (size_t)a.c * a.c;
    if (!wq || !wk || !bq || !bk || (size_t)wq->numel() != cc || (size_t)wk->numel() != cc) {
        set_error("vae: bad or missing '%s.q/k'", key.c_str());
        return ODISE_ERR_STATE;
    }
    std::vector<f16> qk(2 * cc);
    std::vector<float> b(2 * (size_t)a.c);
    for (size_t i = 0; i < cc; ++i) { qk[i] = (f16)wq->data[i]; qk[cc + i] = (f16)wk->data[i]; }
    for (int i = 0; i < a.c; ++i) { b[i] = bq->data[i]; b[a.c + i] = bk->data[i]; }
    a.qk.in = a.c; a.qk.out = 2 * a.c;
    ODISE_TRY(pk.upload(qk.data(), qk.size() * sizeof(f16), (void**)&a.qk.w));
    ODISE_TRY(pk.upload(b.data(), b.size() * sizeof(float), (void**)&a.qk.b));
    ODISE_TRY(pk.linear(key + ".v", a.v, false));
    ODISE_TRY(pk.vec_f32(key + ".v.bias", &a.v_bias, a.c));
    ODISE_TRY(pk.linear(key + ".proj_out", a.proj));
    return ODISE_OK;
}

// rows [r0, r0 + O) of W [*, I] with a LayerNorm (gamma, beta) over I folded in
static int fold_layer_norm(Packer& pk, const float* W, const float* bias, int O, int I, const float* gamma, const float* beta, LinW& lw, float** bias_dev,
                           float** cs_dev) {
    std::vector<f16> wf((size_t)O * I);
    std::vector<float> bf(O), cs(O);
    for (int o = 0; o < O; ++o) {
        double bb = bias[o], c = 0.0;
        for (int i = 0; i < I; ++i) {
            const float w = W[(size_t)o * I + i];
            const f16 h = (f16)(w * gamma[i]);
            wf[(size_t)o * I + i] = h;
            c += (double)(float)h;
            bb += (double)w * beta[i];
        }
        bf[o] = (float)bb;
        cs[o] = (float)c;
    }
    lw.in = I; lw.out = O; lw.b = nullptr;
    ODISE_TRY(pk.upload(wf.data(), wf.size() * sizeof(f16), (void**)&lw.w));
    ODISE_TRY(pk.upload(bf.data(), bf.size() * sizeof(float), (void**)bias_dev));
    ODISE_TRY(pk.upload(cs.data(), cs.size() * sizeof(float), (void**)cs_dev));
    return ODISE_OK;
}

static int build_clip_block(Packer& pk, const std::string& key, ClipBlock& b, int W) {
    ODISE_TRY(pk.norm(key + ".ln_1", b.ln1));
    ODISE_TRY(pk.norm(key + ".ln_2", b.ln2));
    const HostTensor* w = pk.find(key + ".attn.in_proj_weight");
    const HostTensor* bias = pk.find(key + ".attn.in_proj_bias");
    if (!w || !bias || w->numel() != (int64_t)3 * W * W || bias->numel() != 3 * W) {
        set_error("clip: bad or missing '%s.attn.in_proj_*'", key.c_str());
        return ODISE_ERR_STATE;
    }
    const size_t ww = (size_t)W * W;
    std::vector<f16> qk(2 * ww), v(ww);
    for (size_t i = 0; i < 2 * ww; ++i) qk[i] = (f16)w->data[i];
    for (size_t i = 0; i < ww; ++i) v[i] = (f16)w->data[2 * ww + i];
    b.qk.in = W; b.qk.out = 2 * W;
    ODISE_TRY(pk.upload(qk.data(), qk.size() * sizeof(f16), (void**)&b.qk.w));
    ODISE_TRY(pk.upload(bias->data.data(), (size_t)2 * W * sizeof(float), (void**)&b.qk.b));
    b.v.in = W; b.v.out = W; b.v.b = nullptr;
    ODISE_TRY(pk.upload(v.data(), v.size() * sizeof(f16), (void**)&b.v.w));
    ODISE_TRY(pk.upload(bias->data.data() + 2 * W, (size_t)W * sizeof(float), (void**)&b.v_bias));
    ODISE_TRY(pk.linear(key + ".attn.out_proj", b.out));
    ODISE_TRY(pk.linear(key + ".mlp.c_fc", b.fc));
    ODISE_TRY(pk.linear(key + ".mlp.c_proj", b.proj));
    const HostTensor *g1 = pk.find(key + ".ln_1.weight"), *b1 = pk.find(key + ".ln_1.bias");
    const HostTensor *g2 = pk.find(key + ".ln_2.weight"), *b2 = pk.find(key + ".ln_2.bias");
    const HostTensor *fw = pk.find(key + ".mlp.c_fc.weight"), *fb = pk.find(key + ".mlp.c_fc.bias");
    if (!g1 || !b1 || !g2 || !b2 || !fw || !fb || g1->numel() != W || b1->numel() != W || g2->numel() != W || b2->numel() != W ||
        fw->numel() != (int64_t)4 * W * W || fb->numel() != 4 * W) {
        set_error("clip: bad or missing '%s.ln_* / mlp.c_fc'", key.c_str());
        return ODISE_ERR_STATE;
    }
    ODISE_TRY(fold_layer_norm(pk, w->data.data(), bias->data.data(), 2 * W, W, g1->data.data(), b1->data.data(), b.qk_f, &b.qk_f.b, &b.qk_cs));
    ODISE_TRY(fold_layer_norm(pk, w->data.data() + 2 * ww, bias->data.data() + 2 * W, W, W, g1->data.data(), b1->data.data(), b.v_f, &b.v_f_bias, &b.v_cs));
    ODISE_TRY(fold_layer_norm(pk, fw->data.data(), fb->data.data(), 4 * W, W, g2->data.data(), b2->data.data(), b.fc_f, &b.fc_f.b, &b.fc_cs));
    return ODISE_OK;
}

static int extractor_build(odise_hip_ctx* ctx) {
    ModelStore* ms = store_of(ctx);
    extractor_destroy(ms);   // (device-synchronising: a prefetch in flight has finished)
    ms->pf.has_pending = ms->pf.has_ready = ms->pf.use_now = false;   // an encoder result of the previous weights must not be consumed
    ExtractorModel* e = new ExtractorModel();
    ms->extractor = e;
    // ---- UNet --------------------------------------------------------------------------------------------------
    ODISE_TRY(unet_build(ctx, "model.diffusion_model."));
    AllocScope scope(ms, e->owned);
    // ---- VAE ---------------------------------------------------------------------------------------------------
    Packer pk{ctx, ms, "first_stage_model.", ""};
    ODISE_TRY(pk.conv("encoder.conv_in", e->enc_conv_in));
    for (int l = 0; l < 4; ++l) {
        for (int b = 0; b < 2; ++b)
            ODISE_TRY(build_vae_res(pk, "encoder.down." + std::to_string(l) + ".block." + std::to_string(b), e->enc_blocks[l][b]));
        if (l < 3) ODISE_TRY(pk.conv("encoder.down." + std::to_string(l) + ".downsample.conv", e->enc_down[l]));
    }
    ODISE_TRY(build_vae_res(pk, "encoder.mid.block_1", e->enc_mid1));
    ODISE_TRY(build_vae_attn(pk, "encoder.mid.attn_1", e->enc_attn));
    ODISE_TRY(build_vae_res(pk, "encoder.mid.block_2", e->enc_mid2));
    ODISE_TRY(pk.norm("encoder.norm_out", e->enc_norm_out));
    ODISE_TRY(pk.conv("encoder.conv_out", e->enc_conv_out));
    ODISE_TRY(pk.conv("decoder.conv_in", e->dec_conv_in));
    ODISE_TRY(build_vae_res(pk, "decoder.mid.block_1", e->dec_mid1));
    ODISE_TRY(build_vae_attn(pk, "decoder.mid.attn_1", e->dec_attn));
    ODISE_TRY(build_vae_res(pk, "decoder.mid.block_2", e->dec_mid2));
    for (int b = 0; b < 3; ++b) ODISE_TRY(build_vae_res(pk, "decoder.up.3.block." + std::to_string(b), e->dec_l3[b]));
    ODISE_TRY(pk.conv("decoder.up.3.upsample.conv", e->dec_up3));
    for (int b = 0; b < 2; ++b) ODISE_TRY(build_vae_res(pk, "decoder.up.2.block." + std::to_string(b), e->dec_l2[b]));
    {
        const HostTensor *wq = pk.find("quant_conv.weight"), *bq = pk.find("quant_conv.bias"), *wp = pk.find("post_quant_conv.weight"),
                         *bp = pk.find("post_quant_conv.bias");
        if (!wq || !bq || !wp || !bp || wq->numel() != 64 || bq->numel() != 8 || wp->numel() != 16 || bp->numel() != 4) {
            set_error("vae: bad or missing quant_conv / post_quant_conv (expected 8x8 and 4x4 1x1 convs)");
            return ODISE_ERR_STATE;
        }
        for (int c = 0; c < 4; ++c) {
            for (int k = 0; k < 8; ++k) e->lat.wq[c][k] = wq->data[c * 8 + k];
            e->lat.bq[c] = bq->data[c];
            for (int k = 0; k < 4; ++k) e->lat.wp[c][k] = wp->data[c * 4 + k];
            e->lat.bp[c] = bp->data[c];
        }
        e->lat.scale = 0.18215f;
        // "ldm_linear" schedule (gaussian_diffusion.py:125-135), t = 0: alpha_bar_0 = 1 - beta_0
        const double beta0 = pow(sqrt(0.00085), 2.0);
        e->lat.qa = (float)sqrt(1.0 - beta0);
        e->lat.qb = (float)sqrt(1.0 - (1.0 - beta0));
    }
    // ---- CLIP image tower ------------------------------------------------------------------------------------------
    Packer pc{ctx, ms, "clip.visual.", ""};
    ODISE_TRY(pc.conv("conv1", e->clip_conv1, false));
    e->clip_width = e->clip_conv1.cout;
    e->clip_patch = e->clip_conv1.k;
    const HostTensor* pos = pc.find("positional_embedding");
    if (!pos || pos->shape.size() != 2 || pos->shape[1] != e->clip_width) {
        set_error("clip: bad or missing visual.positional_embedding");
        return ODISE_ERR_STATE;
    }
    e->clip_tokens = (int)pos->shape[0];
    {
        const int grid = (int)lround(sqrt((double)(e->clip_tokens - 1)));
        e->clip_image = grid * e->clip_patch;
    }
    ODISE_TRY(pc.vec_f32("positional_embedding", &e->clip_pos, (int64_t)e->clip_tokens * e->clip_width));
    ODISE_TRY(pc.vec_f32("class_embedding", &e->clip_cls, e->clip_width));
    ODISE_TRY(pc.norm("ln_pre", e->clip_ln_pre));
    ODISE_TRY(pc.norm("ln_post", e->clip_ln_post));
    for (int i = 0;; ++i) {
        const std::string key = "transformer.resblocks." + std::to_string(i);
        if (!pc.find(key + ".ln_1.weight")) break;
        ClipBlock b;
        ODISE_TRY(build_clip_block(pc, key, b, e->clip_width));
        e->clip_blocks.push_back(b);
    }
    if (e->clip_blocks.empty()) {
        set_error("clip: no transformer.resblocks found");
        return ODISE_ERR_STATE;
    }
    e->clip_heads = e->clip_width / 64;
    {
        const HostTensor* pr = pc.find("proj");
        if (!pr || pr->shape.size() != 2 || pr->shape[0] != e->clip_width) {
            set_error("clip: bad or missing visual.proj");
            return ODISE_ERR_STATE;
        }
        e->clip_out = (int)pr->shape[1];
        std::vector<f16> wt((size_t)e->clip_out * e->clip_width);
        for (int o = 0; o < e->clip_out; ++o)
            for (int i = 0; i < e->clip_width; ++i) wt[(size_t)o * e->clip_width + i] = (f16)pr->data[(size_t)i * e->clip_out + o];
        e->clip_proj.in = e->clip_width; e->clip_proj.out = e->clip_out; e->clip_proj.b = nullptr;
        ODISE_TRY(pc.upload(wt.data(), wt.size() * sizeof(f16), (void**)&e->clip_proj.w));
    }
    // ---- implicit captioner (trainable part of backbone.feature_extractor) -------------------------------------------
    Packer pf{ctx, ms, "backbone.feature_extractor.", ""};
    ODISE_TRY(pf.linear("clip_project.linear", e->cap_proj));
    e->ctx_dim = e->cap_proj.out;
    {
        const HostTensor *unc = pf.find("ldm_extractor.ldm.uncond_inputs"), *alpha = pf.find("alpha_cond"),
                         *posc = pf.find("clip_project.positional_embedding");
        const int64_t n = (int64_t)77 * e->ctx_dim;
        if (!unc || !alpha || !posc || unc->numel() != n || alpha->numel() != n || posc->numel() != n) {
            set_error("captioner: bad or missing uncond_inputs / alpha_cond / clip_project.positional_embedding");
            return ODISE_ERR_STATE;
        }
        std::vector<float> A1(n), A2(n);
        for (int64_t i = 0; i < n; ++i) {
            A2[i] = tanhf(alpha->data[i]);
            A1[i] = unc->data[i] + A2[i] * posc->data[i];
        }
        ODISE_TRY(pf.upload(A1.data(), n * sizeof(float), (void**)&e->cap_A1));
        ODISE_TRY(pf.upload(A2.data(), n * sizeof(float), (void**)&e->cap_A2));
    }
    {
        const HostTensor *w = pf.find("time_embed_project.linear.weight"), *b = pf.find("time_embed_project.linear.bias"),
                         *post = pf.find("time_embed_project.positional_embedding"), *al = pf.find("alpha_cond_time_embed");
        if (!w || !b || !post || !al || w->shape.size() != 2 || post->numel() != w->shape[0] || al->numel() != w->shape[0]) {
            set_error("captioner: bad or missing time_embed_project / alpha_cond_time_embed (num_timesteps must be 1)");
            return ODISE_ERR_STATE;
        }
        e->ted = (int)w->shape[0];
        const int in = (int)w->shape[1];
        std::vector<f16> wf((size_t)e->ted * in);
        std::vector<float> bf(e->ted);
        for (int o = 0; o < e->ted; ++o) {
            const float t = tanhf(al->data[o]);
            for (int i = 0; i < in; ++i) wf[(size_t)o * in + i] = (f16)(t * w->data[(size_t)o * in + i]);
            bf[o] = t * (b->data[o] + post->data[o]);
        }
        e->cap_time.in = in; e->cap_time.out = e->ted;
        ODISE_TRY(pf.upload(wf.data(), wf.size() * sizeof(f16), (void**)&e->cap_time.w));
        ODISE_TRY(pf.upload(bf.data(), bf.size() * sizeof(float), (void**)&e->cap_time.b));
    }
    {
        const HostTensor* nz = pf.find("ldm_extractor.shared_noise");
        if (!nz || nz->shape.size() != 4 || nz->shape[1] != 4) {
            set_error("extractor: bad or missing ldm_extractor.shared_noise [1,4,h,w]");
            return ODISE_ERR_STATE;
        }
        e->noise_hw = (int)(nz->shape[2] * nz->shape[3]);
        ODISE_TRY(pf.upload(nz->data.data(), nz->data.size() * sizeof(float), (void**)&e->noise));
    }
    e->built = true;
    return ODISE_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// crops [n0, n0 + n) of a batched activation (the GroupNorm statistics of the whole tensor do not carry over)
static Act crop_slice(const Act& a, int n0, int n) {
    Act s = a;
    s.p = a.p + (size_t)n0 * a.h * a.w * a.c;
    s.n = n;
    s.gn_part = nullptr;
    s.gn_blocks = 0;
    return s;
}

// crops per launch of a VAE level whose widest activation has `per_crop_bytes` (ODISE_OPT_VAE_CHUNK_BYTES): the tensor a kernel writes should
// still be in the 256 MiB Infinity Cache when GroupNorm and the next convolution read it.  All 16 crops of a step make 0.27-1.07 GB tensors
// at the 128- / 256- / 512-channel levels of 512^2 .. 128^2 - every GroupNorm pass and every convolution then streams from HBM.  The levels'
// grids stay large in chunks (128 ch @ 512^2: 1024 tiles per crop; 512 ch @ 128^2: 128 per crop, four crops per launch).
static int vae_chunk(const odise_hip_ctx* ctx, int B, size_t per_crop_bytes) {
    if (ctx->vae_chunk_bytes <= 0) return B;
    const int64_t c = ctx->vae_chunk_bytes / (int64_t)per_crop_bytes;
    return (int)std::max<int64_t>(1, std::min<int64_t>(B, c));
}

// out.p == nullptr: allocated here; otherwise the caller's (slice of a batched) tensor is written
static int run_vae_res(Exec& ex, const VaeRes& w, const Act& x, Act& out) {
    // both convs feed a GroupNorm (conv1 -> norm2 here, conv2 (+ skip) -> norm1 of the next block / norm_out): their epilogues
    // reduce the statistics, saving one HBM pass over tensors of up to 1 GB
    if (!out.p) ODISE_TRY(ex.alloc(out, x.n, x.h, x.w, w.c1.cout));
    ODISE_REQUIRE(out.n == x.n && out.h == x.h && out.w == x.w && out.c == w.c1.cout, "vae block: output tensor does not match");
    ODISE_TRY(ex.alloc_gn_stats(out));
    const size_t mk = ex.ms->arena.mark();
    Act t1, h, t2, sk;
    ODISE_TRY(ex.group_norm(x, w.n1, t1, 1e-6f, ODISE_ACT_SILU));
    ODISE_TRY(ex.alloc(h, x.n, x.h, x.w, w.c1.cout));
    ODISE_TRY(ex.alloc_gn_stats(h));
    ODISE_TRY(ex.conv(t1, w.c1, h, 1, 1));
    ODISE_TRY(ex.group_norm(h, w.n2, t2, 1e-6f, ODISE_ACT_SILU));
    const Act* resid = &x;
    if (w.has_nin) {
        ODISE_TRY(ex.conv(x, w.nin, sk, 1, 0));
        resid = &sk;
    }
    ODISE_TRY(ex.conv(t2, w.c2, out, 1, 1, false, resid));
    ex.ms->arena.release(mk);
    return ODISE_OK;
}

// single-head attention of width C over HW tokens: S = (q k^T) C^-0.5 materialised per image (HW x HW fp16), row softmax, P v
static int run_vae_attn(Exec& ex, const VaeAttn& w, const Act& x, Act& out) {
    const int C = w.c;
    const int64_t HW = (int64_t)x.h * x.w, M = x.pixels();
    ODISE_TRY(ex.alloc(out, x.n, x.h, x.w, C));
    const size_t mk = ex.ms->arena.mark();
    Act t;
    ODISE_TRY(ex.group_norm(x, w.norm, t, 1e-6f, ODISE_ACT_NONE));
    const int64_t ldv = round_up(HW, 8);
    ODISE_REQUIRE(HW % 8 == 0, "vae attention: H*W=%lld must be a multiple of 8", (long long)HW);
    f16* qk = (f16*)ex.alloc_bytes((size_t)M * 2 * C * 2);
    f16* vt = (f16*)ex.alloc_bytes((size_t)x.n * C * ldv * 2);
    f16* S = (f16*)ex.alloc_bytes((size_t)x.n * HW * ldv * 2);
    f16* o = (f16*)ex.alloc_bytes((size_t)M * C * 2);
    if (!qk || !vt || !S || !o) return ODISE_ERR_NOMEM;
    ODISE_TRY(ex.linear(t.p, M, w.qk, qk));
    odise_gemm_desc d;
    memset(&d, 0, sizeof(d));  // V^T[b] = Wv n[b]^T + bv (bias along rows)
    d.M = C; d.N = (int)HW; d.K = C;
    d.A = w.v.w; d.lda = C; d.W = t.p; d.ldw = C; d.strideW = HW * C;
    d.C = vt; d.ldc = ldv; d.strideC = (int64_t)C * ldv; d.c_dtype = ODISE_F16;
    d.bias_m = w.v_bias; d.alpha = 1.f; d.batch = x.n;
    ODISE_TRY(ex.gemm(d));
    memset(&d, 0, sizeof(d));  // S[b] = q[b] k[b]^T * C^-0.5
    d.M = (int)HW; d.N = (int)HW; d.K = C;
    d.A = qk; d.lda = 2 * C; d.strideA = HW * 2 * C;
    d.W = qk + C; d.ldw = 2 * C; d.strideW = HW * 2 * C;
    d.C = S; d.ldc = ldv; d.strideC = HW * ldv; d.c_dtype = ODISE_F16;
    d.alpha = 1.0f / sqrtf((float)C); d.batch = x.n;
    ODISE_TRY(ex.gemm(d));
    ODISE_TRY(launch_softmax_rows(ex.ctx, S, S, (int64_t)x.n * HW, (int)HW, ldv, 1.0f));
    memset(&d, 0, sizeof(d));  // o[b] = P[b] v[b]  (W operand = V^T)
    d.M = (int)HW; d.N = C; d.K = (int)HW;
    d.A = S; d.lda = ldv; d.strideA = HW * ldv;
    d.W = vt; d.ldw = ldv; d.strideW = (int64_t)C * ldv;
    d.C = o; d.ldc = C; d.strideC = HW * C; d.c_dtype = ODISE_F16;
    d.alpha = 1.f; d.batch = x.n;
    ODISE_TRY(ex.gemm(d));
    ODISE_TRY(ex.linear(o, M, w.proj, out.p, ODISE_ACT_NONE, x.p));
    ex.ms->arena.release(mk);
    return ODISE_OK;
}

// The CLIP ViT tower on a preprocessed 336x336 NHWC image.  extra = 0: plain image tower, out = class-token embedding
// [B, clip_out] (ClipAdapter._encode_image, clip.py:177-206).  extra = Q > 0: MaskCLIP (clip.py:252-323): Q mask tokens
// (copies of the class token) are appended AFTER the 577 image tokens; they never act as keys, so attention runs with
// Lq = 577 + Q queries over Lk = 577 keys and `mask` [B, 577+Q, ldm] (u8, 1 = not visible) carries the per-(mask, patch)
// visibility; with the same LayerNorm form (ODISE_OPT_CLIP_LN_FOLD) the image-token stream equals the plain tower's.  out = [B, Q, clip_out] (ln_post + proj of the mask tokens).
// Token rows: every image owns TP = round_up(577 + Q, 8) rows of the activation matrices (the tail rows are zero at the input and never read
// as keys, values or results), so that V^T of ALL images is ONE GEMM Wv x n^T -> [width, B*TP] whose column block b*TP.. is image b's
// 16-byte-aligned V^T (the per-image batched form ran at 309 TFLOP/s: 3 column tiles for 577 tokens, 192 tiles on 256 CUs).
// kv != nullptr (extra = 0): the first kv_images images of the batch are MaskCLIP's pictures (engine.h ClipKV): q|k and V^T of EVERY block are
// written in kv's sliding layout, the class-token row after ln_pre goes to kv->cls, and those images leave the tower with the last block's
// projections (no output bit of MaskCLIP depends on the image tokens beyond them).  The other images (the crops of the implicit captioner) run
// the plain tower beside them; with kv_images = img.n there is no output and the pass ends there.
int clip_tower(Exec& ex, const Act& img, int extra, const uint8_t* mask, int64_t ldm, f16* out, ClipKV* kv, int kv_images) {
    ExtractorModel* e = ex.ms->extractor;
    const int B = img.n, S = e->clip_image, Wd = e->clip_width, T = e->clip_tokens, G = S / e->clip_patch;
    const int TA = T + extra;
    const int TP = (int)round_up(TA, 8);
    const size_t mk = ex.ms->arena.mark();
    Act patches;
    ODISE_TRY(ex.conv(img, e->clip_conv1, patches, e->clip_patch, 0, false, nullptr, nullptr, 0, ODISE_ACT_NONE, 0, 0, G, G));
    const int64_t M = (int64_t)B * TP;
    const int64_t ldvt = kv ? kv->ldvt : M;
    const int Bk = kv ? kv_images : 0;   // leading images that only leave keys and values
    ODISE_REQUIRE(!kv || (extra == 0 && Bk >= 1 && Bk <= B && kv->B == Bk && kv->TP == TP), "clip_tower: key / value store laid out for another batch");
    f16* x = (f16*)ex.alloc_bytes((size_t)M * Wd * 2);
    f16* x2 = (f16*)ex.alloc_bytes((size_t)M * Wd * 2);
    f16* n = (f16*)ex.alloc_bytes((size_t)M * Wd * 2);
    f16* qk = kv ? kv->qk : (f16*)ex.alloc_bytes((size_t)M * 2 * Wd * 2);
    f16* vt = kv ? kv->vt : (f16*)ex.alloc_bytes((size_t)Wd * ldvt * 2);
    f16* att = (f16*)ex.alloc_bytes((size_t)M * Wd * 2);
    f16* hid = (f16*)ex.alloc_bytes((size_t)M * 4 * Wd * 2);
    if (!x || !x2 || !n || !qk || !vt || !att || !hid) return ODISE_ERR_NOMEM;
    if (TP > TA) ODISE_CHECK_HIP(hipMemsetAsync(att, 0, (size_t)M * Wd * 2, ex.ctx->stream));  // the attention never writes the tail rows: keep them finite
    ODISE_TRY(launch_clip_assemble(ex.ctx, patches.p, e->clip_cls, e->clip_pos, n, B, T, extra, TP, Wd));
    ODISE_TRY(ex.layer_norm(n, x, M, e->clip_ln_pre, 1e-5f));
    if (kv) ODISE_TRY(launch_broadcast_rows(ex.ctx, x, kv->cls, Wd, 1));   // ln_pre(class_embedding + pos[0]): what every mask token starts from (clip.py:268-270)
    const int heads = e->clip_heads, D = Wd / heads;
    // ln_1 / ln_2 of the blocks never touch HBM as tensors: the GEMM that writes the residual stream (out-proj, c_proj) leaves per-row partial
    // sums next to it, and the GEMMs that would read LN(x) read x with the affine folded into their weights (LnEpi, gemm_epilogue_f16).  8 -> 6
    // launches per block on the chain that sits on the step's critical path.  The first block's ln_1 follows ln_pre and stays a kernel.
    // The statistics live in the math-first epilogue, i.e. on the 256-wide tiles without split-K: the fold is taken where the cost model runs the
    // tower's GEMMs on those tiles anyway (from ~8k tokens: 16 crops).  Below that (MaskCLIP on 4 pictures: 2.7k tokens) the small tiles + two
    // 5 us LayerNorm launches are faster (measured: profiles/r03_lane_scheduling.txt).  The rule is a per-context option (ODISE_OPT_CLIP_LN_FOLD:
    // 0 = by token count, 1 = always, 2 = never), so a caller that needs the same arithmetic whatever the batch pins it.
    const int fold_mode = ex.ctx->clip_ln_fold;
    const bool fold = Wd % 256 == 0 && (fold_mode == 1 || (fold_mode == 0 && M >= 8192));
    const int P = Wd / kLnPartCols;
    float *part_a = nullptr, *part_b = nullptr, *fin = nullptr;
    if (fold) {
        part_a = (float*)ex.alloc_bytes((size_t)M * P * 2 * sizeof(float));
        part_b = (float*)ex.alloc_bytes((size_t)M * P * 2 * sizeof(float));
        fin = (float*)ex.alloc_bytes((size_t)M * 2 * sizeof(float));
        if (!part_a || !part_b || !fin) return ODISE_ERR_NOMEM;
    }
    auto lin = [&](const f16* a, const LinW& w, f16* y, int act, const f16* residual, const LnEpi& ln) -> int {
        odise_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.M = (int)M; d.N = w.out; d.K = w.in;
        d.A = a; d.lda = w.in; d.W = w.w; d.ldw = w.in;
        d.C = y; d.ldc = w.out; d.c_dtype = ODISE_F16;
        d.bias_n = w.b; d.residual = residual; d.ldr = w.out;
        d.act = act; d.alpha = 1.f; d.batch = 1;
        ex.ms->macs += (double)d.M * d.N * d.K;
        return gemm_ln(ex.ctx, &d, ln);
    };
    bool first = true;
    int layer = 0;
    for (const ClipBlock& b : e->clip_blocks) {
        const bool folded1 = fold && !first;   // statistics of x were left by the previous block's c_proj
        first = false;
        if (kv) {
            qk = kv->qk + (size_t)layer * kv->qk_stride;
            vt = kv->vt + (size_t)layer * kv->vt_stride;
            ++layer;
        }
        odise_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.M = Wd; d.N = (int)M; d.K = Wd;  // V^T of every image side by side (only its first T columns are ever read: the image tokens)
        d.lda = Wd; d.ldw = Wd;
        d.C = vt; d.ldc = ldvt; d.c_dtype = ODISE_F16;
        d.alpha = 1.f; d.batch = 1;
        if (folded1) {
            LnEpi l1;
            l1.part = part_b; l1.P = P; l1.inv_c = 1.f / (float)Wd; l1.eps = 1e-5f; l1.colsum = b.qk_cs; l1.final_out = fin;
            ODISE_TRY(lin(x, b.qk_f, qk, ODISE_ACT_NONE, nullptr, l1));
            LnEpi lv;   // swapped operands: the normalised tokens are the rows of W, (r1, rstd) finished by the q|k GEMM above
            lv.fin = fin; lv.rowsum = b.v_cs;
            d.A = b.v_f.w; d.W = x; d.bias_m = b.v_f_bias;
            ex.ms->macs += (double)d.M * d.N * d.K;
            ODISE_TRY(gemm_ln(ex.ctx, &d, lv));
        } else {
            ODISE_TRY(ex.layer_norm(x, n, M, b.ln1, 1e-5f));
            ODISE_TRY(ex.linear(n, M, b.qk, qk));
            d.A = b.v.w; d.W = n; d.bias_m = b.v_bias;
            ODISE_TRY(ex.gemm(d));
        }
        const bool last = &b == &e->clip_blocks.back();
        if (kv && last && Bk == B) {
            ex.ms->arena.release(mk);
            return ODISE_OK;
        }
        odise_attn_desc a;
        memset(&a, 0, sizeof(a));
        // The plain tower's result is ln_post(x[:, 0]) (clip.py:196-206): of the LAST block only the class token's row is ever read, so its
        // attention has one query per image and its out-proj / MLP run on B rows instead of B x 577 (dead work of the reference is not executed:
        // no output bit depends on the other 576 rows; -0.3 ms on the step's critical lane)
        const bool cls_only = extra == 0 && last;
        a.B = B; a.H = heads; a.Lq = cls_only ? 1 : TA; a.Lk = T; a.D = D;
        a.Q = qk; a.ldq = 2 * Wd; a.strideQ = (int64_t)TP * 2 * Wd;
        a.K = qk + Wd; a.ldk = 2 * Wd; a.strideK = (int64_t)TP * 2 * Wd;
        a.Vt = vt; a.ldvt = ldvt; a.strideVt = TP;
        a.O = att; a.ldo = Wd; a.strideO = (int64_t)TP * Wd;
        if (extra > 0) { a.mask = mask; a.ldmask = ldm; a.strideMask = (int64_t)TA * ldm; }
        a.scale = 1.0f / sqrtf((float)D);
        if (Bk > 0 && Bk < B) {
            // pictures and crops as two launches: the crops keep the grid the K / V^T-resident kernel wants ((head, image) pairs in whole rounds
            // of the chip, attn.hip attn_kvres_ok), the few pictures take the tiled kernel; in the last block the pictures are done
            odise_attn_desc ac = a;
            ac.B = B - Bk;
            ac.Q = qk + (size_t)Bk * a.strideQ; ac.K = qk + Wd + (size_t)Bk * a.strideK; ac.Vt = vt + (size_t)Bk * a.strideVt; ac.O = att + (size_t)Bk * a.strideO;
            ODISE_TRY(ex.attention(ac));
            if (!last) {
                a.B = Bk;
                ODISE_TRY(ex.attention(a));
            }
        } else {
            ODISE_TRY(ex.attention(a));
        }
        if (cls_only) {
            const int Bc = B - Bk;   // the images that have an output
            const size_t skip = (size_t)Bk * TP * Wd;
            f16* x2c = (f16*)ex.alloc_bytes((size_t)Bc * Wd * 2);
            f16* nc = (f16*)ex.alloc_bytes((size_t)Bc * Wd * 2);
            f16* hc = (f16*)ex.alloc_bytes((size_t)Bc * 4 * Wd * 2);
            f16* xc = (f16*)ex.alloc_bytes((size_t)Bc * Wd * 2);
            if (!x2c || !nc || !hc || !xc) return ODISE_ERR_NOMEM;
            memset(&d, 0, sizeof(d));   // x2[cls] = x[cls] + attn[cls] Wo^T + bo: the row stride TP*Wd picks token 0 of every image
            d.M = Bc; d.N = Wd; d.K = Wd;
            d.A = att + skip; d.lda = (int64_t)TP * Wd; d.W = b.out.w; d.ldw = Wd;
            d.C = x2c; d.ldc = Wd; d.c_dtype = ODISE_F16; d.bias_n = b.out.b;
            d.residual = x + skip; d.ldr = (int64_t)TP * Wd; d.alpha = 1.f; d.batch = 1;
            ODISE_TRY(ex.gemm(d));
            ODISE_TRY(ex.layer_norm(x2c, nc, Bc, b.ln2, 1e-5f));
            ODISE_TRY(ex.linear(nc, Bc, b.fc, hc, ODISE_ACT_QUICKGELU));
            ODISE_TRY(ex.linear(hc, Bc, b.proj, xc, ODISE_ACT_NONE, x2c));
            ODISE_TRY(ex.layer_norm(xc, nc, Bc, e->clip_ln_post, 1e-5f));
            memset(&d, 0, sizeof(d));
            d.M = Bc; d.N = e->clip_out; d.K = Wd; d.A = nc; d.lda = Wd; d.W = e->clip_proj.w; d.ldw = Wd;
            d.C = out; d.ldc = e->clip_out; d.c_dtype = ODISE_F16; d.alpha = 1.f; d.batch = 1;
            ODISE_TRY(ex.gemm(d));
            ex.ms->arena.release(mk);
            return ODISE_OK;
        }
        if (fold) {
            LnEpi lo, lf, lp;
            lo.stats_out = part_a;
            ODISE_TRY(lin(att, b.out, x2, ODISE_ACT_NONE, x, lo));               // x2 = x + attn (+ row statistics of x2)
            lf.part = part_a; lf.P = P; lf.inv_c = 1.f / (float)Wd; lf.eps = 1e-5f; lf.colsum = b.fc_cs;
            ODISE_TRY(lin(x2, b.fc_f, hid, ODISE_ACT_QUICKGELU, nullptr, lf));   // quick_gelu(ln_2(x2) Wfc^T + b)
            lp.stats_out = part_b;
            ODISE_TRY(lin(hid, b.proj, x, ODISE_ACT_NONE, x2, lp));              // x = x2 + mlp (+ row statistics of x)
        } else {
            ODISE_TRY(ex.linear(att, M, b.out, x2, ODISE_ACT_NONE, x));          // x2 = x + attn
            ODISE_TRY(ex.layer_norm(x2, n, M, b.ln2, 1e-5f));
            ODISE_TRY(ex.linear(n, M, b.fc, hid, ODISE_ACT_QUICKGELU));
            ODISE_TRY(ex.linear(hid, M, b.proj, x, ODISE_ACT_NONE, x2));         // x = x2 + mlp
        }
    }
    ODISE_TRY(ex.layer_norm(x, n, M, e->clip_ln_post, 1e-5f));
    odise_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.N = e->clip_out; d.K = Wd; d.W = e->clip_proj.w; d.ldw = Wd;
    d.C = out; d.ldc = e->clip_out; d.c_dtype = ODISE_F16; d.alpha = 1.f;
    if (extra == 0) {  // class token of every image: row stride TP*Wd picks token 0
        d.M = B; d.A = n; d.lda = (int64_t)TP * Wd; d.batch = 1;
    } else {           // the mask tokens of image b are rows T .. T+extra-1 of its block
        d.M = extra; d.A = n + (size_t)T * Wd; d.lda = Wd; d.strideA = (int64_t)TP * Wd;
        d.strideC = (int64_t)extra * e->clip_out; d.batch = B;
    }
    ODISE_TRY(ex.gemm(d));
    ex.ms->arena.release(mk);
    return ODISE_OK;
}


// ---- MaskCLIP in two passes (engine.h ClipKV; clip.py:252-323) ----------------------------------------------------------------------------
// Reserves and lays out the key / value store for B pictures that run the tower with `others` further images behind them.
static int maskclip_kv_layout(odise_hip_ctx* ctx, ModelStore* ms, int B, int others) {
    const ExtractorModel* e = ms->extractor;
    ClipKV& kv = ms->mclip;
    kv.ready = false;
    const int Wd = e->clip_width, L = (int)e->clip_blocks.size();
    const int TP = (int)round_up(e->clip_tokens, 8);
    const size_t Mp = (size_t)B * TP, Mo = (size_t)others * TP;
    const size_t rows = Mp * L + Mo;   // block l writes rows / columns [l * Mp, l * Mp + Mp + Mo): the tail is overwritten by the blocks after it
    const size_t need = (rows * 2 * Wd + (size_t)Wd * rows + Wd) * sizeof(f16) + 256;
    if (kv.cap < need) {
        // nothing enqueued may still read the old buffer (the mask-token pass of the previous call runs on the main stream, a tower on the second)
        ODISE_CHECK_HIP(hipDeviceSynchronize());
        if (kv.buf) ODISE_CHECK_HIP(hipFree(kv.buf));
        kv.buf = nullptr;
        kv.cap = 0;
        ODISE_CHECK_HIP(hipMalloc(&kv.buf, need + need / 8));
        kv.cap = need + need / 8;
    }
    kv.qk = (f16*)kv.buf;
    kv.vt = kv.qk + rows * 2 * Wd;
    kv.cls = kv.vt + (size_t)Wd * rows;
    kv.B = B; kv.TP = TP; kv.layers = L; kv.width = Wd;
    kv.qk_stride = Mp * 2 * Wd; kv.vt_stride = Mp; kv.ldvt = (int64_t)rows;
    return ODISE_OK;
}

// true when the key / value store for B pictures on their own is (or can be) reserved; false leaves no error behind: the caller runs the one-pass form
bool maskclip_kv_available(odise_hip_ctx* ctx, ModelStore* ms, int B) {
    if (maskclip_kv_layout(ctx, ms, B, 0) == ODISE_OK) return true;
    (void)hipGetLastError();
    return false;
}

// Pass 1 on its own: the image tokens of B pictures [B,3,H,W] in [0,1] (bilinear to 336^2 + CLIP normalisation, clip.py:325-338) through the tower.
int maskclip_image_pass(Exec& ex, const float* image01, int B, int H, int W) {
    ExtractorModel* e = ex.ms->extractor;
    ODISE_TRY(maskclip_kv_layout(ex.ctx, ex.ms, B, 0));
    const int S = e->clip_image;
    const size_t mk = ex.ms->arena.mark();
    Act img;
    int rc = ex.alloc(img, B, S, S, 8);
    if (rc == ODISE_OK) rc = launch_resize_bilinear_norm(ex.ctx, image01, img.p, B, H, W, S);
    if (rc == ODISE_OK) rc = clip_tower(ex, img, 0, nullptr, 0, nullptr, &ex.ms->mclip, B);
    ex.ms->arena.release(mk);
    return rc;
}

// Pass 2: Q mask tokens per picture (all copies of the class token after ln_pre) read the keys / values pass 1 left; mask [B][Q][ldm] u8
// (1 = patch hidden from that mask token; column 0 = the class token, always visible), pictures `stride_mask` bytes apart.  out [B,Q,clip_out].
int maskclip_mask_pass(Exec& ex, int Q, const uint8_t* mask, int64_t ldm, int64_t stride_mask, f16* out) {
    ExtractorModel* e = ex.ms->extractor;
    const ClipKV& kv = ex.ms->mclip;
    const int B = kv.B, TP = kv.TP, Wd = e->clip_width, T = e->clip_tokens, heads = e->clip_heads, D = Wd / heads;
    ODISE_REQUIRE(kv.buf && kv.layers == (int)e->clip_blocks.size() && kv.width == Wd, "maskclip: the image-token pass has not run for this tower");
    const int64_t M = (int64_t)B * Q;
    const size_t mk = ex.ms->arena.mark();
    f16* x = (f16*)ex.alloc_bytes((size_t)M * Wd * 2);
    f16* x2 = (f16*)ex.alloc_bytes((size_t)M * Wd * 2);
    f16* n = (f16*)ex.alloc_bytes((size_t)M * Wd * 2);
    f16* q = (f16*)ex.alloc_bytes((size_t)M * Wd * 2);
    f16* att = (f16*)ex.alloc_bytes((size_t)M * Wd * 2);
    f16* hid = (f16*)ex.alloc_bytes((size_t)M * 4 * Wd * 2);
    if (!x || !x2 || !n || !q || !att || !hid) return ODISE_ERR_NOMEM;
    ODISE_TRY(launch_broadcast_rows(ex.ctx, kv.cls, x, Wd, (int)M));
    int layer = 0;
    for (const ClipBlock& b : e->clip_blocks) {
        ODISE_TRY(ex.layer_norm(x, n, M, b.ln1, 1e-5f));
        LinW wq = b.qk;   // the query rows of the fused q|k projection
        wq.out = Wd;
        ODISE_TRY(ex.linear(n, M, wq, q));
        odise_attn_desc a;
        memset(&a, 0, sizeof(a));
        a.B = B; a.H = heads; a.Lq = Q; a.Lk = T; a.D = D;
        a.Q = q; a.ldq = Wd; a.strideQ = (int64_t)Q * Wd;
        a.K = kv.qk + (size_t)layer * kv.qk_stride + Wd; a.ldk = 2 * Wd; a.strideK = (int64_t)TP * 2 * Wd;
        a.Vt = kv.vt + (size_t)layer * kv.vt_stride; a.ldvt = kv.ldvt; a.strideVt = TP;
        a.O = att; a.ldo = Wd; a.strideO = (int64_t)Q * Wd;
        a.mask = mask; a.ldmask = ldm; a.strideMask = stride_mask;
        a.scale = 1.0f / sqrtf((float)D);
        ODISE_TRY(ex.attention(a));
        ODISE_TRY(ex.linear(att, M, b.out, x2, ODISE_ACT_NONE, x));          // x2 = x + attn
        ODISE_TRY(ex.layer_norm(x2, n, M, b.ln2, 1e-5f));
        ODISE_TRY(ex.linear(n, M, b.fc, hid, ODISE_ACT_QUICKGELU));
        ODISE_TRY(ex.linear(hid, M, b.proj, x, ODISE_ACT_NONE, x2));         // x = x2 + mlp
        ++layer;
    }
    ODISE_TRY(ex.layer_norm(x, n, M, e->clip_ln_post, 1e-5f));
    odise_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.M = (int)M; d.N = e->clip_out; d.K = Wd; d.A = n; d.lda = Wd; d.W = e->clip_proj.w; d.ldw = Wd;
    d.C = out; d.ldc = e->clip_out; d.c_dtype = ODISE_F16; d.alpha = 1.f; d.batch = 1;
    ODISE_TRY(ex.gemm(d));
    ex.ms->arena.release(mk);
    return ODISE_OK;
}

int clip_dims(ModelStore* ms, int* image, int* patch, int* tokens, int* out_dim) {
    ExtractorModel* e = ms->extractor;
    if (!e || !e->built) return ODISE_ERR_STATE;
    if (image) *image = e->clip_image;
    if (patch) *patch = e->clip_patch;
    if (tokens) *tokens = e->clip_tokens;
    if (out_dim) *out_dim = e->clip_out;
    return ODISE_OK;
}

// The implicit captioner's image embedding of the B crops (ldm.py:697-718).  When odise_hip_infer has planned MaskCLIP's image-token pass for its
// pictures (ODISE_OPT_MASKCLIP_PASSES 0), they ride in this tower as its first images: same weights, same token rows, and the GEMM grids of
// 16 crops have room for them in their last round (37 -> 46 row tiles of 256: 592 -> 736 tiles on 256 CUs, three rounds either way).
static int run_clip(Exec& ex, ExtractorModel* e, const float* image, int B, int H, int W, f16* prefix16 /*[B, clip_out]*/) {
    ClipKV& kv = ex.ms->mclip;
    bool ride = kv.planned && !kv.ready && ex.ctx->maskclip_passes == 0;
    if (ride && maskclip_kv_layout(ex.ctx, ex.ms, kv.plan_B, B) != ODISE_OK) {
        // the key / value store could not be reserved: the pictures do not ride; the classification stage falls back (classify.cpp maskclip_tower)
        (void)hipGetLastError();
        ride = false;
    }
    const int Bp = ride ? kv.plan_B : 0;
    const size_t mk = ex.ms->arena.mark();
    const int S = e->clip_image;
    Act img;
    ODISE_TRY(ex.alloc(img, Bp + B, S, S, 8));
    if (ride) ODISE_TRY(launch_resize_bilinear_norm(ex.ctx, kv.plan_image, img.p, Bp, kv.plan_H, kv.plan_W, S));
    ODISE_TRY(launch_clip_preprocess(ex.ctx, image, img.p + (size_t)Bp * S * S * 8, B, H, W, S));
    ODISE_TRY(clip_tower(ex, img, 0, nullptr, 0, prefix16, ride ? &kv : nullptr, Bp));
    if (ride) {
        kv.on_lane2 = false;   // ordered by the lane's join like everything else of this tower
        kv.ready = true;
    }
    ex.ms->arena.release(mk);
    return ODISE_OK;
}

static bool standalone_graph_capture(odise_hip_ctx* ctx) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(ctx->stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

size_t extractor_arena_bytes(int B, int H, int W) {
    // the VAE levels at full resolution dominate (128-channel maps of H x W, ~4 live at a time) + UNet + CLIP
    const size_t per = (size_t)H * W * 128 * 2 * 6 + ((size_t)900 << 20);
    return per * B + ((size_t)256 << 20);
}

const Act* extractor_taps(ModelStore* ms) { return ms->extractor ? ms->extractor->taps : nullptr; }
bool extractor_ready(ModelStore* ms) { return ms->extractor && ms->extractor->built; }

// ---- VAE encoder + latent heads (ldm.py:424-467, 543-566): shared by the call in progress and the prefetch of the next batch ------------------
// `after_level0` (optional) is called once the first level is enqueued (the two-lane step feeds its second lane there, see extractor_launch)
static int run_vae_encoder(Exec& ex, ExtractorModel* e, const float* image, int B, int H, int W, EncoderOut& out, const std::function<int()>* after_level0) {
    odise_hip_ctx* ctx = ex.ctx;
    const int lh = H / 8, lw = W / 8;
    // ---- VAE encoder ------------------------------------------------------------------------------------------------
    Act x;
    ODISE_TRY(ex.alloc(x, B, H, W, 8));
    {
        const float sc[3] = {2.f, 2.f, 2.f}, sh[3] = {-1.f, -1.f, -1.f};  // (img - 0.5) / 0.5  (ldm.py:556)
        ODISE_TRY(launch_image_to_nhwc(ctx, image, x.p, B, 3, H * W, 8, sc, sh));
    }
    // One level = [conv_in] -> two ResBlocks -> [downsample], run over `chunk` crops at a time (vae_chunk): the level's input and output are
    // batched tensors, everything between them lives in the chunk's scope of the arena - the same addresses for every chunk, so the working
    // set of a level is a handful of <= 64 MiB tensors that stay in the Infinity Cache between producer and consumer.  Per-crop arithmetic is
    // what it was (same kernels, same K order; only the row-block partition of the fused GroupNorm sums can follow a different tile choice).
    Act cur = x;
    int flat = 0;
    auto encoder_level = [&](int l) -> int {
        const int h = H >> l, w = W >> l;
        const int cout = e->enc_blocks[l][1].c1.cout;
        const int chunk = vae_chunk(ctx, B, (size_t)h * w * cout * 2);
        const bool tap0 = flat <= 5 && 5 < flat + 2, tap1 = flat <= 7 && 7 < flat + 2;   // the input of block 5 / 7 (ldm.py:437-438) is a batched output
        Act nxt, mid;      // mid: the tensor between the two blocks, batched only when it is a tap
        if (l < 3) ODISE_TRY(ex.alloc(nxt, B, h / 2, w / 2, e->enc_down[l].cout));
        else ODISE_TRY(ex.alloc(nxt, B, h, w, cout));
        const bool scoped = chunk < B;   // all crops at once: nothing is released (the statistics of the level's output feed the next block)
        const bool mid_is_tap = (tap0 && flat + 1 == 5) || (tap1 && flat + 1 == 7);
        if (mid_is_tap) ODISE_TRY(ex.alloc(mid, B, h, w, e->enc_blocks[l][0].c1.cout));
        if (mid_is_tap) (flat + 1 == 5 ? out.tap0 : out.tap1) = mid;
        for (int n0 = 0; n0 < B; n0 += chunk) {
            const int n = std::min(chunk, B - n0);
            const size_t mk = ex.ms->arena.mark();
            Act c0 = crop_slice(cur, n0, n);
            if (l == 0) {
                Act t;   // (conv_c8.hip.  Its statistics epilogue is not used here: the first block's norm1 keeps its own statistics pass, i.e. the
                         //  bits of rounds 1-5 - the parity tests' hard decisions were pinned on those)
                ODISE_TRY(ex.conv(c0, e->enc_conv_in, t, 1, 1));
                c0 = t;
            }
            Act b0 = mid_is_tap ? crop_slice(mid, n0, n) : Act();
            ODISE_TRY(run_vae_res(ex, e->enc_blocks[l][0], c0, b0));
            if (l < 3) {
                Act b1, d = crop_slice(nxt, n0, n);
                ODISE_TRY(run_vae_res(ex, e->enc_blocks[l][1], b0, b1));
                // F.pad (0,1,0,1) + conv3x3 stride 2 pad 0
                ODISE_TRY(ex.conv(b1, e->enc_down[l], d, 2, 0, false, nullptr, nullptr, 0, ODISE_ACT_NONE, 0, 0, h / 2, w / 2));
            } else {
                Act b1 = crop_slice(nxt, n0, n);
                ODISE_TRY(run_vae_res(ex, e->enc_blocks[l][1], b0, b1));
                if (!scoped) nxt = b1;    // keeps the fused GroupNorm statistics for mid.block_1
            }
            // the chunk's temporaries (and the statistics buffers of its slices) are released; batched outputs were allocated below the mark
            if (scoped) ex.ms->arena.release(mk);
        }
        cur = nxt;
        if (scoped || l < 3) { cur.gn_part = nullptr; cur.gn_blocks = 0; }
        flat += 2;
        return ODISE_OK;
    };
    ODISE_TRY(encoder_level(0));
    if (after_level0) ODISE_TRY((*after_level0)());
    for (int l = 1; l < 4; ++l) ODISE_TRY(encoder_level(l));
    {
        Act a, b2, c, nrm, h8;
        ODISE_TRY(run_vae_res(ex, e->enc_mid1, cur, a));
        ODISE_TRY(run_vae_attn(ex, e->enc_attn, a, b2));
        ODISE_TRY(run_vae_res(ex, e->enc_mid2, b2, c));
        ODISE_TRY(ex.group_norm(c, e->enc_norm_out, nrm, 1e-6f, ODISE_ACT_SILU));
        ODISE_TRY(ex.conv(nrm, e->enc_conv_out, h8, 1, 1));
        cur = h8;  // [B, lh, lw, 8]
    }
    // ---- latent: posterior mean * scale, q_sample(t=0), post_quant_conv -----------------------------------------------
    if (e->noise_hw != lh * lw) {
        set_error("extractor: latent %dx%d differs from the shared-noise size (%d elements); only the reference crop size is supported", lh, lw,
                  e->noise_hw);
        return ODISE_ERR_ARG;
    }
    ODISE_TRY(ex.alloc(out.xt, B, lh, lw, 8));
    ODISE_TRY(ex.alloc(out.zdec, B, lh, lw, 8));
    ODISE_TRY(launch_latent_heads(ctx, cur.p, e->noise, out.xt.p, out.zdec.p, nullptr, B, lh * lw, e->lat));
    return ODISE_OK;
}

size_t encoder_arena_bytes(int B, int H, int W) { return ((size_t)H * W * 128 * 2 * 6 + ((size_t)160 << 20)) * B + ((size_t)256 << 20); }

int extractor_encoder_only(odise_hip_ctx* ctx, ModelStore* ms, const float* image, int B, int H, int W, EncoderOut& out) {
    Exec ex{ctx, ms};
    return run_vae_encoder(ex, ms->extractor, image, B, H, W, out, nullptr);
}

// standalone=false: called from the backbone stage, which owns the arena and the MAC counter
// The UNet taps (2..5) are produced on the second lane: a caller that passed join = false consumes the VAE taps (0, 1, 6, 7) first and calls
// this before it touches a UNet tap (stream-side wait, the host never blocks)
int extractor_join(odise_hip_ctx* ctx) {
    if (ctx->lanes == 2 && !standalone_graph_capture(ctx)) ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return ODISE_OK;
}

// ODISE_OPT_MASKCLIP_PASSES 3: the image-token pass odise_hip_infer planned for its pictures as a tower of its own, enqueued by the backbone stage
// once both lanes of the extractor are: on the second lane behind the UNet, i.e. beside the projections, the pixel decoder and the masked decoder
// of the main stream, which waits for ev_mclip before the mask-token pass.  A pass that cannot be placed (one lane, no room in the lane's
// arena) is left to the classification stage, which then runs it in place.
int maskclip_planned_pass(odise_hip_ctx* ctx, ModelStore* ms) {
    ClipKV& kv = ms->mclip;
    if (!kv.planned || kv.ready || ctx->maskclip_passes != 3) return ODISE_OK;
    if (!(ctx->lanes == 2 && !standalone_graph_capture(ctx))) return ODISE_OK;
    const ExtractorModel* e = ms->extractor;
    const size_t rows = (size_t)kv.plan_B * round_up(e->clip_tokens, 8);
    const size_t need = rows * e->clip_width * 2 * 9 + (size_t)kv.plan_B * e->clip_image * e->clip_image * 16 * 2 + ((size_t)16 << 20);
    if (ms->arena2.cap - ms->arena2.off < need) return ODISE_OK;
    Lane2 lane(ctx, ms);
    Exec ex{ctx, ms};
    ODISE_TRY(maskclip_image_pass(ex, kv.plan_image, kv.plan_B, kv.plan_H, kv.plan_W));
    stage_mark(ctx, "lane 2: MaskCLIP image tokens done");
    ODISE_CHECK_HIP(hipEventRecord(ctx->ev_mclip, ctx->stream));
    kv.on_lane2 = true;
    kv.ready = true;
    return ODISE_OK;
}

int extractor_launch(odise_hip_ctx* ctx, ModelStore* ms, const float* image, int B, int H, int W, bool standalone, bool join) {
    ExtractorModel* e = ms->extractor;
    Exec ex{ctx, ms};
    if (standalone) {
        ms->arena.reset();
        ms->macs = 0.0;
    }
    const int lh = H / 8, lw = W / 8;
    // Two independent branches feed the taps: [CLIP image embedding -> conditioning -> UNet] and [VAE encoder -> latent -> VAE decoder];
    // the UNet only needs the latent of the second.  With two lanes the first branch is enqueued on the context's second stream (own arena,
    // own split-K workspace): its many short launches (CLIP GEMMs on 148 of 256 CUs, ~540 UNet kernels) execute in the gaps of the VAE's
    // chip-filling convolutions.  The reference runs the same modules one after the other (ldm.py:697-718, 543-621).
    const bool two = ctx->lanes == 2 && !standalone_graph_capture(ctx);
    if (two) {
        // CLIP activations (16 crops: ~1.5 GB) + UNet (~6 GB) - the extractor's own estimate covers both branches
        ODISE_TRY(ensure_lane2(ctx, ms, extractor_arena_bytes(B, H, W) / 2 + ((size_t)1 << 30)));
        ms->arena2.reset();
        ODISE_CHECK_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));           // the crops (and everything before) are ready
    }
    // ---- implicit captioner conditioning --------------------------------------------------------------------------
    float* cond_inputs = nullptr;
    float* cond_emb = nullptr;
    auto conditioning = [&]() -> int {
        f16* prefix16 = (f16*)ex.alloc_bytes((size_t)B * e->clip_out * 2);
        float* proj = (float*)ex.alloc_bytes((size_t)B * e->ctx_dim * 4);
        cond_inputs = (float*)ex.alloc_bytes((size_t)B * 77 * e->ctx_dim * 4);
        cond_emb = (float*)ex.alloc_bytes((size_t)B * e->ted * 4);
        if (!prefix16 || !proj || !cond_inputs || !cond_emb) return ODISE_ERR_NOMEM;
        ODISE_TRY(run_clip(ex, e, image, B, H, W, prefix16));
        odise_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.M = B; d.N = e->ctx_dim; d.K = e->cap_proj.in;
        d.A = prefix16; d.lda = e->cap_proj.in; d.W = e->cap_proj.w; d.ldw = e->cap_proj.in;
        d.C = proj; d.ldc = e->ctx_dim; d.c_dtype = ODISE_F32; d.bias_n = e->cap_proj.b; d.alpha = 1.f; d.batch = 1;
        ODISE_TRY(ex.gemm(d));
        ODISE_TRY(launch_cond_inputs(ctx, proj, e->cap_A1, e->cap_A2, cond_inputs, B, 77, e->ctx_dim));
        memset(&d, 0, sizeof(d));
        d.M = B; d.N = e->ted; d.K = e->cap_time.in;
        d.A = prefix16; d.lda = e->cap_time.in; d.W = e->cap_time.w; d.ldw = e->cap_time.in;
        d.C = cond_emb; d.ldc = e->ted; d.c_dtype = ODISE_F32; d.bias_n = e->cap_time.b; d.alpha = 1.f; d.batch = 1;
        return ex.gemm(d);
    };
    // ENQUEUE ORDER (two lanes).  The host issues ~1600 launches per step at roughly 20 us each, so the order in which the two lanes are
    // fed decides whether they overlap at all: with the ~450 launches of the CLIP tower enqueued first, the VAE encoder's first kernel
    // reached the GPU 11 ms into the step, and the VAE decoder waited another 10.7 ms behind the ~540 launches of the UNet - each lane
    // idled while the host was busy feeding the other (profiles/r03_lane_scheduling.txt).  The few, chip-filling launches of the VAE go first:
    // encoder, then the CLIP branch, then the decoder, then the UNet (which waits for the encoder's latent on the device anyway).
    // What remains: beside the VAE's convolutions (100-150 us per workgroup on every CU) each of the ~1000 dependent launches of the other
    // lane waits for workgroups to drain, so CLIP takes 31 ms instead of 13 there.  A higher stream priority for that lane is worth -1.7 ms;
    // reserving every 2nd .. 8th CU for it with a CU-masked VAE stream changed nothing (profiles/r03_lane_scheduling.txt).
#ifdef ODISE_TOOLS
    static const bool vae_first = getenv("ODISE_LANE_ORDER_OLD") == nullptr;   // A/B of the enqueue order
#else
    const bool vae_first = true;
#endif
    auto clip_on_lane2 = [&]() -> int {
        Lane2 lane(ctx, ms);                                                  // CLIP branch: independent of the encoder, starts at the fork
        ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_fork, 0));
        stage_mark(ctx, "lane 2: start (CLIP image tower of the crops)");
        ODISE_TRY(conditioning());
        stage_mark(ctx, "lane 2: CLIP + conditioning done");
        return ODISE_OK;
    };
    bool clip_enqueued = false;
    if (!two) ODISE_TRY(conditioning());
    if (two && !vae_first) ODISE_TRY(clip_on_lane2());

    // ---- VAE encoder + latent: computed here, or taken from the prefetch the previous call ran for this batch ---------------------------
    EncoderOut enc;
    const bool prefetched = ms->pf.use_now && !standalone && ms->pf.crops == B;
    ms->pf.use_now = false;   // consumed once, whatever happens below
    if (prefetched) {
        enc = ms->pf.out;
        ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_pf_done, 0));
        if (two && vae_first) { ODISE_TRY(clip_on_lane2()); clip_enqueued = true; }
    } else {
        // the CLIP branch is enqueued behind the first level when the levels run in crop chunks: that level is then ~200 launches (a few ms of
        // host time, ~25 ms of device time), so the second lane's ~450 CLIP launches reach the device while the first is still busy with it
        std::function<int()> hook = [&]() -> int {
            if (two && vae_first && vae_chunk(ctx, B, (size_t)H * W * 128 * 2) < B) {
                ODISE_TRY(clip_on_lane2());
                clip_enqueued = true;
            }
            return ODISE_OK;
        };
        ODISE_TRY(run_vae_encoder(ex, e, image, B, H, W, enc, &hook));
    }
    e->taps[0] = enc.tap0;
    e->taps[1] = enc.tap1;
    const Act xt = enc.xt, zdec = enc.zdec;
    auto unet_on_lane2 = [&]() -> int {   // waits for the latent on the device; its ~540 launches take the host ~10 ms
        Lane2 lane(ctx, ms);
        ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_mid, 0));
        stage_mark(ctx, "lane 2: latent available, UNet starts");
        ODISE_TRY(unet_launch(ctx, ms, ms->unet, nullptr, xt.p, cond_inputs, cond_emb, B, lh, lw, false));
        stage_mark(ctx, "lane 2: UNet done");
        ODISE_CHECK_HIP(hipEventRecord(ctx->ev_join, ctx->stream));
        return ODISE_OK;
    };
    stage_mark(ctx, "extractor: VAE encoder + latent done");
    bool unet_enqueued = false;   // (the CLIP branch may already sit on the second lane - crop chunks, prefetched encoder: then the UNet follows it here, once)
    if (two) {
        ODISE_CHECK_HIP(hipEventRecord(ctx->ev_mid, ctx->stream));            // the latent is ready
        if (vae_first && !clip_enqueued) ODISE_TRY(clip_on_lane2());
        else { ODISE_TRY(unet_on_lane2()); unet_enqueued = true; }
    } else {
        // ---- UNet (t = 0), single lane: before the decoder, as the reference orders its modules
        ODISE_TRY(unet_launch(ctx, ms, ms->unet, nullptr, xt.p, cond_inputs, cond_emb, B, lh, lw, false));
    }
    // ---- VAE decoder up to the last tap ----------------------------------------------------------------------------------
    {
        Act h, a, b2, c;
        ODISE_TRY(ex.conv(zdec, e->dec_conv_in, h, 1, 1));
        ODISE_TRY(run_vae_res(ex, e->dec_mid1, h, a));
        ODISE_TRY(run_vae_attn(ex, e->dec_attn, a, b2));
        ODISE_TRY(run_vae_res(ex, e->dec_mid2, b2, c));
        Act l0, l1, l2, m1;
        ODISE_TRY(run_vae_res(ex, e->dec_l3[0], c, l0));
        ODISE_TRY(run_vae_res(ex, e->dec_l3[1], l0, l1));
        e->taps[6] = l1;  // input of up block 2 (ldm.py:515-516)
        ODISE_TRY(run_vae_res(ex, e->dec_l3[2], l1, l2));
        // upsample + the two blocks at twice the resolution, in crop chunks like the encoder's levels
        const int uh = 2 * l2.h, uw = 2 * l2.w, uc = e->dec_l2[1].c1.cout;
        const int chunk = vae_chunk(ctx, B, (size_t)uh * uw * uc * 2);
        ODISE_TRY(ex.alloc(m1, B, uh, uw, uc));
        for (int n0 = 0; n0 < B; n0 += chunk) {
            const int n = std::min(chunk, B - n0);
            const size_t mk = ex.ms->arena.mark();
            Act up, m0, o = crop_slice(m1, n0, n);
            ODISE_TRY(ex.conv(crop_slice(l2, n0, n), e->dec_up3, up, 1, 1, true));
            ODISE_TRY(run_vae_res(ex, e->dec_l2[0], up, m0));
            ODISE_TRY(run_vae_res(ex, e->dec_l2[1], m0, o));
            ex.ms->arena.release(mk);
        }
        e->taps[7] = m1;  // input of up block 5
    }
    // ---- UNet (t = 0), second lane: enqueued last
    if (two && vae_first && !unet_enqueued) ODISE_TRY(unet_on_lane2());
    {
        const Act* ut = unet_taps(ms);
        for (int i = 0; i < 4; ++i) e->taps[2 + i] = ut[i];
    }
    if (two && join) ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));   // join: the UNet taps are ready for whoever consumes them on the main stream
    e->last_macs = ms->macs;
    return ODISE_OK;
}

static int extractor_forward(odise_hip_ctx* ctx, const float* image, int B, int H, int W) {
    ModelStore* ms = store_of(ctx);
    ExtractorModel* e = ms->extractor;
    if (!e || !e->built) {
        set_error("extractor_forward: call odise_hip_extractor_build first");
        return ODISE_ERR_STATE;
    }
    ODISE_REQUIRE(image && B >= 1 && H >= 64 && W >= 64 && H % 64 == 0 && W % 64 == 0, "extractor_forward: image %dx%d must be a multiple of 64", H, W);
    ODISE_TRY(unet_prepare_timestep(ctx, ms, ms->unet, B, 0));
    ODISE_TRY(ensure_arena(ctx, ms, extractor_arena_bytes(B, H, W)));
    return extractor_launch(ctx, ms, image, B, H, W, true);
}

}  // namespace odise

using namespace odise;

extern "C" int odise_hip_extractor_build(odise_hip_ctx* ctx) {
    ODISE_REQUIRE(ctx, "extractor_build: null context");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));
    return extractor_build(ctx);
}

extern "C" int odise_hip_extractor_forward_nhwc(odise_hip_ctx* ctx, const float* image, int B, int H, int W, void** taps8, int* shapes8x4) {
    ODISE_REQUIRE(ctx, "extractor_forward: null context");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    ODISE_TRY(extractor_forward(ctx, image, B, H, W));
    ExtractorModel* e = store_of(ctx)->extractor;
    for (int i = 0; i < 8; ++i) {
        if (taps8) taps8[i] = e->taps[i].p;
        if (shapes8x4) {
            shapes8x4[4 * i + 0] = e->taps[i].n; shapes8x4[4 * i + 1] = e->taps[i].c;
            shapes8x4[4 * i + 2] = e->taps[i].h; shapes8x4[4 * i + 3] = e->taps[i].w;
        }
    }
    return ODISE_OK;
}

extern "C" int odise_hip_extractor_forward(odise_hip_ctx* ctx, const float* image, int B, int H, int W, float** taps8) {
    ODISE_REQUIRE(ctx && taps8, "extractor_forward: null argument");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    ODISE_TRY(extractor_forward(ctx, image, B, H, W));
    ExtractorModel* e = store_of(ctx)->extractor;
    for (int i = 0; i < 8; ++i) {
        if (!taps8[i]) continue;
        const Act& a = e->taps[i];
        ODISE_TRY(odise_hip_nhwc_f16_to_nchw_f32(ctx, a.p, taps8[i], a.n, a.c, a.h, a.w));
    }
    return ODISE_OK;
}

extern "C" int odise_hip_extractor_last_macs(odise_hip_ctx* ctx, double* macs) {
    ODISE_REQUIRE(ctx && macs, "extractor_last_macs: null argument");
    ExtractorModel* e = store_of(ctx)->extractor;
    *macs = e ? e->last_macs : 0.0;
    return ODISE_OK;
}
