// unet.cpp — SD v1 UNet single-step feature extraction (LdmExtractor.unet_forward, odise/modeling/meta_arch/ldm.py:469-491).
//
// The architecture (ldm UNetModel, v1-inference.yaml: model_channels 320, channel_mult 1-2-4-4, 2 ResBlocks per level,
// SpatialTransformer at 64^2/32^2/16^2 latents, 8 heads, context 768) is restated from SURVEY.md Appendix A.1; weights
// are addressed by their checkpoint keys (model.diffusion_model.* with the prefix stripped).  Taps are the concatenated
// INPUTS of output blocks 2, 5, 8, 11 (ldm.py:485-488); output block 11 itself and `out` are dead (ldm.py:491 discards
// the result) and are not executed.
//
// Execution: NHWC fp16 activations; every conv / linear is the MFMA (implicit-)GEMM of gemm.hip with fused
// bias / time-embedding / residual / GEGLU epilogues; GroupNorm+SiLU and LayerNorm are the streaming kernels of norm.hip;
// attention is attn.hip with V produced pre-transposed by a swapped GEMM.  The forward is a fixed launch sequence on one
// stream (pointer-stable arena) and can be captured into a hipGraph.
#include <math.h>
#include <string.h>

#include "engine.h"

namespace odise {

struct ResBlockW {
    NormW n1, n2;
    ConvW c1, c2, skip;
    bool has_skip = false;
    int cin = 0, cout = 0;
    int emb_off = 0;  // column offset of this block's emb_layers output in the fused projection
};

struct STBlockW {
    int c = 0;
    NormW gn, ln1, ln2, ln3;
    LinW proj_in, proj_out;
    LinW qk1, v1, o1;  // self-attention: to_q|to_k stacked [2C,C], to_v, to_out.0
    LinW q2, k2, v2, o2;  // cross-attention (context 768)
    LinW ff1, ff2;        // GEGLU proj (rows interleaved a/gate) and output
    int ctx_slot = 0;     // index into the precomputed context K / V^T buffers
};

struct UBlock {
    bool has_res = false, has_st = false, has_down = false, has_up = false, is_conv_in = false;
    ResBlockW res;
    STBlockW st;
    ConvW conv;  // conv_in / downsample op / upsample conv
};

struct UNetModel {
    std::vector<void*> owned;   // device weights of this stage (AllocScope)
    bool built = false;
    LinW te0, te2;
    LinW emb_all;  // all ResBlock emb_layers.1 stacked along N
    std::vector<UBlock> in_blocks, out_blocks;
    ResBlockW mid_r1, mid_r2;
    STBlockW mid_st;
    int n_ctx_slots = 0;
    int emb_total = 0;
    int mc = 320, ted = 1280, cdim = 768;  // model_channels, time-embed width, context width (read from the weights)
    // cached timestep embedding input (constant per t)
    int cached_t = -1, cached_B = 0;
    f16* temb_in = nullptr;  // [maxB, 320]
    // graph replay
    bool use_graph = false;
    hipGraphExec_t graph_exec = nullptr;
    int graph_B = 0, graph_h = 0, graph_w = 0;
    const void *graph_x = nullptr, *graph_ctx = nullptr, *graph_ce = nullptr, *graph_arena = nullptr;
    // outputs of the last forward (NHWC f16, arena)
    Act taps[4];
    double last_macs = 0.0;
};

const Act* unet_taps(ModelStore* ms) { return ms->unet ? ms->unet->taps : nullptr; }
double unet_last_macs(ModelStore* ms) { return ms->unet ? ms->unet->last_macs : 0.0; }

void unet_destroy(ModelStore* ms) {
    if (!ms->unet) return;
    if (ms->unet->graph_exec) (void)hipGraphExecDestroy(ms->unet->graph_exec);
    free_allocs(ms->unet->owned);
    if (ms->unet->temb_in) { (void)hipDeviceSynchronize(); (void)hipFree(ms->unet->temb_in); }
    delete ms->unet;
    ms->unet = nullptr;
}

// ---------------------------------------------------------------------------------------------------------------
static int build_res(Packer& pk, const std::string& key, ResBlockW& r, std::vector<const HostTensor*>& emb_w,
                     std::vector<const HostTensor*>& emb_b, int& emb_total, int pk_ted) {
    ODISE_TRY(pk.norm(key + ".in_layers.0", r.n1));
    ODISE_TRY(pk.conv(key + ".in_layers.2", r.c1));
    ODISE_TRY(pk.norm(key + ".out_layers.0", r.n2));
    ODISE_TRY(pk.conv(key + ".out_layers.3", r.c2));
    r.cin = r.c1.cin;
    r.cout = r.c1.cout;
    r.has_skip = pk.find(key + ".skip_connection.weight") != nullptr;
    if (r.has_skip) ODISE_TRY(pk.conv(key + ".skip_connection", r.skip));
    else if (r.cin != r.cout) {
        set_error("unet: '%s' changes channels %d->%d but has no skip_connection", key.c_str(), r.cin, r.cout);
        return ODISE_ERR_STATE;
    }
    const HostTensor* w = pk.find(key + ".emb_layers.1.weight");
    const HostTensor* b = pk.find(key + ".emb_layers.1.bias");
    if (!w || !b || w->shape.size() != 2 || w->shape[0] != r.cout || w->shape[1] != pk_ted) {
        set_error("unet: bad or missing '%s.emb_layers.1'", key.c_str());
        return ODISE_ERR_STATE;
    }
    r.emb_off = emb_total;
    emb_total += r.cout;
    emb_w.push_back(w);
    emb_b.push_back(b);
    return ODISE_OK;
}

static int build_st(Packer& pk, const std::string& key, STBlockW& s, int& n_slots) {
    ODISE_TRY(pk.norm(key + ".norm", s.gn));
    s.c = s.gn.c;
    ODISE_TRY(pk.linear(key + ".proj_in", s.proj_in));
    ODISE_TRY(pk.linear(key + ".proj_out", s.proj_out));
    const std::string tb = key + ".transformer_blocks.0";
    ODISE_TRY(pk.norm(tb + ".norm1", s.ln1));
    ODISE_TRY(pk.norm(tb + ".norm2", s.ln2));
    ODISE_TRY(pk.norm(tb + ".norm3", s.ln3));
    // self attention: stack to_q and to_k into one [2C, C] matrix
    const HostTensor* wq = pk.find(tb + ".attn1.to_q.weight");
    const HostTensor* wk = pk.find(tb + ".attn1.to_k.weight");
    if (!wq || !wk || wq->numel() != (int64_t)s.c * s.c || wk->numel() != (int64_t)s.c * s.c) {
        set_error("unet: bad or missing '%s.attn1.to_q/to_k'", tb.c_str());
        return ODISE_ERR_STATE;
    }
    {
        std::vector<f16> qk((size_t)2 * s.c * s.c);
        for (size_t i = 0; i < (size_t)s.c * s.c; ++i) {
            qk[i] = (f16)wq->data[i];
            qk[(size_t)s.c * s.c + i] = (f16)wk->data[i];
        }
        s.qk1.in = s.c; s.qk1.out = 2 * s.c; s.qk1.b = nullptr;
        ODISE_TRY(pk.upload(qk.data(), qk.size() * sizeof(f16), (void**)&s.qk1.w));
    }
    ODISE_TRY(pk.linear(tb + ".attn1.to_v", s.v1, false));
    ODISE_TRY(pk.linear(tb + ".attn1.to_out.0", s.o1));
    ODISE_TRY(pk.linear(tb + ".attn2.to_q", s.q2, false));
    ODISE_TRY(pk.linear(tb + ".attn2.to_k", s.k2, false));
    ODISE_TRY(pk.linear(tb + ".attn2.to_v", s.v2, false));
    ODISE_TRY(pk.linear(tb + ".attn2.to_out.0", s.o2));
    // GEGLU: proj(x).chunk(2) = (a, gate); interleave rows so that a tile holds (a_j, gate_j) pairs
    const HostTensor* fw = pk.find(tb + ".ff.net.0.proj.weight");
    const HostTensor* fb = pk.find(tb + ".ff.net.0.proj.bias");
    if (!fw || !fb || fw->numel() != (int64_t)8 * s.c * s.c || fb->numel() != 8 * s.c) {
        set_error("unet: bad or missing '%s.ff.net.0.proj'", tb.c_str());
        return ODISE_ERR_STATE;
    }
    {
        const int inner = 4 * s.c;
        std::vector<f16> w((size_t)8 * s.c * s.c);
        std::vector<float> b((size_t)8 * s.c);
        for (int j = 0; j < inner; ++j) {
            for (int k = 0; k < s.c; ++k) {
                w[((size_t)2 * j) * s.c + k] = (f16)fw->data[(size_t)j * s.c + k];
                w[((size_t)2 * j + 1) * s.c + k] = (f16)fw->data[((size_t)inner + j) * s.c + k];
            }
            b[2 * j] = fb->data[j];
            b[2 * j + 1] = fb->data[inner + j];
        }
        s.ff1.in = s.c; s.ff1.out = 8 * s.c;
        ODISE_TRY(pk.upload(w.data(), w.size() * sizeof(f16), (void**)&s.ff1.w));
        ODISE_TRY(pk.upload(b.data(), b.size() * sizeof(float), (void**)&s.ff1.b));
    }
    ODISE_TRY(pk.linear(tb + ".ff.net.2", s.ff2));
    s.ctx_slot = n_slots++;
    return ODISE_OK;
}

int unet_build(odise_hip_ctx* ctx, const char* prefix) {
    ModelStore* ms = store_of(ctx);
    unet_destroy(ms);
    UNetModel* u = new UNetModel();
    ms->unet = u;
    AllocScope scope(ms, u->owned);
    Packer pk{ctx, ms, prefix, ""};
    std::vector<const HostTensor*> emb_w, emb_b;
    ODISE_TRY(pk.linear("time_embed.0", u->te0));
    ODISE_TRY(pk.linear("time_embed.2", u->te2));
    u->mc = u->te0.in;
    u->ted = u->te0.out;
    if (u->te2.in != u->ted || u->te2.out != u->ted || u->mc % 16 != 0) {
        set_error("unet: inconsistent time_embed shapes");
        return ODISE_ERR_STATE;
    }

    // ---- input blocks --------------------------------------------------------------------------------------
    // (channel schedule from the checkpoint itself: a block is Res[+ST] / Downsample / conv_in by which keys exist)
    for (int i = 0; i < 12; ++i) {
        UBlock b;
        const std::string k = "input_blocks." + std::to_string(i);
        if (i == 0) {
            b.is_conv_in = true;
            ODISE_TRY(pk.conv(k + ".0", b.conv));
        } else if (pk.find(k + ".0.op.weight")) {
            b.has_down = true;
            ODISE_TRY(pk.conv(k + ".0.op", b.conv));
        } else {
            b.has_res = true;
            ODISE_TRY(build_res(pk, k + ".0", b.res, emb_w, emb_b, u->emb_total, u->ted));
            if (pk.find(k + ".1.norm.weight")) {
                b.has_st = true;
                ODISE_TRY(build_st(pk, k + ".1", b.st, u->n_ctx_slots));
            }
        }
        u->in_blocks.push_back(b);
    }
    ODISE_TRY(build_res(pk, "middle_block.0", u->mid_r1, emb_w, emb_b, u->emb_total, u->ted));
    ODISE_TRY(build_st(pk, "middle_block.1", u->mid_st, u->n_ctx_slots));
    ODISE_TRY(build_res(pk, "middle_block.2", u->mid_r2, emb_w, emb_b, u->emb_total, u->ted));
    // ---- output blocks 0..10 (block 11 is dead: its INPUT is the last tap) ------------------------------
    for (int i = 0; i < 11; ++i) {
        UBlock b;
        const std::string k = "output_blocks." + std::to_string(i);
        b.has_res = true;
        ODISE_TRY(build_res(pk, k + ".0", b.res, emb_w, emb_b, u->emb_total, u->ted));
        int next = 1;
        if (pk.find(k + ".1.norm.weight")) {
            b.has_st = true;
            ODISE_TRY(build_st(pk, k + ".1", b.st, u->n_ctx_slots));
            next = 2;
        }
        if (pk.find(k + "." + std::to_string(next) + ".conv.weight")) {
            b.has_up = true;
            ODISE_TRY(pk.conv(k + "." + std::to_string(next) + ".conv", b.conv));
        }
        u->out_blocks.push_back(b);
    }
    u->cdim = u->mid_st.k2.in;
    // ---- fused emb_layers projection: one [sum cout, ted] matrix ------------------------------------------
    {
        const size_t ted = (size_t)u->ted;
        std::vector<f16> w((size_t)u->emb_total * ted);
        std::vector<float> bvec((size_t)u->emb_total);
        size_t row = 0;
        for (size_t i = 0; i < emb_w.size(); ++i) {
            const size_t rows = (size_t)emb_w[i]->shape[0];
            for (size_t j = 0; j < rows * ted; ++j) w[row * ted + j] = (f16)emb_w[i]->data[j];
            for (size_t j = 0; j < rows; ++j) bvec[row + j] = emb_b[i]->data[j];
            row += rows;
        }
        u->emb_all.in = (int)ted; u->emb_all.out = u->emb_total;
        ODISE_TRY(pk.upload(w.data(), w.size() * sizeof(f16), (void**)&u->emb_all.w));
        ODISE_TRY(pk.upload(bvec.data(), bvec.size() * sizeof(float), (void**)&u->emb_all.b));
    }
    u->built = true;
    return ODISE_OK;
}

// ---------------------------------------------------------------------------------------------------------------
struct UNetRun {
    Exec ex;
    UNetModel* u;
    int B;
    const float* emb_out;  // [B, emb_total] fp32
    f16* ctx16;            // [B*77, 768]
    std::vector<f16*> ctxK;   // per slot [B,77,C]
    std::vector<f16*> ctxVt;  // per slot [B,C,80]
};

static int run_res(UNetRun& r, const ResBlockW& w, const Act& x, Act& out) {
    Exec& ex = r.ex;
    ODISE_TRY(ex.alloc(out, x.n, x.h, x.w, w.cout));
    const size_t mk = ex.ms->arena.mark();
    Act t1, hcur, t2, sk;
    ODISE_TRY(ex.group_norm(x, w.n1, t1, 1e-5f, ODISE_ACT_SILU));
    ODISE_TRY(ex.conv(t1, w.c1, hcur, 1, 1, false, nullptr, r.emb_out + w.emb_off, r.u->emb_total));
    ODISE_TRY(ex.group_norm(hcur, w.n2, t2, 1e-5f, ODISE_ACT_SILU));
    const Act* resid = &x;
    if (w.has_skip) {
        ODISE_TRY(ex.conv(x, w.skip, sk, 1, 0));
        resid = &sk;
    }
    ODISE_TRY(ex.conv(t2, w.c2, out, 1, 1, false, resid));
    ex.ms->arena.release(mk);
    return ODISE_OK;
}

static int run_st(UNetRun& r, const STBlockW& w, const Act& x, Act& out) {
    Exec& ex = r.ex;
    const int C = w.c, heads = 8, D = C / heads;
    const int64_t HW = (int64_t)x.h * x.w, M = x.pixels();
    ODISE_TRY(ex.alloc(out, x.n, x.h, x.w, C));
    const size_t mk = ex.ms->arena.mark();
    Act t;
    ODISE_TRY(ex.group_norm(x, w.gn, t, 1e-6f, ODISE_ACT_NONE));
    f16* hs = (f16*)ex.alloc_bytes((size_t)M * C * 2);
    f16* nrm = (f16*)ex.alloc_bytes((size_t)M * C * 2);
    f16* qk = (f16*)ex.alloc_bytes((size_t)M * 2 * C * 2);
    const int64_t ldvt = round_up(HW, 8);
    f16* vt = (f16*)ex.alloc_bytes((size_t)x.n * C * ldvt * 2);
    f16* att = (f16*)ex.alloc_bytes((size_t)M * C * 2);
    f16* hs2 = (f16*)ex.alloc_bytes((size_t)M * C * 2);
    f16* ffh = (f16*)ex.alloc_bytes((size_t)M * 4 * C * 2);
    if (!hs || !nrm || !qk || !vt || !att || !hs2 || !ffh) return ODISE_ERR_NOMEM;
    ODISE_TRY(ex.linear(t.p, M, w.proj_in, hs));
    // ---- self attention ---------------------------------------------------------------------------------
    ODISE_TRY(ex.layer_norm(hs, nrm, M, w.ln1, 1e-5f));
    ODISE_TRY(ex.linear(nrm, M, w.qk1, qk));
    {
        odise_gemm_desc d;  // V^T[b] = Wv @ n[b]^T  -> [B, C, HW]
        memset(&d, 0, sizeof(d));
        d.M = C; d.N = (int)HW; d.K = C;
        d.A = w.v1.w; d.lda = C; d.strideA = 0;
        d.W = nrm; d.ldw = C; d.strideW = HW * C;
        d.C = vt; d.ldc = ldvt; d.strideC = (int64_t)C * ldvt; d.c_dtype = ODISE_F16;
        d.alpha = 1.f; d.batch = x.n;
        ODISE_TRY(ex.gemm(d));
        odise_attn_desc a;
        memset(&a, 0, sizeof(a));
        a.B = x.n; a.H = heads; a.Lq = (int)HW; a.Lk = (int)HW; a.D = D;
        a.Q = qk; a.ldq = 2 * C; a.strideQ = HW * 2 * C;
        a.K = qk + C; a.ldk = 2 * C; a.strideK = HW * 2 * C;
        a.Vt = vt; a.ldvt = ldvt; a.strideVt = (int64_t)C * ldvt;
        a.O = att; a.ldo = C; a.strideO = HW * C;
        a.scale = 1.0f / sqrtf((float)D);
        ODISE_TRY(ex.attention(a));
    }
    ODISE_TRY(ex.linear(att, M, w.o1, hs2, ODISE_ACT_NONE, hs));  // hs2 = attn1 + hs
    // ---- cross attention (77 context tokens; K / V^T precomputed per layer) -------------------------------
    ODISE_TRY(ex.layer_norm(hs2, nrm, M, w.ln2, 1e-5f));
    ODISE_TRY(ex.linear(nrm, M, w.q2, qk));  // reuse qk buffer as [M, C]
    {
        odise_attn_desc a;
        memset(&a, 0, sizeof(a));
        a.B = x.n; a.H = heads; a.Lq = (int)HW; a.Lk = 77; a.D = D;
        a.Q = qk; a.ldq = C; a.strideQ = HW * C;
        a.K = r.ctxK[w.ctx_slot]; a.ldk = C; a.strideK = 77 * (int64_t)C;
        a.Vt = r.ctxVt[w.ctx_slot]; a.ldvt = 80; a.strideVt = (int64_t)C * 80;
        a.O = att; a.ldo = C; a.strideO = HW * C;
        a.scale = 1.0f / sqrtf((float)D);
        ODISE_TRY(ex.attention(a));
    }
    ODISE_TRY(ex.linear(att, M, w.o2, hs, ODISE_ACT_NONE, hs2));  // hs = attn2 + hs2
    // ---- GEGLU feed-forward ----------------------------------------------------------------------------
    ODISE_TRY(ex.layer_norm(hs, nrm, M, w.ln3, 1e-5f));
    ODISE_TRY(ex.linear(nrm, M, w.ff1, ffh, ODISE_ACT_NONE, nullptr, true));
    ODISE_TRY(ex.linear(ffh, M, w.ff2, hs2, ODISE_ACT_NONE, hs));  // hs2 = ff + hs
    ODISE_TRY(ex.linear(hs2, M, w.proj_out, out.p, ODISE_ACT_NONE, x.p));
    ex.ms->arena.release(mk);
    return ODISE_OK;
}

int ensure_arena(odise_hip_ctx* ctx, ModelStore* ms, size_t bytes) {
    if (ms->arena.cap >= bytes) return ODISE_OK;
    ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (ms->arena.base) ODISE_CHECK_HIP(hipFree(ms->arena.base));
    ms->arena = Arena();
    ODISE_CHECK_HIP(hipMalloc((void**)&ms->arena.base, bytes));
    ms->arena.cap = bytes;
    return ODISE_OK;
}

// the launch sequence proper (graph-capturable: no allocation, no sync, no host<->device copy)
// x_nhwc (optional): x_t already as NHWC fp16 with 8 channels (4 + zero pad) — the in-library path of the feature
// extractor; otherwise x_t is fp32 NCHW.  standalone=false keeps the caller's arena contents and MAC counter.
int unet_launch(odise_hip_ctx* ctx, ModelStore* ms, UNetModel* u, const float* x_t, const f16* x_nhwc, const float* context,
                const float* cond_emb, int B, int h, int w, bool standalone) {
    UNetRun r;
    r.ex = Exec{ctx, ms};
    r.u = u;
    r.B = B;
    Exec& ex = r.ex;
    if (standalone) {
        ms->arena.reset();
        ms->macs = 0.0;
    }

    // ---- time embedding: emb = time_embed(t_emb) + cond_emb; every ResBlock consumes Linear(SiLU(emb)) -------------
    f16* te_h = (f16*)ex.alloc_bytes((size_t)B * u->ted * 2);
    f16* emb_silu = (f16*)ex.alloc_bytes((size_t)B * u->ted * 2);
    float* emb_out = (float*)ex.alloc_bytes((size_t)B * u->emb_total * 4);
    if (!te_h || !emb_silu || !emb_out) return ODISE_ERR_NOMEM;
    ODISE_TRY(ex.linear(u->temb_in, B, u->te0, te_h, ODISE_ACT_SILU));
    {
        odise_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.M = B; d.N = u->ted; d.K = u->ted;
        d.A = te_h; d.lda = u->ted; d.W = u->te2.w; d.ldw = u->ted;
        d.C = emb_silu; d.ldc = u->ted; d.c_dtype = ODISE_F16;
        d.bias_n = u->te2.b;
        d.rowgroup_add = cond_emb; d.rows_per_group = 1; d.ldg = u->ted;  // emb += cond_emb (ldm.py:474-477)
        d.act = ODISE_ACT_SILU; d.alpha = 1.f; d.batch = 1;
        ODISE_TRY(ex.gemm(d));
        memset(&d, 0, sizeof(d));
        d.M = B; d.N = u->emb_total; d.K = u->ted;
        d.A = emb_silu; d.lda = u->ted; d.W = u->emb_all.w; d.ldw = u->ted;
        d.C = emb_out; d.ldc = u->emb_total; d.c_dtype = ODISE_F32;
        d.bias_n = u->emb_all.b; d.alpha = 1.f; d.batch = 1;
        ODISE_TRY(ex.gemm(d));
    }
    r.emb_out = emb_out;

    // ---- context: fp16 copy + per-layer K and V^T ---------------------------------------------------------------
    r.ctx16 = (f16*)ex.alloc_bytes((size_t)B * 77 * u->cdim * 2);
    if (!r.ctx16) return ODISE_ERR_NOMEM;
    ODISE_TRY(odise_hip_cast_f32_to_f16(ctx, context, r.ctx16, (size_t)B * 77 * u->cdim));
    r.ctxK.assign(u->n_ctx_slots, nullptr);
    r.ctxVt.assign(u->n_ctx_slots, nullptr);
    auto prep_ctx = [&](const STBlockW& s) -> int {
        const int C = s.c;
        f16* k = (f16*)ex.alloc_bytes((size_t)B * 77 * C * 2);
        f16* vt = (f16*)ex.alloc_bytes((size_t)B * C * 80 * 2);
        if (!k || !vt) return ODISE_ERR_NOMEM;
        ODISE_TRY(ex.linear(r.ctx16, (int64_t)B * 77, s.k2, k));
        odise_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.M = C; d.N = 77; d.K = u->cdim;
        d.A = s.v2.w; d.lda = u->cdim; d.strideA = 0;
        d.W = r.ctx16; d.ldw = u->cdim; d.strideW = 77 * (int64_t)u->cdim;
        d.C = vt; d.ldc = 80; d.strideC = (int64_t)C * 80; d.c_dtype = ODISE_F16;
        d.alpha = 1.f; d.batch = B;
        ODISE_TRY(ex.gemm(d));
        r.ctxK[s.ctx_slot] = k;
        r.ctxVt[s.ctx_slot] = vt;
        return ODISE_OK;
    };
    for (auto& b : u->in_blocks)
        if (b.has_st) ODISE_TRY(prep_ctx(b.st));
    ODISE_TRY(prep_ctx(u->mid_st));
    for (auto& b : u->out_blocks)
        if (b.has_st) ODISE_TRY(prep_ctx(b.st));

    // ---- input blocks --------------------------------------------------------------------------------------
    Act x0;
    if (x_nhwc) {
        x0.p = const_cast<f16*>(x_nhwc); x0.n = B; x0.h = h; x0.w = w; x0.c = 8;
    } else {
        ODISE_TRY(ex.alloc(x0, B, h, w, 8));
        ODISE_TRY(odise_hip_nchw_f32_to_nhwc_f16(ctx, x_t, x0.p, B, 4, h, w, 8));
    }
    std::vector<Act> hs;
    Act cur = x0;
    for (auto& b : u->in_blocks) {
        Act nxt;
        if (b.is_conv_in) {
            ODISE_TRY(ex.conv(cur, b.conv, nxt, 1, 1));
        } else if (b.has_down) {
            ODISE_TRY(ex.conv(cur, b.conv, nxt, 2, 1));
        } else {
            ODISE_TRY(run_res(r, b.res, cur, nxt));
            if (b.has_st) {
                Act o2;
                ODISE_TRY(run_st(r, b.st, nxt, o2));
                nxt = o2;
            }
        }
        hs.push_back(nxt);
        cur = nxt;
    }
    // ---- middle ------------------------------------------------------------------------------------------
    {
        Act a, b2, c;
        ODISE_TRY(run_res(r, u->mid_r1, cur, a));
        ODISE_TRY(run_st(r, u->mid_st, a, b2));
        ODISE_TRY(run_res(r, u->mid_r2, b2, c));
        cur = c;
    }
    // ---- output blocks 0..10 + the tap-only concat of block 11 -----------------------------------------------
    int tap_i = 0;
    for (int i = 0; i < 12; ++i) {
        const Act skip = hs.back();
        hs.pop_back();
        Act cat;
        ODISE_TRY(ex.alloc(cat, cur.n, cur.h, cur.w, cur.c + skip.c));
        ODISE_TRY(odise_hip_concat_channels(ctx, cur.p, skip.p, cat.p, (size_t)cur.pixels(), cur.c, skip.c));
        if (i == 2 || i == 5 || i == 8 || i == 11) u->taps[tap_i++] = cat;
        if (i == 11) break;  // block 11 and `out` are dead code in the reference (ldm.py:491)
        const UBlock& b = u->out_blocks[i];
        Act nxt;
        ODISE_TRY(run_res(r, b.res, cat, nxt));
        if (b.has_st) {
            Act o2;
            ODISE_TRY(run_st(r, b.st, nxt, o2));
            nxt = o2;
        }
        if (b.has_up) {
            Act o3;
            ODISE_TRY(ex.conv(nxt, b.conv, o3, 1, 1, true));
            nxt = o3;
        }
        cur = nxt;
    }
    u->last_macs = ms->macs;
    return ODISE_OK;
}

// timestep embedding input: cat[cos(t f), sin(t f)], f_i = exp(-ln(10000) i / half)  (ldm timestep_embedding); cached per t
int unet_prepare_timestep(odise_hip_ctx* ctx, ModelStore* ms, UNetModel* u, int B, int t) {
    if (u->cached_t != t || u->cached_B < B) {
        std::vector<f16> te((size_t)B * u->mc);
        const int half = u->mc / 2;
        for (int i = 0; i < half; ++i) {
            const double f = exp(-log(10000.0) * i / (double)half);
            for (int b = 0; b < B; ++b) {
                te[(size_t)b * u->mc + i] = (f16)(float)cos((double)t * f);
                te[(size_t)b * u->mc + half + i] = (f16)(float)sin((double)t * f);
            }
        }
        ODISE_CHECK_HIP(hipDeviceSynchronize());   // either lane may still read the previous table
        if (u->temb_in) ODISE_CHECK_HIP(hipFree(u->temb_in));
        u->temb_in = nullptr;
        ODISE_CHECK_HIP(hipMalloc((void**)&u->temb_in, te.size() * 2));
        ODISE_CHECK_HIP(hipMemcpy(u->temb_in, te.data(), te.size() * 2, hipMemcpyHostToDevice));
        u->cached_t = t;
        u->cached_B = B;
        if (u->graph_exec) { (void)hipGraphExecDestroy(u->graph_exec); u->graph_exec = nullptr; }
    }
    return ODISE_OK;
}

static int unet_forward(odise_hip_ctx* ctx, const float* x_t, const float* context, const float* cond_emb, int B, int h, int w, int t) {
    ModelStore* ms = store_of(ctx);
    UNetModel* u = ms->unet;
    if (!u || !u->built) {
        set_error("unet_features: call odise_hip_unet_build first");
        return ODISE_ERR_STATE;
    }
    ODISE_REQUIRE(B >= 1 && h >= 8 && w >= 8 && h % 8 == 0 && w % 8 == 0, "unet_features: latent %dx%d (batch %d) must be multiples of 8", h, w, B);
    ODISE_REQUIRE(x_t && context, "unet_features: null device pointer");
    ODISE_TRY(unet_prepare_timestep(ctx, ms, u, B, t));
    // arena: ~0.6 GB per 64x64-latent crop is ample (peak is tracked; see odise_hip_unet_last_macs)
    const size_t per_crop = (size_t)640 << 20;
    const double scale = ((double)h * w) / (64.0 * 64.0);
    ODISE_TRY(ensure_arena(ctx, ms, (size_t)(per_crop * (scale < 0.25 ? 0.25 : scale)) * B + ((size_t)64 << 20)));

    if (!u->use_graph) return unet_launch(ctx, ms, u, x_t, nullptr, context, cond_emb, B, h, w, true);

    const bool same = u->graph_exec && u->graph_B == B && u->graph_h == h && u->graph_w == w && u->graph_x == x_t &&
                      u->graph_ctx == context && u->graph_ce == cond_emb && u->graph_arena == ms->arena.base;
    if (!same) {
        if (u->graph_exec) { (void)hipGraphExecDestroy(u->graph_exec); u->graph_exec = nullptr; }
        hipGraph_t graph = nullptr;
        ODISE_CHECK_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        const int rc = unet_launch(ctx, ms, u, x_t, nullptr, context, cond_emb, B, h, w, true);
        const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
        if (rc != ODISE_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        ODISE_CHECK_HIP(e);
        ODISE_CHECK_HIP(hipGraphInstantiate(&u->graph_exec, graph, nullptr, nullptr, 0));
        ODISE_CHECK_HIP(hipGraphDestroy(graph));
        u->graph_B = B; u->graph_h = h; u->graph_w = w;
        u->graph_x = x_t; u->graph_ctx = context; u->graph_ce = cond_emb; u->graph_arena = ms->arena.base;
    }
    ODISE_CHECK_HIP(hipGraphLaunch(u->graph_exec, ctx->stream));
    return ODISE_OK;
}

}  // namespace odise

using namespace odise;

extern "C" int odise_hip_unet_build(odise_hip_ctx* ctx) {
    ODISE_REQUIRE(ctx, "unet_build: null context");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));
    return unet_build(ctx, "");
}

extern "C" int odise_hip_unet_use_graph(odise_hip_ctx* ctx, int enable) {
    ODISE_REQUIRE(ctx, "unet_use_graph: null context");
    ModelStore* ms = store_of(ctx);
    if (!ms->unet) {
        set_error("unet_use_graph: call odise_hip_unet_build first");
        return ODISE_ERR_STATE;
    }
    ms->unet->use_graph = enable != 0;
    return ODISE_OK;
}

extern "C" int odise_hip_unet_features_nhwc(odise_hip_ctx* ctx, const float* x_t, const float* context, const float* cond_emb, int B,
                                            int h, int w, int t, void** taps4) {
    ODISE_REQUIRE(ctx, "unet_features: null context");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    ODISE_TRY(unet_forward(ctx, x_t, context, cond_emb, B, h, w, t));
    if (taps4) {
        ModelStore* ms = store_of(ctx);
        for (int i = 0; i < 4; ++i) taps4[i] = ms->unet->taps[i].p;
    }
    return ODISE_OK;
}

extern "C" int odise_hip_unet_features(odise_hip_ctx* ctx, const float* x_t, const float* context, const float* cond_emb, int B, int h,
                                       int w, int t, float* tap_u2, float* tap_u5, float* tap_u8, float* tap_u11) {
    ODISE_REQUIRE(ctx, "unet_features: null context");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // the caller may be a new host thread, or hold another device current
    ODISE_TRY(unet_forward(ctx, x_t, context, cond_emb, B, h, w, t));
    ModelStore* ms = store_of(ctx);
    float* outs[4] = {tap_u2, tap_u5, tap_u8, tap_u11};
    for (int i = 0; i < 4; ++i) {
        if (!outs[i]) continue;
        const Act& a = ms->unet->taps[i];
        ODISE_TRY(odise_hip_nhwc_f16_to_nchw_f32(ctx, a.p, outs[i], a.n, a.c, a.h, a.w));
    }
    return ODISE_OK;
}

extern "C" int odise_hip_unet_last_macs(odise_hip_ctx* ctx, double* macs) {
    ODISE_REQUIRE(ctx && macs, "unet_last_macs: null argument");
    ModelStore* ms = store_of(ctx);
    *macs = ms->unet ? ms->unet->last_macs : 0.0;
    return ODISE_OK;
}
