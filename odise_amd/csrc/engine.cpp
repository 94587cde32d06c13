// engine.cpp — weight store, packer and launch helpers (see engine.h).
#include <stdlib.h>

#include "engine.h"

#include <string.h>

namespace odise {

void unet_destroy(ModelStore* ms);       // unet.cpp
void extractor_destroy(ModelStore* ms);  // extractor.cpp
void maskgen_destroy(ModelStore* ms);    // maskgen.cpp
void classify_destroy(ModelStore* ms);   // classify.cpp

ModelStore* store_of(odise_hip_ctx* ctx) {
    if (!ctx->models) ctx->models = new ModelStore();
    return (ModelStore*)ctx->models;
}

void free_allocs(std::vector<void*>& owned) {
    if (owned.empty()) return;
    (void)hipDeviceSynchronize();   // a kernel of either lane may still read these weights
    for (void* p : owned) (void)hipFree(p);
    owned.clear();
}

void models_destroy(odise_hip_ctx* ctx) {
    if (!ctx->models) return;
    ModelStore* ms = (ModelStore*)ctx->models;
    unet_destroy(ms);
    extractor_destroy(ms);
    maskgen_destroy(ms);
    classify_destroy(ms);
    for (void* p : ms->dev_allocs) (void)hipFree(p);
    if (ms->arena.base) (void)hipFree(ms->arena.base);
    if (ms->arena2.base) (void)hipFree(ms->arena2.base);
    if (ms->mclip.buf) (void)hipFree(ms->mclip.buf);
    for (Arena& a : ms->pf.arena)
        if (a.base) (void)hipFree(a.base);
    delete ms;
    ctx->models = nullptr;
}

int ensure_prefetch_lane(odise_hip_ctx* ctx, ModelStore* ms, int slot, size_t arena_bytes) {
    if (!ctx->stream3) {
        // all or nothing: the lane's four resources are created into locals and committed together, so a failure half way (the workspace is
        // ws_bytes of HBM) leaves no half-initialised lane behind for the next call to mistake for a complete one
        int lo = 0, hi = 0;   // (least, greatest) priority
        ODISE_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        const int e8 = ctx->prefetch_cu_eighths;
        hipStream_t st = nullptr;
        void* ws = nullptr;
        hipEvent_t go = nullptr, done = nullptr;
        hipError_t err;
        if (e8 >= 1 && e8 <= 7) {
            // a CU-masked stream: e8 of every 8 compute units (bit i of the mask = CU i; the same pattern in every byte spreads over the XCDs /
            // shader engines whatever the enumeration), so the batch in progress always finds CUs no prefetch workgroup occupies.  HIP offers no
            // priority or flags with a CU mask: this stream has NORMAL priority and default flags (it synchronises with the legacy NULL stream)
            uint32_t mask[8];
            const uint32_t byte = (1u << e8) - 1u;
            for (uint32_t& m : mask) m = byte * 0x01010101u;
            err = hipExtStreamCreateWithCUMask(&st, 8, mask);
        } else {
            err = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo);   // never ahead of the batch in progress
        }
        if (err == hipSuccess) err = hipMalloc(&ws, ctx->ws_bytes);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&go, hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        if (err != hipSuccess) {
            if (done) (void)hipEventDestroy(done);
            if (go) (void)hipEventDestroy(go);
            if (ws) (void)hipFree(ws);
            if (st) (void)hipStreamDestroy(st);
            set_error("prefetch lane: %s", hipGetErrorString(err));
            return ODISE_ERR_HIP;
        }
        ctx->stream3 = st; ctx->ws3 = ws; ctx->ev_pf_go = go; ctx->ev_pf_done = done;
    }
    Arena& a = ms->pf.arena[slot];
    if (a.cap < arena_bytes) {
        ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream3));
        ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (a.base) ODISE_CHECK_HIP(hipFree(a.base));
        a = Arena();
        ODISE_CHECK_HIP(hipMalloc((void**)&a.base, arena_bytes));
        a.cap = arena_bytes;
    }
    return ODISE_OK;
}

int ensure_lane2(odise_hip_ctx* ctx, ModelStore* ms, size_t arena_bytes) {
    if (!ctx->stream2) {
        int lo = 0, hi = 0;   // (least, greatest) priority: numerically, greatest <= least
        ODISE_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        // The CLIP -> UNet lane is the step's critical path (the UNet starts when the encoder's latent exists and ~540 dependent launches
        // follow); its workgroups go ahead of the VAE's chip-filling convolutions whenever both are ready: -1.7 ms per step, same box
        // (profiles/r03_lane_scheduling.txt)
        int prio = hi;
#ifdef ODISE_TOOLS
        if (getenv("ODISE_LANE2_NORMAL_PRIORITY")) prio = lo;   // A/B
#endif
        ODISE_CHECK_HIP(hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio));
        ODISE_CHECK_HIP(hipMalloc(&ctx->ws2, ctx->ws_bytes));
        ODISE_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        ODISE_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_mid, hipEventDisableTiming));
        ODISE_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        ODISE_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_mclip, hipEventDisableTiming));
    }
    if (ms->arena2.cap < arena_bytes) {
        ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream2));
        if (ms->arena2.base) ODISE_CHECK_HIP(hipFree(ms->arena2.base));
        ms->arena2 = Arena();
        ODISE_CHECK_HIP(hipMalloc((void**)&ms->arena2.base, arena_bytes));
        ms->arena2.cap = arena_bytes;
    }
    return ODISE_OK;
}

// ---------------------------------------------------------------------------------------------------------------
const HostTensor* Packer::find(const std::string& key) {
    auto it = ms->host.find(prefix + key);
    if (it == ms->host.end()) {
        if (missing.empty()) missing = prefix + key;
        return nullptr;
    }
    return &it->second;
}

int Packer::upload(const void* host, size_t bytes, void** dev) {
    ODISE_CHECK_HIP(hipMalloc(dev, bytes ? bytes : 16));
    ms->track(*dev);
    ODISE_CHECK_HIP(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return ODISE_OK;
}

int Packer::vec_f32(const std::string& key, float** out, int64_t expect) {
    const HostTensor* t = find(key);
    if (!t) {
        set_error("weights: missing key '%s'", (prefix + key).c_str());
        return ODISE_ERR_STATE;
    }
    if (t->numel() != expect) {
        set_error("weights: '%s' has %lld elements, expected %lld", (prefix + key).c_str(), (long long)t->numel(), (long long)expect);
        return ODISE_ERR_STATE;
    }
    return upload(t->data.data(), t->data.size() * sizeof(float), (void**)out);
}

int Packer::conv(const std::string& key, ConvW& out, bool bias) {
    const HostTensor* t = find(key + ".weight");
    if (!t || t->shape.size() != 4) {
        set_error("weights: missing or non-4D conv weight '%s.weight'", (prefix + key).c_str());
        return ODISE_ERR_STATE;
    }
    const int O = (int)t->shape[0], I = (int)t->shape[1], KH = (int)t->shape[2], KW = (int)t->shape[3];
    if (KH != KW) {
        set_error("weights: non-square conv kernel '%s'", (prefix + key).c_str());
        return ODISE_ERR_STATE;
    }
    const int Ip = (int)round_up(I, 8);
    std::vector<f16> packed((size_t)O * KH * KW * Ip, (f16)0.f);
    for (int o = 0; o < O; ++o)
        for (int i = 0; i < I; ++i)
            for (int y = 0; y < KH; ++y)
                for (int x = 0; x < KW; ++x)
                    packed[(((size_t)o * KH + y) * KW + x) * Ip + i] = (f16)t->data[(((size_t)o * I + i) * KH + y) * KW + x];
    out.cin = I; out.cin_pad = Ip; out.cout = O; out.k = KH;
    ODISE_TRY(upload(packed.data(), packed.size() * sizeof(f16), (void**)&out.w));
    out.b = nullptr;
    if (bias) ODISE_TRY(vec_f32(key + ".bias", &out.b, O));
    return ODISE_OK;
}

int Packer::linear(const std::string& key, LinW& out, bool bias) {
    const HostTensor* t = find(key + ".weight");
    if (!t || t->shape.size() < 2) {
        set_error("weights: missing linear weight '%s.weight'", (prefix + key).c_str());
        return ODISE_ERR_STATE;
    }
    const int O = (int)t->shape[0], I = (int)t->shape[1];
    if (t->numel() != (int64_t)O * I || I % 8 != 0) {
        set_error("weights: '%s.weight' is not a [O,I] / [O,I,1,1] matrix with I %% 8 == 0", (prefix + key).c_str());
        return ODISE_ERR_STATE;
    }
    std::vector<f16> packed((size_t)O * I);
    for (size_t i = 0; i < packed.size(); ++i) packed[i] = (f16)t->data[i];
    out.in = I; out.out = O;
    ODISE_TRY(upload(packed.data(), packed.size() * sizeof(f16), (void**)&out.w));
    out.b = nullptr;
    if (bias) ODISE_TRY(vec_f32(key + ".bias", &out.b, O));
    return ODISE_OK;
}

int Packer::norm(const std::string& key, NormW& out) {
    const HostTensor* t = find(key + ".weight");
    if (!t) {
        set_error("weights: missing norm weight '%s.weight'", (prefix + key).c_str());
        return ODISE_ERR_STATE;
    }
    out.c = (int)t->numel();
    ODISE_TRY(vec_f32(key + ".weight", &out.g, out.c));
    ODISE_TRY(vec_f32(key + ".bias", &out.b, out.c));
    return ODISE_OK;
}

// ---------------------------------------------------------------------------------------------------------------
void* Exec::alloc_bytes(size_t bytes) {
    void* p = ms->arena.alloc(bytes);
    if (!p) set_error("arena exhausted: need %zu more bytes (capacity %zu, used %zu)", bytes, ms->arena.cap, ms->arena.off);
    return p;
}

int Exec::alloc(Act& a, int n, int h, int w, int c) {
    a.n = n; a.h = h; a.w = w; a.c = c;
    a.p = (f16*)alloc_bytes((size_t)a.elems() * sizeof(f16));
    return a.p ? ODISE_OK : ODISE_ERR_NOMEM;
}

int Exec::gemm(const odise_gemm_desc& d) {
    ms->macs += (double)d.M * d.N * d.K * (d.batch > 1 ? d.batch : 1);
    return odise_hip_gemm(ctx, &d);
}

int Exec::attention(const odise_attn_desc& d) {
    ms->macs += 2.0 * d.B * d.H * (double)d.Lq * d.Lk * d.D;
    return odise_hip_attention(ctx, &d);
}

int Exec::conv(const Act& x, const ConvW& w, Act& y, int stride, int pad, bool upsample, const Act* residual,
               const float* per_image_add, int64_t pia_ld, int act, int pad_t, int pad_l, int oh, int ow) {
    if (x.c != w.cin_pad) {
        set_error("conv: input has %d channels, weight expects %d", x.c, w.cin_pad);
        return ODISE_ERR_ARG;
    }
    if (pad < 0) pad = w.k / 2;
    if (pad_t < 0) pad_t = pad;
    if (pad_l < 0) pad_l = pad;
    const int hin = upsample ? 2 * x.h : x.h, win = upsample ? 2 * x.w : x.w;
    if (oh < 0) oh = (hin + 2 * pad - w.k) / stride + 1;
    if (ow < 0) ow = (win + 2 * pad - w.k) / stride + 1;
    if (!y.p) ODISE_TRY(alloc(y, x.n, oh, ow, w.cout));
    odise_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.N = x.n; d.H = x.h; d.W = x.w; d.Cin = x.c;
    d.Cout = w.cout; d.KH = w.k; d.KW = w.k; d.stride = stride; d.pad_t = pad_t; d.pad_l = pad_l; d.OH = oh; d.OW = ow;
    d.upsample2x = upsample ? 1 : 0;
    d.X = x.p; d.Wt = w.w; d.Y = y.p; d.y_dtype = ODISE_F16;
    d.bias = w.b;
    d.residual = residual ? residual->p : nullptr;
    d.per_image_add = per_image_add;
    d.per_image_add_ld = pia_ld;
    d.act = act;
    ms->macs += (double)x.n * oh * ow * w.cout * w.k * w.k * w.cin;
    y.gn_blocks = 0;
#ifdef ODISE_TOOLS
    if (y.gn_part && getenv("ODISE_NO_GN_FUSION")) return odise_hip_conv2d(ctx, &d);   // A/B: separate GroupNorm statistics pass
#endif
    if (y.gn_part) return conv_forced(ctx, &d, -1, 0, y.gn_part, &y.gn_blocks);
    return odise_hip_conv2d(ctx, &d);
}

int Exec::alloc_gn_stats(Act& a) {
    // at most one row block per 64 output pixels (the smallest tile) -> [n][blocks][c][2] fp32
    const int64_t blocks = ceil_div((int64_t)a.h * a.w, 64);
    a.gn_part = (float*)alloc_bytes((size_t)a.n * blocks * a.c * 2 * sizeof(float));
    a.gn_blocks = 0;
    return a.gn_part ? ODISE_OK : ODISE_ERR_NOMEM;
}

int Exec::linear(const f16* x, int64_t M, const LinW& w, f16* y, int act, const f16* residual, bool geglu) {
    odise_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.M = (int)M; d.N = w.out; d.K = w.in;
    d.A = x; d.lda = w.in;
    d.W = w.w; d.ldw = w.in;
    d.C = y; d.ldc = geglu ? w.out / 2 : w.out; d.c_dtype = ODISE_F16;
    d.bias_n = w.b;
    d.residual = residual; d.ldr = d.ldc;
    d.act = act; d.geglu = geglu ? 1 : 0; d.alpha = 1.f; d.batch = 1;
    return gemm(d);
}

int Exec::group_norm(const Act& x, const NormW& w, Act& y, float eps, int act) {
    if (!y.p) ODISE_TRY(alloc(y, x.n, x.h, x.w, x.c));
    if (x.gn_part && x.gn_blocks > 0)  // the conv that wrote x already reduced it per channel and row block
        return group_norm_from_colpart(ctx, x.p, y.p, w.g, w.b, x.n, x.h * x.w, x.c, 32, eps, act, x.gn_part, x.gn_blocks);
    return odise_hip_group_norm(ctx, x.p, y.p, w.g, w.b, x.n, x.h * x.w, x.c, 32, eps, act);
}

int Exec::layer_norm(const f16* x, f16* y, int64_t rows, const NormW& w, float eps) {
    return odise_hip_layer_norm(ctx, x, y, w.g, w.b, (int)rows, w.c, eps);
}

}  // namespace odise

using namespace odise;

extern "C" int odise_hip_load_weight(odise_hip_ctx* ctx, const char* name, const float* host_data, const int64_t* shape, int ndim) {
    ODISE_REQUIRE(ctx && name && host_data && (ndim == 0 || shape), "load_weight: null argument");
    ODISE_REQUIRE(ndim >= 0 && ndim <= 4, "load_weight: rank %d not in [0,4]", ndim);
    ModelStore* ms = store_of(ctx);
    HostTensor t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        ODISE_REQUIRE(shape[i] >= 0, "load_weight: negative dim");
        t.shape.push_back(shape[i]);
        n *= shape[i];
    }
    t.data.assign(host_data, host_data + n);
    ms->host[std::string(name)] = std::move(t);
    return ODISE_OK;
}

extern "C" int odise_hip_clear_host_weights(odise_hip_ctx* ctx) {
    ODISE_REQUIRE(ctx, "clear_host_weights: null context");
    store_of(ctx)->host.clear();
    return ODISE_OK;
}
