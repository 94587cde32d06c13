// api.cpp — context, error reporting and memory/stream plumbing of libodise_hip.so.
#include <stdarg.h>
#include <time.h>
#include <string.h>

#include <mutex>
#include <vector>

#include <algorithm>
#include "common.h"
#include "../../include/odise_hip_tools.h"

namespace odise {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void models_destroy(odise_hip_ctx* ctx);  // unet.cpp
}  // namespace odise

using namespace odise;

extern "C" const char* odise_hip_last_error(void) { return g_err; }
extern "C" int odise_hip_version(void) { return 100; }

extern "C" int odise_hip_create(int device, odise_hip_ctx** out) {
    ODISE_REQUIRE(out != nullptr, "create: null out pointer");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        set_error("create: no HIP device available (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
        return ODISE_ERR_HIP;
    }
    ODISE_REQUIRE(device >= 0 && device < count, "create: device %d out of range [0,%d)", device, count);
    ODISE_CHECK_HIP(hipSetDevice(device));
    odise_hip_ctx* c = new odise_hip_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    ODISE_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        int optin = 0;
        if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) optin = 0;
        const size_t m = std::max<size_t>({(size_t)optin, prop.sharedMemPerBlock, prop.maxSharedMemoryPerMultiProcessor, (size_t)65536});
        c->max_lds_optin = (int)std::min<size_t>(m, (size_t)1 << 30);
    }
    ODISE_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    c->ws_bytes = (size_t)256 << 20;
    ODISE_CHECK_HIP(hipMalloc(&c->ws, c->ws_bytes));
    ODISE_CHECK_HIP(hipMalloc(&c->zeros, 256));
    ODISE_CHECK_HIP(hipMemset(c->zeros, 0, 256));
    ODISE_CHECK_HIP(hipEventCreate(&c->ev0));
    ODISE_CHECK_HIP(hipEventCreate(&c->ev1));
    *out = c;
    return ODISE_OK;
}

extern "C" int odise_hip_destroy(odise_hip_ctx* ctx) {
    if (!ctx) return ODISE_OK;
    (void)hipSetDevice(ctx->device);  // teardown: nothing useful can be done about a failing call here
    (void)hipStreamSynchronize(ctx->stream);
    models_destroy(ctx);
    odise::jpeg_release(ctx);
    odise::comm_release(ctx);
    odise::probe_release(ctx);
    odise::launch_log_release(ctx);
    odise::stage_log_release(ctx);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->ws2) (void)hipFree(ctx->ws2);
    if (ctx->ws3) (void)hipFree(ctx->ws3);
    if (ctx->ev_pf_go) (void)hipEventDestroy(ctx->ev_pf_go);
    if (ctx->ev_pf_done) (void)hipEventDestroy(ctx->ev_pf_done);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_mid) (void)hipEventDestroy(ctx->ev_mid);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_mclip) (void)hipEventDestroy(ctx->ev_mclip);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->zeros) (void)hipFree(ctx->zeros);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return ODISE_OK;
}

extern "C" int odise_hip_set_stream(odise_hip_ctx* ctx, void* hip_stream) {
    ODISE_REQUIRE(ctx, "set_stream: null context");
    ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (hip_stream == nullptr) {
        if (!ctx->own_stream) {
            ODISE_CHECK_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
            ctx->own_stream = true;
        }
        return ODISE_OK;
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return ODISE_OK;
}

extern "C" int odise_hip_malloc(odise_hip_ctx* ctx, size_t bytes, void** dptr) {
    ODISE_REQUIRE(ctx && dptr, "malloc: null argument");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));
    *dptr = nullptr;
    if (bytes == 0) bytes = 16;
    ODISE_CHECK_HIP(hipMalloc(dptr, bytes));
    return ODISE_OK;
}
extern "C" int odise_hip_free(odise_hip_ctx* ctx, void* dptr) {
    ODISE_REQUIRE(ctx, "free: null context");
    if (!dptr) return ODISE_OK;
    ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ODISE_CHECK_HIP(hipFree(dptr));
    return ODISE_OK;
}
extern "C" int odise_hip_memcpy_h2d(odise_hip_ctx* ctx, void* dst, const void* src, size_t bytes) {
    ODISE_REQUIRE(ctx && (bytes == 0 || (dst && src)), "memcpy_h2d: null argument");
    if (bytes == 0) return ODISE_OK;
    ODISE_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return ODISE_OK;
}
extern "C" int odise_hip_memcpy_d2h(odise_hip_ctx* ctx, void* dst, const void* src, size_t bytes) {
    ODISE_REQUIRE(ctx && (bytes == 0 || (dst && src)), "memcpy_d2h: null argument");
    if (bytes == 0) return ODISE_OK;
    ODISE_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return ODISE_OK;
}
extern "C" int odise_hip_memset(odise_hip_ctx* ctx, void* dst, int value, size_t bytes) {
    ODISE_REQUIRE(ctx && (bytes == 0 || dst), "memset: null argument");
    if (bytes == 0) return ODISE_OK;
    ODISE_CHECK_HIP(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return ODISE_OK;
}
extern "C" int odise_hip_sync(odise_hip_ctx* ctx) {
    ODISE_REQUIRE(ctx, "sync: null context");
    ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return ODISE_OK;
}
extern "C" int odise_hip_timer_start(odise_hip_ctx* ctx) {
    ODISE_REQUIRE(ctx, "timer_start: null context");
    ODISE_CHECK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return ODISE_OK;
}
extern "C" int odise_hip_timer_stop(odise_hip_ctx* ctx, float* ms) {
    ODISE_REQUIRE(ctx && ms, "timer_stop: null argument");
    ODISE_CHECK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    ODISE_CHECK_HIP(hipEventSynchronize(ctx->ev1));
    ODISE_CHECK_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return ODISE_OK;
}
extern "C" int odise_hip_device_info(odise_hip_ctx* ctx, char* name_buf, int buf_len, int* cu_count, size_t* hbm_bytes) {
    ODISE_REQUIRE(ctx, "device_info: null context");
    hipDeviceProp_t prop;
    ODISE_CHECK_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (name_buf && buf_len > 0) {
        snprintf(name_buf, buf_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return ODISE_OK;
}
extern "C" int odise_hip_set_option(odise_hip_ctx* ctx, int option, int64_t value) {
    ODISE_REQUIRE(ctx, "set_option: null context");
    switch (option) {
        case ODISE_OPT_CLIP_LN_FOLD:
            ODISE_REQUIRE(value >= 0 && value <= 2, "set_option: CLIP_LN_FOLD takes 0 (by token count), 1 (always) or 2 (never)");
            ctx->clip_ln_fold = (int)value;
            return ODISE_OK;
        case ODISE_OPT_ATTN_KV_RESIDENT:
            ODISE_REQUIRE(value >= 0 && value <= 7 && !(value & 1), "set_option: ATTN_KV_RESIDENT takes 0 (the library's rules), 2 (never the K/V-resident kernel), 4 (never the pipelined self-attention kernel) or 6");
            ctx->attn_kv_resident = (int)value;
            return ODISE_OK;
        case ODISE_OPT_VAE_CHUNK_BYTES:
            ODISE_REQUIRE(value >= 0, "set_option: VAE_CHUNK_BYTES must be >= 0");
            ctx->vae_chunk_bytes = value;
            return ODISE_OK;
        case ODISE_OPT_PREFETCH_CU_EIGHTHS:
            ODISE_REQUIRE(value >= 0 && value <= 8, "set_option: PREFETCH_CU_EIGHTHS takes 0..8");
            ODISE_REQUIRE(!ctx->stream3, "set_option: PREFETCH_CU_EIGHTHS must be set before the first prefetch (the stream exists already)");
            ctx->prefetch_cu_eighths = (int)value;
            return ODISE_OK;
        case ODISE_OPT_PREFETCH_START:
            ODISE_REQUIRE(value == 0 || value == 1, "set_option: PREFETCH_START takes 0 (behind the VAE lane) or 1 (behind the backbone)");
            ctx->prefetch_start = (int)value;
            return ODISE_OK;
        case ODISE_OPT_MASKCLIP_PASSES:
            ODISE_REQUIRE(value >= 0 && value <= 3, "set_option: MASKCLIP_PASSES takes 0 (image tokens in the crops' tower), 1 (two passes in place), 2 (one pass) or 3 (image tokens as a tower of their own on the second lane)");
            ctx->maskclip_passes = (int)value;
            return ODISE_OK;
        default:
            set_error("set_option: unknown option %d", option);
            return ODISE_ERR_ARG;
    }
}
extern "C" int odise_hip_get_option(odise_hip_ctx* ctx, int option, int64_t* value) {
    ODISE_REQUIRE(ctx && value, "get_option: null argument");
    switch (option) {
        case ODISE_OPT_CLIP_LN_FOLD: *value = ctx->clip_ln_fold; return ODISE_OK;
        case ODISE_OPT_VAE_CHUNK_BYTES: *value = ctx->vae_chunk_bytes; return ODISE_OK;
        case ODISE_OPT_ATTN_KV_RESIDENT: *value = ctx->attn_kv_resident; return ODISE_OK;
        case ODISE_OPT_PREFETCH_CU_EIGHTHS: *value = ctx->prefetch_cu_eighths; return ODISE_OK;
        case ODISE_OPT_PREFETCH_START: *value = ctx->prefetch_start; return ODISE_OK;
        case ODISE_OPT_MASKCLIP_PASSES: *value = ctx->maskclip_passes; return ODISE_OK;
        default:
            set_error("get_option: unknown option %d", option);
            return ODISE_ERR_ARG;
    }
}

namespace odise {
struct StageLog {
    std::vector<const char*> names;
    std::vector<hipEvent_t> events;   // pooled: events[i] belongs to names[i] for i < names.size()
    std::vector<double> host_us;
};
static double host_now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
void stage_mark(odise_hip_ctx* ctx, const char* name) {
    StageLog* s = (StageLog*)ctx->stages;
    if (!s) return;
    const size_t i = s->names.size();
    if (i >= s->events.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return;
        s->events.push_back(e);
    }
    if (hipEventRecord(s->events[i], ctx->stream) != hipSuccess) return;
    s->names.push_back(name);
    s->host_us.push_back(host_now_us());
}
void stage_log_release(odise_hip_ctx* ctx) {
    StageLog* s = (StageLog*)ctx->stages;
    if (!s) return;
    for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
    delete s;
    ctx->stages = nullptr;
}
void launch_log_push(odise_hip_ctx* ctx, const LaunchRec& r) { ((std::vector<LaunchRec>*)ctx->launch_log)->push_back(r); }
void launch_log_release(odise_hip_ctx* ctx) {
    delete (std::vector<LaunchRec>*)ctx->launch_log;
    ctx->launch_log = nullptr;
}
void probe_release(odise_hip_ctx* ctx) {
    LaunchProbe* p = (LaunchProbe*)ctx->probe;
    if (!p) return;
    for (int i = 0; i < 2 * p->cap; ++i) (void)hipEventDestroy(p->ev[i]);
    delete[] p->ev;
    delete p;
    ctx->probe = nullptr;
}
}  // namespace odise
extern "C" int odise_hip_probe_arm(odise_hip_ctx* ctx, int conv, int M, int N, int K, int max_launches) {
    ODISE_REQUIRE(ctx && max_launches >= 1 && max_launches <= 4096, "probe_arm: 1 <= max_launches <= 4096");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));
    LaunchProbe* p = (LaunchProbe*)ctx->probe;
    if (p && p->cap < max_launches) { probe_release(ctx); p = nullptr; }
    if (!p) {
        p = new LaunchProbe();
        p->cap = max_launches;
        p->ev = new hipEvent_t[2 * (size_t)max_launches]();
        ctx->probe = p;
        for (int i = 0; i < 2 * max_launches; ++i) ODISE_CHECK_HIP(hipEventCreate(&p->ev[i]));
    }
    p->conv = conv; p->M = M; p->N = N; p->K = K;
    p->n = 0;
    p->armed = true;
    return ODISE_OK;
}
extern "C" int odise_hip_probe_read(odise_hip_ctx* ctx, float* us_out, int cap, int* n_launches) {
    ODISE_REQUIRE(ctx && n_launches, "probe_read: null argument");
    LaunchProbe* p = (LaunchProbe*)ctx->probe;
    if (!p) {
        set_error("probe_read: no probe armed on this context");
        return ODISE_ERR_STATE;
    }
    p->armed = false;
    *n_launches = p->n;
    for (int i = 0; i < p->n && i < cap; ++i) {
        float ms = 0.f;
        ODISE_CHECK_HIP(hipEventSynchronize(p->ev[2 * i + 1]));
        ODISE_CHECK_HIP(hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]));
        if (us_out) us_out[i] = ms * 1000.f;
    }
    return ODISE_OK;
}

extern "C" int odise_hip_stage_timeline(odise_hip_ctx* ctx, int on) {
    ODISE_REQUIRE(ctx, "stage_timeline: null context");
    if (!on) { stage_log_release(ctx); return ODISE_OK; }
    StageLog* s = (StageLog*)ctx->stages;
    if (!s) ctx->stages = s = new StageLog();
    s->names.clear();       // start / restart: the events are reused
    s->host_us.clear();
    return ODISE_OK;
}
extern "C" int odise_hip_stage_timeline_read(odise_hip_ctx* ctx, char* names, int names_cap, float* gpu_ms, double* host_ms, int cap, int* n) {
    ODISE_REQUIRE(ctx && n, "stage_timeline_read: null argument");
    StageLog* s = (StageLog*)ctx->stages;
    *n = s ? (int)s->names.size() : 0;
    if (!s || s->names.empty()) return ODISE_OK;
    ODISE_CHECK_HIP(hipDeviceSynchronize());
    int off = 0;
    for (int i = 0; i < *n && i < cap; ++i) {
        float ms = 0.f;
        ODISE_CHECK_HIP(hipEventElapsedTime(&ms, s->events[0], s->events[i]));
        if (gpu_ms) gpu_ms[i] = ms;
        if (host_ms) host_ms[i] = (s->host_us[i] - s->host_us[0]) * 1e-3;
        if (names) off += snprintf(names + off, off < names_cap ? names_cap - off : 0, "%s\n", s->names[i]);
    }
    return ODISE_OK;
}
extern "C" int odise_hip_launch_log(odise_hip_ctx* ctx, int on) {
    ODISE_REQUIRE(ctx, "launch_log: null context");
    launch_log_release(ctx);
    if (on) ctx->launch_log = new std::vector<LaunchRec>();
    return ODISE_OK;
}
extern "C" int odise_hip_launch_log_read(odise_hip_ctx* ctx, int* out6, int cap, int* n) {
    ODISE_REQUIRE(ctx && n, "launch_log_read: null argument");
    std::vector<LaunchRec>* v = (std::vector<LaunchRec>*)ctx->launch_log;
    *n = v ? (int)v->size() : 0;
    for (int i = 0; v && out6 && i < *n && i < cap; ++i) {
        const LaunchRec& r = (*v)[i];
        int* o = out6 + 6 * (size_t)i;
        o[0] = r.conv; o[1] = r.M; o[2] = r.N; o[3] = r.K; o[4] = r.tile; o[5] = r.split;
    }
    return ODISE_OK;
}

// ABI self-description used by tests/test_lib_abi.py to validate the ctypes mirrors
extern "C" int odise_hip_set_lanes(odise_hip_ctx* ctx, int lanes) {
    ODISE_REQUIRE(ctx && (lanes == 1 || lanes == 2), "set_lanes: 1 or 2");
    ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->lanes = lanes;
    return ODISE_OK;
}
extern "C" int odise_hip_sizeof_gemm_desc(void) { return (int)sizeof(odise_gemm_desc); }
extern "C" int odise_hip_sizeof_conv_desc(void) { return (int)sizeof(odise_conv_desc); }
extern "C" int odise_hip_sizeof_attn_desc(void) { return (int)sizeof(odise_attn_desc); }
extern "C" int odise_hip_sizeof_post_desc(void) { return (int)sizeof(odise_post_desc); }
extern "C" int odise_hip_sizeof_infer_desc(void) { return (int)sizeof(odise_infer_desc); }
