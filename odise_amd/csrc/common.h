// common.h — shared host/device helpers for libodise_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>

#include "../../include/odise_hip.h"

namespace odise {

typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void set_error(const char* fmt, ...);

#define ODISE_CHECK_HIP(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            ::odise::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,          \
                               hipGetErrorString(_e));                                     \
            return ODISE_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

#define ODISE_REQUIRE(cond, ...)                                                           \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            ::odise::set_error(__VA_ARGS__);                                               \
            return ODISE_ERR_ARG;                                                          \
        }                                                                                  \
    } while (0)

#define ODISE_TRY(expr)                \
    do {                               \
        int _rc = (expr);              \
        if (_rc != ODISE_OK) return _rc; \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// x * sigmoid(kx) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the IEEE division the compiler emits for `/` (two
// v_div_scale, v_rcp, four fmas, v_div_fmas, v_div_fixup per element): the SiLU / QuickGELU epilogues and the GroupNorm apply pass run
// this once per output element.  At -inf the product is -0 either way.
__device__ __forceinline__ float mul_sigmoid(float x, float kx) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-kx)); }

// GELU (erf form, torch's default) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. <= 1e-7 * |x| on the result, far
// inside the f16 rounding of every tensor this feeds): 18 VALU operations instead of the ~38 of the device library's two-range erff.  The
// GEGLU epilogue of the UNet feed-forward GEMMs evaluates it once per output element (0.25 G evaluations per step).
__device__ __forceinline__ float gelu_erf(float v) {
    const float x = v * 0.70710678118654752f;
    const float a = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * a);
    const float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - p * __expf(-a * a);
    return 0.5f * v * (1.0f + copysignf(r, x));
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ODISE_ACT_SILU: return mul_sigmoid(v, v);
        case ODISE_ACT_RELU: return v > 0.f ? v : 0.f;
        case ODISE_ACT_GELU: return gelu_erf(v);
        case ODISE_ACT_QUICKGELU: return mul_sigmoid(v, 1.702f * v);
        default: return v;
    }
}

}  // namespace odise

// The context: one device, one stream, a split-K workspace, an activation arena and the weight store.
struct odise_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cu_count = 256;
    int max_lds_optin = 65536;   // the most LDS a block may opt into on this device (160 KiB on gfx950; the largest of what the runtime reports, api.cpp)
    // split-K / scratch workspace
    void* ws = nullptr;
    size_t ws_bytes = 0;
    void* zeros = nullptr;  // 256 zero bytes
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void* models = nullptr;  // odise::ModelStore* (weights + unet), see unet.cpp
    // jpeg.hip: pinned host staging for entropy-decoded coefficients, device coefficients + planes, upload-complete event
    void* jpeg_host = nullptr;
    size_t jpeg_host_bytes = 0;
    void* jpeg_dev = nullptr;
    size_t jpeg_dev_bytes = 0;
    hipEvent_t jpeg_ev = nullptr;
    // second lane (engine.h Lane2): the feature extractor runs its two independent branches - CLIP conditioning -> UNet, and VAE encoder ->
    // VAE decoder - on two streams with separate split-K workspaces, joined by events
    hipStream_t stream2 = nullptr;
    void* ws2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_mid = nullptr, ev_join = nullptr;
    hipEvent_t ev_mclip = nullptr;   // the MaskCLIP image-token pass enqueued on the second lane is done (engine.h ClipKV)
    int lanes = 2;  // 1 = everything on the one stream (tools A/B: odise_hip_set_lanes)
    // encoder prefetch (engine.h Prefetch, odise_hip_infer_prefetch): a third, lowest-priority stream with its own split-K workspace; the next
    // batch's VAE encoder runs there behind ev_pf_go (recorded when the current batch's VAE lane is done) and publishes ev_pf_done
    int prefetch_cu_eighths = 0;     // ODISE_OPT_PREFETCH_CU_EIGHTHS
    int prefetch_start = 1;          // ODISE_OPT_PREFETCH_START
    hipStream_t stream3 = nullptr;
    void* ws3 = nullptr;
    hipEvent_t ev_pf_go = nullptr, ev_pf_done = nullptr;
    void* comm = nullptr;  // odise::Comm* (comm.cpp): RCCL communicator + exchange stream, created by odise_hip_comm_init
    // per-context execution options (odise_hip_set_option, include/odise_hip.h); read on the host when a stage is enqueued
    int clip_ln_fold = 0;            // ODISE_OPT_CLIP_LN_FOLD: 0 = by token count, 1 = always, 2 = never
    int maskclip_passes = 0;         // ODISE_OPT_MASKCLIP_PASSES: 0 = image tokens ride in the crops' tower, 1 = two passes in place, 2 = one pass, 3 = own tower on the second lane
    int attn_kv_resident = 0;        // ODISE_OPT_ATTN_KV_RESIDENT: 0 = by the library's rules (attn.hip attn_kvres_ok / attn_sa_ok), bit 1 (2) = never the K/V-resident
                                     // kernel, bit 2 (4) = never the pipelined self-attention kernel
    int64_t vae_chunk_bytes = 0;     // ODISE_OPT_VAE_CHUNK_BYTES: crops per VAE launch so that one activation stays below this (0 = all crops at once, the default)
    void* probe = nullptr;           // odise::LaunchProbe* (api.cpp): HIP events around the launches of one kernel shape (odise_hip_probe_*)
    void* stages = nullptr;          // odise::StageLog* while odise_hip_stage_timeline is on: (name, HIP event on the current stream, host clock) at stage boundaries
    void* launch_log = nullptr;      // std::vector<odise::LaunchRec>* while odise_hip_launch_log is on: (shape, tile, split-K) of every GEMM / conv launch
};

namespace odise {
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE property of a kernel: remember it per (kernel instantiation, device), not per
// process, and let any number of host threads race to set it (the call is idempotent; the bit is published after it succeeded).
struct LdsAttrOnce {
    std::atomic<uint64_t> done{0};   // bit d: set on device d (devices >= 64 set it on every launch)
};
static inline int ensure_dyn_lds(odise_hip_ctx* ctx, LdsAttrOnce& once, const void* kernel, int bytes) {
    const int d = ctx->device;
    if (d >= 0 && d < 64 && ((once.done.load(std::memory_order_acquire) >> d) & 1)) return ODISE_OK;
    ODISE_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (d >= 0 && d < 64) once.done.fetch_or(1ull << d, std::memory_order_release);
    return ODISE_OK;
}
}  // namespace odise

namespace odise {
// odise_hip_probe_arm / _read: HIP events around every launch of ONE GEMM / convolution shape, recorded on the stream the launch goes to
// (whichever lane that is), so bench.py can report the dominant kernel's duration as it runs INSIDE the timed step, beside the other lane
struct LaunchProbe {
    int conv = 0, M = 0, N = 0, K = 0;
    int cap = 0, n = 0;
    hipEvent_t* ev = nullptr;   // 2 * cap events (start, stop)
    bool armed = false;
};
static inline LaunchProbe* probe_match(odise_hip_ctx* ctx, bool conv, int M, int N, int K) {
    LaunchProbe* p = (LaunchProbe*)ctx->probe;
    return (p && p->armed && p->n < p->cap && (p->conv != 0) == conv && p->M == M && p->N == N && p->K == K) ? p : nullptr;
}
void probe_release(odise_hip_ctx* ctx);
// stage boundaries of a model call for the measurement tools (tools/stage_timeline.py): no-op unless the timeline is on
void stage_mark(odise_hip_ctx* ctx, const char* name);
void stage_log_release(odise_hip_ctx* ctx);
struct LaunchRec { int conv, M, N, K, tile, split; };
void launch_log_push(odise_hip_ctx* ctx, const LaunchRec& r);
void launch_log_release(odise_hip_ctx* ctx);
// LayerNorm folded into the GEMMs around it (gemm.hip gemm_epilogue_f16; used by the CLIP towers): statistics of the rows a GEMM writes come out
// of its epilogue (`stats_out`), the GEMMs that would read LN(x) read x with gamma / beta folded into their weights and finish the
// normalisation per row (`part` + `colsum`) or, with swapped operands, per column (`fin` + `rowsum`)
constexpr int kLnPartCols = 64;   // columns per partial (sum, sum of squares) of a row: the wave column of the 8-phase 256x256 tile, half of the ping-pong tiles'
struct LnEpi {
    const float* part = nullptr;   // [M][P][2] partial (sum, sum of squares) per row of A
    int P = 0;
    float inv_c = 0.f, eps = 0.f;
    const float* colsum = nullptr; // [N]
    float* final_out = nullptr;    // [M][2] (-mean * rstd, rstd), optional
    const float* fin = nullptr;    // [N][2] per row of W (swapped form)
    const float* rowsum = nullptr; // [M]
    float* stats_out = nullptr;    // [M][N / kLnPartCols][2]
};
bool conv3_c8_ok(const odise_conv_desc* d);                                                                     // conv_c8.hip
int launch_conv3_c8(odise_hip_ctx* ctx, const odise_conv_desc* d, float* gn_stats, int* stats_blocks);
int gemm_forced(odise_hip_ctx* ctx, const odise_gemm_desc* d, int force_tile, int force_split, const LnEpi* ln = nullptr);   // force_tile < 0: the cost model's choice
int gemm_ln(odise_hip_ctx* ctx, const odise_gemm_desc* d, const LnEpi& ln);   // 256x256 ping-pong tile, math-first epilogue
void jpeg_release(odise_hip_ctx* ctx);
void comm_release(odise_hip_ctx* ctx);
}
