// eval_ops.hip — input resize and evaluator reductions of the evaluation loop on the device.
//
//   odise_hip_resize_bilinear_u8   detectron2 T.ResizeShortestEdge -> ResizeTransform -> PIL.Image.resize(BILINEAR) on uint8 images
//                                  (configs/common/data/pano_open_d2_eval.py:74-107).  Pillow's 8-bit resampler is reproduced bit for
//                                  bit: double-precision triangle-filter coefficients normalised per output pixel, rounded to 22-bit
//                                  fixed point on the host, horizontal pass then vertical pass with a uint8 intermediate, accumulate
//                                  from 1 << 21, shift, clip.
//   odise_hip_u8_hwc_to_f32_chw    the DatasetMapper's HWC uint8 -> CHW float step (times `scale`, e.g. 1/255 for pixel_std 255).
//   odise_hip_semantic_confusion   detectron2 SemSegEvaluator.process (odise/evaluation/d2_evaluator.py:63): argmax over classes,
//                                  (K+1)^2 confusion counts (rows = prediction, ignore label mapped to K by the caller).
//   odise_hip_pair_histogram       the per-pixel part of panopticapi pq_compute_single_core (COCOPanopticEvaluator,
//                                  d2_evaluator.py:49): co-occurrence counts of (ground-truth segment, predicted segment) indices.
// All of it is HBM-bound integer / byte work: coalesced row-major sweeps, per-block LDS histograms flushed with integer atomics
// (deterministic: integer addition commutes).
#include <math.h>

#include <vector>

#include "common.h"

namespace odise {

constexpr int kPrecisionBits = 32 - 8 - 2;

struct ResampleCoeffs {
    std::vector<int> bounds;  // [out][2] (xmin, count)
    std::vector<int> kk;      // [out][ksize]
    int ksize = 0;
};

// Pillow precompute_coeffs + normalize_coeffs_8bpc, bilinear filter (support 1), whole input range
static void precompute_coeffs(int in_size, int out_size, ResampleCoeffs& c) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    c.ksize = (int)ceil(support) * 2 + 1;
    c.bounds.assign((size_t)out_size * 2, 0);
    c.kk.assign((size_t)out_size * c.ksize, 0);
    std::vector<double> k(c.ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < c.ksize; ++x) k[x] = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double w = a < 1.0 ? 1.0 - a : 0.0;
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        c.bounds[2 * xx] = xmin;
        c.bounds[2 * xx + 1] = xmax;
        for (int x = 0; x < c.ksize; ++x)
            c.kk[(size_t)xx * c.ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << kPrecisionBits)) : (int)(0.5 + k[x] * (1 << kPrecisionBits));
    }
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= kPrecisionBits;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one thread per output byte (row, ox, c); src row pitch W*C
__global__ void __launch_bounds__(256) resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ bounds,
                                                        const int* __restrict__ kk, int H, int W, int OW, int C, int ksize) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)H * OW * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int ox = (int)((idx / C) % OW);
    const int y = (int)(idx / ((int64_t)C * OW));
    const int xmin = bounds[2 * ox], n = bounds[2 * ox + 1];
    const int* k = kk + (size_t)ox * ksize;
    const uint8_t* s = src + ((int64_t)y * W + xmin) * C + c;
    int acc = 1 << (kPrecisionBits - 1);
    for (int x = 0; x < n; ++x) acc += (int)s[(int64_t)x * C] * k[x];
    dst[idx] = clip8(acc);
}

// one thread per output byte (oy, x*C + c); rows of W*C bytes
__global__ void __launch_bounds__(256) resample_v_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ bounds,
                                                        const int* __restrict__ kk, int OH, int row_bytes, int ksize) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)OH * row_bytes;
    if (idx >= total) return;
    const int xb = (int)(idx % row_bytes);
    const int oy = (int)(idx / row_bytes);
    const int ymin = bounds[2 * oy], n = bounds[2 * oy + 1];
    const int* k = kk + (size_t)oy * ksize;
    const uint8_t* s = src + (int64_t)ymin * row_bytes + xb;
    int acc = 1 << (kPrecisionBits - 1);
    for (int y = 0; y < n; ++y) acc += (int)s[(int64_t)y * row_bytes] * k[y];
    dst[idx] = clip8(acc);
}

__global__ void __launch_bounds__(256) u8_hwc_to_f32_chw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int HW, int C, float scale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)HW * C) return;
    const int c = (int)(idx / HW);
    const int p = (int)(idx - (int64_t)c * HW);
    dst[idx] = (float)src[(int64_t)p * C + c] * scale;
}

// dst [C,Hp,Wp]: the image in the top-left corner, zeros elsewhere (ImageList.from_tensors)
__global__ void __launch_bounds__(256) u8_hwc_to_f32_chw_pad_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int H, int W, int C,
                                                                   int Hp, int Wp, float scale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t plane = (int64_t)Hp * Wp;
    if (idx >= plane * C) return;
    const int c = (int)(idx / plane);
    const int64_t p = idx - c * plane;
    const int y = (int)(p / Wp), x = (int)(p - (int64_t)y * Wp);
    dst[idx] = (y < H && x < W) ? (float)src[((int64_t)y * W + x) * C + c] * scale : 0.f;
}

// per pixel: first-maximum argmax over K planes (torch.argmax semantics for distinct values; ties -> lowest class), count (pred, gt)
__global__ void __launch_bounds__(256) semantic_confusion_kernel(const float* __restrict__ sem, const int* __restrict__ gt, int K, int npix,
                                                                unsigned long long* __restrict__ conf) {
    extern __shared__ unsigned int hist[];  // [(K+1)*(K+1)] when it fits, else unused
    const int n = (K + 1) * (K + 1);
    const bool use_lds = n <= 12288;
    if (use_lds) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) hist[i] = 0;
        __syncthreads();
    }
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        float best = sem[p];
        int bi = 0;
        for (int k = 1; k < K; ++k) {
            const float v = sem[(int64_t)k * npix + p];
            if (v > best) { best = v; bi = k; }
        }
        int g = gt[p];
        g = (g < 0 || g > K) ? K : g;
        const int cell = bi * (K + 1) + g;
        if (use_lds) atomicAdd(&hist[cell], 1u);
        else atomicAdd(&conf[cell], 1ull);
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            if (hist[i]) atomicAdd(&conf[i], (unsigned long long)hist[i]);
    }
}

__global__ void __launch_bounds__(256) pair_histogram_kernel(const int* __restrict__ a, const int* __restrict__ b, int npix, int na, int nb,
                                                            unsigned int* __restrict__ out) {
    extern __shared__ unsigned int hist[];
    const int n = na * nb;
    const bool use_lds = n <= 12288;
    if (use_lds) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) hist[i] = 0;
        __syncthreads();
    }
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const int ia = a[p], ib = b[p];
        if ((unsigned)ia < (unsigned)na && (unsigned)ib < (unsigned)nb) {
            if (use_lds) atomicAdd(&hist[ia * nb + ib], 1u);
            else atomicAdd(&out[ia * nb + ib], 1u);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            if (hist[i]) atomicAdd(&out[i], hist[i]);
    }
}

static int upload_coeffs(odise_hip_ctx* ctx, const ResampleCoeffs& c, int** d_bounds, int** d_kk) {
    ODISE_CHECK_HIP(hipMalloc((void**)d_bounds, c.bounds.size() * sizeof(int)));
    ODISE_CHECK_HIP(hipMalloc((void**)d_kk, c.kk.size() * sizeof(int)));
    ODISE_CHECK_HIP(hipMemcpyAsync(*d_bounds, c.bounds.data(), c.bounds.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    ODISE_CHECK_HIP(hipMemcpyAsync(*d_kk, c.kk.data(), c.kk.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    return ODISE_OK;
}

}  // namespace odise

using namespace odise;

extern "C" int odise_hip_resize_bilinear_u8(odise_hip_ctx* ctx, const void* src, int H, int W, int C, void* dst, int OH, int OW) {
    ODISE_REQUIRE(ctx && src && dst, "resize_bilinear_u8: null argument");
    ODISE_REQUIRE(H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "resize_bilinear_u8: bad dims");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));
    const uint8_t* cur = (const uint8_t*)src;
    uint8_t* tmp = nullptr;
    int *bh = nullptr, *kh = nullptr, *bv = nullptr, *kv = nullptr;
    int rc = ODISE_OK;
    if (OW != W) {  // horizontal pass first (ImagingResampleInner); its output is the final image when the height is unchanged
        ResampleCoeffs c;
        precompute_coeffs(W, OW, c);
        rc = upload_coeffs(ctx, c, &bh, &kh);
        uint8_t* out = (uint8_t*)dst;
        if (rc == ODISE_OK && OH != H) {
            if (hipMalloc((void**)&tmp, (size_t)H * OW * C) != hipSuccess) { set_error("resize_bilinear_u8: out of memory"); rc = ODISE_ERR_NOMEM; }
            out = tmp;
        }
        if (rc == ODISE_OK) {
            const int64_t total = (int64_t)H * OW * C;
            hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, cur, out, bh, kh, H, W, OW, C, c.ksize);
            cur = out;
        }
    }
    if (rc == ODISE_OK && OH != H) {
        ResampleCoeffs c;
        precompute_coeffs(H, OH, c);
        rc = upload_coeffs(ctx, c, &bv, &kv);
        if (rc == ODISE_OK) {
            const int row_bytes = OW * C;
            const int64_t total = (int64_t)OH * row_bytes;
            hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, cur, (uint8_t*)dst, bv, kv, OH, row_bytes, c.ksize);
        }
    } else if (rc == ODISE_OK && OW == W) {
        if (hipMemcpyAsync(dst, src, (size_t)H * W * C, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = ODISE_ERR_HIP;
    }
    // the coefficient tables and the intermediate image are per-call temporaries: wait for the stream, then free them
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == ODISE_OK) { set_error("resize_bilinear_u8: stream error"); rc = ODISE_ERR_HIP; }
    for (void* p : {(void*)tmp, (void*)bh, (void*)kh, (void*)bv, (void*)kv})
        if (p) (void)hipFree(p);
    return rc;
}

extern "C" int odise_hip_u8_hwc_to_f32_chw(odise_hip_ctx* ctx, const void* src, float* dst, int H, int W, int C, float scale) {
    ODISE_REQUIRE(ctx && src && dst && H > 0 && W > 0 && C > 0, "u8_hwc_to_f32_chw: bad argument");
    const int64_t total = (int64_t)H * W * C;
    hipLaunchKernelGGL(u8_hwc_to_f32_chw_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, (const uint8_t*)src, dst, H * W, C, scale);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

extern "C" int odise_hip_u8_hwc_to_f32_chw_padded(odise_hip_ctx* ctx, const void* src, float* dst, int H, int W, int C, int Hp, int Wp, float scale) {
    ODISE_REQUIRE(ctx && src && dst && H > 0 && W > 0 && C > 0 && Hp >= H && Wp >= W, "u8_hwc_to_f32_chw_padded: bad argument");
    const int64_t total = (int64_t)Hp * Wp * C;
    hipLaunchKernelGGL(u8_hwc_to_f32_chw_pad_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, (const uint8_t*)src, dst, H, W, C,
                       Hp, Wp, scale);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

extern "C" int odise_hip_semantic_confusion(odise_hip_ctx* ctx, const float* sem_seg, const int* gt, int K, int npix, int64_t* conf) {
    ODISE_REQUIRE(ctx && sem_seg && gt && conf && K >= 1 && npix > 0, "semantic_confusion: bad argument");
    const int n = (K + 1) * (K + 1);
    const size_t lds = n <= 12288 ? (size_t)n * sizeof(unsigned int) : 0;
    const int blocks = (int)std::min<int64_t>(ceil_div(npix, 256), 4 * ctx->cu_count);
    hipLaunchKernelGGL(semantic_confusion_kernel, dim3(blocks), dim3(256), lds, ctx->stream, sem_seg, gt, K, npix, (unsigned long long*)conf);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

extern "C" int odise_hip_pair_histogram(odise_hip_ctx* ctx, const int* a, const int* b, int npix, int na, int nb, int* hist) {
    ODISE_REQUIRE(ctx && a && b && hist && npix > 0 && na > 0 && nb > 0, "pair_histogram: bad argument");
    const int n = na * nb;
    const size_t lds = n <= 12288 ? (size_t)n * sizeof(unsigned int) : 0;
    const int blocks = (int)std::min<int64_t>(ceil_div(npix, 256), 4 * ctx->cu_count);
    hipLaunchKernelGGL(pair_histogram_kernel, dim3(blocks), dim3(256), lds, ctx->stream, a, b, npix, na, nb, (unsigned int*)hist);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
