// elementwise.hip — layout conversion, cast, concat and mask-pooling helpers (HBM-bound streaming kernels).
#include "common.h"

namespace odise {

// x [N,C,HW] f32 -> y [N,HW,Cpad] f16 (channels >= C zero filled).  Lanes run along HW so reads are coalesced.
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ x, f16* __restrict__ y, int C, int HW,
                                                          int Cpad) {
    const int n = blockIdx.z;
    const int c8 = blockIdx.y;  // group of 8 output channels
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    f16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = c8 * 8 + i;
        o[i] = c < C ? (f16)x[((int64_t)n * C + c) * HW + p] : (f16)0.f;
    }
    *reinterpret_cast<f16x8*>(y + ((int64_t)n * HW + p) * Cpad + c8 * 8) = o;
}

// x [N,HW,C] f16 -> y [N,C,HW] f32
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const f16* __restrict__ x, float* __restrict__ y, int C, int HW) {
    const int n = blockIdx.z;
    const int c8 = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const f16x8 v = *reinterpret_cast<const f16x8*>(x + ((int64_t)n * HW + p) * C + c8 * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) y[((int64_t)n * C + c8 * 8 + i) * HW + p] = (float)v[i];
}

__global__ void __launch_bounds__(256) cast_f32_f16_kernel(const float* __restrict__ x, f16* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (f16)x[i];
}
__global__ void __launch_bounds__(256) cast_f16_f32_kernel(const f16* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (float)x[i];
}

// y[p, :] = cat(a[p, :Ca], b[p, :Cb]) in 16-byte chunks
__global__ void __launch_bounds__(256) concat_kernel(const f16* __restrict__ a, const f16* __restrict__ b, f16* __restrict__ y,
                                                    size_t pixels, int Ca, int Cb) {
    const int C8 = (Ca + Cb) >> 3, Ca8 = Ca >> 3;
    const size_t total = pixels * (size_t)C8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C8;
        const int c = (int)(i - p * C8);
        const f16x8 v = c < Ca8 ? *reinterpret_cast<const f16x8*>(a + p * Ca + (size_t)c * 8)
                                : *reinterpret_cast<const f16x8*>(b + p * Cb + (size_t)(c - Ca8) * 8);
        *reinterpret_cast<f16x8*>(y + i * 8) = v;
    }
}

// MaskPooling prologue (odise.py:953-954): m01 = (sigmoid(mask) > 0.5); inv[row] = 1/(sum m01 + 1e-8)
// one block per (b,q) row of HW logits
__global__ void __launch_bounds__(256) mask_binarize_kernel(const float* __restrict__ mask, f16* __restrict__ m01,
                                                           float* __restrict__ inv, int HW) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float* mr = mask + row * HW;
    f16* orow = m01 + row * HW;
    float cnt = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const float b = (1.f / (1.f + expf(-mr[i]))) > 0.5f ? 1.f : 0.f;
        orow[i] = (f16)b;
        cnt += b;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) inv[row] = 1.f / (red[0] + red[1] + red[2] + red[3] + 1e-8f);
}

static int grid_for(size_t n, int per_block) { return (int)std::min<size_t>((n + per_block - 1) / per_block, 8192); }

}  // namespace odise

using namespace odise;

extern "C" int odise_hip_nchw_f32_to_nhwc_f16(odise_hip_ctx* ctx, const float* x, void* y, int N, int C, int H, int W, int Cpad) {
    ODISE_REQUIRE(ctx && x && y, "nchw_f32_to_nhwc_f16: null argument");
    ODISE_REQUIRE(Cpad % 8 == 0 && Cpad >= C && C > 0, "nchw_f32_to_nhwc_f16: Cpad=%d must be a multiple of 8 and >= C=%d", Cpad, C);
    if (N == 0) return ODISE_OK;
    const int HW = H * W;
    dim3 grid((unsigned)ceil_div(HW, 256), (unsigned)(Cpad / 8), (unsigned)N);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, ctx->stream, x, (f16*)y, C, HW, Cpad);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

extern "C" int odise_hip_nhwc_f16_to_nchw_f32(odise_hip_ctx* ctx, const void* x, float* y, int N, int C, int H, int W) {
    ODISE_REQUIRE(ctx && x && y, "nhwc_f16_to_nchw_f32: null argument");
    ODISE_REQUIRE(C % 8 == 0 && C > 0, "nhwc_f16_to_nchw_f32: C=%d must be a multiple of 8", C);
    if (N == 0) return ODISE_OK;
    const int HW = H * W;
    dim3 grid((unsigned)ceil_div(HW, 256), (unsigned)(C / 8), (unsigned)N);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, ctx->stream, (const f16*)x, y, C, HW);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

extern "C" int odise_hip_cast_f32_to_f16(odise_hip_ctx* ctx, const float* x, void* y, size_t n) {
    ODISE_REQUIRE(ctx && (n == 0 || (x && y)), "cast_f32_to_f16: null argument");
    if (n == 0) return ODISE_OK;
    hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, x, (f16*)y, n);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

extern "C" int odise_hip_cast_f16_to_f32(odise_hip_ctx* ctx, const void* x, float* y, size_t n) {
    ODISE_REQUIRE(ctx && (n == 0 || (x && y)), "cast_f16_to_f32: null argument");
    if (n == 0) return ODISE_OK;
    hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, (const f16*)x, y, n);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

extern "C" int odise_hip_concat_channels(odise_hip_ctx* ctx, const void* a, const void* b, void* y, size_t pixels, int Ca, int Cb) {
    ODISE_REQUIRE(ctx && a && b && y, "concat_channels: null argument");
    ODISE_REQUIRE(Ca % 8 == 0 && Cb % 8 == 0 && Ca > 0 && Cb > 0, "concat_channels: channel counts must be multiples of 8");
    if (pixels == 0) return ODISE_OK;
    const size_t total = pixels * (size_t)((Ca + Cb) / 8);
    hipLaunchKernelGGL(concat_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, (const f16*)a, (const f16*)b, (f16*)y,
                       pixels, Ca, Cb);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

// MaskPooling.forward (odise/modeling/meta_arch/odise.py:937-963):
//   pooled[b,q,c] = sum_hw x[b,c,hw] * m01[b,q,hw] / (sum_hw m01[b,q,hw] + 1e-8)
// realised as one split-K MFMA GEMM per image: A = m01 [Q,HW] (exact in fp16), W = x [C,HW] cast to fp16,
// fp32 accumulate, per-row 1/denominator in the epilogue.
extern "C" int odise_hip_mask_pooling(odise_hip_ctx* ctx, const float* x, const float* mask, float* pooled, int B, int C, int Q,
                                      int HW) {
    ODISE_REQUIRE(ctx && x && mask && pooled, "mask_pooling: null argument");
    ODISE_REQUIRE(B >= 0 && C > 0 && Q > 0 && HW > 0, "mask_pooling: bad dims");
    ODISE_REQUIRE(HW % 8 == 0, "mask_pooling: H*W=%d must be a multiple of 8", HW);
    if (B == 0) return ODISE_OK;
    f16 *x16 = nullptr, *m16 = nullptr;
    float* inv = nullptr;
    ODISE_CHECK_HIP(hipMallocAsync((void**)&x16, (size_t)B * C * HW * 2, ctx->stream));
    ODISE_CHECK_HIP(hipMallocAsync((void**)&m16, (size_t)B * Q * HW * 2, ctx->stream));
    ODISE_CHECK_HIP(hipMallocAsync((void**)&inv, (size_t)B * Q * 4, ctx->stream));
    int rc = odise_hip_cast_f32_to_f16(ctx, x, x16, (size_t)B * C * HW);
    if (rc == ODISE_OK) {
        hipLaunchKernelGGL(mask_binarize_kernel, dim3((unsigned)(B * Q)), dim3(256), 0, ctx->stream, mask, m16, inv, HW);
        if (hipGetLastError() != hipSuccess) rc = ODISE_ERR_HIP;
    }
    for (int b = 0; b < B && rc == ODISE_OK; ++b) {
        odise_gemm_desc d = {};
        d.M = Q; d.N = C; d.K = HW;
        d.A = m16 + (size_t)b * Q * HW; d.lda = HW;
        d.W = x16 + (size_t)b * C * HW; d.ldw = HW;
        d.C = pooled + (size_t)b * Q * C; d.ldc = C; d.c_dtype = ODISE_F32;
        d.scale_m = inv + (size_t)b * Q;
        d.alpha = 1.f; d.batch = 1;
        rc = odise_hip_gemm(ctx, &d);
    }
    (void)hipFreeAsync(x16, ctx->stream);
    (void)hipFreeAsync(m16, ctx->stream);
    (void)hipFreeAsync(inv, ctx->stream);
    return rc;
}
