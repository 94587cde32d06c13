// decoder_ops.hip — streaming kernels of the backbone projection / stitching stage and of the Mask2Former heads.
// All are HBM/L2-bound gathers or elementwise passes over NHWC fp16 maps; 16-byte accesses per lane.
//   crop_extract        slide-window crops of the input image (feature_extractor.py:216-224)
//   upsample_nearest    F.interpolate(..., size) default mode (feature_extractor.py:165-168)
//   stitch_crops        overlap-add of per-crop features / count_mats (feature_extractor.py:229-248)
//   add_vec_table       src + level_embed (+ positional table)       (msdeformattn.py:71-75, odise.py:657-660)
//   msda_prepare        softmax over (levels*points) + sampling locations (ms_deform_attn.py:103-110)
//   bilinear_add        cur_fpn + bilinear(out[-1])                   (msdeformattn.py:347)
//   mask_binarize_f16   MaskPooling prologue on fp16 logits           (odise.py:949-955)
//   attn_mask           bilinear(mask logits -> level size).sigmoid() < 0.5, rows that mask everything are cleared
//                       (odise.py:760-774, 683)
#include "engine.h"

namespace odise {

__global__ void __launch_bounds__(256) crop_extract_kernel(const float* __restrict__ img, float* __restrict__ crops, int C, int H, int W,
                                                          int S, int K, const int* __restrict__ boxes /*[K][2] y1,x1*/, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % S);
        int64_t t = idx / S;
        const int y = (int)(t % S); t /= S;
        const int c = (int)(t % C); t /= C;
        const int k = (int)(t % K);
        const int64_t b = t / K;
        crops[idx] = img[((b * C + c) * H + boxes[2 * k] + y) * W + boxes[2 * k + 1] + x];
    }
}

// Slide windows smaller than the network input (images whose short side is below 512, feature_extractor.py:197-215): the s x s window
// is resized to S x S with bicubic interpolation - torchvision T.Resize(BICUBIC) on a tensor = F.interpolate(mode="bicubic",
// align_corners=False, antialias off; feature_extractor.py:69-77, 146-149): cubic convolution with A = -0.75, source coordinate
// (dst + 0.5) * s / S - 0.5, taps clamped to the WINDOW (the resize sees the cropped tensor, not the image around it).
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
    const float A = -0.75f;
    float x = t + 1.f;
    w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
    x = t;
    w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = 1.f - t;
    w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = 2.f - t;
    w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}
__global__ void __launch_bounds__(256) crop_resize_bicubic_kernel(const float* __restrict__ img, float* __restrict__ crops, int C, int H, int W, int s,
                                                                 int S, int K, const int* __restrict__ boxes, int64_t total) {
    const float scale = (float)s / (float)S;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % S);
        int64_t t = idx / S;
        const int y = (int)(t % S); t /= S;
        const int c = (int)(t % C); t /= C;
        const int k = (int)(t % K);
        const int64_t b = t / K;
        const float sy = ((float)y + 0.5f) * scale - 0.5f, sx = ((float)x + 0.5f) * scale - 0.5f;
        const int iy = (int)floorf(sy), ix = (int)floorf(sx);
        float wy[4], wx[4];
        cubic_coeffs(sy - (float)iy, wy);
        cubic_coeffs(sx - (float)ix, wx);
        const float* src = img + ((b * C + c) * H + boxes[2 * k]) * (int64_t)W + boxes[2 * k + 1];
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = min(max(iy - 1 + j, 0), s - 1);
            float row = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) row += wx[i] * src[(int64_t)yy * W + min(max(ix - 1 + i, 0), s - 1)];
            acc += wy[j] * row;
        }
        crops[idx] = acc;
    }
}

// y [N,OH,OW,C] = x [N,H,W,C] nearest (src = floor(dst * in / out))
__global__ void __launch_bounds__(256) upsample_nearest_kernel(const f16* __restrict__ x, f16* __restrict__ y, int H, int W, int OH, int OW,
                                                              int C8, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C8);
        int64_t t = idx / C8;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int64_t n = t / OH;
        const int sy = min((int)floorf(oy * ((float)H / OH)), H - 1), sx = min((int)floorf(ox * ((float)W / OW)), W - 1);
        *reinterpret_cast<f16x8*>(y + idx * 8) = *reinterpret_cast<const f16x8*>(x + (((n * H + sy) * W + sx) * C8 + c) * 8);
    }
}

// out [B,OH,OW,C] = sum over crops covering the pixel of feat [(b*K+k), ch, cw, C] / count
__global__ void __launch_bounds__(256) stitch_kernel(const f16* __restrict__ feat, f16* __restrict__ out, float* __restrict__ out_nchw,
                                                    int K, const int* __restrict__ boxes /*[K][2] in feature pixels*/, int ch, int cw,
                                                    int OH, int OW, int C8, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C8);
        int64_t t = idx / C8;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int64_t b = t / OH;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        int cnt = 0;
        for (int k = 0; k < K; ++k) {
            const int yy = oy - boxes[2 * k], xx = ox - boxes[2 * k + 1];
            if (yy >= 0 && yy < ch && xx >= 0 && xx < cw) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(feat + ((((b * K + k) * ch + yy) * cw + xx) * (int64_t)C8 + c) * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
                ++cnt;
            }
        }
        const float inv = cnt > 0 ? 1.f / (float)cnt : 0.f;
        f16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (f16)(acc[i] * inv);
        if (out) *reinterpret_cast<f16x8*>(out + idx * 8) = o;
        if (out_nchw) {
#pragma unroll
            for (int i = 0; i < 8; ++i) out_nchw[((b * (int64_t)C8 * 8 + c * 8 + i) * OH + oy) * OW + ox] = acc[i] * inv;
        }
    }
}

// y[b,p,:] = x[b,p,:] + vec[:] (+ table[p,:])
__global__ void __launch_bounds__(256) add_vec_table_kernel(const f16* __restrict__ x, const float* __restrict__ vec,
                                                           const float* __restrict__ table, f16* __restrict__ y, int P, int C8,
                                                           int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C8) * 8;
        const int p = (int)((idx / C8) % P);
        const f16x8 v = *reinterpret_cast<const f16x8*>(x + idx * 8);
        f16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = (float)v[i] + (vec ? vec[c + i] : 0.f);
            if (table) t += table[(int64_t)p * C8 * 8 + c + i];
            o[i] = (f16)t;
        }
        *reinterpret_cast<f16x8*>(y + idx * 8) = o;
    }
}

// y[b, :] = x[:] for b < B (16-byte pieces): the learned queries broadcast over the batch (odise.py:663-664) in ONE launch - four
// hipMemcpyAsync device-to-device copies of 51 KB cost ~60 us each through the blit path
__global__ void __launch_bounds__(256) broadcast_rows_kernel(const f16* __restrict__ x, f16* __restrict__ y, int64_t n8, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<f16x8*>(y)[idx] = reinterpret_cast<const f16x8*>(x)[idx % n8];
}

struct MsdaPrep {
    int L, P, M;
    int H[8], W[8], start[8];
};

// off [R, M*L*P*2] f32, aw [R, M*L*P] f32 (R = B*Lq rows; query q = r % Lq) -> loc [R,M,L,P,2], w [R,M,L,P]
// reference point of query q = centre of its own cell at its own level, identical for every sampled level (valid ratios 1)
// LC, PC: compile-time levels / points (0, 0 = runtime values from g: generic, arrays spill to scratch)
template <int LC, int PC>
__global__ void __launch_bounds__(256) msda_prepare_kernel(const float* __restrict__ off, const float* __restrict__ aw, float* __restrict__ loc,
                                                          float* __restrict__ w, int Lq, int64_t rows_heads, MsdaPrep g) {
    constexpr bool FIXED = LC > 0;
    constexpr int NE = FIXED ? LC * PC : 32;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < rows_heads; idx += (int64_t)gridDim.x * blockDim.x) {
        const int L = FIXED ? LC : g.L, P = FIXED ? PC : g.P;
        const int64_t r = idx / g.M;
        const int q = (int)(r % Lq);
        int lvl = 0;
        for (int l = 1; l < L; ++l)
            if (q >= g.start[l]) lvl = l;
        const int local = q - g.start[lvl];
        const float ref_y = ((float)(local / g.W[lvl]) + 0.5f) / (float)g.H[lvl];
        const float ref_x = ((float)(local % g.W[lvl]) + 0.5f) / (float)g.W[lvl];
        const int LP = L * P;
        const float* a = aw + idx * LP;
        const float* o = off + idx * LP * 2;
        float e[NE], ov[2 * NE];
        if (FIXED && (NE & 3) == 0) {
            // a lane's 4*LP + 8*LP bytes are contiguous and 16-byte aligned: vector loads (scalar ones waste 3/4 of every request)
#pragma unroll
            for (int i = 0; i < NE; i += 4) {
                const float4 t = *reinterpret_cast<const float4*>(a + i);
                e[i] = t.x; e[i + 1] = t.y; e[i + 2] = t.z; e[i + 3] = t.w;
            }
#pragma unroll
            for (int i = 0; i < 2 * NE; i += 4) {
                const float4 t = *reinterpret_cast<const float4*>(o + i);
                ov[i] = t.x; ov[i + 1] = t.y; ov[i + 2] = t.z; ov[i + 3] = t.w;
            }
        } else {
            for (int i = 0; i < LP; ++i) e[i] = a[i];
            for (int i = 0; i < 2 * LP; ++i) ov[i] = o[i];
        }
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (i < LP) mx = fmaxf(mx, e[i]);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (i < LP) { e[i] = expf(e[i] - mx); sum += e[i]; }
        const float inv = 1.f / sum;
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (i < LP) {
                const int l = i / P;
                e[i] *= inv;
                ov[2 * i + 0] = ref_x + ov[2 * i + 0] / (float)g.W[l];
                ov[2 * i + 1] = ref_y + ov[2 * i + 1] / (float)g.H[l];
            }
        float* wo = w + idx * LP;
        float* lo = loc + idx * LP * 2;
        if (FIXED && (NE & 3) == 0) {
#pragma unroll
            for (int i = 0; i < NE; i += 4) *reinterpret_cast<float4*>(wo + i) = make_float4(e[i], e[i + 1], e[i + 2], e[i + 3]);
#pragma unroll
            for (int i = 0; i < 2 * NE; i += 4) *reinterpret_cast<float4*>(lo + i) = make_float4(ov[i], ov[i + 1], ov[i + 2], ov[i + 3]);
        } else {
            for (int i = 0; i < LP; ++i) wo[i] = e[i];
            for (int i = 0; i < 2 * LP; ++i) lo[i] = ov[i];
        }
    }
}

__device__ __forceinline__ void bilinear_setup(int o, int in, int out, int& i0, int& i1, float& t) {
    // F.interpolate(mode="bilinear", align_corners=False): src = max((o + 0.5) * in/out - 0.5, 0)
    float s = ((float)o + 0.5f) * ((float)in / (float)out) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i0 = i0 < in - 1 ? i0 : in - 1;
    i1 = i0 < in - 1 ? i0 + 1 : i0;
    t = s - (float)i0;
}

// y [N,OH,OW,C] = a [N,OH,OW,C] + bilinear(b [N,H,W,C] -> OH x OW); a = nullptr: the resized map alone
__global__ void __launch_bounds__(256) bilinear_add_kernel(const f16* __restrict__ a, const f16* __restrict__ b, f16* __restrict__ y, int H,
                                                          int W, int OH, int OW, int C8, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C8);
        int64_t t = idx / C8;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int64_t n = t / OH;
        int y0, y1, x0, x1;
        float ty, tx;
        bilinear_setup(oy, H, OH, y0, y1, ty);
        bilinear_setup(ox, W, OW, x0, x1, tx);
        const f16* bb = b + n * H * W * (int64_t)C8 * 8 + c * 8;
        const f16x8 v00 = *reinterpret_cast<const f16x8*>(bb + ((int64_t)y0 * W + x0) * C8 * 8);
        const f16x8 v01 = *reinterpret_cast<const f16x8*>(bb + ((int64_t)y0 * W + x1) * C8 * 8);
        const f16x8 v10 = *reinterpret_cast<const f16x8*>(bb + ((int64_t)y1 * W + x0) * C8 * 8);
        const f16x8 v11 = *reinterpret_cast<const f16x8*>(bb + ((int64_t)y1 * W + x1) * C8 * 8);
        f16x8 va = {0, 0, 0, 0, 0, 0, 0, 0};
        if (a) va = *reinterpret_cast<const f16x8*>(a + idx * 8);
        f16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float top = (float)v00[i] + tx * ((float)v01[i] - (float)v00[i]);
            const float bot = (float)v10[i] + tx * ((float)v11[i] - (float)v10[i]);
            o[i] = (f16)((float)va[i] + top + ty * (bot - top));
        }
        *reinterpret_cast<f16x8*>(y + idx * 8) = o;
    }
}

// one block per (b,q) row of HW fp16 logits: m01 = sigmoid(x) > 0.5, inv = 1 / (sum m01 + 1e-8)
__global__ void __launch_bounds__(256) mask_binarize_f16_kernel(const f16* __restrict__ mask, f16* __restrict__ m01, float* __restrict__ inv,
                                                               int HW) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const f16* mr = mask + row * HW;
    f16* orow = m01 + row * HW;
    float cnt = 0.f;
    for (int i = threadIdx.x * 8; i < HW; i += blockDim.x * 8) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(mr + i);
        f16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float b = (1.f / (1.f + expf(-(float)v[k]))) > 0.5f ? 1.f : 0.f;
            o[k] = (f16)b;
            cnt += b;
        }
        *reinterpret_cast<f16x8*>(orow + i) = o;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) inv[row] = 1.f / (red[0] + red[1] + red[2] + red[3] + 1e-8f);
}

// one block per (b,q): logits [H,W] (fp16, or the fp32 accumulators of the prediction GEMM) -> u8 mask [oh*ow] (1 = key masked out); a row that
// masks every key is cleared
template <typename T>
__global__ void __launch_bounds__(256) attn_mask_kernel(const T* __restrict__ logits, uint8_t* __restrict__ out, int H, int W, int oh, int ow,
                                                       int64_t ldm) {
    __shared__ int red[4];
    const int64_t row = blockIdx.x;
    const T* lr = logits + row * H * W;
    uint8_t* orow = out + row * ldm;
    const int n = oh * ow;
    int masked = 0;
    if (H == oh && W == ow) {
        // the prediction was computed at this level already (maskgen.cpp predictor_forward): the resize is the identity (both taps coincide, weight 0:
        // v00 + 0 * (v01 - v00) = v00 for finite logits), so the loop below reduces to the threshold
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float v = (float)lr[i];
            const int m = (1.f / (1.f + expf(-v))) < 0.5f ? 1 : 0;
            orow[i] = (uint8_t)m;
            masked += m;
        }
    } else
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int oy = i / ow, ox = i - oy * ow;
        int y0, y1, x0, x1;
        float ty, tx;
        bilinear_setup(oy, H, oh, y0, y1, ty);
        bilinear_setup(ox, W, ow, x0, x1, tx);
        const float v00 = (float)lr[y0 * W + x0], v01 = (float)lr[y0 * W + x1], v10 = (float)lr[y1 * W + x0], v11 = (float)lr[y1 * W + x1];
        const float top = v00 + tx * (v01 - v00), bot = v10 + tx * (v11 - v10);
        const float v = top + ty * (bot - top);
        const int m = (1.f / (1.f + expf(-v))) < 0.5f ? 1 : 0;
        orow[i] = (uint8_t)m;
        masked += m;
    }
    for (int i = n + threadIdx.x; i < ldm; i += blockDim.x) orow[i] = 1;  // padding columns never visible
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) masked += __shfl_xor(masked, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = masked;
    __syncthreads();
    const int total = red[0] + red[1] + red[2] + red[3];
    if (total == n) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) orow[i] = 0;
    }
}

static int g1(int64_t n) { return (int)std::min<int64_t>(ceil_div(n, 256), 8192); }

int launch_crop_extract(odise_hip_ctx* ctx, const float* img, float* crops, int B, int C, int H, int W, int S, int K, const int* boxes_dev) {
    const int64_t total = (int64_t)B * K * C * S * S;
    hipLaunchKernelGGL(crop_extract_kernel, dim3(g1(total)), dim3(256), 0, ctx->stream, img, crops, C, H, W, S, K, boxes_dev, total);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_crop_resize_bicubic(odise_hip_ctx* ctx, const float* img, float* crops, int B, int C, int H, int W, int s, int S, int K, const int* boxes_dev) {
    const int64_t total = (int64_t)B * K * C * S * S;
    hipLaunchKernelGGL(crop_resize_bicubic_kernel, dim3(g1(total)), dim3(256), 0, ctx->stream, img, crops, C, H, W, s, S, K, boxes_dev, total);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_upsample_nearest(odise_hip_ctx* ctx, const f16* x, f16* y, int N, int H, int W, int OH, int OW, int C) {
    const int64_t total = (int64_t)N * OH * OW * (C / 8);
    hipLaunchKernelGGL(upsample_nearest_kernel, dim3(g1(total)), dim3(256), 0, ctx->stream, x, y, H, W, OH, OW, C / 8, total);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_stitch(odise_hip_ctx* ctx, const f16* feat, f16* out, float* out_nchw, int B, int K, const int* boxes_dev, int ch, int cw, int OH,
                  int OW, int C) {
    const int64_t total = (int64_t)B * OH * OW * (C / 8);
    hipLaunchKernelGGL(stitch_kernel, dim3(g1(total)), dim3(256), 0, ctx->stream, feat, out, out_nchw, K, boxes_dev, ch, cw, OH, OW, C / 8, total);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_broadcast_rows(odise_hip_ctx* ctx, const f16* x, f16* y, int64_t n, int B) {   // n % 8 == 0, 16-byte aligned
    const int64_t n8 = n / 8, total = n8 * B;
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3(g1(total)), dim3(256), 0, ctx->stream, x, y, n8, total);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_add_vec_table(odise_hip_ctx* ctx, const f16* x, const float* vec, const float* table, f16* y, int64_t N, int P, int C) {
    const int64_t total = N * P * (C / 8);
    hipLaunchKernelGGL(add_vec_table_kernel, dim3(g1(total)), dim3(256), 0, ctx->stream, x, vec, table, y, P, C / 8, total);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_msda_prepare(odise_hip_ctx* ctx, const float* off, const float* aw, float* loc, float* w, int B, int Lq, int M, int L, int P,
                        const int* Hs, const int* Ws, const int* starts) {
    ODISE_REQUIRE(L * P <= 32 && L <= 8, "msda_prepare: levels*points must be <= 32");
    MsdaPrep g;
    g.L = L; g.P = P; g.M = M;
    for (int l = 0; l < L; ++l) { g.H[l] = Hs[l]; g.W[l] = Ws[l]; g.start[l] = starts[l]; }
    const int64_t rh = (int64_t)B * Lq * M;
    if (L == 3 && P == 4) hipLaunchKernelGGL((msda_prepare_kernel<3, 4>), dim3(g1(rh)), dim3(256), 0, ctx->stream, off, aw, loc, w, Lq, rh, g);
    else hipLaunchKernelGGL((msda_prepare_kernel<0, 0>), dim3(g1(rh)), dim3(256), 0, ctx->stream, off, aw, loc, w, Lq, rh, g);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_bilinear_add(odise_hip_ctx* ctx, const f16* a, const f16* b, f16* y, int N, int H, int W, int OH, int OW, int C) {
    const int64_t total = (int64_t)N * OH * OW * (C / 8);
    hipLaunchKernelGGL(bilinear_add_kernel, dim3(g1(total)), dim3(256), 0, ctx->stream, a, b, y, H, W, OH, OW, C / 8, total);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_mask_binarize_f16(odise_hip_ctx* ctx, const f16* mask, f16* m01, float* inv, int64_t rows, int HW) {
    ODISE_REQUIRE(HW % 8 == 0, "mask_binarize: H*W must be a multiple of 8");
    hipLaunchKernelGGL(mask_binarize_f16_kernel, dim3((unsigned)rows), dim3(256), 0, ctx->stream, mask, m01, inv, HW);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_attn_mask(odise_hip_ctx* ctx, const f16* logits, uint8_t* out, int64_t rows, int H, int W, int oh, int ow, int64_t ldm) {
    hipLaunchKernelGGL(attn_mask_kernel<f16>, dim3((unsigned)rows), dim3(256), 0, ctx->stream, logits, out, H, W, oh, ow, ldm);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_attn_mask_f32(odise_hip_ctx* ctx, const float* logits, uint8_t* out, int64_t rows, int H, int W, int oh, int ow, int64_t ldm) {
    hipLaunchKernelGGL(attn_mask_kernel<float>, dim3((unsigned)rows), dim3(256), 0, ctx->stream, logits, out, H, W, oh, ow, ldm);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

}  // namespace odise
