// misc.hip — small streaming kernels around the GEMM stages of the feature extractor:
//   * image_to_nhwc:        [B,3,H,W] f32 -> NHWC f16 with per-channel affine (LdmExtractor.forward normalisation, ldm.py:556)
//   * resize_bicubic_norm:  CLIP preprocess (clip.py:94): bicubic short-side resize (align_corners=False, A=-0.75, no
//                           antialias — torchvision 0.14.1 tensor path) + centre crop + mean/std normalise -> NHWC f16
//   * softmax_rows:         fp16 row softmax with fp32 math (VAE AttnBlock, single head of width 512)
//   * clip_assemble_tokens: [cls | patches] + positional embedding (clip.py:179-190)
//   * cond_inputs:          uncond + tanh(alpha) * (proj + pos)   (ldm.py:706-709), folded to A1 + A2 * proj
//   * latent_heads:         quant_conv mean * scale -> q_sample(t) -> x_t ; post_quant_conv(mean) -> decoder input
//                           (ldm.py:459-467, 535-538, 577-598; gaussian_diffusion.py:275-292)
#include "engine.h"

namespace odise {

__global__ void __launch_bounds__(256) image_to_nhwc_kernel(const float* __restrict__ x, f16* __restrict__ y, int C, int HW, int Cpad,
                                                           float s0, float s1, float s2, float b0, float b1, float b2) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float sc[3] = {s0, s1, s2}, sh[3] = {b0, b1, b2};
    for (int c8 = 0; c8 < Cpad / 8; ++c8) {
        f16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c8 * 8 + i;
            o[i] = c < C ? (f16)(x[((int64_t)n * C + c) * HW + p] * sc[c < 3 ? c : 0] + sh[c < 3 ? c : 0]) : (f16)0.f;
        }
        *reinterpret_cast<f16x8*>(y + ((int64_t)n * HW + p) * Cpad + c8 * 8) = o;
    }
}

__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// out [B, S, S, 8] f16 (channels 3..7 zero) from x [B,3,H,W] f32; resized size (RH,RW), crop offset (top,left)
__global__ void __launch_bounds__(256) resize_bicubic_norm_kernel(const float* __restrict__ x, f16* __restrict__ y, int H, int W, int RH,
                                                                 int RW, int S, int top, int left, float m0, float m1, float m2,
                                                                 float is0, float is1, float is2) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S * S) return;
    const int oy = p / S + top, ox = p % S + left;
    const float A = -0.75f;
    const float sy = (float)H / (float)RH, sx = (float)W / (float)RW;
    const float fy = (oy + 0.5f) * sy - 0.5f, fx = (ox + 0.5f) * sx - 0.5f;
    const float yfl = floorf(fy), xfl = floorf(fx);
    const int iy = (int)yfl, ix = (int)xfl;
    const float ty = fy - yfl, tx = fx - xfl;
    float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
    float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
    const float mean[3] = {m0, m1, m2}, istd[3] = {is0, is1, is2};
    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool identity = (RH == H && RW == W);
    for (int c = 0; c < 3; ++c) {
        const float* xc = x + ((int64_t)n * 3 + c) * H * W;
        float acc = 0.f;
        if (identity) {
            acc = xc[(int64_t)oy * W + ox];
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int yy = min(max(iy - 1 + a, 0), H - 1);
                float row = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int xx = min(max(ix - 1 + b, 0), W - 1);
                    row += wx[b] * xc[(int64_t)yy * W + xx];
                }
                acc += wy[a] * row;
            }
        }
        o[c] = (f16)((acc - mean[c]) * istd[c]);
    }
    *reinterpret_cast<f16x8*>(y + ((int64_t)n * S * S + p) * 8) = o;
}

// y[r, :] = softmax(scale * x[r, :]); one 256-thread block per row, row kept in registers (cols <= 256*8*NV)
template <int NV>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const f16* __restrict__ x, f16* __restrict__ y, int cols, int64_t ld,
                                                          float scale_log2e) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const f16* xr = x + row * ld;
    f16* yr = y + row * ld;
    f16x8 v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (tid + 256 * k) * 8;
        if (c < cols) {
            v[k] = *reinterpret_cast<const f16x8*>(xr + c);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c + i < cols) mx = fmaxf(mx, (float)v[k][i]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float e[NV][8];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (tid + 256 * k) * 8;
        if (c < cols) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                e[k][i] = (c + i < cols) ? exp2f(((float)v[k][i] - mx) * scale_log2e) : 0.f;
                sum += e[k][i];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (tid + 256 * k) * 8;
        if (c < cols) {
            f16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (f16)(e[k][i] * inv);
            if (c + 8 <= cols) {
                *reinterpret_cast<f16x8*>(yr + c) = o;
            } else {
                for (int i = 0; c + i < cols; ++i) yr[c + i] = o[i];
            }
        }
    }
}

// tok[b, t, :] = (t == 0 ? cls : patches[b, t-1, :]) + pos[t, :]     (f16 out); every image owns TP >= T + extra rows, the tail rows are zero
__global__ void __launch_bounds__(256) clip_assemble_kernel(const f16* __restrict__ patches, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, f16* __restrict__ tok, int T, int extra, int TP,
                                                           int Cw, int64_t total8) {
    const int C8 = Cw >> 3;
    const int TA = T + extra;  // tokens T..TA-1 are MaskCLIP mask tokens = copies of the class token (clip.py:268-270)
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total8; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C8) * 8;
        const int64_t bt = idx / C8;
        const int t = (int)(bt % TP);
        const int64_t b = bt / TP;
        f16x8 o;
        if (t >= TA) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (f16)0.f;
        } else if (t == 0 || t >= T) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (f16)(cls[c + i] + pos[c + i]);
        } else {
            const f16x8 p = *reinterpret_cast<const f16x8*>(patches + (b * (T - 1) + (t - 1)) * Cw + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (f16)((float)p[i] + pos[(int64_t)t * Cw + c + i]);
        }
        *reinterpret_cast<f16x8*>(tok + idx * 8) = o;
    }
}

// out[b,t,c] = A1[t,c] + A2[t,c] * proj[b,c]   (fp32)
__global__ void __launch_bounds__(256) cond_inputs_kernel(const float* __restrict__ proj, const float* __restrict__ A1,
                                                         const float* __restrict__ A2, float* __restrict__ out, int T, int Cw,
                                                         int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Cw);
        const int64_t bt = idx / Cw;
        const int t = (int)(bt % T);
        const int64_t b = bt / T;
        out[idx] = A1[(int64_t)t * Cw + c] + A2[(int64_t)t * Cw + c] * proj[b * Cw + c];
    }
}

// h [B,P,8] f16 (VAE conv_out) -> x_t [B,P,8] f16 (4 channels + zero pad), zdec [B,P,8] f16, latent [B,4,P] f32 (optional)
__global__ void __launch_bounds__(256) latent_heads_kernel(const f16* __restrict__ h, const float* __restrict__ noise, f16* __restrict__ xt,
                                                          f16* __restrict__ zdec, float* __restrict__ latent, int P, int64_t total,
                                                          LatentW w) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(idx % P);
        const int64_t b = idx / P;
        const f16x8 v = *reinterpret_cast<const f16x8*>(h + idx * 8);
        float mean[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = w.bq[c];
#pragma unroll
            for (int k = 0; k < 8; ++k) a += w.wq[c][k] * (float)v[k];
            mean[c] = a;
        }
        f16x8 ox = {0, 0, 0, 0, 0, 0, 0, 0}, oz = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float lat = w.scale * mean[c];                       // latent = scale_factor * posterior.mean
            if (latent) latent[(b * 4 + c) * P + p] = lat;
            ox[c] = (f16)(w.qa * lat + w.qb * noise[(int64_t)c * P + p]);  // q_sample with the shared noise
            float z = w.bp[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) z += w.wp[c][k] * ((1.0f / w.scale) * (w.scale * mean[k]));  // post_quant_conv(latent / scale)
            oz[c] = (f16)z;
        }
        *reinterpret_cast<f16x8*>(xt + idx * 8) = ox;
        *reinterpret_cast<f16x8*>(zdec + idx * 8) = oz;
    }
}

static int grid1d(int64_t n) { return (int)std::min<int64_t>(ceil_div(n, 256), 4096); }

int launch_image_to_nhwc(odise_hip_ctx* ctx, const float* x, f16* y, int N, int C, int HW, int Cpad, const float* scale3,
                         const float* shift3) {
    dim3 grid((unsigned)ceil_div(HW, 256), (unsigned)N);
    hipLaunchKernelGGL(image_to_nhwc_kernel, grid, dim3(256), 0, ctx->stream, x, y, C, HW, Cpad, scale3[0], scale3[1], scale3[2],
                       shift3[0], shift3[1], shift3[2]);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

int launch_clip_preprocess(odise_hip_ctx* ctx, const float* x, f16* y, int N, int H, int W, int S) {
    int RH, RW;
    if (H <= W) { RH = S; RW = (int)((int64_t)S * W / H); } else { RH = (int)((int64_t)S * H / W); RW = S; }
    const int top = (int)lround((RH - S) / 2.0), left = (int)lround((RW - S) / 2.0);
    dim3 grid((unsigned)ceil_div(S * S, 256), (unsigned)N);
    hipLaunchKernelGGL(resize_bicubic_norm_kernel, grid, dim3(256), 0, ctx->stream, x, y, H, W, RH, RW, S, top, left, 0.48145466f,
                       0.4578275f, 0.40821073f, 1.f / 0.26862954f, 1.f / 0.26130258f, 1.f / 0.27577711f);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

int launch_softmax_rows(odise_hip_ctx* ctx, const f16* x, f16* y, int64_t rows, int cols, int64_t ld, float scale) {
    ODISE_REQUIRE(cols <= 256 * 8 * 4 && ld % 8 == 0, "softmax_rows: cols=%d must be <= 8192 and ld a multiple of 8", cols);
    const float s = scale * 1.4426950408889634f;
    if (cols <= 2048) hipLaunchKernelGGL(softmax_rows_kernel<1>, dim3((unsigned)rows), dim3(256), 0, ctx->stream, x, y, cols, ld, s);
    else if (cols <= 4096) hipLaunchKernelGGL(softmax_rows_kernel<2>, dim3((unsigned)rows), dim3(256), 0, ctx->stream, x, y, cols, ld, s);
    else hipLaunchKernelGGL(softmax_rows_kernel<4>, dim3((unsigned)rows), dim3(256), 0, ctx->stream, x, y, cols, ld, s);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

int launch_clip_assemble(odise_hip_ctx* ctx, const f16* patches, const float* cls, const float* pos, f16* tok, int B, int T, int extra,
                         int TP, int Cw) {
    ODISE_REQUIRE(TP >= T + extra, "clip_assemble: %d rows per image < %d tokens", TP, T + extra);
    const int64_t total8 = (int64_t)B * TP * (Cw / 8);
    hipLaunchKernelGGL(clip_assemble_kernel, dim3(grid1d(total8)), dim3(256), 0, ctx->stream, patches, cls, pos, tok, T, extra, TP, Cw, total8);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

int launch_cond_inputs(odise_hip_ctx* ctx, const float* proj, const float* A1, const float* A2, float* out, int B, int T, int Cw) {
    const int64_t total = (int64_t)B * T * Cw;
    hipLaunchKernelGGL(cond_inputs_kernel, dim3(grid1d(total)), dim3(256), 0, ctx->stream, proj, A1, A2, out, T, Cw, total);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

int launch_latent_heads(odise_hip_ctx* ctx, const f16* h, const float* noise, f16* xt, f16* zdec, float* latent, int B, int P,
                        const LatentW& w) {
    const int64_t total = (int64_t)B * P;
    hipLaunchKernelGGL(latent_heads_kernel, dim3(grid1d(total)), dim3(256), 0, ctx->stream, h, noise, xt, zdec, latent, P, total, w);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

}  // namespace odise
