// 3x3 convolution of an 8-channel NHWC image into 128 channels: AutoencoderKL's encoder.conv_in (3 real channels, ldm.py:556-560; 16 crops
// of 512^2 per step).  K = 9 taps x 8 channels = 72, so the launch is a 1 GB write and nothing else - but as an implicit GEMM through the
// generic tile it took 606 us (the LDS round trip of an im2col whose rows are 144 bytes), and the GroupNorm that follows needed its own
// statistics pass over the gigabyte (the generic tile has no statistics epilogue).
// Here the MFMA operands are loaded straight from memory: with K ordered tap-major, the 8 channels of one tap of one pixel ARE the 8
// consecutive k values a lane of v_mfma_f32_16x16x32_f16 holds (lane l: row / column l & 15, k = 8 (l >> 4) ..), i.e. one 16-byte load of
// the input pixel (y + dy, x + dx).  Three MFMAs cover the 9 taps (4 + 4 + 1, the rest zero).  The weights are the row operand (rows =
// output channels) and stay in registers for the whole block; the pixels are the columns, so a lane ends up with 4 consecutive channels
// of ONE pixel per MFMA.  Stored as such (8 bytes into each of 16 pixel rows per instruction) the launch wrote 2.4 TB/s; the wave's 32
// pixels x 128 channels = 8 KB of CONTIGUOUS NHWC output therefore pass through a wave-private LDS tile and leave as 16-byte pieces, 1 KB per
// instruction.  The MFMAs see the same k groups in the same order as the implicit GEMM: outputs are bit-identical to it.  The epilogue can
// also reduce the per-channel (sum, sum of squares) of the block's fp16 outputs - the statistics a following GroupNorm reads (gemm.hip
// GemmEpi::gn_stats has the same contract); the extractor does not request them (extractor.cpp: the first norm keeps its own pass).
#include "common.h"
#include "engine.h"

namespace odise {

constexpr int C8_PIX = 256;   // pixels per block: 4 waves x 4 groups of 16
constexpr int C8_PITCH = 272;  // bytes per pixel row of the LDS tile (256 + 16: the 8-byte writes of 16 pixel lanes fall on different banks, rows stay 16-byte aligned)

template <int CT, bool STATS>   // Cout = 16 * CT
__global__ void __launch_bounds__(256) conv3_c8_kernel(const f16* __restrict__ x, const f16* __restrict__ w, const float* __restrict__ bias,
                                                      f16* __restrict__ y, float* __restrict__ gn_part, int H, int W) {
    constexpr int Cout = 16 * CT;
    __shared__ float red[STATS ? 4 * Cout * 2 : 1];
    __shared__ __attribute__((aligned(16))) char otile[4][32 * C8_PITCH];   // per wave: 32 pixels x Cout fp16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // weights: row r of channel tile t, taps 4 s + g
    f16x8 wf[CT][3];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int tap = 4 * s + g;
            wf[t][s] = tap < 9 ? *reinterpret_cast<const f16x8*>(w + (size_t)(t * 16 + r) * 72 + tap * 8) : zero8;
        }
    float bs[CT][4];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) bs[t][i] = bias ? bias[t * 16 + g * 4 + i] : 0.f;
    // statistics: taken where the tile leaves LDS - a lane's 16-byte pieces are always the same 8 channels (64 lanes = 4 pixels x Cout / 8 pieces)
    static_assert(Cout / 8 == 16, "the read-back below assumes 16 pieces per pixel");
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    const f16* xn = x + (size_t)n * HW * 8;
    f16* yn = y + (size_t)n * HW * Cout;
    const int64_t p0 = (int64_t)blockIdx.x * C8_PIX + wave * 64;
    // the taps of the wave's four pixel groups: all 12 loads in flight before the first MFMA
    f16x8 xf[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t p = p0 + q * 16 + r;
        const int py = (int)(p / W), px = (int)(p - (int64_t)py * W);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int tap = 4 * s + g;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int yy = py + dy, xx = px + dx;
            const bool ok = tap < 9 && p < HW && yy >= 0 && yy < H && xx >= 0 && xx < W;
            xf[q][s] = ok ? *reinterpret_cast<const f16x8*>(xn + ((int64_t)yy * W + xx) * 8) : zero8;
        }
    }
    char* ot = otile[wave];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][s], xf[q][s], acc, 0, 0, 0);
            typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
            f16x4v o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (f16)(acc[i] + bs[t][i]);
            *reinterpret_cast<f16x4v*>(ot + ((q & 1) * 16 + r) * C8_PITCH + (t * 16 + g * 4) * 2) = o;
        }
        if (q & 1) {   // two groups = 32 consecutive pixels = 32 * Cout * 2 contiguous bytes of the output
            const int64_t pb = p0 + (q - 1) * 16;
            constexpr int PIECES = 32 * Cout * 2 / 16;   // 16-byte pieces, Cout / 8 per pixel
#pragma unroll
            for (int k = 0; k < PIECES / 64; ++k) {
                const int piece = k * 64 + lane;
                const int pr = piece / (Cout / 8), pc = piece - pr * (Cout / 8);
                const f16x8 v = *reinterpret_cast<const f16x8*>(ot + pr * C8_PITCH + pc * 16);
                if (pb + pr < HW) {
                    *reinterpret_cast<f16x8*>(yn + (pb + pr) * Cout + pc * 8) = v;
                    if (STATS) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float f = (float)v[i];
                            s1[i] += f;
                            s2[i] += f * f;
                        }
                    }
                }
            }
        }
    }
    if (STATS) {
        // lanes l, l + 16, l + 32, l + 48 hold the same 8 channels (pieces (l & 15) of four pixels); then the four waves through LDS
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s1[i] += __shfl_xor(s1[i], 16);
            s2[i] += __shfl_xor(s2[i], 16);
            s1[i] += __shfl_xor(s1[i], 32);
            s2[i] += __shfl_xor(s2[i], 32);
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane * 8 + i;
                red[(wave * Cout + c) * 2] = s1[i];
                red[(wave * Cout + c) * 2 + 1] = s2[i];
            }
        }
        __syncthreads();
        for (int i = tid; i < Cout * 2; i += 256) {
            const float a = red[i] + red[Cout * 2 + i] + red[2 * Cout * 2 + i] + red[3 * Cout * 2 + i];
            gn_part[(((int64_t)n * gridDim.x + blockIdx.x) * Cout) * 2 + i] = a;
        }
    }
}

bool conv3_c8_ok(const odise_conv_desc* d) {
    return d->Cin == 8 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 && d->OH == d->H && d->OW == d->W &&
           !d->upsample2x && d->y_dtype == ODISE_F16 && !d->residual && !d->per_image_add && d->act == ODISE_ACT_NONE && d->Cout == 128 &&
           ((int64_t)d->H * d->W) % C8_PIX == 0 && d->N >= 1 && d->N < 65536 && (((uintptr_t)d->X | (uintptr_t)d->Wt | (uintptr_t)d->Y) & 15) == 0;
}

// gn_stats (optional): [N][H*W / 256][Cout][2] fp32; *stats_blocks = row blocks per image
int launch_conv3_c8(odise_hip_ctx* ctx, const odise_conv_desc* d, float* gn_stats, int* stats_blocks) {
    const int blocks = (int)(((int64_t)d->H * d->W) / C8_PIX);
    dim3 grid((unsigned)blocks, (unsigned)d->N);
    if (gn_stats)
        hipLaunchKernelGGL((conv3_c8_kernel<8, true>), grid, dim3(256), 0, ctx->stream, (const f16*)d->X, (const f16*)d->Wt, d->bias, (f16*)d->Y, gn_stats, d->H,
                           d->W);
    else
        hipLaunchKernelGGL((conv3_c8_kernel<8, false>), grid, dim3(256), 0, ctx->stream, (const f16*)d->X, (const f16*)d->Wt, d->bias, (f16*)d->Y, nullptr, d->H,
                           d->W);
    ODISE_CHECK_HIP(hipGetLastError());
    if (stats_blocks) *stats_blocks = gn_stats ? blocks : 0;
    return ODISE_OK;
}

}  // namespace odise
