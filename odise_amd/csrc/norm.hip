// norm.hip — GroupNorm(+SiLU/ReLU) and LayerNorm for NHWC / token-major fp16 activations, fp32 statistics.
//
// GroupNorm32 of the SD UNet ResBlock / SpatialTransformer (computed in fp32 and cast back — SURVEY.md
// Appendix A.1), the VAE `Normalize` (eps 1e-6), detectron2 get_norm("GN") of the projection BottleneckBlocks
// (feature_extractor.py:53-66) and the pixel decoder convs.  HBM/L2-bound streaming, two launches:
//   1. gn_partial_kernel: grid (chunks, N).  Each lane owns one 16-byte vector (8 channels) of a pixel and strides
//      over the chunk's pixels; per-channel (sum, sumsq) go through LDS and are folded per group in a fixed order
//      (deterministic, no atomics) into [N, chunks, G, 2] fp32 partials.
//   2. gn_apply_kernel: every block folds the (small) partials of its image with all 256 lanes, builds per-channel
//      scale/shift in LDS and streams x -> act(x*scale+shift) with 16-byte loads/stores.
// LayerNorm keeps a row in registers (one wavefront per row, <= 3 vectors per lane) so x is read once.
#include "common.h"
#include <stdlib.h>

namespace odise {

constexpr int GN_MAX_C = 4096;
static int g_gn_chunk_factor = 2;     // chunks of the statistics pass = cu_count * factor / N per image: two fat blocks per CU (measured: UNet at 16 crops 21.5 ms at 8, 21.3 at 2; tools: odise_hip_gn_tuning)
static int g_ln_rows8 = 32768;        // rows from which a LayerNorm of C <= 512 takes 8 rows per wavefront (8 loads in flight per lane): UNet at 16 crops 20.94 -> 20.81 ms (tools: odise_hip_gn_tuning(-rows, ..))
static int g_gn_fold_in_apply = 1;    // 1: the apply kernel folds the chunk partials itself when there are <= 64 per image (no finalize launch)

__global__ void __launch_bounds__(256) gn_partial_kernel(const f16* __restrict__ x, float* __restrict__ partial, int HW, int C,
                                                        int G, int pix_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) float sred[];  // [2][PL][C]
    const int n = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int tid = threadIdx.x;
    const int V = C >> 3;                   // 16-byte vectors per pixel
    const int VW = V < 256 ? V : 256;       // vectors handled side by side
    const int PL = 256 / VW;                // pixel lanes
    float* ssum = sred;
    float* ssq = sred + PL * C;
    const int p_begin = chunk * pix_per_chunk;
    const int p_end = min(HW, p_begin + pix_per_chunk);
    const int pl = tid / VW;
    if (pl < PL) {
        for (int v = tid - pl * VW; v < V; v += VW) {
            float s[8], q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
            const f16* xb = x + ((int64_t)n * HW) * C + v * 8;
            int p = p_begin + pl;
            // four independent 16-byte loads in flight per lane (the accumulation chain is short; the loop is latency-bound otherwise)
            for (; p + 3 * PL < p_end; p += 4 * PL) {
                f16x8 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f16x8*>(xb + (int64_t)(p + u * PL) * C);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float a = (float)t[u][i];
                        s[i] += a;
                        q[i] += a * a;
                    }
            }
            for (; p < p_end; p += PL) {
                const f16x8 t = *reinterpret_cast<const f16x8*>(xb + (int64_t)p * C);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float a = (float)t[i];
                    s[i] += a;
                    q[i] += a * a;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                ssum[pl * C + v * 8 + i] = s[i];
                ssq[pl * C + v * 8 + i] = q[i];
            }
        }
    }
    __syncthreads();
    const int cpg = C / G;
    if (tid < G) {
        float s = 0.f, q = 0.f;
        for (int t = 0; t < PL; ++t)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
                s += ssum[t * C + c];
                q += ssq[t * C + c];
            }
        float* o = partial + (((int64_t)n * nchunks + chunk) * G + tid) * 2;
        o[0] = s;
        o[1] = q;
    }
}

// grid (N); block 256: folds the chunk partials of one image in a fixed order (fp64) -> stats[n][g] = (mean, rstd)
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats, int HW, int C, int G,
                                                         int nchunks, float eps) {
    __shared__ double red[2 * 256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / G;
    // lane (g = tid % G, sub = tid / G) sums chunks sub, sub+nsub, ...; then a fixed-order fold over sub
    const int nsub = 256 / G;
    const int gI = tid % G, sub = tid / G;
    double s = 0.0, q = 0.0;
    if (sub < nsub) {
        for (int ch = sub; ch < nchunks; ch += nsub) {
            const float* o = partial + (((int64_t)n * nchunks + ch) * G + gI) * 2;
            s += (double)o[0];
            q += (double)o[1];
        }
    }
    red[2 * tid] = s;
    red[2 * tid + 1] = q;
    __syncthreads();
    if (tid < G) {
        s = q = 0.0;
        for (int u = 0; u < nsub; ++u) {
            s += red[2 * (u * G + tid)];
            q += red[2 * (u * G + tid) + 1];
        }
        const double cnt = (double)HW * cpg;
        const double mu = s / cnt;
        double var = q / cnt - mu * mu;
        if (var < 0.0) var = 0.0;
        stats[((int64_t)n * G + tid) * 2] = (float)mu;
        stats[((int64_t)n * G + tid) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// grid (N); block 256: statistics fused into the producing conv's epilogue arrive as per-channel (sum, sumsq) of every row block:
// colpart [N][nblk][C][2] -> stats[n][g] = (mean, rstd).  One block per (group, image): thread t folds row blocks t, t + 256, ... of its
// group's cpg channels (2 * cpg contiguous floats per row block) in fp64, then a fixed-order tree over the 256 partial sums.  (One block
// per image took 19.5 us on the 512-channel VAE tensors - 1 MB read by 256 threads with an 8-byte stride; 29 launches per step.)
__global__ void __launch_bounds__(256) gn_finalize_cols_kernel(const float* __restrict__ colpart, float* __restrict__ stats, int HW, int C, int G,
                                                              int nblk, float eps) {
    __shared__ double red[2 * 256];
    const int gI = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / G;
    double s = 0.0, q = 0.0;
    for (int b = tid; b < nblk; b += 256) {
        const float* o = colpart + (((int64_t)n * nblk + b) * C + (int64_t)gI * cpg) * 2;
        if ((cpg & 1) == 0) {
            for (int c = 0; c < cpg; c += 2) {
                const float4 v = *reinterpret_cast<const float4*>(o + 2 * c);  // (sum, sumsq) of two channels; 16-byte aligned: C % 8 == 0, cpg even
                s += (double)v.x + (double)v.z;
                q += (double)v.y + (double)v.w;
            }
        } else {
            for (int c = 0; c < cpg; ++c) {
                s += (double)o[2 * c];
                q += (double)o[2 * c + 1];
            }
        }
    }
    red[2 * tid] = s;
    red[2 * tid + 1] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) {
            red[2 * tid] += red[2 * (tid + w)];
            red[2 * tid + 1] += red[2 * (tid + w) + 1];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double cnt = (double)HW * cpg;
        const double mu = red[0] / cnt;
        double var = red[1] / cnt - mu * mu;
        if (var < 0.0) var = 0.0;
        stats[((int64_t)n * G + gI) * 2] = (float)mu;
        stats[((int64_t)n * G + gI) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// grid (blocks_per_image, N); block 256; every thread normalises up to four 16-byte vectors (all loads issued first)
__global__ void __launch_bounds__(256) gn_apply_kernel(const f16* __restrict__ x, f16* __restrict__ y,
                                                      const float* __restrict__ stats, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int HW, int C, int G,
                                                      int act, const f16* __restrict__ residual,
                                                      const f16* __restrict__ accum, int nchunks, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sss[];  // scale[C], shift[C]
    float* scale = sss;
    float* shift = sss + C;
    const int n = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / G;
    __shared__ double red[2 * 256];
    __shared__ float gstat[2 * 256];
    if (nchunks > 0) {
        // Round 6: `stats` holds the chunk partials [N][nchunks][G][2] of gn_partial_kernel and every block folds its image's (a few KB, L2-hot)
        // itself, in gn_finalize_kernel's order (the same bits): one launch of ~14 us less per GroupNorm in the UNet's dependent chain
        const int nsub = 256 / G;
        const int gI = tid % G, sub = tid / G;
        double s = 0.0, q = 0.0;
        if (sub < nsub) {
            for (int ch = sub; ch < nchunks; ch += nsub) {
                const float2 o = *reinterpret_cast<const float2*>(stats + (((int64_t)n * nchunks + ch) * G + gI) * 2);
                s += (double)o.x;
                q += (double)o.y;
            }
        }
        red[2 * tid] = s;
        red[2 * tid + 1] = q;
        __syncthreads();
        if (tid < G) {
            s = q = 0.0;
            for (int u = 0; u < nsub; ++u) {
                s += red[2 * (u * G + tid)];
                q += red[2 * (u * G + tid) + 1];
            }
            const double cnt = (double)HW * cpg;
            const double mu = s / cnt;
            double var = q / cnt - mu * mu;
            if (var < 0.0) var = 0.0;
            gstat[2 * tid] = (float)mu;
            gstat[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
        }
        __syncthreads();
    }
    for (int c = tid; c < C; c += blockDim.x) {
        const int gI = c / cpg;
        const float mu = nchunks > 0 ? gstat[2 * gI] : stats[((int64_t)n * G + gI) * 2];
        const float rs = nchunks > 0 ? gstat[2 * gI + 1] : stats[((int64_t)n * G + gI) * 2 + 1];
        const float sc = rs * (gamma ? gamma[c] : 1.f);
        scale[c] = sc;
        shift[c] = (beta ? beta[c] : 0.f) - mu * sc;
    }
    __syncthreads();
    // stream: 32-bit vector indices (HW * C / 8 < 2^31, checked by the launcher); four vectors per thread per sweep, loads first
    const int C8 = C >> 3;
    const int total = HW * C8;
    const f16* xb = x + (int64_t)n * HW * C;
    f16* yb = y + (int64_t)n * HW * C;
    const f16* rb = residual ? residual + (int64_t)n * HW * C : nullptr;
    const f16* ab = accum ? accum + (int64_t)n * HW * C : nullptr;
    const int stride = gridDim.x * blockDim.x;
    for (int base = blockIdx.x * blockDim.x + tid; base < total; base += 4 * stride) {
        f16x8 v[4], r[4], a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * stride;
            const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            v[u] = r[u] = a[u] = z;
            if (idx < total) {
                v[u] = *reinterpret_cast<const f16x8*>(xb + (int64_t)idx * 8);
                if (rb) r[u] = *reinterpret_cast<const f16x8*>(rb + (int64_t)idx * 8);
                if (ab) a[u] = *reinterpret_cast<const f16x8*>(ab + (int64_t)idx * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * stride;
            if (idx >= total) break;
            const int c0 = (idx % C8) * 8;
            const float4 s0 = *reinterpret_cast<const float4*>(scale + c0), s1 = *reinterpret_cast<const float4*>(scale + c0 + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(shift + c0), h1 = *reinterpret_cast<const float4*>(shift + c0 + 4);
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            float t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = (float)v[u][i] * sc[i] + sh[i] + (float)r[u][i];  // y = act(gn(x) + residual) + accum
            if (act == ODISE_ACT_SILU) {
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = mul_sigmoid(t[i], t[i]);
            } else if (act != ODISE_ACT_NONE) {
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = act_apply(t[i], act);
            }
            f16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (f16)(t[i] + (float)a[u][i]);
            __builtin_nontemporal_store(o, reinterpret_cast<f16x8*>(yb + (int64_t)idx * 8));  // streamed once: keep it out of the way of the L2-resident operands of the next conv
        }
    }
}

// One wavefront handles R rows at a time (all R row loads are issued before the first reduction, for memory-level
// parallelism); NV = 16-byte vectors per lane kept in registers (C <= 512*NV); x is read exactly once.
template <int NV, int R>
__global__ void __launch_bounds__(256) layer_norm_kernel(const f16* __restrict__ x, f16* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    f16x8 v[R][NV];
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = 0.f;
        const bool rok = row0 + r < rows;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (lane + 64 * k) * 8;
            if (rok && c < C) v[r][k] = *reinterpret_cast<const f16x8*>(x + (int64_t)(row0 + r) * C + c);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (lane + 64 * k) * 8;
            if (c < C) {
#pragma unroll
                for (int i = 0; i < 8; ++i) s[r] += (float)v[r][k][i];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] += __shfl_xor(s[r], o);
    }
    float q[R], mu[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mu[r] = s[r] / (float)C;
        q[r] = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (lane + 64 * k) * 8;
            if (c < C) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float d = (float)v[r][k][i] - mu[r];
                    q[r] += d * d;
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] += __shfl_xor(q[r], o);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (lane + 64 * k) * 8;
        if (c < C) {
            float gm[8], bt[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                gm[i] = gamma ? gamma[c + i] : 1.f;
                bt[i] = beta ? beta[c + i] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (row0 + r < rows) {
                    const float rs = rsqrtf(q[r] / (float)C + eps);
                    f16x8 o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = (f16)(((float)v[r][k][i] - mu[r]) * rs * gm[i] + bt[i]);
                    *reinterpret_cast<f16x8*>(y + (int64_t)(row0 + r) * C + c) = o;
                }
            }
        }
    }
}

// C <= 256 (the 256-channel pixel decoder and masked decoder: 44 MB launches over 86 016 rows): a row is at most 32 vectors, so the kernel above
// leaves half of every wavefront idle (2.2 TB/s measured).  Here each HALF-wave owns R rows: twice the rows and bytes in flight per wave.  The
// reductions are the same butterflies over 32 lanes in the same order (the full-wave form adds the idle half's zeros first): results are
// bit-identical.
// y2 (optional) = y + table[row % P] from the ROUNDED y: the next layer's query = LayerNorm output + positional table (msdeformattn.py:108-110,
// odise.py:700-716) leaves with it instead of re-reading it in a kernel of its own (the bits add_vec_table_kernel would write).
template <int R>
__global__ void __launch_bounds__(256) layer_norm_half_kernel(const f16* __restrict__ x, f16* __restrict__ y, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int rows, int C, float eps, f16* __restrict__ y2,
                                                             const float* __restrict__ table, int P) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int row0 = ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half) * R;
    const int c = l31 * 8;
    const bool cok = c < C;
    f16x8 v[R];
    float s[R], q[R], mu[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        v[r] = z;
        if (cok && row0 + r < rows) v[r] = *reinterpret_cast<const f16x8*>(x + (int64_t)(row0 + r) * C + c);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = 0.f;
        if (cok) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s[r] += (float)v[r][i];
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] += __shfl_xor(s[r], o);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mu[r] = s[r] / (float)C;
        q[r] = 0.f;
        if (cok) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = (float)v[r][i] - mu[r];
                q[r] += d * d;
            }
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] += __shfl_xor(q[r], o);
    }
    if (!cok) return;
    float gm[8], bt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        gm[i] = gamma ? gamma[c + i] : 1.f;
        bt[i] = beta ? beta[c + i] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (row0 + r < rows) {
            const float rs = rsqrtf(q[r] / (float)C + eps);
            f16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (f16)(((float)v[r][i] - mu[r]) * rs * gm[i] + bt[i]);
            *reinterpret_cast<f16x8*>(y + (int64_t)(row0 + r) * C + c) = o;
            if (y2) {
                const float* tr = table + (int64_t)((row0 + r) % P) * C + c;
                f16x8 o2;
#pragma unroll
                for (int i = 0; i < 8; ++i) o2[i] = (f16)((float)o[i] + tr[i]);
                *reinterpret_cast<f16x8*>(y2 + (int64_t)(row0 + r) * C + c) = o2;
            }
        }
    }
}

}  // namespace odise

extern "C" int odise_hip_group_norm_ex(odise_hip_ctx* ctx, const void* x, void* y, const float* gamma, const float* beta, int N,
                                       int HW, int C, int groups, float eps, int act, const void* residual, const void* accum);

extern "C" int odise_hip_group_norm(odise_hip_ctx* ctx, const void* x, void* y, const float* gamma, const float* beta, int N,
                                    int HW, int C, int groups, float eps, int act) {
    return odise_hip_group_norm_ex(ctx, x, y, gamma, beta, N, HW, C, groups, eps, act, nullptr, nullptr);
}

// y = act(GroupNorm(x) + residual) + accum   (residual / accum: optional fp16 tensors shaped like x)
extern "C" int odise_hip_group_norm_ex(odise_hip_ctx* ctx, const void* x, void* y, const float* gamma, const float* beta, int N,
                                       int HW, int C, int groups, float eps, int act, const void* residual, const void* accum) {
    using namespace odise;
    ODISE_REQUIRE(ctx && x && y, "group_norm: null argument");
    ODISE_REQUIRE(N >= 0 && HW > 0 && C > 0 && groups > 0, "group_norm: bad dims");
    ODISE_REQUIRE(C % groups == 0 && C % 8 == 0 && C <= GN_MAX_C, "group_norm: C=%d must be a multiple of 8 and of groups=%d, <= %d", C, groups, GN_MAX_C);
    ODISE_REQUIRE(groups <= 256, "group_norm: groups=%d > 256", groups);
    if (N == 0) return ODISE_OK;
    const int V = C / 8;
    const int VW = V < 256 ? V : 256;
    const int PL = 256 / VW;
    // chunking: ~2 blocks per CU over the batch, every pixel lane gets at least ~2 pixels, at most 512 chunks per image
    ODISE_REQUIRE((int64_t)HW * (C / 8) < (1ll << 31) - (1 << 24), "group_norm: image too large");
    int64_t want = std::max<int64_t>(1, (int64_t)ctx->cu_count * g_gn_chunk_factor / N);
    int64_t maxc = std::max<int64_t>(1, HW / (4 * PL));
    int nchunks = (int)std::min<int64_t>(std::min<int64_t>(want, maxc), 256);
    const int ppc = (int)ceil_div(HW, nchunks);
    nchunks = (int)ceil_div(HW, ppc);
    const size_t pbytes = ((size_t)N * nchunks * groups * 2 + (size_t)N * groups * 2) * sizeof(float);
    ODISE_REQUIRE(pbytes <= ctx->ws_bytes, "group_norm: workspace too small");
    float* partial = (float*)ctx->ws;
    float* stats = partial + (size_t)N * nchunks * groups * 2;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunks, N), dim3(256), 2 * (size_t)PL * C * sizeof(float), ctx->stream, (const f16*)x,
                       partial, HW, C, groups, ppc);
    ODISE_CHECK_HIP(hipGetLastError());
    // the apply blocks fold the partials themselves when there are few of them (<= 64 chunks: 16 KB per image, L2-hot); otherwise one finalize launch
    const bool fold_in_apply = g_gn_fold_in_apply && nchunks <= 64;
    if (!fold_in_apply) {
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(256), 0, ctx->stream, partial, stats, HW, C, groups, nchunks, eps);
        ODISE_CHECK_HIP(hipGetLastError());
    }
    const int64_t total = (int64_t)HW * (C / 8);
    const int bpi = (int)std::max<int64_t>(1, ceil_div(total, 256 * 4));  // four vectors per thread
    hipLaunchKernelGGL(gn_apply_kernel, dim3(bpi, N), dim3(256), 2 * (size_t)C * sizeof(float), ctx->stream, (const f16*)x, (f16*)y,
                       fold_in_apply ? partial : stats, gamma, beta, HW, C, groups, act, (const f16*)residual, (const f16*)accum, fold_in_apply ? nchunks : 0, eps);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

// GroupNorm whose statistics were produced by the conv that wrote x (gemm.hip: GemmEpi::gn_stats): finalize + apply only
namespace odise {
int group_norm_from_colpart(odise_hip_ctx* ctx, const void* x, void* y, const float* gamma, const float* beta, int N, int HW, int C, int groups,
                            float eps, int act, const float* colpart, int nblk) {
    ODISE_REQUIRE(ctx && x && y && colpart && nblk > 0, "group_norm: null argument");
    ODISE_REQUIRE(C % groups == 0 && C % 8 == 0 && C <= GN_MAX_C && groups <= 256, "group_norm: bad channel / group count");
    ODISE_REQUIRE((int64_t)HW * (C / 8) < (1ll << 31) - (1 << 24), "group_norm: image too large");
    ODISE_REQUIRE((size_t)N * groups * 2 * sizeof(float) <= ctx->ws_bytes, "group_norm: workspace too small");
    float* stats = (float*)ctx->ws;
    hipLaunchKernelGGL(gn_finalize_cols_kernel, dim3(groups, N), dim3(256), 0, ctx->stream, colpart, stats, HW, C, groups, nblk, eps);
    ODISE_CHECK_HIP(hipGetLastError());
    const int64_t total = (int64_t)HW * (C / 8);
    const int bpi = (int)std::max<int64_t>(1, ceil_div(total, 256 * 4));
    hipLaunchKernelGGL(gn_apply_kernel, dim3(bpi, N), dim3(256), 2 * (size_t)C * sizeof(float), ctx->stream, (const f16*)x, (f16*)y, stats, gamma,
                       beta, HW, C, groups, act, (const f16*)nullptr, (const f16*)nullptr, 0, eps);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
}  // namespace odise

namespace odise {
// y = LayerNorm(x); y2 (optional, C <= 256 only) = y + table[row % P]
int layer_norm_add_table(odise_hip_ctx* ctx, const void* x, void* y, const float* gamma, const float* beta, int rows, int C, float eps, f16* y2,
                         const float* table, int P) {
    ODISE_REQUIRE(ctx && x && y, "layer_norm: null argument");
    ODISE_REQUIRE(!y2 || (C <= 256 && table && P >= 1), "layer_norm: the second output needs C <= 256 and a table");
    ODISE_REQUIRE(rows >= 0 && C > 0 && C % 8 == 0 && C <= 4096, "layer_norm: C=%d must be a positive multiple of 8, <= 4096", C);
    if (rows == 0) return ODISE_OK;
    auto launch = [&](auto kern, int R) {
        const dim3 grid((unsigned)ceil_div(rows, 4 * R));
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, ctx->stream, (const f16*)x, (f16*)y, gamma, beta, rows, C, eps);
    };
    int many_rows = 2048;  // four rows per wavefront (all loads in flight first) from here on; measured on the step: never -2.1 ms, from 8192 rows -0.2 ms
#ifdef ODISE_TOOLS
    if (const char* e = getenv("ODISE_LN_MANY_ROWS")) many_rows = atoi(e);   // A/B of the rows-per-wavefront switch
#endif
    const bool many = rows >= many_rows;
    if (C <= 256) {   // a row fits half a wavefront: two rows per wave pass (layer_norm_half_kernel)
        if (many) hipLaunchKernelGGL(layer_norm_half_kernel<4>, dim3((unsigned)ceil_div(rows, 4 * 2 * 4)), dim3(256), 0, ctx->stream, (const f16*)x, (f16*)y, gamma, beta, rows, C, eps, y2, table, P);
        else hipLaunchKernelGGL(layer_norm_half_kernel<1>, dim3((unsigned)ceil_div(rows, 4 * 2)), dim3(256), 0, ctx->stream, (const f16*)x, (f16*)y, gamma, beta, rows, C, eps, y2, table, P);
    } else if (C <= 512) { if (rows >= g_ln_rows8) launch(layer_norm_kernel<1, 8>, 8); else if (many) launch(layer_norm_kernel<1, 4>, 4); else launch(layer_norm_kernel<1, 1>, 1); }
    else if (C <= 1024) { if (many) launch(layer_norm_kernel<2, 4>, 4); else launch(layer_norm_kernel<2, 1>, 1); }
    else if (C <= 2048) { if (many) launch(layer_norm_kernel<4, 2>, 2); else launch(layer_norm_kernel<4, 1>, 1); }
    else launch(layer_norm_kernel<8, 1>, 1);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
}  // namespace odise

extern "C" int odise_hip_layer_norm(odise_hip_ctx* ctx, const void* x, void* y, const float* gamma, const float* beta,
                                    int rows, int C, float eps) {
    return odise::layer_norm_add_table(ctx, x, y, gamma, beta, rows, C, eps, nullptr, nullptr, 1);
}

// tools hook (include/odise_hip_tools.h): chunk density of the GroupNorm statistics pass and whether the apply kernel finishes the statistics itself
extern "C" int odise_hip_gn_tuning(int chunk_factor, int fold_in_apply) {
    if (chunk_factor >= 1) odise::g_gn_chunk_factor = chunk_factor;
    if (chunk_factor < 0) odise::g_ln_rows8 = -chunk_factor;
    odise::g_gn_fold_in_apply = fold_in_apply != 0;
    return 0;
}
