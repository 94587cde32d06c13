// classify_ops.hip — kernels of the open-vocabulary classification and post-processing stages.
//   resize_bilinear_norm   MaskCLIP.get_mask_embed image resize (F.interpolate bilinear, clip.py:327-332) + CLIP normalise
//   maskclip_token_mask    bilinear(mask logits -> 336^2).sigmoid() -> max_pool(patch) < 0.5  (clip.py:333, 290-302) as the u8
//                          visibility mask of the [577 image | Q mask] token layout used by clip_tower()
//   l2_normalize_rows      F.normalize(x, dim=-1)
//   classify_rows          synonym max-ensemble, softmaxes, geometric ensemble with the MaskCLIP logits, null-probability merge
//                          (helper.py:79-109; odise.py:181-207, 300-323, 1506-1536)
//   postprocess_pixels     fused mask upsample (x4 bilinear to the padded size, crop, bilinear to the output size:
//                          odise.py:326-331 + sem_seg_postprocess) + sigmoid + panoptic argmax + area counters
//                          (maskformer_model.py:286-320) writing the pixel-major sigmoid matrix for the semantic GEMM
//   column_stats / panoptic_write / instance_masks
#include "engine.h"

namespace odise {

__device__ __forceinline__ void bil_setup(int o, int in, int out, int& i0, int& i1, float& t) {
    float s = ((float)o + 0.5f) * ((float)in / (float)out) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i0 = i0 < in - 1 ? i0 : in - 1;
    i1 = i0 < in - 1 ? i0 + 1 : i0;
    t = s - (float)i0;
}

// x [B,3,H,W] f32 in [0,1] -> y [B,S,S,8] f16: bilinear (align_corners=False) to SxS, then (v - mean) / std
__global__ void __launch_bounds__(256) resize_bilinear_norm_kernel(const float* __restrict__ x, f16* __restrict__ y, int H, int W, int S) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= S * S) return;
    const int oy = p / S, ox = p - oy * S;
    int y0, y1, x0, x1;
    float ty, tx;
    bil_setup(oy, H, S, y0, y1, ty);
    bil_setup(ox, W, S, x0, x1, tx);
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
    const float istd[3] = {1.f / 0.26862954f, 1.f / 0.26130258f, 1.f / 0.27577711f};
    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < 3; ++c) {
        const float* xc = x + ((int64_t)n * 3 + c) * H * W;
        const float top = xc[(int64_t)y0 * W + x0] + tx * (xc[(int64_t)y0 * W + x1] - xc[(int64_t)y0 * W + x0]);
        const float bot = xc[(int64_t)y1 * W + x0] + tx * (xc[(int64_t)y1 * W + x1] - xc[(int64_t)y1 * W + x0]);
        o[c] = (f16)(((top + ty * (bot - top)) - mean[c]) * istd[c]);
    }
    *reinterpret_cast<f16x8*>(y + ((int64_t)n * S * S + p) * 8) = o;
}

// one block per (b, token row) of the [T image | Q mask] layout; logits [B,Q,h,w] f16.
// image rows: everything visible.  mask row q: col 0 (class token) visible, col 1+p visible iff max over the patch of the
// bilinearly resized mask probability >= 0.5.
__global__ void __launch_bounds__(256) maskclip_token_mask_kernel(const f16* __restrict__ logits, uint8_t* __restrict__ out, int Q, int h,
                                                                 int w, int S, int patch, int T, int64_t ldm) {
    const int TA = T + Q;
    const int b = blockIdx.x / TA, row = blockIdx.x % TA;
    uint8_t* orow = out + ((int64_t)b * TA + row) * ldm;
    if (row < T) {
        for (int i = threadIdx.x; i < ldm; i += blockDim.x) orow[i] = i < T ? 0 : 1;
        return;
    }
    const int q = row - T, G = S / patch;
    const f16* lr = logits + ((int64_t)b * Q + q) * h * w;
    for (int p = threadIdx.x; p < G * G; p += blockDim.x) {
        const int py = p / G, px = p - py * G;
        float mx = -INFINITY;
        for (int dy = 0; dy < patch; ++dy) {
            int y0, y1;
            float ty;
            bil_setup(py * patch + dy, h, S, y0, y1, ty);
            for (int dx = 0; dx < patch; ++dx) {
                int x0, x1;
                float tx;
                bil_setup(px * patch + dx, w, S, x0, x1, tx);
                const float v00 = (float)lr[y0 * w + x0], v01 = (float)lr[y0 * w + x1], v10 = (float)lr[y1 * w + x0], v11 = (float)lr[y1 * w + x1];
                const float top = v00 + tx * (v01 - v00), bot = v10 + tx * (v11 - v10);
                mx = fmaxf(mx, top + ty * (bot - top));
            }
        }
        orow[1 + p] = (1.f / (1.f + expf(-mx))) < 0.5f ? 1 : 0;
    }
    if (threadIdx.x == 0) orow[0] = 0;
    for (int i = T + threadIdx.x; i < ldm; i += blockDim.x) orow[i] = 1;
}

// y[r,:] = x[r,:] / max(||x[r,:]||, 1e-12); one wavefront per row; TIn in {f16, float}; output f16
template <typename TIn>
__global__ void __launch_bounds__(256) l2_normalize_kernel(const TIn* __restrict__ x, f16* __restrict__ y, int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TIn* xr = x + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = (float)xr[c]; s += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
    for (int c = lane; c < C; c += 64) y[row * C + c] = (f16)((float)xr[c] * inv);
}

// one block per (b,q) row.  L1 [rows, K1tot+1] cosines vs the category text bank (+ null as last column); L2 [rows, K2tot] cosines
// vs the MaskCLIP text bank; seg [K+1] group offsets (shared by both banks); ovl [K]; out [rows, K+1] log-probabilities.
// binary (optional) [rows, 2]: learned (object, no-object) logits of CaptionODISE's class_embed - when given the no-object probability
// is softmax(binary)[1] (odise.py:559-565) instead of the null-text probability of CategoryODISE (odise.py:313-314).
__global__ void __launch_bounds__(256) classify_rows_kernel(const float* __restrict__ L1, const float* __restrict__ L2, const int* __restrict__ seg,
                                                           const int* __restrict__ ovl, const float* __restrict__ binary, float* __restrict__ out,
                                                           int K, int Ktot, float ls1, float ls2, float alpha, float beta) {
    extern __shared__ float sm[];  // a[K], b[K], open[K], red[8]
    float* a = sm;
    float* bq = sm + K;
    float* op = sm + 2 * K;
    float* red = sm + 3 * K;
    const int64_t row = blockIdx.x;
    const float* l1 = L1 + row * (Ktot + 1);
    const float* l2 = L2 + row * Ktot;
    const int tid = threadIdx.x;
    for (int k = tid; k < K; k += blockDim.x) {  // max over synonyms (helper.py:96-100)
        float m1 = -INFINITY, m2 = -INFINITY;
        for (int i = seg[k]; i < seg[k + 1]; ++i) { m1 = fmaxf(m1, l1[i]); m2 = fmaxf(m2, l2[i]); }
        a[k] = m1 * ls1;
        bq[k] = m2 * ls2;
    }
    __syncthreads();
    const float nul = l1[Ktot] * ls1;
    auto block_max = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    };
    auto block_sum = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    // log-softmax of a (open logits) and of b (MaskCLIP logits)
    float ma = -INFINITY, mb = -INFINITY;
    for (int k = tid; k < K; k += blockDim.x) { ma = fmaxf(ma, a[k]); mb = fmaxf(mb, bq[k]); }
    ma = block_max(ma);
    mb = block_max(mb);
    float sa = 0.f, sb = 0.f;
    for (int k = tid; k < K; k += blockDim.x) { sa += expf(a[k] - ma); sb += expf(bq[k] - mb); }
    sa = block_sum(sa);
    sb = block_sum(sb);
    const float lsa = ma + logf(sa), lsb = mb + logf(sb);
    // geometric ensemble (odise.py:1514-1536): log(p^(1-w) q^w), w = alpha on categories seen in training, beta otherwise
    float mo = -INFINITY;
    for (int k = tid; k < K; k += blockDim.x) {
        const float w = ovl[k] ? alpha : beta;
        const float v = (1.f - w) * (a[k] - lsa) + w * (bq[k] - lsb);
        op[k] = v;
        mo = fmaxf(mo, v);
    }
    mo = block_max(mo);
    float so = 0.f;
    for (int k = tid; k < K; k += blockDim.x) so += expf(op[k] - mo);
    so = block_sum(so);
    // null probability from softmax over [a, null] (odise.py:313-314), or the learned binary head of the caption variant
    const float mn = fmaxf(ma, nul);
    float pn = expf(nul - mn) / (sa * expf(ma - mn) + expf(nul - mn));
    if (binary) {
        const float b0 = binary[row * 2], b1 = binary[row * 2 + 1], mb2 = fmaxf(b0, b1);
        pn = expf(b1 - mb2) / (expf(b0 - mb2) + expf(b1 - mb2));
    }
    float* orow = out + row * (K + 1);
    for (int k = tid; k < K; k += blockDim.x) orow[k] = logf(expf(op[k] - mo) / so * (1.f - pn) + 1e-8f);
    if (tid == 0) orow[K] = logf(pn + 1e-8f);
}

__device__ __forceinline__ float sample_stage1(const f16* lr, int w4, int h4, int y, int x, int ph, int pw) {
    int y0, y1, x0, x1;
    float ty, tx;
    bil_setup(y, h4, ph, y0, y1, ty);
    bil_setup(x, w4, pw, x0, x1, tx);
    const float v00 = (float)lr[y0 * w4 + x0], v01 = (float)lr[y0 * w4 + x1], v10 = (float)lr[y1 * w4 + x0], v11 = (float)lr[y1 * w4 + x1];
    const float top = v00 + tx * (v01 - v00), bot = v10 + tx * (v11 - v10);
    return top + ty * (bot - top);
}

// logit of query q at output pixel (oy, ox): bilinear(crop(bilinear(logits -> padded size)) -> output size)
__device__ __forceinline__ float sample_mask(const f16* lr, const PostGeom& g, int oy, int ox) {
    if (g.oh == g.ih && g.ow == g.iw) return sample_stage1(lr, g.w4, g.h4, oy, ox, g.ph, g.pw);
    int y0, y1, x0, x1;
    float ty, tx;
    bil_setup(oy, g.ih, g.oh, y0, y1, ty);
    bil_setup(ox, g.iw, g.ow, x0, x1, tx);
    const float v00 = sample_stage1(lr, g.w4, g.h4, y0, x0, g.ph, g.pw), v01 = sample_stage1(lr, g.w4, g.h4, y0, x1, g.ph, g.pw);
    const float v10 = sample_stage1(lr, g.w4, g.h4, y1, x0, g.ph, g.pw), v11 = sample_stage1(lr, g.w4, g.h4, y1, x1, g.ph, g.pw);
    const float top = v00 + tx * (v01 - v00), bot = v10 + tx * (v11 - v10);
    return top + ty * (bot - top);
}

// thread per output pixel.  kscore[q] = panoptic score of query q if it is kept (label != null, score > threshold) else < 0.
// S [npix, Qpad] f16 sigmoid (optional); ids [npix] int32 = argmax kept query | (sigmoid>=0.5 ? 1<<16 : 0), -1 if nothing kept;
// counts [3][Q] int32: mask_area (argmax == q), original_area (sigmoid >= 0.5), intersection.
__global__ void __launch_bounds__(256) postprocess_pixels_kernel(const f16* __restrict__ logits, const float* __restrict__ kscore,
                                                                f16* __restrict__ S, int* __restrict__ ids, int* __restrict__ counts,
                                                                PostGeom g) {
    extern __shared__ int hist[];  // [3][Q]
    const int Q = g.Q;
    for (int i = threadIdx.x; i < 3 * Q; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int npix = g.oh * g.ow;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = p < npix;
    const int oy = ok ? p / g.ow : 0, ox = ok ? p - oy * g.ow : 0;
    float best = -1.f;
    int best_q = -1;
    bool best_pos = false;
    for (int q0 = 0; q0 < g.Qpad; q0 += 8) {
        f16x8 sv = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = q0 + j;
            if (q < Q) {  // uniform branch
                const float v = ok ? sample_mask(logits + (int64_t)q * g.h4 * g.w4, g, oy, ox) : -1.f;
                const float sg = 1.f / (1.f + expf(-v));
                sv[j] = (f16)sg;
                const float ks = kscore[q];
                const bool pos = sg >= 0.5f;
                if (ks >= 0.f) {
                    const unsigned long long bal = __ballot(ok && pos);
                    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&hist[Q + q], __popcll(bal));
                    const float pv = ks * sg;
                    if (ok && pv > best) { best = pv; best_q = q; best_pos = pos; }
                }
            }
        }
        if (S && ok) *reinterpret_cast<f16x8*>(S + (int64_t)p * g.Qpad + q0) = sv;
    }
    if (ok) {
        if (ids) ids[p] = best_q < 0 ? -1 : (best_q | (best_pos ? (1 << 16) : 0));
        if (best_q >= 0) {
            atomicAdd(&hist[best_q], 1);
            if (best_pos) atomicAdd(&hist[2 * Q + best_q], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * Q; i += blockDim.x)
        if (hist[i]) atomicAdd(&counts[i], hist[i]);
}

// Same outputs as postprocess_pixels_kernel for the common geometry (output size = image size, mask logits at exactly 1/4 of the
// padded size): the x4 bilinear upsampling has a fixed tap pattern - output column 4c+k reads cells (c-1, c) with t = .625 / .875
// for k = 0, 1 and (c, c+1) with t = .125 / .375 for k = 2, 3 (indices clamped at the border, which reproduces the clamped source
// coordinate exactly) - so one thread produces the 4 pixels of a cell column from 2 rows x 3 cells per query: 6 coalesced 2-byte
// loads per query instead of 16 gathers, the same arithmetic per pixel, identical results.
__global__ void __launch_bounds__(256) postprocess_pixels_x4_kernel(const f16* __restrict__ logits, const float* __restrict__ kscore,
                                                                   f16* __restrict__ S, int* __restrict__ ids, int* __restrict__ counts,
                                                                   PostGeom g) {
    extern __shared__ int hist[];  // [3][Q]
    const int Q = g.Q;
    for (int i = threadIdx.x; i < 3 * Q; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int cw = (g.ow + 3) >> 2;  // cell columns per output row
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool okt = t < g.oh * cw;
    const int oy = okt ? t / cw : 0, cx = okt ? t - oy * cw : 0;
    const int cy = oy >> 2, ky = oy & 3;
    const int ra = max(ky < 2 ? cy - 1 : cy, 0), rb = min(ky < 2 ? cy : cy + 1, g.h4 - 1);
    const float ty = ky == 0 ? 0.625f : ky == 1 ? 0.875f : ky == 2 ? 0.125f : 0.375f;
    const int cl = max(cx - 1, 0), cr = min(cx + 1, g.w4 - 1);
    const int oa = ra * g.w4, ob = rb * g.w4;
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ok[k] = okt && (4 * cx + k) < g.ow;
    float best[4] = {-1.f, -1.f, -1.f, -1.f};
    int best_q[4] = {-1, -1, -1, -1};
    bool best_pos[4] = {false, false, false, false};
    const int64_t p0 = (int64_t)oy * g.ow + 4 * cx;
    const int plane = g.h4 * g.w4;
    for (int q0 = 0; q0 < g.Qpad; q0 += 8) {
        f16x8 sv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sv[k] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = q0 + j;
            if (q < Q) {  // uniform branch
                const f16* lr = logits + (int64_t)q * plane;
                const float al = (float)lr[oa + cl], ac = (float)lr[oa + cx], ar = (float)lr[oa + cr];
                const float bl = (float)lr[ob + cl], bc = (float)lr[ob + cx], br = (float)lr[ob + cr];
                const float ks = kscore[q];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float tx = k == 0 ? 0.625f : k == 1 ? 0.875f : k == 2 ? 0.125f : 0.375f;
                    const float v00 = k < 2 ? al : ac, v01 = k < 2 ? ac : ar, v10 = k < 2 ? bl : bc, v11 = k < 2 ? bc : br;
                    const float top = v00 + tx * (v01 - v00), bot = v10 + tx * (v11 - v10);
                    const float v = ok[k] ? top + ty * (bot - top) : -1.f;
                    const float sg = 1.f / (1.f + expf(-v));
                    sv[k][j] = (f16)sg;
                    const bool pos = sg >= 0.5f;
                    if (ks >= 0.f) {
                        const unsigned long long bal = __ballot(ok[k] && pos);
                        if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&hist[Q + q], __popcll(bal));
                        const float pv = ks * sg;
                        if (ok[k] && pv > best[k]) { best[k] = pv; best_q[k] = q; best_pos[k] = pos; }
                    }
                }
            }
        }
        if (S) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ok[k]) *reinterpret_cast<f16x8*>(S + (p0 + k) * g.Qpad + q0) = sv[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (ok[k]) {
            if (ids) ids[p0 + k] = best_q[k] < 0 ? -1 : (best_q[k] | (best_pos[k] ? (1 << 16) : 0));
            if (best_q[k] >= 0) {
                atomicAdd(&hist[best_q[k]], 1);
                if (best_pos[k]) atomicAdd(&hist[2 * Q + best_q[k]], 1);
            }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * Q; i += blockDim.x)
        if (hist[i]) atomicAdd(&counts[i], hist[i]);
}

// per-query sum over pixels of sigmoid * [sigmoid > 0.5] and count of [sigmoid > 0.5] from S (instance mask scores,
// maskformer_model.py:376-377): block partials [nblocks][2][Qpad], folded in fixed order by column_fold_kernel
__global__ void __launch_bounds__(256) column_stats_kernel(const f16* __restrict__ S, float* __restrict__ partial, int npix, int Qpad,
                                                          int pix_per_block) {
    extern __shared__ float cs[];  // [PL][2][Qpad]
    const int V = Qpad >> 3, PL = 256 / V;
    const int v = threadIdx.x % V, pl = threadIdx.x / V;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
    float s[8], c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = c[i] = 0.f;
    if (pl < PL) {
        for (int p = p0 + pl; p < p1; p += PL) {
            const f16x8 t = *reinterpret_cast<const f16x8*>(S + (int64_t)p * Qpad + v * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float a = (float)t[i];
                if (a > 0.5f) { s[i] += a; c[i] += 1.f; }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            cs[(pl * 2 + 0) * Qpad + v * 8 + i] = s[i];
            cs[(pl * 2 + 1) * Qpad + v * 8 + i] = c[i];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Qpad; i += blockDim.x) {
        float a = 0.f;
        for (int t = 0; t < PL; ++t) a += cs[(t * 2 + i / Qpad) * Qpad + (i % Qpad)];
        partial[(int64_t)blockIdx.x * 2 * Qpad + i] = a;
    }
}
// one wavefront per output column: lanes stride over the block partials, then a fixed-order butterfly (deterministic)
__global__ void __launch_bounds__(256) column_fold_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblocks, int n) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    double a = 0.0;
    for (int b = lane; b < nblocks; b += 64) a += (double)partial[(int64_t)b * n + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) out[i] = (float)a;
}

// seg[p] = map[q] where ids[p] = q | flag and the pixel is inside mask q (flag) ; 0 otherwise  (maskformer_model.py:321-333)
__global__ void __launch_bounds__(256) panoptic_write_kernel(const int* __restrict__ ids, const int* __restrict__ map, int* __restrict__ seg,
                                                            int npix) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const int v = ids[p];
    seg[p] = (v >= 0 && (v & (1 << 16))) ? map[v & 0xffff] : 0;
}

// out[n, p] = (upsampled logit of query idx[n] at pixel p > 0) ? 1 : 0   (maskformer_model.py:371)
__global__ void __launch_bounds__(256) instance_masks_kernel(const f16* __restrict__ logits, const int* __restrict__ idx, float* __restrict__ out,
                                                            PostGeom g) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int npix = g.oh * g.ow;
    if (p >= npix) return;
    const int oy = p / g.ow, ox = p - oy * g.ow;
    const float v = sample_mask(logits + (int64_t)idx[n] * g.h4 * g.w4, g, oy, ox);
    out[(int64_t)n * npix + p] = v > 0.f ? 1.f : 0.f;
}

// x4 specialisation of instance_masks_kernel (same tap pattern as postprocess_pixels_x4_kernel): a thread writes the 4 pixels of a
// cell column as one 16-byte store
__global__ void __launch_bounds__(256) instance_masks_x4_kernel(const f16* __restrict__ logits, const int* __restrict__ idx,
                                                               float* __restrict__ out, PostGeom g) {
    const int n = blockIdx.y;
    const int cw = g.ow >> 2;  // ow % 4 == 0 on this path
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.oh * cw) return;
    const int oy = t / cw, cx = t - oy * cw;
    const int cy = oy >> 2, ky = oy & 3;
    const int ra = max(ky < 2 ? cy - 1 : cy, 0), rb = min(ky < 2 ? cy : cy + 1, g.h4 - 1);
    const float ty = ky == 0 ? 0.625f : ky == 1 ? 0.875f : ky == 2 ? 0.125f : 0.375f;
    const int cl = max(cx - 1, 0), cr = min(cx + 1, g.w4 - 1);
    const f16* lr = logits + (int64_t)idx[n] * g.h4 * g.w4;
    const float al = (float)lr[ra * g.w4 + cl], ac = (float)lr[ra * g.w4 + cx], ar = (float)lr[ra * g.w4 + cr];
    const float bl = (float)lr[rb * g.w4 + cl], bc = (float)lr[rb * g.w4 + cx], br = (float)lr[rb * g.w4 + cr];
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float tx = k == 0 ? 0.625f : k == 1 ? 0.875f : k == 2 ? 0.125f : 0.375f;
        const float v00 = k < 2 ? al : ac, v01 = k < 2 ? ac : ar, v10 = k < 2 ? bl : bc, v11 = k < 2 ? bc : br;
        const float top = v00 + tx * (v01 - v00), bot = v10 + tx * (v11 - v10);
        o[k] = (top + ty * (bot - top)) > 0.f ? 1.f : 0.f;
    }
    *reinterpret_cast<float4*>(out + (int64_t)n * g.oh * g.ow + (int64_t)oy * g.ow + 4 * cx) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---- launchers -------------------------------------------------------------------------------------------------------
int launch_resize_bilinear_norm(odise_hip_ctx* ctx, const float* x, f16* y, int B, int H, int W, int S) {
    dim3 grid((unsigned)ceil_div(S * S, 256), (unsigned)B);
    hipLaunchKernelGGL(resize_bilinear_norm_kernel, grid, dim3(256), 0, ctx->stream, x, y, H, W, S);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_maskclip_token_mask(odise_hip_ctx* ctx, const f16* logits, uint8_t* out, int B, int Q, int h, int w, int S, int patch, int T,
                               int64_t ldm) {
    hipLaunchKernelGGL(maskclip_token_mask_kernel, dim3((unsigned)(B * (T + Q))), dim3(256), 0, ctx->stream, logits, out, Q, h, w, S, patch, T, ldm);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_l2_normalize_f16(odise_hip_ctx* ctx, const f16* x, f16* y, int64_t rows, int C) {
    hipLaunchKernelGGL(l2_normalize_kernel<f16>, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, ctx->stream, x, y, rows, C);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_l2_normalize_f32(odise_hip_ctx* ctx, const float* x, f16* y, int64_t rows, int C) {
    hipLaunchKernelGGL(l2_normalize_kernel<float>, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, ctx->stream, x, y, rows, C);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_classify_rows(odise_hip_ctx* ctx, const float* L1, const float* L2, const int* seg, const int* ovl, const float* binary, float* out, int64_t rows, int K,
                         int Ktot, float ls1, float ls2, float alpha, float beta) {
    hipLaunchKernelGGL(classify_rows_kernel, dim3((unsigned)rows), dim3(256), (3 * (size_t)K + 8) * sizeof(float), ctx->stream, L1, L2, seg, ovl, binary, out,
                       K, Ktot, ls1, ls2, alpha, beta);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_postprocess_pixels(odise_hip_ctx* ctx, const f16* logits, const float* kscore, f16* S, int* ids, int* counts, const PostGeom& g) {
    const int npix = g.oh * g.ow;
    if (g.oh == g.ih && g.ow == g.iw && g.ph == 4 * g.h4 && g.pw == 4 * g.w4 && !getenv("ODISE_POST_GENERIC")) {
        const int nthreads = g.oh * ((g.ow + 3) / 4);
        hipLaunchKernelGGL(postprocess_pixels_x4_kernel, dim3((unsigned)ceil_div(nthreads, 256)), dim3(256), 3 * (size_t)g.Q * sizeof(int),
                           ctx->stream, logits, kscore, S, ids, counts, g);
        ODISE_CHECK_HIP(hipGetLastError());
        return ODISE_OK;
    }
    hipLaunchKernelGGL(postprocess_pixels_kernel, dim3((unsigned)ceil_div(npix, 256)), dim3(256), 3 * (size_t)g.Q * sizeof(int), ctx->stream, logits,
                       kscore, S, ids, counts, g);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_column_stats(odise_hip_ctx* ctx, const f16* S, float* partial, float* out2, int npix, int Qpad) {
    const int V = Qpad / 8, PL = 256 / V;
    const int nblocks = 512, ppb = (int)ceil_div(npix, nblocks);
    const int nb = (int)ceil_div(npix, ppb);
    hipLaunchKernelGGL(column_stats_kernel, dim3(nb), dim3(256), (size_t)PL * 2 * Qpad * sizeof(float), ctx->stream, S, partial, npix, Qpad, ppb);
    ODISE_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(column_fold_kernel, dim3((unsigned)ceil_div(2 * Qpad, 4)), dim3(256), 0, ctx->stream, partial, out2, nb, 2 * Qpad);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_panoptic_write(odise_hip_ctx* ctx, const int* ids, const int* map, int* seg, int npix) {
    hipLaunchKernelGGL(panoptic_write_kernel, dim3((unsigned)ceil_div(npix, 256)), dim3(256), 0, ctx->stream, ids, map, seg, npix);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_instance_masks(odise_hip_ctx* ctx, const f16* logits, const int* idx, float* out, int n, const PostGeom& g) {
    if (n == 0) return ODISE_OK;
    if (g.oh == g.ih && g.ow == g.iw && g.ph == 4 * g.h4 && g.pw == 4 * g.w4 && g.ow % 4 == 0 && ((uintptr_t)out & 15) == 0 &&
        !getenv("ODISE_POST_GENERIC")) {
        dim3 grid4((unsigned)ceil_div(g.oh * (g.ow / 4), 256), (unsigned)n);
        hipLaunchKernelGGL(instance_masks_x4_kernel, grid4, dim3(256), 0, ctx->stream, logits, idx, out, g);
        ODISE_CHECK_HIP(hipGetLastError());
        return ODISE_OK;
    }
    dim3 grid((unsigned)ceil_div(g.oh * g.ow, 256), (unsigned)n);
    hipLaunchKernelGGL(instance_masks_kernel, grid, dim3(256), 0, ctx->stream, logits, idx, out, g);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

}  // namespace odise
