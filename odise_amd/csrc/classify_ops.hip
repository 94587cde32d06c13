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
// A thread owns a COLUMN of the resized S x S mask (its x taps are fixed, the y taps of a row are the same for the whole block: scalar work) and
// walks it one patch row at a time; the column maxima of a patch row meet in LDS.  Same taps, same interpolation order per sample as one thread
// per patch (round 5: 196 samples x 4 strided loads each, 300 us for 4 pictures on the serial tail) - a maximum does not care about the order.
__global__ void __launch_bounds__(256) maskclip_token_mask_kernel(const f16* __restrict__ logits, uint8_t* __restrict__ out, int Q, int h,
                                                                 int w, int S, int patch, int T, int64_t ldm, int plain) {
    extern __shared__ float colmax[];   // [S]
    const int TA = T + Q;
    const int b = blockIdx.x / TA, row = blockIdx.x % TA;
    uint8_t* orow = out + ((int64_t)b * TA + row) * ldm;
    if (row < T) {
        for (int i = threadIdx.x; i < ldm; i += blockDim.x) orow[i] = i < T ? 0 : 1;
        return;
    }
    const int q = row - T, G = S / patch;
    const f16* lr = logits + ((int64_t)b * Q + q) * h * w;
    if (S <= 2 * 256 && !plain) {
        // (the CLIP input: 336 columns.)  A column's horizontal interpolant of source row y, H(y) = v[y][x0] + tx (v[y][x1] - v[y][x0]), is the
        // `top` of one sample row and the `bot` of the one before (consecutive sample rows share source rows: 336 samples over 256 rows), so the
        // two held rows are reused whenever the next sample row names them again - the same expression, evaluated once: the same bits.
        int x0[2], x1[2];
        float tx[2], h0[2], h1[2];
        bool cok[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ox = threadIdx.x + 256 * c;
            cok[c] = ox < S;
            bil_setup(cok[c] ? ox : S - 1, w, S, x0[c], x1[c], tx[c]);
            h0[c] = h1[c] = 0.f;
        }
        int hy0 = -1, hy1 = -1;   // the source rows h0 / h1 hold (block-uniform)
        for (int py = 0; py < G; ++py) {
            float mx[2] = {-INFINITY, -INFINITY};
            for (int dy = 0; dy < patch; ++dy) {
                int y0, y1;
                float ty;
                bil_setup(py * patch + dy, h, S, y0, y1, ty);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (!cok[c]) continue;
                    float n0, n1;
                    if (y0 == hy0) n0 = h0[c];
                    else if (y0 == hy1) n0 = h1[c];
                    else {
                        const float a = (float)lr[y0 * w + x0[c]], bb = (float)lr[y0 * w + x1[c]];
                        n0 = a + tx[c] * (bb - a);
                    }
                    if (y1 == y0) n1 = n0;
                    else if (y1 == hy1) n1 = h1[c];
                    else if (y1 == hy0) n1 = h0[c];
                    else {
                        const float a = (float)lr[y1 * w + x0[c]], bb = (float)lr[y1 * w + x1[c]];
                        n1 = a + tx[c] * (bb - a);
                    }
                    h0[c] = n0;
                    h1[c] = n1;
                    mx[c] = fmaxf(mx[c], n0 + ty * (n1 - n0));
                }
                hy0 = y0;
                hy1 = y1;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (cok[c]) colmax[threadIdx.x + 256 * c] = mx[c];
            __syncthreads();
            for (int px = threadIdx.x; px < G; px += blockDim.x) {
                float m = colmax[px * patch];
                for (int dx = 1; dx < patch; ++dx) m = fmaxf(m, colmax[px * patch + dx]);
                orow[1 + py * G + px] = (1.f / (1.f + expf(-m))) < 0.5f ? 1 : 0;
            }
            __syncthreads();
        }
    } else
    for (int py = 0; py < G; ++py) {
        for (int ox = threadIdx.x; ox < S; ox += blockDim.x) {
            int x0, x1;
            float tx;
            bil_setup(ox, w, S, x0, x1, tx);
            float mx = -INFINITY;
            for (int dy = 0; dy < patch; ++dy) {
                int y0, y1;
                float ty;
                bil_setup(py * patch + dy, h, S, y0, y1, ty);
                const float v00 = (float)lr[y0 * w + x0], v01 = (float)lr[y0 * w + x1], v10 = (float)lr[y1 * w + x0], v11 = (float)lr[y1 * w + x1];
                const float top = v00 + tx * (v01 - v00), bot = v10 + tx * (v11 - v10);
                mx = fmaxf(mx, top + ty * (bot - top));
            }
            colmax[ox] = mx;
        }
        __syncthreads();
        for (int px = threadIdx.x; px < G; px += blockDim.x) {
            float m = colmax[px * patch];
            for (int dx = 1; dx < patch; ++dx) m = fmaxf(m, colmax[px * patch + dx]);
            orow[1 + py * G + px] = (1.f / (1.f + expf(-m))) < 0.5f ? 1 : 0;
        }
        __syncthreads();
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

// sigmoid of the per-pixel pass: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the library expf and the IEEE division (~20 VALU
// instructions less per pixel and query, of ~70); the result is rounded to fp16 for S and compared against 0.5 for the panoptic flag, both far
// coarser.  All three forms of the pass (generic, x4, tiled) share it, so they stay bit-identical to each other.
__device__ __forceinline__ float post_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// fp16 sigmoid for the pixel-major matrix S.  The instance head counts a pixel as inside the mask iff its LOGIT is positive
// (maskformer_model.py:371 `mask_pred > 0`) and averages sigmoid over exactly those pixels (:376-377); sigmoid(v) for 0 < v < ~1e-3
// rounds to 0.5 in fp16, so such pixels get the next representable value above 0.5: `S > 0.5` then reproduces `logit > 0` exactly.
__device__ __forceinline__ f16 sig_f16(float sg, float v) {
    const f16 h = (f16)sg;
    const bool lift = (v > 0.f) & !(h > (f16)0.5f);   // bitwise: one select, no branch
    return lift ? (f16)0.50048828125f : h;
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
                const float sg = post_sigmoid(v);
                sv[j] = sig_f16(sg, v);
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
                    const float sg = post_sigmoid(v);
                    sv[k][j] = sig_f16(sg, v);
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

// ---- tiled form of postprocess_pixels_x4_kernel ------------------------------------------------------------------------------------
// The thread-per-cell-column kernel above walks all Q queries in one thread (6 loads, 4 sigmoids per query, 100 times in a row) and
// writes the pixel-major matrix S as 16-byte pieces of 208-byte rows, 13 separate partial-line stores per pixel: 375 us per 1024^2 image,
// 0.58 TB/s of its 218 MB (profiles/r02_bench_full_by_shape.txt).  Here a block owns 256 consecutive pixels of one output row (64 cell
// columns) and ALL queries: the 13 groups of 8 queries are dealt round-robin to the block's four wavefronts (4x the threads of the old
// form), every thread evaluates its 4 pixels x 8 queries per group with the same arithmetic as before - bit-identical results - and
// puts the fp16 sigmoids into an LDS image of the tile's S rows; the tile is then written out as ONE contiguous 53 KB run of the matrix
// (256 pixels x 208 bytes are adjacent in memory), 16 bytes per lane, whole 128-byte lines.  The panoptic arg-max is finished across the
// four query slices through LDS (ties to the lowest query, like the sequential walk), the area counters as before.
constexpr float kStatUnit = 2048.0f;  // instance statistics: fp16 values above 0.5 are whole multiples of 2^-11 (column_stats_kernel)
constexpr int PT_PIX = 256;          // pixels of a tile
constexpr int PT_CELLS = 64;         // cell columns of a tile (one per lane)
// Round 6: the semantic head rides in the same kernel (`sem` != nullptr).  sem_seg[c, p] = sum_q P[q, c] sigmoid(mask)[q, p]
// (maskformer_model.py:280-284) was a GEMM over the pixel-major matrix S this kernel had just written: 218 MB out, 218 MB back in, then 558 MB
// of scores per 1024^2 picture - 353 us at 1.8 TB/s behind the 204 us pixel pass.  The tile's S rows are still in LDS when the block is done
// with them: every wave takes 64 of the tile's pixels as the column operand of v_mfma_f32_32x32x16_f16 (their S rows in registers, exactly the
// operand form of semantic_argmax_kernel below: same products, same order), walks the class tiles of PT = P^T [K, Qpad] from the L2 and stores
// whole 128-byte lines of the class rows.  S is still written when the instance head needs it (column_stats_kernel).
template <int KS>   // k-steps of 16 of the fused semantic product: Qpad <= 16 * KS (KS = 0: no semantic epilogue compiled in)
__global__ void __launch_bounds__(256) postprocess_pixels_x4_tiled_kernel(const f16* __restrict__ logits, const float* __restrict__ kscore,
                                                                         f16* __restrict__ S, int* __restrict__ ids, int* __restrict__ counts,
                                                                         PostGeom g, int tiles_per_row, const f16* __restrict__ PT,
                                                                         float* __restrict__ sem, int K, unsigned int* __restrict__ stats_partial) {
    extern __shared__ __attribute__((aligned(16))) char psm[];
    const int Q = g.Q, Qpad = g.Qpad;
    const int pitch = Qpad * 2 + 8;                      // bytes per pixel row of the LDS image: 8-byte aligned, rows 4 pixels apart fall 2-way on the banks
    char* tile = psm;                                    // [PT_PIX][pitch]
    float* red_v = reinterpret_cast<float*>(psm + (size_t)PT_PIX * pitch);   // [4][PT_PIX] best score of each query slice
    int* red_q = reinterpret_cast<int*>(red_v + 4 * PT_PIX);                // [4][PT_PIX] its query | pos << 16, -1 = none
    int* hist = red_q + 4 * PT_PIX;                                         // [3][Q]
    unsigned int* st = reinterpret_cast<unsigned int*>(hist + 3 * Q);       // [2][Qpad] instance statistics of the tile (stats_partial)
    const int tid = threadIdx.x, lane = tid & 63, slice = tid >> 6;
    for (int i = tid; i < 3 * Q + 2 * Qpad; i += 256) hist[i] = 0;
    const int oy = blockIdx.x / tiles_per_row, cx0 = (blockIdx.x - oy * tiles_per_row) * PT_CELLS;
    const int cw = (g.ow + 3) >> 2;
    const int cx = cx0 + lane;
    const bool okt = cx < cw;
    const int cxc = okt ? cx : cw - 1;
    const int cy = oy >> 2, ky = oy & 3;
    const int ra = max(ky < 2 ? cy - 1 : cy, 0), rb = min(ky < 2 ? cy : cy + 1, g.h4 - 1);
    const float ty = ky == 0 ? 0.625f : ky == 1 ? 0.875f : ky == 2 ? 0.125f : 0.375f;
    const int cl = max(cxc - 1, 0), cr = min(cxc + 1, g.w4 - 1);
    const int oa = ra * g.w4, ob = rb * g.w4;
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ok[k] = okt && (4 * cx + k) < g.ow;
    float best[4] = {-1.f, -1.f, -1.f, -1.f};
    int best_q[4] = {-1, -1, -1, -1};
    int best_pos[4] = {0, 0, 0, 0};   // int, not bool: selects instead of exec-masked moves
    const int plane = g.h4 * g.w4;
    __syncthreads();   // hist is zero
    for (int q0 = slice * 8; q0 < Qpad; q0 += 32) {
        f16x8 sv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sv[k] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        // the six taps and the score of ALL eight queries of the group first (clamped query index: always in range, wave-uniform): 56 loads in
        // flight per lane.  Issued one query at a time - load, wait, sigmoid, next - the pass was a chain of ~26 memory latencies per wave with
        // two waves per SIMD to hide them (LDS bounds the occupancy): 300 us of a pass whose VALU work is ~80.
        f16 tap[8][6];
        float ksv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int qc = min(q0 + j, Q - 1);
            const f16* lr = logits + (int64_t)qc * plane;
            tap[j][0] = lr[oa + cl]; tap[j][1] = lr[oa + cxc]; tap[j][2] = lr[oa + cr];
            tap[j][3] = lr[ob + cl]; tap[j][4] = lr[ob + cxc]; tap[j][5] = lr[ob + cr];
            ksv[j] = kscore[qc];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = q0 + j;
            if (q < Q) {  // uniform branch
                const float al = (float)tap[j][0], ac = (float)tap[j][1], ar = (float)tap[j][2];
                const float bl = (float)tap[j][3], bc = (float)tap[j][4], br = (float)tap[j][5];
                const float ks = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ksv[j])));   // wave-uniform by construction: keep the test scalar
                int npos = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float tx = k == 0 ? 0.625f : k == 1 ? 0.875f : k == 2 ? 0.125f : 0.375f;
                    const float v00 = k < 2 ? al : ac, v01 = k < 2 ? ac : ar, v10 = k < 2 ? bl : bc, v11 = k < 2 ? bc : br;
                    const float top = v00 + tx * (v01 - v00), bot = v10 + tx * (v11 - v10);
                    const float v = ok[k] ? top + ty * (bot - top) : -1.f;
                    const float sg = post_sigmoid(v);
                    sv[k][j] = sig_f16(sg, v);
                    const bool pos = sg >= 0.5f;
                    if (ks >= 0.f) {   // wave-uniform
                        npos += __popcll(__ballot(ok[k] & pos));
                        const float pv = ks * sg;
                        const bool upd = ok[k] & (pv > best[k]);
                        best[k] = upd ? pv : best[k];
                        best_q[k] = upd ? q : best_q[k];
                        best_pos[k] = upd ? (int)pos : best_pos[k];
                    }
                }
                if (ks >= 0.f && lane == 0 && npos) atomicAdd(&hist[Q + q], npos);
            }
        }
        if (S || (KS > 0 && sem) || stats_partial) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
                char* dst = tile + (size_t)(4 * lane + k) * pitch + q0 * 2;
                f16x4v lo = {sv[k][0], sv[k][1], sv[k][2], sv[k][3]}, hi = {sv[k][4], sv[k][5], sv[k][6], sv[k][7]};
                *reinterpret_cast<f16x4v*>(dst) = lo;
                *reinterpret_cast<f16x4v*>(dst + 8) = hi;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red_v[slice * PT_PIX + 4 * lane + k] = best[k];
        red_q[slice * PT_PIX + 4 * lane + k] = best_q[k] < 0 ? -1 : (best_q[k] | (best_pos[k] << 16));
    }
    __syncthreads();
    // ---- panoptic arg-max across the four query slices: thread per pixel of the tile
    {
        const int px = 4 * cx0 + tid;                    // output column
        if (px < g.ow) {
            float bv = -1.f;
            int bq = -1;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const float v = red_v[sl * PT_PIX + tid];
                const int qq = red_q[sl * PT_PIX + tid];
                if (qq >= 0 && (v > bv || (v == bv && (qq & 0xffff) < (bq & 0xffff)))) { bv = v; bq = qq; }
            }
            if (ids) ids[(int64_t)oy * g.ow + px] = bq;
            if (bq >= 0) {
                atomicAdd(&hist[bq & 0xffff], 1);
                if (bq & (1 << 16)) atomicAdd(&hist[2 * Q + (bq & 0xffff)], 1);
            }
        }
    }
    // ---- instance statistics of the tile (column_stats_kernel's sums, taken from the LDS image instead of a second pass over S in HBM)
    if (stats_partial) {
        const int V = Qpad >> 3, PL = 256 / V;
        const int v = tid % V, pl = tid / V;
        const int npx = min(PT_PIX, g.ow - 4 * cx0);
        if (pl < PL) {
            unsigned int s[8], c[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = c[i] = 0u;
            for (int p = pl; p < npx; p += PL) {
                typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
                const char* src = tile + (size_t)p * pitch + v * 16;
                const f16x4v lo = *reinterpret_cast<const f16x4v*>(src), hi = *reinterpret_cast<const f16x4v*>(src + 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float a = (float)(i < 4 ? lo[i & 3] : hi[i & 3]);
                    if (a > 0.5f) { s[i] += (unsigned int)(a * kStatUnit); c[i] += 1u; }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c[i]) {
                    atomicAdd(&st[v * 8 + i], s[i]);
                    atomicAdd(&st[Qpad + v * 8 + i], c[i]);
                }
        }
    }
    // ---- the tile's S rows: one contiguous run of the matrix
    if (S) {
        const int npx = min(PT_PIX, g.ow - 4 * cx0);
        const int cpp = Qpad >> 3;                       // 16-byte chunks per pixel row
        const int nchunks = npx * cpp;
        f16* dst = S + ((int64_t)oy * g.ow + 4 * cx0) * Qpad;
        for (int j = tid; j < nchunks; j += 256) {
            const int pxl = j / cpp, part = j - pxl * cpp;
            typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
            const char* src = tile + (size_t)pxl * pitch + part * 16;
            const f16x4v lo = *reinterpret_cast<const f16x4v*>(src), hi = *reinterpret_cast<const f16x4v*>(src + 8);
            f16x8 o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<f16x8*>(dst + (int64_t)j * 8) = o;
        }
    }
    // ---- semantic scores of the tile's pixels from the LDS image (the tile is complete since the barrier above)
    if (KS > 0 && sem) {
        const int hi = lane >> 5, l31 = lane & 31;
        const int64_t npix = (int64_t)g.oh * g.ow;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int pxl = slice * 64 + half * 32 + l31;          // pixel of the tile
            const int px = 4 * cx0 + pxl;                          // output column
            const bool pok = px < g.ow;
            f16x8 pf[KS > 0 ? KS : 1];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int ko = ks * 16 + hi * 8;
                pf[ks] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (pok && ko + 8 <= Qpad) {
                    typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
                    const char* src = tile + (size_t)pxl * pitch + ko * 2;
                    const f16x4v lo = *reinterpret_cast<const f16x4v*>(src), hh = *reinterpret_cast<const f16x4v*>(src + 8);
                    pf[ks] = f16x8{lo[0], lo[1], lo[2], lo[3], hh[0], hh[1], hh[2], hh[3]};
                }
            }
            float* out = sem + (int64_t)oy * g.ow + px;
            for (int c0 = 0; c0 < K; c0 += 32) {
                const int c = c0 + l31;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int ko = ks * 16 + hi * 8;
                    f16x8 cf = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (c < K && ko + 8 <= Qpad) cf = *reinterpret_cast<const f16x8*>(PT + (int64_t)c * Qpad + ko);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cf, pf[ks], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cls = c0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (cls < K && pok) __builtin_nontemporal_store(acc[r], out + (int64_t)cls * npix);   // lanes 0..31: 32 consecutive pixels of one class row = one 128-byte line
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 3 * Q; i += 256)
        if (hist[i]) atomicAdd(&counts[i], hist[i]);
    if (stats_partial)
        for (int i = tid; i < 2 * Qpad; i += 256) stats_partial[(int64_t)blockIdx.x * 2 * Qpad + i] = st[i];
}

// per-query sum over pixels of sigmoid * [sigmoid > 0.5] and count of [sigmoid > 0.5] from S (instance mask scores,
// maskformer_model.py:376-377).  The fp16 values above 0.5 are whole multiples of 2^-11, so the sums are taken as INTEGERS in that unit: exact,
// whatever the order - every form of the pixel pass (this kernel over S, or the tiled pass's own statistics epilogue) gives the same bits.
// Block partials [nblocks][2][Qpad] u32 (a block covers < 2^20 pixels), folded by column_fold_kernel.
__global__ void __launch_bounds__(256) column_stats_kernel(const f16* __restrict__ S, unsigned int* __restrict__ partial, int npix, int Qpad,
                                                          int pix_per_block) {
    extern __shared__ unsigned int cs[];  // [2][Qpad]
    const int V = Qpad >> 3, PL = 256 / V;
    const int v = threadIdx.x % V, pl = threadIdx.x / V;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
    for (int i = threadIdx.x; i < 2 * Qpad; i += blockDim.x) cs[i] = 0u;
    __syncthreads();
    unsigned int s[8], c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = c[i] = 0u;
    if (pl < PL) {
        for (int p = p0 + pl; p < p1; p += PL) {
            const f16x8 t = *reinterpret_cast<const f16x8*>(S + (int64_t)p * Qpad + v * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float a = (float)t[i];
                if (a > 0.5f) { s[i] += (unsigned int)(a * kStatUnit); c[i] += 1u; }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (c[i]) {
                atomicAdd(&cs[v * 8 + i], s[i]);
                atomicAdd(&cs[Qpad + v * 8 + i], c[i]);
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Qpad; i += blockDim.x) partial[(int64_t)blockIdx.x * 2 * Qpad + i] = cs[i];
}
// one wavefront per output column: lanes stride over the block partials (64-bit integer sums: exact); out [2][Qpad] f32 = (sum, count)
__global__ void __launch_bounds__(256) column_fold_kernel(const unsigned int* __restrict__ partial, float* __restrict__ out, int nblocks, int n) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    unsigned long long a = 0ull;
    for (int b = lane; b < nblocks; b += 64) a += partial[(int64_t)b * n + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int lo = __shfl_xor((unsigned int)a, o), hi = __shfl_xor((unsigned int)(a >> 32), o);
        a += ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 0) out[i] = i < n / 2 ? (float)((double)a * (1.0 / (double)kStatUnit)) : (float)a;
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
                                                            PostGeom g, const int* __restrict__ n_dev) {
    const int n = blockIdx.y;
    if (n_dev && n >= *n_dev) return;  // launched for the maximum count; the selection's size lives on the device
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
                                                               float* __restrict__ out, PostGeom g, const int* __restrict__ n_dev) {
    const int n = blockIdx.y;
    if (n_dev && n >= *n_dev) return;
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


// ---- device-side decisions of the three inference heads (maskformer_model.py:286-380) -------------------------------------------------
// The reference takes them on the host from device tensors (`.item()` per segment, maskformer_model.py:312-340); round 1 of this
// library took them in numpy after a read-back of mask_cls.  Here they stay on the device, so one model call is one stream of kernels
// and a single small read-back (segment / instance tables) at the end, and the multi-GPU record is written where it is produced.

// One block per image.  mask_cls [Q, K+1] log-probabilities -> F.softmax again (maskformer_model.py:287, 281, 349), then
//   kscore [Q]       max probability if the query is kept (label != K and score > threshold, :289-290), -1 otherwise
//   label  [Q]       argmax (first maximum)
//   semT   [K, Qpad] f16 probabilities of the K real classes, transposed (A operand of the semantic GEMM, :281-283); optional
//   probs  [Q, K]    f32 probabilities (instance top-k, :349); optional
__global__ void __launch_bounds__(256) post_decide_kernel(const float* __restrict__ mask_cls, float* __restrict__ kscore, int* __restrict__ label,
                                                         f16* __restrict__ semT, float* __restrict__ probs, int Q, int Qpad, int K,
                                                         float object_mask_threshold, int64_t cls_stride, int64_t out_stride_q, int64_t semT_stride,
                                                         int64_t probs_stride) {
    const int b = blockIdx.x;
    const float* mc = mask_cls + (int64_t)b * cls_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (semT && blockIdx.y == 0) {   // zero the padding columns q in [Q, Qpad)
        f16* st = semT + (int64_t)b * semT_stride;
        for (int i = threadIdx.x; i < K * (Qpad - Q); i += blockDim.x) st[(int64_t)(i / (Qpad - Q)) * Qpad + Q + i % (Qpad - Q)] = (f16)0.f;
    }
    for (int q = blockIdx.y * 4 + wave; q < Q; q += 4 * gridDim.y) {   // one wavefront per query (grid.y = Q / 4: a single block per image took 71 us on the serial tail)
        const float* row = mc + (int64_t)q * (K + 1);
        float m = -INFINITY;
        for (int k = lane; k <= K; k += 64) m = fmaxf(m, row[k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float ssum = 0.f;
        for (int k = lane; k <= K; k += 64) ssum += expf(row[k] - m);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o);
        float best = -1.f;
        int best_k = 0x7fffffff;
        for (int k = lane; k <= K; k += 64) {
            const float pr = expf(row[k] - m) / ssum;
            if (k < K) {
                if (semT) semT[(int64_t)b * semT_stride + (int64_t)k * Qpad + q] = (f16)pr;
                if (probs) probs[(int64_t)b * probs_stride + (int64_t)q * K + k] = pr;
            }
            if (pr > best) { best = pr; best_k = k; }   // ascending k per lane: the first maximum of the lane
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int ok = __shfl_xor(best_k, o);
            if (ob > best || (ob == best && ok < best_k)) { best = ob; best_k = ok; }
        }
        if (lane == 0) {
            const bool keep = best_k != K && best > object_mask_threshold;
            kscore[(int64_t)b * out_stride_q + q] = keep ? best : -1.f;
            label[(int64_t)b * out_stride_q + q] = best_k;
        }
    }
}

// One thread per image walks the kept queries in order (maskformer_model.py:312-340): areas from the integer counters of the
// per-pixel pass, overlap test, stuff merging by class, sequential segment ids.  map [Q] = segment id of every query (0 = none);
// table = n_segments | (id, isthing, category_id) x n  (the tail of the image's prediction record, odise_amd/distributed.py).
__global__ void panoptic_decide_kernel(const int* __restrict__ cnt, const float* __restrict__ ks, const int* __restrict__ lb,
                                       const uint8_t* __restrict__ isthing, int* __restrict__ mp, int* __restrict__ table, int Q, int K,
                                       double overlap_threshold, int max_segments, int* __restrict__ stuff) {
    // stuff[k] = segment id of a stuff class already emitted, 0 = none
    for (int k = threadIdx.x; k < K; k += blockDim.x) stuff[k] = 0;
    __syncthreads();
    if (threadIdx.x != 0) return;
    int current = 0, n = 0;
    for (int q = 0; q < Q; ++q) {
        mp[q] = 0;
        if (!(ks[q] >= 0.f)) continue;
        const int cls = lb[q];
        const int mask_area = cnt[q], original_area = cnt[Q + q], inter = cnt[2 * Q + q];
        if (mask_area > 0 && original_area > 0 && inter > 0) {
            if ((double)mask_area / (double)original_area < overlap_threshold) continue;
            const bool thing = isthing[cls] != 0;
            if (!thing) {
                if (stuff[cls]) { mp[q] = stuff[cls]; continue; }
                stuff[cls] = current + 1;
            }
            ++current;
            mp[q] = current;
            if (table && n < max_segments) {
                table[1 + 3 * n] = current;
                table[2 + 3 * n] = thing ? 1 : 0;
                table[3 + 3 * n] = cls;
            }
            ++n;
        }
    }
    if (table) {
        const int m = n < max_segments ? n : max_segments;
        table[0] = m;
        for (int i = m; i < max_segments; ++i) table[1 + 3 * i] = table[2 + 3 * i] = table[3 + 3 * i] = 0;
    }
}

// Instance head (maskformer_model.py:344-380): one block per image (blockIdx.x; one launch for the batch) selects the top-k of the Q*K class probabilities (radix select on
// the float bits - probabilities are non-negative, so unsigned order = float order), sorts them (score descending, flat index
// ascending on ties), keeps "thing" classes when the panoptic head is on (:363-369) and multiplies with the mask score
// sum(sigmoid * [logit > 0]) / (count + 1e-6) (:376-377) from the per-pixel pass.
//   table [1 + 2*topk] int32: n | query index x topk | class x topk ;  scores [topk] f32 (entries >= n are zero)
__global__ void __launch_bounds__(1024) instance_topk_kernel(const float* __restrict__ probs_all, const float* __restrict__ inst_stats_all,
                                                            const uint8_t* __restrict__ isthing, int* __restrict__ table_all, float* __restrict__ scores_all,
                                                            int Q, int Qpad, int K, int topk, int things_only) {
    const float* probs = probs_all + (size_t)blockIdx.x * Q * K;
    const float* inst_stats = inst_stats_all + (size_t)blockIdx.x * 2 * Qpad;
    int* table = table_all + (size_t)blockIdx.x * (1 + 2 * topk);
    float* scores = scores_all + (size_t)blockIdx.x * topk;
    extern __shared__ unsigned int sm_u[];
    unsigned int* hist = sm_u;                 // [256]
    unsigned int* sel_key = sm_u + 256;        // [topk]
    int* sel_idx = (int*)(sm_u + 256 + topk);  // [topk]
    __shared__ unsigned int s_prefix, s_need, s_count, s_base;
    const float* pr = probs;
    const unsigned int* keys = reinterpret_cast<const unsigned int*>(pr);
    const int N = Q * K;
    const int k_sel = topk < N ? topk : N;
    const int tid = threadIdx.x, nt = blockDim.x;
    // the keys this thread looks at, once (Q * K <= 16 x the block: the released vocabularies): the four digit passes and the compaction below read
    // them again and again, and each pass was a chain of load - wait - atomic per key (13 memory latencies per pass: most of the kernel's 66 us)
    constexpr int KREG = 16;
    const bool small = N <= KREG * nt;
    unsigned int kreg[KREG];
#pragma unroll
    for (int r = 0; r < KREG; ++r) {
        const int i = tid + r * nt;
        kreg[r] = (small && i < N) ? keys[i] : 0u;
    }
    // ---- radix select: the k_sel-th largest key
    if (tid == 0) { s_prefix = 0; s_need = (unsigned)k_sel; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 256; i += nt) hist[i] = 0;
        __syncthreads();
        const unsigned int prefix = s_prefix;
        const unsigned int himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        if (small) {
#pragma unroll
            for (int r = 0; r < KREG; ++r) {
                const unsigned int kx = kreg[r];
                if (tid + r * nt < N && (kx & himask) == prefix) atomicAdd(&hist[(kx >> shift) & 255u], 1u);
            }
        } else {
            for (int i = tid; i < N; i += nt) {
                const unsigned int kx = keys[i];
                if ((kx & himask) == prefix) atomicAdd(&hist[(kx >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {
            // the largest digit d with  sum of hist[d' > d] < need <= sum of hist[d' >= d]  (d = 0 if none: the serial walk's fall-through).
            // Lane l owns digits 4l .. 4l+3; `above` = everything in higher lanes (inclusive suffix scan over the lanes minus its own four).
            const unsigned int need = s_need;
            const unsigned int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const unsigned int own = h0 + h1 + h2 + h3;
            unsigned int suf = own;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int up = __shfl_down(suf, o);
                if (tid + o < 64) suf += up;
            }
            const unsigned int above = suf - own;
            // within the lane, from digit 4l+3 downwards
            int d = -1;
            unsigned int acc = above;
            if (acc < need) {
                if (acc + h3 >= need) d = 4 * tid + 3;
                else if (acc + h3 + h2 >= need) { d = 4 * tid + 2; acc += h3; }
                else if (acc + h3 + h2 + h1 >= need) { d = 4 * tid + 1; acc += h3 + h2; }
                else if (acc + own >= need) { d = 4 * tid; acc += h3 + h2 + h1; }
            }
            // exactly one lane finds its digit unless the total is short of `need` (then digit 0 with everything above it, as the serial walk)
            const unsigned long long found = __ballot(d >= 0);
            if (found == 0ull) {
                if (tid == 0) { s_need = need - (suf - h0); s_prefix = prefix; }
            } else if (d >= 0) {
                s_need = need - acc;                     // how many of digit d (and the finer digits below) are still needed
                s_prefix = prefix | ((unsigned)d << shift);
            }
        }
        __syncthreads();
    }
    const unsigned int T = s_prefix;          // key of the k_sel-th largest element
    const unsigned int need_eq = s_need;      // elements equal to T that belong to the selection (taken in index order)
    // ---- compaction: keys > T fill sel[0, c_gt), the first need_eq keys == T (in index order) fill sel[c_gt, k_sel)
    const unsigned int c_gt = (unsigned)k_sel - need_eq;
    if (tid == 0) { s_count = 0; s_base = 0; }
    __syncthreads();
    for (int base = 0, rr = 0; base < N; base += nt, ++rr) {
        const int i = base + tid;
        unsigned int kx = 0u;
        if (small) {   // kreg[rr] without a dynamic register index: rr is block-uniform
#pragma unroll
            for (int r = 0; r < KREG; ++r) kx = r == rr ? kreg[r] : kx;
        } else if (i < N) {
            kx = keys[i];
        }
        const bool gt = i < N && kx > T, eq = i < N && kx == T;
        const unsigned long long bg = __ballot(gt), be = __ballot(eq);
        const int lane = tid & 63, w = tid >> 6;
        __shared__ unsigned int wg[16], we[16];
        if (lane == 0) { wg[w] = __popcll(bg); we[w] = __popcll(be); }
        __syncthreads();
        unsigned int og = 0, oe = 0, tg = 0, te = 0;   // exclusive offsets of this thread, totals of the chunk
        for (int j = 0; j < (nt >> 6); ++j) {
            if (j < w) { og += wg[j]; oe += we[j]; }
            tg += wg[j];
            te += we[j];
        }
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        og += __popcll(bg & below);
        oe += __popcll(be & below);
        const unsigned int g0 = s_count, e0 = s_base;
        if (gt) {
            sel_key[g0 + og] = kx;
            sel_idx[g0 + og] = i;
        } else if (eq && e0 + oe < need_eq) {
            sel_key[c_gt + e0 + oe] = kx;
            sel_idx[c_gt + e0 + oe] = i;
        }
        __syncthreads();
        if (tid == 0) { s_count = g0 + tg; s_base = e0 + te; }
        __syncthreads();
    }
    // ---- rank sort of the k_sel selected entries (score descending, index ascending), thing filter, mask scores
    int* tb = table;
    float* sc = scores;
    const float* st = inst_stats;
    // rank of entry t = number of entries that precede it
    for (int t = tid; t < k_sel; t += nt) {
        const unsigned int kt = sel_key[t];
        const int it = sel_idx[t];
        int rank = 0;
        for (int u = 0; u < k_sel; ++u) {
            const unsigned int ku = sel_key[u];
            const int iu = sel_idx[u];
            rank += (ku > kt || (ku == kt && iu < it)) ? 1 : 0;
        }
        // stash (rank -> entry) in the output table temporarily: query index slot holds the flat index
        tb[1 + rank] = it;
    }
    for (int t = k_sel + tid; t < topk; t += nt) tb[1 + t] = -1;
    __syncthreads();
    // the kept entries in rank order: chunks of `nt` ranks, a block-wide exclusive count of the kept ones before each (every chunk reads its
    // flat indices before any thread of it writes: an entry moves to a slot at or below its own rank, i.e. one already read)
    __shared__ unsigned int wk[16];
    __shared__ int s_n;
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int base = 0; base < k_sel; base += nt) {
        const int t = base + tid;
        const int flat = t < k_sel ? tb[1 + t] : -1;
        int q = 0, c = 0;
        bool keep = false;
        if (flat >= 0) {
            q = flat / K;
            c = flat - q * K;
            keep = !(things_only && !isthing[c]);
        }
        const unsigned long long bk = __ballot(keep);
        const int lane = tid & 63, w = tid >> 6;
        if (lane == 0) wk[w] = __popcll(bk);
        __syncthreads();
        unsigned int before = 0, total = 0;
        for (int j = 0; j < (nt >> 6); ++j) {
            if (j < w) before += wk[j];
            total += wk[j];
        }
        before += __popcll(bk & (lane ? (~0ull >> (64 - lane)) : 0ull));
        const int n0 = s_n;
        if (keep) {
            const float s = pr[flat];
            const float mask_score = st[q] / (st[Qpad + q] + 1e-6f);
            tb[1 + n0 + before] = q;
            tb[1 + topk + n0 + before] = c;
            sc[n0 + before] = s * mask_score;
        }
        __syncthreads();
        if (tid == 0) s_n = n0 + (int)total;
        __syncthreads();
    }
    const int n = s_n;
    for (int t = n + tid; t < topk; t += nt) { tb[1 + t] = 0; tb[1 + topk + t] = 0; sc[t] = 0.f; }
    if (tid == 0) tb[0] = n;
}


// ---- semantic argmax without the [K, H, W] tensor (A-847 at 1280x1280 would materialise 5.5 GB of fp32) --------
// out[p] = argmax_k sum_q P[q,k] * sigmoid(mask)[q,p]  (maskformer_model.py:280-284 followed by the evaluator's `.argmax(dim=0)`,
// detectron2 SemSegEvaluator.process reached from odise/evaluation/d2_evaluator.py:63).  MFMA with the CLASS tile as the row operand:
// the 32x32 result then has classes along the registers and the pixel along the lane, so the running (max, argmax) over all K classes
// is per-lane register work - no cross-lane reduction until the two lane halves are merged once at the end.  A wave owns 32 pixels:
// their S rows (Qpad fp16 each) stay in registers for the whole class loop; PT ([K, Qpad] fp16, <= 250 KB) streams from L2.
template <int KS>  // k-steps of 16: Qpad <= 16 * KS
__global__ void __launch_bounds__(256) semantic_argmax_kernel(const f16* __restrict__ S, const f16* __restrict__ PT, int* __restrict__ out, int npix,
                                                             int Qpad, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int64_t p = ((int64_t)blockIdx.x * 4 + wave) * 32 + l31;
    f16x8 pf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int ko = ks * 16 + hi * 8;
        pf[ks] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (p < npix && ko + 8 <= Qpad) pf[ks] = *reinterpret_cast<const f16x8*>(S + p * Qpad + ko);
    }
    float best = -INFINITY;
    int best_k = 0;
    for (int c0 = 0; c0 < K; c0 += 32) {
        const int c = c0 + l31;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ko = ks * 16 + hi * 8;
            f16x8 cf = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c < K && ko + 8 <= Qpad) cf = *reinterpret_cast<const f16x8*>(PT + (int64_t)c * Qpad + ko);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cf, pf[ks], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cls = c0 + (r & 3) + 8 * (r >> 2) + 4 * hi;   // ascending in r: `>` keeps the first maximum
            if (cls < K && acc[r] > best) { best = acc[r]; best_k = cls; }
        }
    }
    const float ob = __shfl_xor(best, 32);
    const int ok = __shfl_xor(best_k, 32);
    if (ob > best || (ob == best && ok < best_k)) { best = ob; best_k = ok; }
    if (hi == 0 && p < npix) out[p] = best_k;
}


// dst fp32 [3,Hp,Wp] = src / 255 in the top-left h x w corner, zeros elsewhere: the (x - pixel_mean) / pixel_std normalisation
// (mean 0, std 255: configs/common/models/odise_with_label.py) followed by ImageList.from_tensors (odise.py:238-244).
// layout 0: uint8 [h,w,3]; 1: uint8 [3,h,w]; 2: fp32 [3,h,w]
__global__ void __launch_bounds__(256) image_pad_kernel(const void* __restrict__ src, int layout, int h, int w, float* __restrict__ dst, int Hp, int Wp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t plane = (int64_t)Hp * Wp;
    if (idx >= plane * 3) return;
    const int c = (int)(idx / plane);
    const int64_t p = idx - c * plane;
    const int y = (int)(p / Wp), x = (int)(p - (int64_t)y * Wp);
    float v = 0.f;
    if (y < h && x < w) {
        if (layout == 0) v = (float)((const uint8_t*)src)[((int64_t)y * w + x) * 3 + c];
        else if (layout == 1) v = (float)((const uint8_t*)src)[((int64_t)c * h + y) * w + x];
        else v = ((const float*)src)[((int64_t)c * h + y) * w + x];
        v = v / 255.0f;   // a true division, like the reference's (x - 0) / 255 and x / 255.0
    }
    dst[idx] = v;
}

// ---- launchers -------------------------------------------------------------------------------------------------------
static int g_post_generic = 0;  // tools hook (odise_hip_post_generic): 1 = never take the x4 specialisations (bit-equality tests), 2 = x4 without the tiled form
int launch_resize_bilinear_norm(odise_hip_ctx* ctx, const float* x, f16* y, int B, int H, int W, int S) {
    dim3 grid((unsigned)ceil_div(S * S, 256), (unsigned)B);
    hipLaunchKernelGGL(resize_bilinear_norm_kernel, grid, dim3(256), 0, ctx->stream, x, y, H, W, S);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
static int g_token_mask_plain = 0;   // tools hook (odise_hip_maskclip_token_mask): 1 = every sample row interpolates its two source rows anew (bit-compare)
int launch_maskclip_token_mask(odise_hip_ctx* ctx, const f16* logits, uint8_t* out, int B, int Q, int h, int w, int S, int patch, int T,
                               int64_t ldm) {
    ODISE_REQUIRE(S % patch == 0 && S <= 8192, "maskclip_token_mask: image %d / patch %d", S, patch);
    hipLaunchKernelGGL(maskclip_token_mask_kernel, dim3((unsigned)(B * (T + Q))), dim3(256), (size_t)S * sizeof(float), ctx->stream, logits, out, Q, h, w, S, patch, T, ldm,
                       g_token_mask_plain);
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
static size_t tiled_lds(const PostGeom& g) {
    return (size_t)PT_PIX * (g.Qpad * 2 + 8) + 8 * (size_t)PT_PIX * 4 + 3 * (size_t)g.Q * sizeof(int) + 2 * (size_t)g.Qpad * sizeof(unsigned int);
}
// true when the pixel pass of this geometry runs its tiled form (an exact 4x upsampling whose tile fits 64 KB of LDS): that form can leave the
// instance statistics itself (stats_partial: `postprocess_pixels_stat_blocks` block partials for column_fold), so S need not exist for them
bool postprocess_pixels_tiled(const PostGeom& g) {
    return g.oh == g.ih && g.ow == g.iw && g.ph == 4 * g.h4 && g.pw == 4 * g.w4 && g_post_generic == 0 && tiled_lds(g) <= 64 * 1024;
}
int postprocess_pixels_stat_blocks(const PostGeom& g) { return g.oh * (int)ceil_div((g.ow + 3) / 4, PT_CELLS); }
// ... and the semantic scores too (Qpad <= 112): the caller then passes PT / sem to launch_postprocess_pixels and skips the semantic GEMM
bool postprocess_pixels_fuses_semantic(const PostGeom& g) { return postprocess_pixels_tiled(g) && g.Qpad <= 112 && g.ow % 32 == 0; }
int launch_postprocess_pixels(odise_hip_ctx* ctx, const f16* logits, const float* kscore, f16* S, int* ids, int* counts, const PostGeom& g, const f16* PT,
                              float* sem, int K, unsigned int* stats_partial) {
    const int npix = g.oh * g.ow;
    ODISE_REQUIRE(!sem || (PT && K > 0 && postprocess_pixels_fuses_semantic(g)), "postprocess_pixels: this geometry cannot fuse the semantic head");
    ODISE_REQUIRE(!stats_partial || postprocess_pixels_tiled(g), "postprocess_pixels: this geometry cannot leave the instance statistics");
    if (g.oh == g.ih && g.ow == g.iw && g.ph == 4 * g.h4 && g.pw == 4 * g.w4 && g_post_generic != 1) {
        const size_t lds = tiled_lds(g);
        if (g_post_generic != 2 && lds <= 64 * 1024) {   // tiled form (two blocks per CU); odise_hip_post_generic(2) keeps the thread-per-cell-column form
            const int tiles_per_row = (int)ceil_div((g.ow + 3) / 4, PT_CELLS);
            if (sem) hipLaunchKernelGGL(postprocess_pixels_x4_tiled_kernel<7>, dim3((unsigned)(g.oh * tiles_per_row)), dim3(256), lds, ctx->stream, logits, kscore, S,
                                        ids, counts, g, tiles_per_row, PT, sem, K, stats_partial);
            else hipLaunchKernelGGL(postprocess_pixels_x4_tiled_kernel<0>, dim3((unsigned)(g.oh * tiles_per_row)), dim3(256), lds, ctx->stream, logits, kscore, S, ids,
                                    counts, g, tiles_per_row, nullptr, nullptr, 0, stats_partial);
            ODISE_CHECK_HIP(hipGetLastError());
            return ODISE_OK;
        }
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
int launch_column_stats(odise_hip_ctx* ctx, const f16* S, unsigned int* partial, float* out2, int npix, int Qpad) {
    const int nblocks = 512, ppb = (int)ceil_div(npix, nblocks);
    const int nb = (int)ceil_div(npix, ppb);
    hipLaunchKernelGGL(column_stats_kernel, dim3(nb), dim3(256), (size_t)2 * Qpad * sizeof(unsigned int), ctx->stream, S, partial, npix, Qpad, ppb);
    ODISE_CHECK_HIP(hipGetLastError());
    return launch_column_fold(ctx, partial, out2, nb, Qpad);
}
// out2 [2][Qpad] f32 = (sum of sigmoid over the mask's pixels, their count) from nb block partials [nb][2][Qpad] u32 (column_stats_kernel, or the
// tiled pixel pass's statistics epilogue)
int launch_column_fold(odise_hip_ctx* ctx, const unsigned int* partial, float* out2, int nb, int Qpad) {
    hipLaunchKernelGGL(column_fold_kernel, dim3((unsigned)ceil_div(2 * Qpad, 4)), dim3(256), 0, ctx->stream, partial, out2, nb, 2 * Qpad);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_post_decide(odise_hip_ctx* ctx, const float* mask_cls, float* kscore, int* label, f16* semT, float* probs, int B, int Q, int Qpad, int K,
                       float object_mask_threshold) {
    hipLaunchKernelGGL(post_decide_kernel, dim3((unsigned)B, (unsigned)ceil_div(Q, 4)), dim3(256), 0, ctx->stream, mask_cls, kscore, label, semT, probs, Q, Qpad, K,
                       object_mask_threshold, (int64_t)Q * (K + 1), (int64_t)Q, (int64_t)K * Qpad, (int64_t)Q * K);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_panoptic_decide(odise_hip_ctx* ctx, const int* counts, const float* kscore, const int* label, const uint8_t* isthing, int* map, int* table,
                           int Q, int K, double overlap_threshold, int max_segments, int* stuff_scratch) {
    hipLaunchKernelGGL(panoptic_decide_kernel, dim3(1), dim3(256), 0, ctx->stream, counts, kscore, label, isthing, map, table, Q, K, overlap_threshold,
                       max_segments, stuff_scratch);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_instance_topk(odise_hip_ctx* ctx, const float* probs, const float* inst_stats, const uint8_t* isthing, int* table, float* scores, int B,
                         int Q, int Qpad, int K, int topk, int things_only) {
    ODISE_REQUIRE(topk >= 1 && topk <= 4096, "instance_topk: topk %d out of range", topk);
    const size_t lds = (256 + 2 * (size_t)topk) * sizeof(unsigned int);
    // probs [B][Q*K], inst_stats [B][2*Qpad], table [B][1 + 2*topk], scores [B][topk]
    hipLaunchKernelGGL(instance_topk_kernel, dim3((unsigned)B), dim3(1024), lds, ctx->stream, probs, inst_stats, isthing, table, scores, Q, Qpad, K, topk,
                       things_only);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_semantic_argmax(odise_hip_ctx* ctx, const f16* S, const f16* PT, int* out, int npix, int Qpad, int K) {
    const unsigned blocks = (unsigned)ceil_div(npix, 128);
    ODISE_REQUIRE(Qpad % 8 == 0 && Qpad <= 304, "semantic_argmax: %d queries unsupported", Qpad);
    if (Qpad <= 112) hipLaunchKernelGGL(semantic_argmax_kernel<7>, dim3(blocks), dim3(256), 0, ctx->stream, S, PT, out, npix, Qpad, K);
    else if (Qpad <= 208) hipLaunchKernelGGL(semantic_argmax_kernel<13>, dim3(blocks), dim3(256), 0, ctx->stream, S, PT, out, npix, Qpad, K);
    else hipLaunchKernelGGL(semantic_argmax_kernel<19>, dim3(blocks), dim3(256), 0, ctx->stream, S, PT, out, npix, Qpad, K);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_image_pad(odise_hip_ctx* ctx, const void* src, int layout, int h, int w, float* dst, int Hp, int Wp) {
    hipLaunchKernelGGL(image_pad_kernel, dim3((unsigned)ceil_div((int64_t)3 * Hp * Wp, 256)), dim3(256), 0, ctx->stream, src, layout, h, w, dst, Hp, Wp);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_panoptic_write(odise_hip_ctx* ctx, const int* ids, const int* map, int* seg, int npix) {
    hipLaunchKernelGGL(panoptic_write_kernel, dim3((unsigned)ceil_div(npix, 256)), dim3(256), 0, ctx->stream, ids, map, seg, npix);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
int launch_instance_masks(odise_hip_ctx* ctx, const f16* logits, const int* idx, float* out, int n, const PostGeom& g, const int* n_dev) {
    if (n == 0) return ODISE_OK;
    if (g.oh == g.ih && g.ow == g.iw && g.ph == 4 * g.h4 && g.pw == 4 * g.w4 && g.ow % 4 == 0 && ((uintptr_t)out & 15) == 0 &&
        g_post_generic != 1) {
        dim3 grid4((unsigned)ceil_div(g.oh * (g.ow / 4), 256), (unsigned)n);
        hipLaunchKernelGGL(instance_masks_x4_kernel, grid4, dim3(256), 0, ctx->stream, logits, idx, out, g, n_dev);
        ODISE_CHECK_HIP(hipGetLastError());
        return ODISE_OK;
    }
    dim3 grid((unsigned)ceil_div(g.oh * g.ow, 256), (unsigned)n);
    hipLaunchKernelGGL(instance_masks_kernel, grid, dim3(256), 0, ctx->stream, logits, idx, out, g, n_dev);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

}  // namespace odise

extern "C" int odise_hip_post_generic(int on) { odise::g_post_generic = on; return 0; }
// test hook (include/odise_hip_tools.h): MaskCLIP's visibility rows [B][T + Q][ldm] u8 from mask logits [B,Q,h,w] f16; plain = 1: the form without
// the reuse of interpolated source rows
extern "C" int odise_hip_maskclip_token_mask(odise_hip_ctx* ctx, const void* logits_f16, void* out_u8, int B, int Q, int h, int w, int S, int patch, int T,
                                             int64_t ldm, int plain) {
    ODISE_REQUIRE(ctx && logits_f16 && out_u8 && B >= 1 && Q >= 1 && h >= 1 && w >= 1 && patch >= 1 && T == (S / patch) * (S / patch) + 1 && ldm >= T,
                  "maskclip_token_mask: bad argument");
    odise::g_token_mask_plain = plain ? 1 : 0;
    const int rc = odise::launch_maskclip_token_mask(ctx, (const odise::f16*)logits_f16, (uint8_t*)out_u8, B, Q, h, w, S, patch, T, ldm);
    odise::g_token_mask_plain = 0;
    return rc;
}
