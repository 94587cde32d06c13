// msda.hip — multi-scale deformable attention forward for gfx950.
//
// Replaces MSDA.ms_deform_attn_forward of the reference
// (third_party/Mask2Former/mask2former/modeling/pixel_decoder/ops/src/cuda/ms_deform_attn_cuda.cu:25-85,
//  kernel semantics ms_deform_im2col_cuda.cuh:242-303, bilinear helper :38-89):
//   out[b,q,m,c] = sum_l sum_p w[b,q,m,l,p] * bilinear(value_l[b,:,m,c], loc*(W_l,H_l) - 0.5)
// with zero padding outside (-1,H)x(-1,W).
//
// MI355X design: the op is an L2/HBM gather.  value is [B,S,M,D] so the D channels of a head are
// contiguous; each lane owns VEC consecutive channels (16 B for f32, 8 B for f16) and D/VEC lanes
// share one (q,m) pair, so a 64-wide wavefront covers 64*VEC/D (q,m) pairs and every corner fetch is
// a contiguous D*sizeof(T)-byte segment.  Sampling locations / weights of a (q,m) pair are read once
// per lane group (same-address broadcast).  One launch for the whole batch: the reference's
// im2col_step chunking only bounded its temporary; there is no temporary here.
#include "engine.h"

namespace odise {

struct MsdaLevels {
    int H[8];
    int W[8];
    int start[8];
};

template <typename T, int VEC>
struct VecIO;

template <>
struct VecIO<float, 4> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <>
struct VecIO<float, 1> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[1]) { *p = v[0]; }
};
template <>
struct VecIO<f16, 4> {
    static __device__ __forceinline__ void load(const f16* p, float (&v)[4]) {
        const f16x4 t = *reinterpret_cast<const f16x4*>(p);
        v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    }
    static __device__ __forceinline__ void store(f16* p, const float (&v)[4]) {
        f16x4 t;
        t[0] = (f16)v[0]; t[1] = (f16)v[1]; t[2] = (f16)v[2]; t[3] = (f16)v[3];
        *reinterpret_cast<f16x4*>(p) = t;
    }
};
template <>
struct VecIO<f16, 1> {
    static __device__ __forceinline__ void load(const f16* p, float (&v)[1]) { v[0] = (float)*p; }
    static __device__ __forceinline__ void store(f16* p, const float (&v)[1]) { *p = (f16)v[0]; }
};

template <typename T, int VEC>
__global__ void __launch_bounds__(256) msda_forward_kernel(const T* __restrict__ value, const float* __restrict__ loc,
                                                          const float* __restrict__ attw, T* __restrict__ out,
                                                          MsdaLevels lv, int64_t total, int S, int M, int D, int Lq,
                                                          int L, int P) {
    const int dchunks = D / VEC;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int dc = (int)(idx % dchunks);
        const int64_t qm = idx / dchunks;  // ((b*Lq + q)*M + m)
        const int m = (int)(qm % M);
        const int64_t bq = qm / M;
        const int b = (int)(bq / Lq);
        const float* lp = loc + qm * (int64_t)(L * P * 2);
        const float* wp = attw + qm * (int64_t)(L * P);
        const T* vb = value + ((int64_t)b * S * M + m) * D + dc * VEC;
        const int64_t pix_stride = (int64_t)M * D;

        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

        for (int l = 0; l < L; ++l) {
            const int Hl = lv.H[l], Wl = lv.W[l];
            const T* vl = vb + (int64_t)lv.start[l] * pix_stride;
            for (int p = 0; p < P; ++p) {
                const float lx = lp[(l * P + p) * 2 + 0];
                const float ly = lp[(l * P + p) * 2 + 1];
                const float w = wp[l * P + p];
                const float h_im = ly * (float)Hl - 0.5f;
                const float w_im = lx * (float)Wl - 0.5f;
                if (h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl) {
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const int h_low = (int)hf, w_low = (int)wf;
                    const int h_high = h_low + 1, w_high = w_low + 1;
                    const float lh = h_im - hf, lw = w_im - wf;
                    const float hh = 1.f - lh, hw = 1.f - lw;
                    float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) v1[i] = v2[i] = v3[i] = v4[i] = 0.f;
                    if (h_low >= 0 && w_low >= 0) VecIO<T, VEC>::load(vl + ((int64_t)h_low * Wl + w_low) * pix_stride, v1);
                    if (h_low >= 0 && w_high <= Wl - 1) VecIO<T, VEC>::load(vl + ((int64_t)h_low * Wl + w_high) * pix_stride, v2);
                    if (h_high <= Hl - 1 && w_low >= 0) VecIO<T, VEC>::load(vl + ((int64_t)h_high * Wl + w_low) * pix_stride, v3);
                    if (h_high <= Hl - 1 && w_high <= Wl - 1) VecIO<T, VEC>::load(vl + ((int64_t)h_high * Wl + w_high) * pix_stride, v4);
                    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const float val = w1 * v1[i] + w2 * v2[i] + w3 * v3[i] + w4 * v4[i];
                        acc[i] += w * val;
                    }
                }
            }
        }
        VecIO<T, VEC>::store(out + qm * D + dc * VEC, acc);
    }
}

// ---- pixel-decoder form (round 6): softmax of the 12 attention logits and the sampling locations computed in the kernel ------------------
// The encoder layer of msdeformattn.py:92-131 produces raw offsets [R, M, L, P, 2] and attention logits [R, M, L, P] (two GEMMs) and
// MSDeformAttn.forward (ops/modules/ms_deform_attn.py:98-125) turns them into locations (reference point + offset / (W_l, H_l)) and softmax
// weights before the native op.  That step was a kernel of its own (msda_prepare_kernel: 99 MB written and read back per layer at 1024^2 x 4).
// Here the gather kernel does it per (query, head) - the 4 lanes that share a pair load the same 36 floats (one broadcast request each) -
// with the prepare kernel's arithmetic in its order, then the reference's fixed summation order over (level, point, corner): bit-identical
// to prepare + gather.  Blocks are renumbered so that each XCD (private 4 MiB L2; the dispatcher deals consecutive workgroups round-robin over
// the 8 XCDs) walks ONE contiguous eighth of the query range: counters of round 4 showed 373 MB fetched per launch for a 44 MB value tensor -
// every XCD's L2 pulled all of it, because neighbouring queries (which sample neighbouring pixels) sat on eight different L2s.
template <int L, int P, int VEC>
__global__ void __launch_bounds__(256) msda_fused_kernel(const f16* __restrict__ value, const float* __restrict__ off, const float* __restrict__ aw,
                                                        f16* __restrict__ out, MsdaLevels lv, int64_t total, int S, int M, int Lq, int ld_off, int ld_aw) {
    constexpr int D = 32, dchunks = D / VEC, LP = L * P;
    typedef _Float16 fvec __attribute__((ext_vector_type(VEC)));
    // XCD-aware block order (bijective for any grid size, as in gemm.hip)
    const int nb = gridDim.x, bid = blockIdx.x;
    const int qd = nb >> 3, r = nb & 7, xcd = bid & 7, bi = bid >> 3;
    const int logical = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + bi;
    const int64_t idx = (int64_t)logical * 256 + threadIdx.x;
    if (idx >= total) return;
    const int dc = (int)(idx % dchunks);
    const int64_t qm = idx / dchunks;  // ((b*Lq + q)*M + m)
    const int m = (int)(qm % M);
    const int64_t bq = qm / M;
    const int b = (int)(bq / Lq), q = (int)(bq - (int64_t)b * Lq);
    // reference point: the centre of the query's own cell at its own level (valid ratios 1: identical for every sampled level)
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < L; ++l)
        if (q >= lv.start[l]) lvl = l;
    const int local = q - lv.start[lvl];
    const float ref_y = ((float)(local / lv.W[lvl]) + 0.5f) / (float)lv.H[lvl];
    const float ref_x = ((float)(local % lv.W[lvl]) + 0.5f) / (float)lv.W[lvl];
    float e[LP], ov[2 * LP];
    {
        const float* a = aw + bq * ld_aw + m * LP;          // (ld: floats per query row - the two projections may be column blocks of one GEMM's output)
        const float* o = off + bq * ld_off + m * LP * 2;
#pragma unroll
        for (int i = 0; i < LP; i += 4) {
            const float4 t = *reinterpret_cast<const float4*>(a + i);
            e[i] = t.x; e[i + 1] = t.y; e[i + 2] = t.z; e[i + 3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < 2 * LP; i += 4) {
            const float4 t = *reinterpret_cast<const float4*>(o + i);
            ov[i] = t.x; ov[i + 1] = t.y; ov[i + 2] = t.z; ov[i + 3] = t.w;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < LP; ++i) mx = fmaxf(mx, e[i]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) { e[i] = expf(e[i] - mx); sum += e[i]; }
    const float inv = 1.f / sum;
    const f16* vb = value + ((int64_t)b * S * M + m) * D + dc * VEC;
    const int64_t pix_stride = (int64_t)M * D;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    // Branch-free per level: the 16 corner loads of a level's four points are issued back to back from CLAMPED (always valid) addresses and an
    // out-of-range corner / point gets weight 0 instead of being skipped (0 x finite = 0, and adding it leaves the sum's bits unchanged): the
    // reference kernel's guarded loads (ms_deform_im2col_cuda.cuh:38-89) cost a divergent branch and a memory round trip per corner.
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int Hl = lv.H[l], Wl = lv.W[l];
        const f16* vl = vb + (int64_t)lv.start[l] * pix_stride;
        fvec c[P][4];
        float cw[P][4], pw[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int i = l * P + p;
            const float lx = ref_x + ov[2 * i + 0] / (float)Wl;
            const float ly = ref_y + ov[2 * i + 1] / (float)Hl;
            const float h_im = ly * (float)Hl - 0.5f;
            const float w_im = lx * (float)Wl - 0.5f;
            const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_low = (int)fminf(fmaxf(hf, -1.f), (float)Hl), w_low = (int)fminf(fmaxf(wf, -1.f), (float)Wl);   // (clamped before the conversion: a far-off point must not overflow int)
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - hf, lw = w_im - wf;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const bool t = h_low >= 0, bt = h_high <= Hl - 1, lf = w_low >= 0, rt = w_high <= Wl - 1;
            const int hl = min(max(h_low, 0), Hl - 1), hh_i = min(max(h_high, 0), Hl - 1), wl = min(max(w_low, 0), Wl - 1), wh = min(max(w_high, 0), Wl - 1);
            c[p][0] = *reinterpret_cast<const fvec*>(vl + ((int64_t)hl * Wl + wl) * pix_stride);
            c[p][1] = *reinterpret_cast<const fvec*>(vl + ((int64_t)hl * Wl + wh) * pix_stride);
            c[p][2] = *reinterpret_cast<const fvec*>(vl + ((int64_t)hh_i * Wl + wl) * pix_stride);
            c[p][3] = *reinterpret_cast<const fvec*>(vl + ((int64_t)hh_i * Wl + wh) * pix_stride);
            cw[p][0] = (in && t && lf) ? hh * hw : 0.f;
            cw[p][1] = (in && t && rt) ? hh * lw : 0.f;
            cw[p][2] = (in && bt && lf) ? lh * hw : 0.f;
            cw[p][3] = (in && bt && rt) ? lh * lw : 0.f;
            pw[p] = in ? e[i] * inv : 0.f;
        }
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float val = cw[p][0] * (float)c[p][0][k] + cw[p][1] * (float)c[p][1][k] + cw[p][2] * (float)c[p][2][k] + cw[p][3] * (float)c[p][3][k];
                acc[k] += pw[p] * val;
            }
    }
    fvec o;
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = (f16)acc[k];
    *reinterpret_cast<fvec*>(out + qm * D + dc * VEC) = o;
}

// value f16 [B, S, M, 32]; off f32 [B*Lq, M*L*P*2] and aw f32 [B*Lq, M*L*P] as the two projections leave them; out f16 [B*Lq, M*32].
// L = 3 levels x P = 4 points (the released configuration); anything else: launch_msda_prepare + odise_hip_ms_deform_attn_forward
bool msda_fused_ok(int M, int D, int L, int P) { return D == 32 && L == 3 && P == 4 && M >= 1; }
// Lanes per (query, head) pair: 4 lanes x 16 bytes (VEC = 8) measured 175 us per layer at 4 x 1024^2 against 250 us with 8 lanes x 8 bytes and 317 us for
// prepare + native op (tools/msda_bench.py, profiles/r06_msda_variants.txt) - half the load instructions per gathered byte is worth more than the
// occupancy (178 VGPRs: two waves per SIMD; forcing three by a launch bound changed nothing).
static int g_msda_vec = 8;   // tools: odise_hip_msda_fused_forward fused = 1 measures the 8-lane form
int launch_msda_fused(odise_hip_ctx* ctx, const f16* value, const float* off, const float* aw, f16* out, const int* Hs, const int* Ws, const int* starts, int B,
                      int S, int M, int Lq, int ld_off, int ld_aw) {
    if (ld_off <= 0) ld_off = M * 3 * 4 * 2;
    if (ld_aw <= 0) ld_aw = M * 3 * 4;
    ODISE_REQUIRE(ld_off % 4 == 0 && ld_aw % 4 == 0 && ((uintptr_t)off & 15) == 0 && ((uintptr_t)aw & 15) == 0, "msda: offsets / logits must keep 16-byte rows");
    MsdaLevels lv;
    for (int l = 0; l < 3; ++l) { lv.H[l] = Hs[l]; lv.W[l] = Ws[l]; lv.start[l] = starts[l]; }
    const int64_t total = (int64_t)B * Lq * M * (32 / g_msda_vec);
    if (total == 0) return ODISE_OK;
    ODISE_REQUIRE(ceil_div(total, 256) < (1ll << 31), "msda: too many queries for one launch");
    if (g_msda_vec == 8) hipLaunchKernelGGL((msda_fused_kernel<3, 4, 8>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, value, off, aw, out, lv, total, S, M, Lq, ld_off, ld_aw);
    else hipLaunchKernelGGL((msda_fused_kernel<3, 4, 4>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, value, off, aw, out, lv, total, S, M, Lq, ld_off, ld_aw);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

template <typename T>
static int launch_msda(hipStream_t stream, const void* value, const float* loc, const float* attw, void* out,
                       const MsdaLevels& lv, int B, int S, int M, int D, int Lq, int L, int P) {
    const int threads = 256;
    if (D % 4 == 0) {
        const int64_t total = (int64_t)B * Lq * M * (D / 4);
        if (total == 0) return ODISE_OK;
        const int blocks = (int)std::min<int64_t>(ceil_div(total, threads), 1 << 20);
        hipLaunchKernelGGL((msda_forward_kernel<T, 4>), dim3(blocks), dim3(threads), 0, stream, (const T*)value, loc, attw,
                           (T*)out, lv, total, S, M, D, Lq, L, P);
    } else {
        const int64_t total = (int64_t)B * Lq * M * D;
        if (total == 0) return ODISE_OK;
        const int blocks = (int)std::min<int64_t>(ceil_div(total, threads), 1 << 20);
        hipLaunchKernelGGL((msda_forward_kernel<T, 1>), dim3(blocks), dim3(threads), 0, stream, (const T*)value, loc, attw,
                           (T*)out, lv, total, S, M, D, Lq, L, P);
    }
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

}  // namespace odise

extern "C" int odise_hip_ms_deform_attn_forward(odise_hip_ctx* ctx, const void* value, const int64_t* spatial_shapes,
                                                const int64_t* level_start_index, const float* sampling_loc,
                                                const float* attn_weight, int B, int S, int M, int D, int Lq, int L, int P,
                                                int im2col_step, int value_dtype, void* out) {
    using namespace odise;
    ODISE_REQUIRE(ctx != nullptr, "ms_deform_attn_forward: null context");
    ODISE_REQUIRE(L >= 1 && L <= 8, "ms_deform_attn_forward: n_levels=%d not in [1,8]", L);
    ODISE_REQUIRE(B >= 0 && S >= 0 && M >= 1 && D >= 1 && Lq >= 0 && P >= 1, "ms_deform_attn_forward: bad dims");
    ODISE_REQUIRE(spatial_shapes && level_start_index, "ms_deform_attn_forward: spatial_shapes/level_start_index must be host arrays");
    // the reference asserts batch % im2col_step_ == 0 with im2col_step_ = min(batch, im2col_step)
    if (B > 0) {
        const int step = std::min(B, im2col_step);
        ODISE_REQUIRE(step >= 1 && B % step == 0, "batch(%d) must divide im2col_step(%d)", B, step);
    }
    ODISE_REQUIRE(value_dtype == ODISE_F32 || value_dtype == ODISE_F16, "ms_deform_attn_forward: dtype must be f32|f16");
    MsdaLevels lv;
    int64_t tot = 0;
    for (int l = 0; l < L; ++l) {
        lv.H[l] = (int)spatial_shapes[2 * l + 0];
        lv.W[l] = (int)spatial_shapes[2 * l + 1];
        lv.start[l] = (int)level_start_index[l];
        ODISE_REQUIRE(lv.H[l] >= 1 && lv.W[l] >= 1, "ms_deform_attn_forward: empty level %d", l);
        ODISE_REQUIRE(lv.start[l] >= 0 && (int64_t)lv.start[l] + (int64_t)lv.H[l] * lv.W[l] <= S,
                      "ms_deform_attn_forward: level %d exceeds S=%d", l, S);
        tot += (int64_t)lv.H[l] * lv.W[l];
    }
    (void)tot;
    if (B == 0 || Lq == 0) return ODISE_OK;
    ODISE_REQUIRE(value && sampling_loc && attn_weight && out, "ms_deform_attn_forward: null device pointer");
    if (value_dtype == ODISE_F32)
        return launch_msda<float>(ctx->stream, value, sampling_loc, attn_weight, out, lv, B, S, M, D, Lq, L, P);
    return launch_msda<f16>(ctx->stream, value, sampling_loc, attn_weight, out, lv, B, S, M, D, Lq, L, P);
}

// test hook (include/odise_hip_tools.h): the pixel decoder's fused form on caller-provided raw projections, and the two-kernel form it replaces
extern "C" int odise_hip_msda_fused_forward(odise_hip_ctx* ctx, const void* value, const float* off, const float* aw, const int* hs3, const int* ws3, int B, int M,
                                            int fused, void* out, float* loc_scratch, float* w_scratch) {
    using namespace odise;
    ODISE_REQUIRE(ctx && value && off && aw && hs3 && ws3 && out && B >= 1 && M >= 1, "msda_fused_forward: bad argument");
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));
    int starts[3], Lq = 0;
    for (int l = 0; l < 3; ++l) { starts[l] = Lq; Lq += hs3[l] * ws3[l]; }
    if (fused) {
        g_msda_vec = fused == 1 ? 4 : 8;
        const int rc = launch_msda_fused(ctx, (const f16*)value, off, aw, (f16*)out, hs3, ws3, starts, B, Lq, M, Lq);
        g_msda_vec = 8;
        return rc;
    }
    ODISE_REQUIRE(loc_scratch && w_scratch, "msda_fused_forward: the two-kernel form needs loc / w scratch");
    ODISE_TRY(launch_msda_prepare(ctx, off, aw, loc_scratch, w_scratch, B, Lq, M, 3, 4, hs3, ws3, starts));
    int64_t ss[6], ls[3];
    for (int l = 0; l < 3; ++l) { ss[2 * l] = hs3[l]; ss[2 * l + 1] = ws3[l]; ls[l] = starts[l]; }
    return odise_hip_ms_deform_attn_forward(ctx, value, ss, ls, loc_scratch, w_scratch, B, Lq, M, 32, Lq, 3, 4, 128, ODISE_F16, out);
}
