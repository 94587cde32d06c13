// attn.hip — fused (flash-style) multi-head attention forward for gfx950, fp16 in / fp32 softmax.
//
// Covers the UNet SpatialTransformer self-attention (tokens 4096/1024/256/64, d_head 40/80/160) and
// cross-attention (77 context keys) — ldm CrossAttention, SURVEY.md Appendix A.1 — the CLIP ViT-L/14
// blocks (577/677 tokens, d 64, optional per-(query,key) visibility mask, clip.py:252-280) and the
// Mask2Former decoder's masked cross-attention (100 queries, d 32, odise.py:760-774).
//
// CDNA4 mapping (wave64, v_mfma_f32_32x32x16_f16):
//   * block = 4 wavefronts = 128 queries; each wave owns 32 queries; KV tiles of 64 keys staged in LDS.
//   * S^T = K·Q^T ("swapped" product): the MFMA result has column = query = lane&31, so one lane holds
//     16 of the 32 keys of a tile for ONE query -> row max / row sum are in-register reductions plus a
//     single cross-half exchange (lane ^ 32); no LDS round trip for P.
//   * O^T = V^T·P^T: the exponentiated S^T registers ARE the B operand (8 keys per lane for its query),
//     V arrives pre-transposed ([H*D, keys], produced by the swapped projection GEMM) so the A operand is
//     two ds_read_b64 per MFMA; the accumulator again has column = query, so the online-softmax rescale
//     is a per-lane scalar multiply.
//   * K rows are padded by 16 B and V^T rows by 8 B so the ds_read_b128 / ds_read_b64 lane groups are
//     bank-conflict free (odd multiples of the access width).
#include "common.h"

namespace odise {

struct AttnArgs {
    int B, H, Lq, Lk, D;
    const f16* Q; int64_t ldq, strideQ;
    const f16* K; int64_t ldk, strideK;
    const f16* Vt; int64_t ldvt, strideVt;
    f16* O; int64_t ldo, strideO;
    const uint8_t* mask; int64_t ldmask, strideMask;
    float scale_log2e;
    int nsplit;   // > 1: keys are split over `nsplit` blocks per (query block, head, batch); partial results go to `part`
    float* part;  // [B, H, nsplit, Lq, D + 2] fp32: unnormalised O^T rows, running max (log2 domain), running sum
};

template <int DPAD>
__global__ void __launch_bounds__(256) attn_kernel(AttnArgs a) {
    constexpr int KS = DPAD / 16;          // MFMA k-steps of QK^T
    constexpr int DT = (DPAD + 31) / 32;   // 32-row d tiles of O^T
    constexpr int KROW = DPAD + 8;         // halves per K row in LDS
    constexpr int VROW = 64 + 4;           // halves per V^T row in LDS
    constexpr int KSLOTS = DPAD / 8;       // 16-byte slots per K row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* Ks = reinterpret_cast<f16*>(smem);                       // [64][KROW]
    f16* Vs = reinterpret_cast<f16*>(smem + 64 * KROW * 2);       // [DT*32][VROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int split = a.nsplit > 1 ? blockIdx.x % a.nsplit : 0;
    const int qblock = a.nsplit > 1 ? blockIdx.x / a.nsplit : blockIdx.x;
    const int q = qblock * 128 + wave * 32 + l31;  // this lane's query
    const bool qok = q < a.Lq;
    const int D = a.D;

    const f16* Qb = a.Q + (int64_t)b * a.strideQ + (int64_t)h * D;
    const f16* Kb = a.K + (int64_t)b * a.strideK + (int64_t)h * D;
    const f16* Vb = a.Vt + (int64_t)b * a.strideVt + (int64_t)h * D * a.ldvt;
    const uint8_t* Mb = a.mask ? a.mask + (int64_t)b * a.strideMask + (int64_t)(qok ? q : 0) * a.ldmask : nullptr;

    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // Q fragments (B operand of S^T): Q[q][16s + 8hi .. +7]
    f16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int d0 = s * 16 + hi * 8;
        qf[s] = zero8;
        if (qok && d0 < D) qf[s] = *reinterpret_cast<const f16x8*>(Qb + (int64_t)q * a.ldq + d0);
    }

    f32x16 ot[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (a.Lk + 63) / 64;
    int kt_begin = 0, kt_end = ntiles;
    if (a.nsplit > 1) {
        const int per = (ntiles + a.nsplit - 1) / a.nsplit;
        kt_begin = split * per;
        kt_end = min(ntiles, kt_begin + per);
    }
    // K / V^T tiles are fetched one tile ahead into registers (the loads of tile kt+1 fly while tile kt is multiplied) and written
    // to LDS after the barrier that retires the previous tile's reads
    constexpr int NKR = (64 * KSLOTS + 255) / 256;  // 16-byte K slots per thread
    constexpr int NVR = DT;                         // 16-byte V^T slots per thread (DT*32 rows x 8 slots / 256 threads)
    f16x8 kreg[NKR], vreg[NVR];
    auto fetch = [&](int kt) {
        const int kv0 = kt * 64;
#pragma unroll
        for (int u = 0; u < NKR; ++u) {
            const int c = tid + u * 256;
            const int key = c / KSLOTS, sl = c - key * KSLOTS;
            kreg[u] = zero8;
            if (c < 64 * KSLOTS && kv0 + key < a.Lk && sl * 8 < D) kreg[u] = *reinterpret_cast<const f16x8*>(Kb + (int64_t)(kv0 + key) * a.ldk + sl * 8);
        }
#pragma unroll
        for (int u = 0; u < NVR; ++u) {
            const int c = tid + u * 256;
            const int d = c >> 3, sl = c & 7;
            const int k0 = kv0 + sl * 8;
            f16x8 v = zero8;
            if (d < D && k0 < a.Lk) {
                v = *reinterpret_cast<const f16x8*>(Vb + (int64_t)d * a.ldvt + k0);
                if (k0 + 8 > a.Lk) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (k0 + i >= a.Lk) v[i] = (f16)0.f;
                }
            }
            vreg[u] = v;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < NKR; ++u) {
            const int c = tid + u * 256;
            const int key = c / KSLOTS, sl = c - key * KSLOTS;
            if (c < 64 * KSLOTS) *reinterpret_cast<f16x8*>(Ks + key * KROW + sl * 8) = kreg[u];
        }
#pragma unroll
        for (int u = 0; u < NVR; ++u) {
            const int c = tid + u * 256;
            const int d = c >> 3, sl = c & 7;
            // V^T row stride is 136 B: write as two 8-byte halves (8-byte aligned)
            const f16x8 v = vreg[u];
            f16x4 lo4 = {v[0], v[1], v[2], v[3]}, hi4 = {v[4], v[5], v[6], v[7]};
            *reinterpret_cast<f16x4*>(Vs + d * VROW + sl * 8) = lo4;
            *reinterpret_cast<f16x4*>(Vs + d * VROW + sl * 8 + 4) = hi4;
        }
    };
    if (kt_begin < kt_end) fetch(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int kv0 = kt * 64;
        __syncthreads();  // previous tile fully consumed
        stage();
        __syncthreads();
        if (kt + 1 < kt_end) fetch(kt + 1);

        // ---- S^T = K Q^T for two 32-key tiles ----
        f32x16 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(Ks + (t * 32 + l31) * KROW + s * 16 + hi * 8);
                st[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], st[t], 0, 0, 0);
            }
        }
        // ---- scale, mask, online softmax (this lane: one query, 32 of the 64 keys) ----
        // At d_head 40..64 this section, not the MFMAs, bounds the kernel (4096^2 scores per head: ~8 VALU slots each against
        // 2 x 16 MFMA cycles per 32x32 tile), so the common case - no mask, tile fully inside Lk - runs a lean path: row max on
        // the raw scores, one FMA (scale and max subtraction) + one v_exp_f32 per score.
        const bool general = (Mb != nullptr) || (kv0 + 64 > a.Lk);  // wave-uniform
        float mx = -INFINITY;
        if (general) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int kbase = kv0 + t * 32 + 8 * g + 4 * hi;  // keys kbase .. kbase+3  <-> regs 4g..4g+3
                    uint32_t mbits = 0;
                    if (Mb && kbase < a.Lk) mbits = *reinterpret_cast<const uint32_t*>(Mb + kbase);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float s = st[t][4 * g + i];
                        const bool dead = (kbase + i >= a.Lk) || ((mbits >> (8 * i)) & 0xff);
                        s = dead ? -INFINITY : s;
                        st[t][4 * g + i] = s;
                        mx = fmaxf(mx, s);
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * a.scale_log2e;  // scale > 0: max commutes with the scaling (-inf stays -inf)
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);  // m_run = -inf -> 0
        // two scores per VALU op where a packed form exists (v_pk_fma_f32, v_pk_add_f32); v_exp_f32 has none
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 sc2 = {a.scale_log2e, a.scale_log2e}, nm2 = {-m_safe, -m_safe};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 x = {st[t][r], st[t][r + 1]};
                const f32x2 y = __builtin_elementwise_fma(x, sc2, nm2);  // masked: fma(-inf) = -inf -> p = 0
                const f32x2 p = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
                st[t][r] = p[0];
                st[t][r + 1] = p[1];
                ps2 += p;
            }
        const float psum = ps2[0] + ps2[1];
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {  // the running max settles after a few tiles: skip the rescale then
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[t][r] *= alpha;
        }

        // ---- O^T += V^T P^T : per (key tile t, half-step s2) one MFMA per d tile ----
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8 pf;
#pragma unroll
                for (int i = 0; i < 8; ++i) pf[i] = (f16)st[t][8 * s2 + i];
                // keys (local to the 64-key tile) held by this lane half for this step
                const int kA = t * 32 + 16 * s2 + 4 * hi;  // regs 8*s2+0..3
                const int kB = kA + 8;                      // regs 8*s2+4..7
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const f16* vr = Vs + (dt * 32 + l31) * VROW;
                    const f16x4 va = *reinterpret_cast<const f16x4*>(vr + kA);
                    const f16x4 vb = *reinterpret_cast<const f16x4*>(vr + kB);
                    const f16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                    ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, ot[dt], 0, 0, 0);
                }
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (a.nsplit > 1) {
        // ---- split-KV: hand the unnormalised partial to attn_combine_kernel ----
        if (qok) {
            float* P = a.part + ((((int64_t)b * a.H + h) * a.nsplit + split) * a.Lq + q) * (D + 2);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = dt * 32 + 8 * g + 4 * hi;
                    if (d0 < D) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) P[d0 + i] = ot[dt][4 * g + i];
                    }
                }
            if (hi == 0) {
                P[D] = m_run;
                P[D + 1] = l_tot;
            }
        }
        return;
    }
    // ---- epilogue: normalise and store O[q][h*D + d] ----
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (qok) {
        f16* Ob = a.O + (int64_t)b * a.strideO + (int64_t)q * a.ldo + (int64_t)h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = dt * 32 + 8 * g + 4 * hi;
                if (d0 < D) {
                    f16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (f16)(ot[dt][4 * g + i] * inv);
                    *reinterpret_cast<f16x4*>(Ob + d0) = o;
                }
            }
        }
    }
}

// ---- keys and values resident in LDS (CLIP ViT-L/14@336: 577 keys, d_head 64) -------------------------------------------------------------
// The tiled kernel above restages every 64-key K / V^T tile once per 128-query block (5 times per head for 577 queries, two barriers per
// tile) and rounds both 577s up to 640.  Here ONE block owns a (head, image): K (608 x 64, 16-byte slots XOR-swizzled) and V^T (64 x 612)
// of that head are loaded into LDS once - 156 160 of the CU's 163 840 bytes - and after a single barrier the block's 16 waves walk their
// 32-query tiles over 19 key tiles of 32 straight out of LDS: no further barrier, no restaging, 608 keys instead of 640.  16 crops x 16
// heads = 256 blocks = one per CU; with fewer (head, image) pairs (MaskCLIP: 4 pictures) the query tiles of a pair are split over
// `qsplit` blocks, each loading the pair's K / V^T (L2 hits).  Same arithmetic per score as the tiled kernel (fp32 scale + running max,
// exp2, fp16 P into the second MFMA); the running max is updated per 32 keys instead of per 64, so results agree to fp32 rounding of the
// rescale, not bit for bit.
// LDS reads: K fragment = ds_read_b128 of slot (2s + hi) of row (32t + lane%32); physical slot = slot ^ ((row >> 1) & 7) makes the 16 lanes
// of every ds_read_b128 group ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md LDS table) hit 16 distinct 16-byte positions of the 256-byte
// LDS line.  V^T rows are 612 halves = 306 dwords apart: 306 mod 64 = 50, so the 32 lanes of a ds_read_b64 group cover all 64 banks.
constexpr int KVR_LKP = 608;             // keys held (19 tiles of 32)
constexpr int KVR_VROW = KVR_LKP + 4;    // halves per V^T row
constexpr int KVR_LDS = KVR_LKP * 64 * 2 + 64 * KVR_VROW * 2;

template <int KVR_WAVES>
__global__ void __launch_bounds__(64 * KVR_WAVES) attn_kvres_kernel(AttnArgs a, int qsplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* Ks = reinterpret_cast<f16*>(smem);                            // [608][64], slot-swizzled
    f16* Vs = reinterpret_cast<f16*>(smem + KVR_LKP * 64 * 2);         // [64][612]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y, part = blockIdx.x;
    constexpr int D = 64;
    const f16* Qb = a.Q + (int64_t)b * a.strideQ + (int64_t)h * D;
    const f16* Kb = a.K + (int64_t)b * a.strideK + (int64_t)h * D;
    const f16* Vb = a.Vt + (int64_t)b * a.strideVt + (int64_t)h * D * a.ldvt;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nkt = (a.Lk + 31) >> 5;          // key tiles of 32
    const int lkp = nkt * 32;
    // ---- K and V^T of this (head, image) -> LDS (rows / columns beyond Lk are zero: their scores are masked, 0 x V must stay 0).
    // All of a thread's 16-byte loads are in flight before its first LDS write (two batches - K, then V^T - of PER loads each: 5 with the 16 waves instantiated): the prologue costs two memory
    // round trips, not twenty (a load -> write loop waits for vmcnt(0) every iteration: measured 15-20 of the kernel's 54 us).
    constexpr int NT = 64 * KVR_WAVES;
    constexpr int PER = (KVR_LKP * 8 + NT - 1) / NT;      // 16-byte slots per thread and operand (608 * 8 / 1024 -> 5 at 16 waves)
    const int vslots = lkp >> 3;                           // 16-byte slots per V^T row
    {
        f16x8 buf[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = tid + u * NT;
            const int key = c >> 3, sl = c & 7;
            // unconditional load from a clamped (always valid) address, zeroed by a select afterwards: a branch per load would make the
            // compiler wait for each load before issuing the next
            const f16x8 t = *reinterpret_cast<const f16x8*>(Kb + (int64_t)min(key, a.Lk - 1) * a.ldk + sl * 8);
            buf[u] = key < a.Lk ? t : zero8;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = tid + u * NT;
            const int key = c >> 3, sl = c & 7;
            if (c < lkp * 8) *reinterpret_cast<f16x8*>(Ks + key * 64 + ((sl ^ ((key >> 1) & 7)) << 3)) = buf[u];
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = tid + u * NT;
            const int d = c / vslots, sl = c - d * vslots;
            const int k0 = sl * 8;
            // (ldvt >= round_up(Lk, 8): the slot that straddles Lk is readable; slots beyond it are clamped to it and zeroed)
            const int dc = min(d, 63), kc = min(k0, (a.Lk - 1) & ~7);
            f16x8 v = *reinterpret_cast<const f16x8*>(Vb + (int64_t)dc * a.ldvt + kc);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (k0 + i < a.Lk) ? v[i] : (f16)0.f;
            buf[u] = v;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = tid + u * NT;
            const int d = c / vslots, sl = c - d * vslots;
            const int k0 = sl * 8;
            if (c < 64 * vslots) {
                const f16x8 v = buf[u];
                const f16x4 lo4 = {v[0], v[1], v[2], v[3]}, hi4 = {v[4], v[5], v[6], v[7]};
                *reinterpret_cast<f16x4*>(Vs + d * KVR_VROW + k0) = lo4;       // rows are 8-byte aligned (1224 B apart)
                *reinterpret_cast<f16x4*>(Vs + d * KVR_VROW + k0 + 4) = hi4;
            }
        }
    }
    __syncthreads();
    // ---- this block's query tiles, dealt round-robin to the waves ----
    const int nqt = (a.Lq + 31) >> 5;
    const int per = (nqt + qsplit - 1) / qsplit;
    const int qt_end = min(nqt, (part + 1) * per);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    for (int qt = part * per + wave; qt < qt_end; qt += KVR_WAVES) {
        const int q = qt * 32 + l31;
        const bool qok = q < a.Lq;
        const uint8_t* Mb = a.mask ? a.mask + (int64_t)b * a.strideMask + (int64_t)(qok ? q : 0) * a.ldmask : nullptr;
        f16x8 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = zero8;
            if (qok) qf[s] = *reinterpret_cast<const f16x8*>(Qb + (int64_t)q * a.ldq + s * 16 + hi * 8);
        }
        f32x16 ot[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
            const int kv0 = kt * 32;
            // S^T tile = K[32 keys] . Q^T[32 queries]
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            const int krow = kv0 + l31;
            const f16* kr = Ks + krow * 64;
            const int ksw = (krow >> 1) & 7;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(kr + (((2 * s + hi) ^ ksw) << 3));
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], st, 0, 0, 0);
            }
            // scale, mask, online softmax: this lane holds 16 of the tile's 32 keys for ONE query (regs 4g..4g+3 <-> keys kv0 + 8g + 4hi ..)
            const bool general = (Mb != nullptr) || (kv0 + 32 > a.Lk);   // wave-uniform
            float mx;
            if (general) {
                mx = -INFINITY;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int kbase = kv0 + 8 * g + 4 * hi;
                    uint32_t mbits = 0;
                    if (Mb && kbase < a.Lk) mbits = *reinterpret_cast<const uint32_t*>(Mb + kbase);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool dead = (kbase + i >= a.Lk) || ((mbits >> (8 * i)) & 0xff);
                        const float sv = dead ? -INFINITY : st[4 * g + i];
                        st[4 * g + i] = sv;
                        mx = fmaxf(mx, sv);
                    }
                }
            } else {
                mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[r]), st[r + 1]);   // v_max3_f32
                mx = fmaxf(mx, st[15]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32)) * a.scale_log2e;
            const float m_new = fmaxf(m_run, mx);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
            const f32x2 sc2 = {a.scale_log2e, a.scale_log2e}, nm2 = {-m_safe, -m_safe};
            f32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 x = {st[r], st[r + 1]};
                const f32x2 y = __builtin_elementwise_fma(x, sc2, nm2);
                const f32x2 p = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
                st[r] = p[0];
                st[r + 1] = p[1];
                ps2 += p;
            }
            l_run = l_run * alpha + (ps2[0] + ps2[1]);
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[t][r] *= alpha;
            }
            // O^T += V^T[64 d][32 keys] . P^T
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8 pf;
#pragma unroll
                for (int i = 0; i < 8; ++i) pf[i] = (f16)st[8 * s2 + i];
                const int kA = kv0 + 16 * s2 + 4 * hi, kB = kA + 8;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const f16* vr = Vs + (dt * 32 + l31) * KVR_VROW;
                    const f16x4 va = *reinterpret_cast<const f16x4*>(vr + kA);
                    const f16x4 vb = *reinterpret_cast<const f16x4*>(vr + kB);
                    const f16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                    ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, ot[dt], 0, 0, 0);
                }
            }
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
        if (qok) {
            f16* Ob = a.O + (int64_t)b * a.strideO + (int64_t)q * a.ldo + (int64_t)h * D;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (f16)(ot[dt][4 * g + i] * inv);
                    *reinterpret_cast<f16x4*>(Ob + dt * 32 + 8 * g + 4 * hi) = o;
                }
        }
    }
}

// Taken where one block per (head, image) - or a whole number of query splits of it - fills the chip in (nearly) whole rounds: 16 crops x 16
// heads = 256 pairs = one round.  288 pairs (18 crops) would run two rounds for 1.125 of work, and a pair split over few blocks pays the
// 148 KB prologue per block for too few query tiles (MaskCLIP on 4 pictures): the tiled kernel keeps those (tools/attn_bench.py).
static bool attn_kvres_ok(const odise_hip_ctx* ctx, const AttnArgs& a) {
    const int cus = ctx->cu_count;
    if (ctx->attn_kv_resident & 2) return false;   // ODISE_OPT_ATTN_KV_RESIDENT bit 1: never
    if (ctx->max_lds_optin < KVR_LDS) return false; // a device (or partition) that cannot give one block 152.5 KiB of LDS keeps the tiled kernel
    if (!(a.D == 64 && a.Lk >= 256 && a.Lk <= KVR_LKP && a.Lq >= 256)) return false;   // (100 mask-token queries per pair: 22.9 us against 19.0 tiled, tools/attn_bench.py)

    const int64_t pairs = (int64_t)a.B * a.H;
    if (pairs < cus) return false;
    const double rounds = (double)pairs / cus;
    return ceil(rounds) / rounds <= 1.07;
}

template <int KVR_WAVES>
static int launch_attn_kvres(odise_hip_ctx* ctx, AttnArgs& a) {
    static LdsAttrOnce once;  // tracked per device inside
    ODISE_TRY(ensure_dyn_lds(ctx, once, (const void*)attn_kvres_kernel<KVR_WAVES>, KVR_LDS));
    // one block per (head, image) when those fill the chip; otherwise the query tiles of a pair are split so that every CU gets a block
    // (a block needs at least one query tile per wave to pay for loading K / V^T)
    const int64_t pairs = (int64_t)a.B * a.H;
    const int nqt = (int)ceil_div(a.Lq, 32);
    int qsplit = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->cu_count / pairs, ceil_div(nqt, KVR_WAVES)));
    a.nsplit = 1;
    a.part = nullptr;
    dim3 grid((unsigned)qsplit, (unsigned)a.H, (unsigned)a.B);
    hipLaunchKernelGGL(attn_kvres_kernel<KVR_WAVES>, grid, dim3(64 * KVR_WAVES), KVR_LDS, ctx->stream, a, qsplit);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}

// ---- software-pipelined self-attention (round 6): UNet SpatialTransformer self-attention, 4096 / 1024 tokens, d_head 40 / 80 ---------------
// The tiled kernel above runs a wave's tile as a serial chain - stage, barrier, S^T MFMAs, softmax on the VALU, O^T MFMAs, barrier - and at
// d_head 40 the VALU part alone (32 scores per lane and 64-key tile: max3, packed scale, v_exp_f32 at quarter rate, packed sum, packed
// convert: ~800 issue cycles) is 1.8x the 14 MFMAs (448 cycles): measured 886 us on the 16-crop 64^2 level, i.e. ~1700 cycles per wave
// tile - VALU, MFMA, LDS round trips and two barriers back to back.  Here the S^T MFMAs of tile t+1 are issued BEFORE the softmax of tile
// t (a second accumulator set, +32 VGPRs) and the O^T MFMAs of tile t right after it, so the matrix pipe works underneath the VALU stretch
// of the same wave instead of between two of them; K is double- and V^T triple-buffered in LDS so that ONE barrier per tile orders
// everything: at the top of iteration t every wave has finished iteration t-1 (its reads of K[t] and V[t-1]), tile t+2 goes from the
// prefetch registers into the buffers those reads released, and tile t+3's global loads are issued.  Same arithmetic per score in the
// same order as attn_kernel (bit-identical results).  Preconditions (attn_sa_ok): no mask, no key split, Lk % 128 == 0,
// Lq % 128 == 0.
template <int DPAD>
__global__ void __launch_bounds__(256) attn_sa_kernel(AttnArgs a) {
    constexpr int KS = DPAD / 16;          // MFMA k-steps of QK^T
    constexpr int DT = (DPAD + 31) / 32;   // 32-row d tiles of O^T
    constexpr int KROW = DPAD + 8;         // halves per K row in LDS
    constexpr int VROW = 64 + 4;           // halves per V^T row in LDS
    constexpr int KSLOTS = DPAD / 8;       // 16-byte slots per K row
    constexpr int KHALVES = 64 * KROW, VHALVES = DT * 32 * VROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* Ks = reinterpret_cast<f16*>(smem);                  // [2][64][KROW]
    f16* Vs = Ks + 2 * KHALVES;                              // [3][DT*32][VROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + l31;  // this lane's query (always < Lq)
    const int D = a.D;
    const f16* Qb = a.Q + (int64_t)b * a.strideQ + (int64_t)h * D;
    const f16* Kb = a.K + (int64_t)b * a.strideK + (int64_t)h * D;
    const f16* Vb = a.Vt + (int64_t)b * a.strideVt + (int64_t)h * D * a.ldvt;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    f16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int d0 = s * 16 + hi * 8;
        qf[s] = zero8;
        if (d0 < D) qf[s] = *reinterpret_cast<const f16x8*>(Qb + (int64_t)q * a.ldq + d0);
    }
    f32x16 ot[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int ntiles = a.Lk >> 6;

    constexpr int NKR = (64 * KSLOTS + 255) / 256;  // 16-byte K slots per thread
    constexpr int NVR = DT;                         // 16-byte V^T slots per thread
    f16x8 kreg[NKR], vreg[NVR];
    // Unconditional loads from clamped (always valid) addresses, zeroed by a select: a branch per load would make the compiler wait for
    // each load before it issues the next one.  The pad slots (d >= D) are staged as zeros every tile.
    int64_t koff[NKR], voff[NVR];
    bool kval[NKR], vval[NVR];
#pragma unroll
    for (int u = 0; u < NKR; ++u) {
        const int c = tid + u * 256;
        const int key = c / KSLOTS, sl = c - key * KSLOTS;
        kval[u] = c < 64 * KSLOTS && sl * 8 < D;
        koff[u] = (int64_t)min(key, 63) * a.ldk + min(sl * 8, D - 8);
    }
#pragma unroll
    for (int u = 0; u < NVR; ++u) {
        const int c = tid + u * 256;
        const int d = c >> 3, sl = c & 7;
        vval[u] = d < D;
        voff[u] = (int64_t)min(d, D - 1) * a.ldvt + sl * 8;
    }
    auto fetch = [&](int kt) {
        const f16* Kt = Kb + (int64_t)kt * 64 * a.ldk;
        const f16* Vt_ = Vb + kt * 64;
#pragma unroll
        for (int u = 0; u < NKR; ++u) kreg[u] = *reinterpret_cast<const f16x8*>(Kt + koff[u]);
#pragma unroll
        for (int u = 0; u < NVR; ++u) vreg[u] = *reinterpret_cast<const f16x8*>(Vt_ + voff[u]);
    };
    auto stage = [&](f16* Kd, f16* Vd) {
#pragma unroll
        for (int u = 0; u < NKR; ++u) {
            const int c = tid + u * 256;
            const int key = c / KSLOTS, sl = c - key * KSLOTS;
            if (c < 64 * KSLOTS) *reinterpret_cast<f16x8*>(Kd + key * KROW + sl * 8) = kval[u] ? kreg[u] : zero8;   // (wave-uniform predicate)
        }
#pragma unroll
        for (int u = 0; u < NVR; ++u) {
            const int c = tid + u * 256;
            const int d = c >> 3, sl = c & 7;
            const f16x8 v = vval[u] ? vreg[u] : zero8;
            const f16x4 lo4 = {v[0], v[1], v[2], v[3]}, hi4 = {v[4], v[5], v[6], v[7]};
            *reinterpret_cast<f16x4*>(Vd + d * VROW + sl * 8) = lo4;       // V^T rows are 136 B apart: two 8-byte halves
            *reinterpret_cast<f16x4*>(Vd + d * VROW + sl * 8 + 4) = hi4;
        }
    };
    // S^T = K Q^T for the two 32-key halves of a tile
    auto qk = [&](const f16* Kt, f32x16 (&st)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(Kt + (t * 32 + l31) * KROW + s * 16 + hi * 8);
                st[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], st[t], 0, 0, 0);
            }
        }
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // online softmax of one tile's scores (this lane: one query, 32 of the 64 keys) and O^T += V^T P^T
    auto softmax_pv = [&](f32x16 (&st)[2], const f16* Vt_) {
        float mx = fmaxf(fmaxf(st[0][0], st[0][1]), st[0][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[0][r]), st[0][r + 1]);
        mx = fmaxf(fmaxf(mx, st[0][15]), st[1][0]);
#pragma unroll
        for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, st[1][r]), st[1][r + 1]);
        mx = fmaxf(mx, st[1][15]);
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * a.scale_log2e;
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // m_run = -inf -> 0 (m_new is finite: no mask)
        const f32x2 sc2 = {a.scale_log2e, a.scale_log2e}, nm2 = {-m_new, -m_new};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 x = {st[t][r], st[t][r + 1]};
                const f32x2 y = __builtin_elementwise_fma(x, sc2, nm2);
                const f32x2 p = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
                st[t][r] = p[0];
                st[t][r + 1] = p[1];
                ps2 += p;
            }
        l_run = l_run * alpha + (ps2[0] + ps2[1]);
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[t][r] *= alpha;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8 pf;
#pragma unroll
                for (int i = 0; i < 8; ++i) pf[i] = (f16)st[t][8 * s2 + i];
                const int kA = t * 32 + 16 * s2 + 4 * hi, kB = kA + 8;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const f16* vr = Vt_ + (dt * 32 + l31) * VROW;
                    const f16x4 va = *reinterpret_cast<const f16x4*>(vr + kA);
                    const f16x4 vb = *reinterpret_cast<const f16x4*>(vr + kB);
                    const f16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                    ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, ot[dt], 0, 0, 0);
                }
            }
        }
    };

    // ---- prologue: tiles 0 and 1 into LDS, tile 2 into the prefetch registers, S^T of tile 0
    fetch(0);
    stage(Ks, Vs);
    fetch(1);
    stage(Ks + KHALVES, Vs + VHALVES);
    fetch(min(2, ntiles - 1));
    __syncthreads();
    f32x16 sa[2], sb[2];
    qk(Ks, sa);
    // iteration t: scores of tile t are in `cur` (issued one iteration ago), tile t+1 is computed into `nxt`.  vcur / vfree: the V^T buffers of
    // tile t and of tile t+2 (= the one tile t-1 released), walked as run-time offsets so that the loop is NOT unrolled by its buffer period
    // (2 x 3 copies of a 600-instruction body would not fit the instruction cache)
    int vcur = 0, vfree = 2 * VHALVES;
    auto iteration = [&](int t, f32x16 (&cur)[2], f32x16 (&nxt)[2]) {
        __syncthreads();   // every wave is done with iteration t-1: K[t] (buffer t & 1) and V[t-1] are free, K / V of tile t+1 are visible
        // The body is branch-free: in the last iterations the staging writes land in buffers nobody reads any more, the prefetch re-reads the
        // last tile and the S^T of the tile "after the last" is computed from stale (finite) LDS and never used - cheaper than the loop
        // versions the compiler builds around three conditions (19k lines of ISA, beyond the instruction cache).
        stage(Ks + (t & 1) * KHALVES, Vs + vfree);
        fetch(min(t + 3, ntiles - 1));
        qk(Ks + ((t + 1) & 1) * KHALVES, nxt);
        __builtin_amdgcn_sched_barrier(0);   // the S^T MFMAs of tile t+1 stay ahead of the softmax of tile t
        softmax_pv(cur, Vs + vcur);
        vfree = vcur;                         // V[t] is released when every wave has passed the next barrier; tile t+3 goes there
        vcur = vcur + VHALVES == 3 * VHALVES ? 0 : vcur + VHALVES;
    };
#pragma unroll 1
    for (int t = 0; t < ntiles; t += 2) {     // Lk % 128 == 0: whole pairs of tiles
        iteration(t, sa, sb);
        iteration(t + 1, sb, sa);
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    f16* Ob = a.O + (int64_t)b * a.strideO + (int64_t)q * a.ldo + (int64_t)h * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = dt * 32 + 8 * g + 4 * hi;
            if (d0 < D) {
                f16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (f16)(ot[dt][4 * g + i] * inv);
                *reinterpret_cast<f16x4*>(Ob + d0) = o;
            }
        }
    }
}

template <int DPAD>
static int launch_attn_sa(odise_hip_ctx* ctx, AttnArgs& a) {
    constexpr int DT = (DPAD + 31) / 32;
    constexpr int LDS = 2 * 64 * (DPAD + 8) * 2 + 3 * DT * 32 * 68 * 2;
    a.nsplit = 1;
    a.part = nullptr;
    dim3 grid((unsigned)(a.Lq / 128), (unsigned)a.H, (unsigned)a.B);
    hipLaunchKernelGGL((attn_sa_kernel<DPAD>), grid, dim3(256), LDS, ctx->stream, a);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
// the pipelined kernel takes the self-attention shapes that fill the chip: whole 128-query blocks and 64-key tiles, no mask
static bool attn_sa_ok(const odise_hip_ctx* ctx, const AttnArgs& a) {
    if (ctx->attn_kv_resident & 4) return false;   // ODISE_OPT_ATTN_KV_RESIDENT bit 2: never (A/B, tests)
    if (a.mask || a.Lq % 128 != 0 || a.Lk % 128 != 0 || a.Lk < 256 || a.D > 80 || a.D <= 32) return false;
    // Measured (tools/attn_unet_bench.py, profiles/r06_attention_pipelined.txt): with up to ~8 blocks per CU the pipelined kernel wins (61.6 against 75.5 us
    // on one crop's 64^2 level, 78.3 against 84.7 on 16 crops' 32^2 level); on the 16-crop 64^2 level (16 blocks per CU) the tiled kernel's third wave
    // per SIMD (138 against 190 VGPRs) is worth more than the overlap inside a wave (717 against 750 us): both sit on the VALU + MFMA issue time
    // of the scores (~44 cycles per 64 scores and SIMD), which neither form overlaps.
    const int64_t blocks = (int64_t)(a.Lq / 128) * a.H * a.B;
    return blocks >= ctx->cu_count && blocks <= 8 * (int64_t)ctx->cu_count;
}

// O[q] = sum_s O_s 2^(m_s - M) / sum_s l_s 2^(m_s - M): one thread per (row, 4 channels)
__global__ void __launch_bounds__(256) attn_combine_kernel(AttnArgs a) {
    const int D = a.D, D4 = D >> 2;
    const int64_t total = (int64_t)a.B * a.H * a.Lq * D4;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int d0 = (int)(idx % D4) * 4;
    const int64_t row = idx / D4;  // (b*H + h)*Lq + q
    const int q = (int)(row % a.Lq);
    const int64_t bh = row / a.Lq;
    const int h = (int)(bh % a.H), b = (int)(bh / a.H);
    const float* P = a.part + ((bh * a.nsplit) * a.Lq + q) * (D + 2);
    const int64_t sstride = (int64_t)a.Lq * (D + 2);
    float M = -INFINITY;
    for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, P[s * sstride + D]);
    float o[4] = {0.f, 0.f, 0.f, 0.f}, L = 0.f;
    if (M > -INFINITY) {
        for (int s = 0; s < a.nsplit; ++s) {
            const float* Ps = P + s * sstride;
            const float w = exp2f(Ps[D] - M);  // m_s = -inf -> 0
            L += Ps[D + 1] * w;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] += Ps[d0 + i] * w;
        }
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    f16x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (f16)(o[i] * inv);
    *reinterpret_cast<f16x4*>(a.O + (int64_t)b * a.strideO + (int64_t)q * a.ldo + (int64_t)h * D + d0) = r;
}

template <int DPAD>
static int launch_attn(odise_hip_ctx* ctx, AttnArgs& a) {
    constexpr int DT = (DPAD + 31) / 32;
    const size_t lds = 64 * (DPAD + 8) * 2 + (size_t)DT * 32 * 68 * 2;
    // few queries against many keys (masked cross-attention of the mask decoder: 100 queries x up to 16384 keys): split the keys
    // over blocks until the CUs are covered twice, then fold the partials
    const int64_t qblocks = ceil_div(a.Lq, 128), base = qblocks * a.H * a.B;
    const int ntiles = (int)ceil_div(a.Lk, 64);
    a.nsplit = 1;
    a.part = nullptr;
    if (base < ctx->cu_count && ntiles >= 8 && a.D % 4 == 0) {
        int ns = (int)std::min<int64_t>(ceil_div(2 * (int64_t)ctx->cu_count, base), ntiles / 4);
        while (ns > 1 && (size_t)a.B * a.H * ns * a.Lq * (a.D + 2) * sizeof(float) > ctx->ws_bytes) --ns;
        if (ns > 1) {
            const int per = (int)ceil_div(ntiles, ns);
            a.nsplit = (int)ceil_div(ntiles, per);  // no empty splits
            a.part = (float*)ctx->ws;
        }
    }
    dim3 grid((unsigned)(qblocks * a.nsplit), (unsigned)a.H, (unsigned)a.B);
    hipLaunchKernelGGL((attn_kernel<DPAD>), grid, dim3(256), lds, ctx->stream, a);
    ODISE_CHECK_HIP(hipGetLastError());
    if (a.nsplit > 1) {
        const int64_t total = (int64_t)a.B * a.H * a.Lq * (a.D / 4);
        hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream, a);
        ODISE_CHECK_HIP(hipGetLastError());
    }
    return ODISE_OK;
}

}  // namespace odise

extern "C" int odise_hip_attention(odise_hip_ctx* ctx, const odise_attn_desc* d) {
    using namespace odise;
    ODISE_REQUIRE(ctx && d, "attention: null argument");
    ODISE_REQUIRE(d->B >= 0 && d->H >= 1 && d->Lq >= 0 && d->Lk >= 1 && d->D >= 8, "attention: bad dims");
    ODISE_REQUIRE(d->scale > 0.f, "attention: scale must be positive");
    ODISE_REQUIRE(d->D % 8 == 0 && d->D <= 160, "attention: head dim %d must be a multiple of 8 and <= 160", d->D);
    if (d->B == 0 || d->Lq == 0) return ODISE_OK;
    ODISE_REQUIRE(d->Q && d->K && d->Vt && d->O, "attention: null device pointer");
    ODISE_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldvt % 8 == 0 && d->ldo % 4 == 0, "attention: leading dims must keep 16-byte rows");
    ODISE_REQUIRE(d->ldvt >= round_up(d->Lk, 8), "attention: ldvt=%lld must be >= Lk rounded up to 8", (long long)d->ldvt);
    ODISE_REQUIRE(!d->mask || (d->ldmask % 4 == 0 && d->ldmask >= round_up(d->Lk, 4)), "attention: ldmask must be a multiple of 4 and >= Lk");
    ODISE_REQUIRE((d->D * d->H) % 4 == 0, "attention: H*D must be a multiple of 4");
    AttnArgs a;
    a.B = d->B; a.H = d->H; a.Lq = d->Lq; a.Lk = d->Lk; a.D = d->D;
    a.Q = (const f16*)d->Q; a.ldq = d->ldq; a.strideQ = d->strideQ;
    a.K = (const f16*)d->K; a.ldk = d->ldk; a.strideK = d->strideK;
    a.Vt = (const f16*)d->Vt; a.ldvt = d->ldvt; a.strideVt = d->strideVt;
    a.O = (f16*)d->O; a.ldo = d->ldo; a.strideO = d->strideO;
    a.mask = d->mask; a.ldmask = d->ldmask; a.strideMask = d->strideMask;
    a.scale_log2e = d->scale * 1.4426950408889634f;
    const int D = d->D;
    // 16 waves per block (four per SIMD) measured best: 48.1 us against 52.2 (8 waves), 51.4 (12) and 54.4 (tiled kernel) on the 16-crop tower, 91.1
    // against 96.6 (tiled) on 32 crops (tools/attn_bench.py, profiles/r04_attention_kv_resident.txt); only that form is instantiated
    if (attn_kvres_ok(ctx, a)) return launch_attn_kvres<16>(ctx, a);
    if (attn_sa_ok(ctx, a)) {
        if (D <= 48) return launch_attn_sa<48>(ctx, a);
        if (D <= 64) return launch_attn_sa<64>(ctx, a);
        return launch_attn_sa<80>(ctx, a);
    }
    if (D <= 32) return launch_attn<32>(ctx, a);
    if (D <= 48) return launch_attn<48>(ctx, a);
    if (D <= 64) return launch_attn<64>(ctx, a);
    if (D <= 80) return launch_attn<80>(ctx, a);
    if (D <= 96) return launch_attn<96>(ctx, a);
    if (D <= 128) return launch_attn<128>(ctx, a);
    return launch_attn<160>(ctx, a);
}
