// gemm.hip — fp16 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   C[M,N] = epilogue(alpha * A[M,K] * W[N,K]^T)          (dense; every Linear / 1x1 conv of the path)
//   Y[n,oy,ox,co] = epilogue(sum_{ky,kx,ci} X[n,iy,ix,ci] * Wt[co,ky,kx,ci])   (CONV=true; NHWC, A gathered on the fly)
//
// These are the dense conv / im2col and Linear GEMMs of the SD-v1 UNet ResBlock / SpatialTransformer
// (ldm.py:469-491 -> ldm UNetModel), the VAE, CLIP and Mask2Former heads.
//
// CDNA4 mapping (v_mfma_f32_32x32x16_f16, fp32 accumulate, block tile BM x BN x 64):
//   * gemm_kernel      - 4- or 8-wave tiles 64x64 .. 256x320; A and W tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4),
//                        double-buffered, one barrier per K-tile; small / ragged problems, split-K over grid.z.
//   * gemm_pp_kernel   - 256x256 / 256x320 / 512x128 tiles, 8 waves in two groups staggered by a barrier (one multiplies while the
//                        other feeds), DMA in flight across barriers with counted vmcnt.
//   * gemm_pp2_kernel  - same, with the next phase's fragment reads issued under the MFMAs.
//   * conv3_halo_kernel- 3x3 convs: A fragments come from an LDS-resident 18x18 input patch fetched once per 64-channel chunk.
//   LDS rows are 128 B (64 halves); 16-byte slots are XOR-swizzled (applied to the DMA source address, the LDS image of a DMA
//   instruction being lane-linear) so that the 16-lane groups of ds_read_b128 hit 16 distinct slots of the 256-B bank row.
//   The epilogue goes through LDS (fp32) so that bias / time-embedding broadcast / activation / GEGLU / residual run on 8 consecutive
//   output channels per lane and stores are 16-byte coalesced.  Tile, split-K and kernel generation are chosen by launch_gemm's
//   cost model.  Every variant keeps the same fp32 summation order: results are bit-identical across kernels for a given K order.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

// Timing-only ablation switches (GemmArgs::dbg: drop the operand DMA / fragment reads / epilogue, freeze the K walk) and the
// ODISE_GEMM_FLAGS / ODISE_GEMM_FREEZE_K environment switches exist only in the measurement build of the library
// (`python -m odise_amd.build --tools` -> libodise_hip_tools.so, -DODISE_TOOLS); the product build compiles them out.
#ifdef ODISE_TOOLS
#define ODISE_ABLATE(g, bit) ((g).dbg & (bit))
#else
#define ODISE_ABLATE(g, bit) false
#endif

namespace odise {

struct GemmEpi {
    void* C;
    int64_t ldc;
    int c_dtype;
    const float* bias_n;
    const float* bias_m;
    const float* scale_m;
    const f16* residual;
    int64_t ldr;
    const float* rowgroup_add;
    int rows_per_group;
    int64_t ldg;
    int act;
    int geglu;
    float alpha;
    int64_t strideC, strideR;
    int fast;  // 1: every vector access of a full 8-column chunk is aligned -> epi_fast8 (set by launch_gemm)
    int f16path;  // 1: fp16 output whose 16-byte row chunks are all aligned and whole -> the math-first epilogue (gemm_epilogue_f16; set by launch_gemm)
    // LayerNorm folded into the GEMMs around it (gemm_ln, math-first epilogue only; see gemm_epilogue_f16)
    const float* ln_part;     // consumer, LN over the rows of A: [M][ln_P][2] partial (sum, sum of squares) of every row of the LN input
    int ln_P;
    float ln_inv_c, ln_eps;
    const float* ln_colsum;   // ... [N]: sum over k of the gamma-folded fp16 weights of column n
    float* ln_final_out;      // ... optional [M][2] = (-mean * rstd, rstd) of every row, written by the blocks of the first column tile
    const float* ln_final;    // consumer, LN over the rows of W (swapped GEMM): [N][2] = (-mean * rstd, rstd) per output column
    const float* ln_rowsum;   // ... [M]: sum over k of the folded weights of output row m
    float* ln_stats_out;      // producer: [M][ceil(N / kLnPartCols)][2] partial (sum, sum of squares) of every output row (of the rounded fp16 values)
    float* gn_stats;  // optional [row blocks][N][2]: per-channel (sum, sum of squares) of the block's fp16 outputs (GroupNorm statistics
                      // fused into the producing conv; set by launch_gemm only when the chosen kernel supports it)
};

struct ConvGeom {
    int H, W, Cin, KH, KW, stride, pad_t, pad_l, OH, OW, ups;
    int chunk_major;  // K-tiles walk (Cin chunk outer, tap inner): see prep_tile
    int halo_tx, halo_ty;  // conv3_halo_kernel: 16x16 output patches per image row / column (0 = not a halo launch)
};

struct GemmArgs {
    int M, N, K;
    const f16* A;
    int64_t lda, strideA;
    const f16* W;
    int64_t ldw, strideW;
    GemmEpi epi;
    ConvGeom cg;
    int splitk;
    int ktiles_per_split;
    float* ws;
    const f16* zeros;  // >= 16 zero bytes (source of padded / out-of-range operand slots)
    int dbg;           // ablation switches (tools only): 1 = no operand DMA after the first tile, 2 = no fragment reads after the first
    int stats_blocks;  // out: row blocks per image of the fused GroupNorm statistics (0 = not produced, epi.gn_stats was cleared)
    int epi_block;     // 1: never the wave-private epilogue (launch_gemm_select: ODISE_GEMM_FLAGS 32768)
};

// Workgroup barrier that only orders LDS traffic.  `__syncthreads()` also drains the vector-memory counter, i.e. it waits for
// every outstanding global STORE of the wave (vmcnt counts stores on CDNA4); between the epilogue passes that costs one HBM
// write round trip per pass (measured: 61 of 89 us on a 65536x640x320 GEMM).  The staging buffer is only touched by ds_* ops.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// 16-byte global -> LDS DMA (global_load_lds_dwordx4): LDS address = wave-uniform `lds_base` + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

// Applies the epilogue to 8 consecutive columns (n..n+7) of row m and stores them.
__device__ __forceinline__ void epi_store8(const GemmEpi& e, float (&v)[8], int m, int n, int N, int z) {
    const int nvalid = (N - n) < 8 ? (N - n) : 8;
    if (nvalid <= 0) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float x = v[i] * e.alpha;
        if (e.scale_m) x *= e.scale_m[m];
        if (i < nvalid) {
            if (e.bias_n) x += e.bias_n[n + i];
            if (e.bias_m) x += e.bias_m[m];
            if (e.rowgroup_add) x += e.rowgroup_add[(int64_t)(m / e.rows_per_group) * e.ldg + n + i];
        }
        v[i] = x;
    }
    if (e.geglu) {
        // columns are (a, gate) pairs; output has N/2 columns
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = v[2 * i] * gelu_erf(v[2 * i + 1]);
        const int no = n >> 1;
        const int nov = nvalid >> 1;
        if (e.c_dtype == ODISE_F16) {
            f16* c = (f16*)e.C + (int64_t)z * e.strideC + (int64_t)m * e.ldc + no;
            if (nov == 4 && ((e.ldc & 3) == 0)) {
                f16x4 t;
                t[0] = (f16)o[0]; t[1] = (f16)o[1]; t[2] = (f16)o[2]; t[3] = (f16)o[3];
                *reinterpret_cast<f16x4*>(c) = t;
            } else {
                for (int i = 0; i < nov; ++i) c[i] = (f16)o[i];
            }
        } else {
            float* c = (float*)e.C + (int64_t)z * e.strideC + (int64_t)m * e.ldc + no;
            for (int i = 0; i < nov; ++i) c[i] = o[i];
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = act_apply(v[i], e.act);
    if (e.residual) {
        const f16* r = e.residual + (int64_t)z * e.strideR + (int64_t)m * e.ldr + n;
        if (nvalid == 8 && ((e.ldr & 7) == 0)) {
            const f16x8 t = *reinterpret_cast<const f16x8*>(r);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += (float)t[i];
        } else {
            for (int i = 0; i < nvalid; ++i) v[i] += (float)r[i];
        }
    }
    if (e.c_dtype == ODISE_F16) {
        f16* c = (f16*)e.C + (int64_t)z * e.strideC + (int64_t)m * e.ldc + n;
        if (nvalid == 8 && ((e.ldc & 7) == 0)) {
            f16x8 t;
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = (f16)v[i];
            *reinterpret_cast<f16x8*>(c) = t;
        } else {
            for (int i = 0; i < nvalid; ++i) c[i] = (f16)v[i];
        }
    } else {
        float* c = (float*)e.C + (int64_t)z * e.strideC + (int64_t)m * e.ldc + n;
        if (nvalid == 8 && ((e.ldc & 3) == 0)) {
            *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            for (int i = 0; i < nvalid; ++i) c[i] = v[i];
        }
    }
}

// Block tile BM x BN x 64 computed by WAVES_M x WAVES_N wavefronts (each a (BM/WAVES_M) x (BN/WAVES_N) sub-tile of
// 32x32x16 MFMAs).  Instantiated as 4-wave tiles (128x128, 64x128, 64x64: small problems, with split-K) and 8-wave
// tiles (256x320 for the 320*k channel counts of the SD UNet, 256x256, 256x128): at 256 rows one K-tile carries
// 2048-2560 MFMA cycles per SIMD, which covers an L2-miss round trip with a single tile of LDS-DMA prefetch in flight,
// and halves the operand bytes per flop (29 B/clk/CU at peak vs the 64 B/clk/CU L1 limit).
// Lean epilogue for the common case (all 8 columns in range, 16-byte aligned rows, fp16 output, no GEGLU / per-row terms):
// every load is issued before the arithmetic, no per-element branches.  The generic epi_store8 above costs ~300 instructions per
// 8 outputs and made the epilogue instruction-bound (61 of 89 us on a 65536x640x320 GEMM).
__device__ __forceinline__ void epi_fast8(const GemmEpi& e, float (&v)[8], int m, int n, int z, float* rounded = nullptr) {
    float b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = 0.f;
    if (e.bias_n) {
        const float4 b0 = *reinterpret_cast<const float4*>(e.bias_n + n);
        const float4 b1 = *reinterpret_cast<const float4*>(e.bias_n + n + 4);
        b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
    }
    if (e.rowgroup_add) {  // per-image vector (time embedding): one row-group index per 8 outputs
        const float* rg = e.rowgroup_add + (int64_t)((unsigned)m / (unsigned)e.rows_per_group) * e.ldg + n;
        const float4 r0 = *reinterpret_cast<const float4*>(rg);
        const float4 r1 = *reinterpret_cast<const float4*>(rg + 4);
        b[0] += r0.x; b[1] += r0.y; b[2] += r0.z; b[3] += r0.w; b[4] += r1.x; b[5] += r1.y; b[6] += r1.z; b[7] += r1.w;
    }
    float alpha = e.alpha;
    if (e.scale_m) alpha *= e.scale_m[m];
    if (e.bias_m) {
        const float bm = e.bias_m[m];
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] += bm;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = v[i] * alpha + b[i];
    if (e.geglu) {
        // columns are (a, gate) pairs; the output has N/2 columns (no activation / residual on this path)
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = v[2 * i] * gelu_erf(v[2 * i + 1]);
        const int64_t off = (int64_t)z * e.strideC + (int64_t)m * e.ldc + (n >> 1);
        if (e.c_dtype == ODISE_F16) {
            f16x4 t;
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = (f16)o[i];
            *reinterpret_cast<f16x4*>((f16*)e.C + off) = t;
        } else {
            *reinterpret_cast<float4*>((float*)e.C + off) = make_float4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    f16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
    if (e.residual) r = *reinterpret_cast<const f16x8*>(e.residual + (int64_t)z * e.strideR + (int64_t)m * e.ldr + n);
    if (e.act == ODISE_ACT_SILU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = mul_sigmoid(v[i], v[i]);
    } else if (e.act == ODISE_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
    } else if (e.act == ODISE_ACT_QUICKGELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = mul_sigmoid(v[i], 1.702f * v[i]);
    } else if (e.act == ODISE_ACT_GELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = gelu_erf(v[i]);
    }
    const int64_t off = (int64_t)z * e.strideC + (int64_t)m * e.ldc + n;
    if (e.c_dtype == ODISE_F16) {
        f16x8 t;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = (f16)(v[i] + (float)r[i]);
        *reinterpret_cast<f16x8*>((f16*)e.C + off) = t;
        if (rounded) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rounded[i] = (float)t[i];
        }
    } else {
        float* c = (float*)e.C + off;
        *reinterpret_cast<float4*>(c) = make_float4(v[0] + (float)r[0], v[1] + (float)r[1], v[2] + (float)r[2], v[3] + (float)r[3]);
        *reinterpret_cast<float4*>(c + 4) = make_float4(v[4] + (float)r[4], v[5] + (float)r[5], v[6] + (float)r[6], v[7] + (float)r[7]);
    }
}

// ---- epilogue through LDS (fp32 staging) in passes of WG wave-rows = WG * (BM / WAVES_M) rows x BN columns: the waves of those
// rows dump their accumulators, then ALL threads read 8 consecutive columns per lane, apply the epilogue and store 16 bytes.
// WG is as large as the kernel's LDS allows (epi_wave_rows): fewer passes = fewer barriers and more waves staging at once.
// The caller guarantees that every wave is done with the operand tiles (barrier) and that no LDS-DMA is outstanding.
// LDS requested by the two kernel families: the operand stages, stretched (when that keeps the blocks-per-CU count) so that the
// epilogue can stage more wave-rows per pass
constexpr int plain_lds_bytes(int BM, int BN, int WAVES_M) {
    const int stage = (BM + BN) * 64 * 2 * 2;
    const int all_rows = BM * (BN + 4) * 4;  // the whole tile staged in one pass
    // 4-wave tiles share a CU: only stretch while the resident block count (160 KiB / LDS) is unchanged
    const int f32 = (all_rows > stage && (160 * 1024) / all_rows == (160 * 1024) / stage) ? all_rows : stage;
    const int f16_all = BM * (BN + 8) * 2;   // fp16 staging of the whole tile (math-first epilogue), same residency rule
    return (f16_all > f32 && f16_all <= 160 * 1024 && (160 * 1024) / f16_all == (160 * 1024) / f32) ? f16_all : f32;
}
constexpr int pp_lds_bytes(int BM, int BN, int WAVES_M) {
    const int stage = (BM + BN) * 64 * 2 * 2;
    const int half_rows = (BM / 2) * (BN + 4) * 4;  // one block per CU anyway: stage half the tile per pass when it fits
    const int f32 = (half_rows > stage && half_rows <= 160 * 1024) ? half_rows : stage;
    const int f16_all = BM * (BN + 8) * 2;
    return (f16_all > f32 && f16_all <= 160 * 1024) ? f16_all : f32;
}
constexpr int halo_lds_bytes(int BN) {
    const int layout = 2 * BN * 64 * 2 + 2 * (41 + 1) * 1024;  // two B stages + two halo buffers (41 groups + a dummy one)
    const int all = 256 * (BN + 4) * 4;                        // the whole 256-row tile staged in one epilogue pass when that fits (BN = 128) ...
    const int epi = all <= 160 * 1024 ? all : all / 2;         // ... else two wave-rows per pass
    return layout > epi ? layout : epi;
}
// ---- fp16 staging of the math-first epilogue (gemm_epilogue_f16): (BN + 8) halves per row, EW wave-rows per pass
constexpr int epi16_bytes(int BM, int BN, int WAVES_M, int EW) { return EW * (BM / WAVES_M) * (BN + 8) * 2; }
constexpr int epi16_wave_rows(int BM, int BN, int WAVES_M, int lds_bytes) {
    int ew = lds_bytes / ((BM / WAVES_M) * (BN + 8) * 2);
    if (ew < 1) ew = 1;
    if (ew > WAVES_M) ew = WAVES_M;
    while (WAVES_M % ew) --ew;
    return ew;
}
// Where the math-first form runs (0 = keep the fp32-staged epilogue).  Measured on MI355X (tools/epi16_ab.py, profiles/r03_epilogue_f16_ab.txt):
// both forms are bound by the write burst of a round - every CU stores its tile at the same time, ~2.5 TB/s across the chip - so the
// shorter LDS / VALU path only pays where the tile is staged in ONE pass: +4 % on the 3x3 convolutions (halo tiles), +6..15 % on the
// GEGLU GEMMs (256x256), +15..35 % on 256x128; the two-pass 256x320 tile loses 15-30 % (half of the waves idle per pass and all stores at
// the end, where the fp32-staged form streams them out between its four passes), and the small 4-wave tiles are within +-8 % either way.
constexpr int epi16_rows_if_enabled(int BM, int BN, int WAVES_M, int lds_bytes) {
    return (BN != 320 && BM * BN >= 256 * 128) ? epi16_wave_rows(BM, BN, WAVES_M, lds_bytes) : 0;
}
constexpr int max_int(int a, int b) { return a > b ? a : b; }
constexpr int halo4_lds_bytes(int BN) {
    const int layout = 2 * BN * 64 * 2 + (41 + 1) * 1024;     // two weight stages + ONE halo buffer (41 groups + a dummy one)
    const int f16_all = 256 * (BN + 8) * 2;                   // fp16 staging of the whole tile
    const int f32_row = 64 * (BN + 4) * 4;                    // one wave-row of the fp32-staged form
    return max_int(max_int(layout, f16_all <= 80 * 1024 ? f16_all : 0), f32_row);
}
constexpr int epi_wave_rows(int BM, int BN, int WAVES_M, int lds_bytes) {
    const int per_row = (BM / WAVES_M) * (BN + 4) * 4;
    int wg = lds_bytes / per_row;
    if (wg < 1) wg = 1;
    if (wg > WAVES_M) wg = WAVES_M;
    while (WAVES_M % wg) --wg;
    return wg;
}
constexpr int epi_lds_bytes(int BM, int BN, int WAVES_M, int WG) { return WG * (BM / WAVES_M) * (BN + 4) * 4; }

// ---- how a wave's sub-tile sits in its accumulator registers -------------------------------------------------------------------------
// Per 32x32 block (p, j) of the sub-tile a lane owns NR rows x NC column quads; quad (rr, cc) = registers 4 (rr NC + cc) .. + 3 of the
// block's 16 = four consecutive columns of one row (the kernels multiply with the MFMA operands swapped, see gemm_epilogue).
//   L16 = false: one v_mfma_f32_32x32x16_f16 tile - row l & 31, quads 8 cc + 4 (l >> 5)                     (f32x16 acc[TM][TN])
//   L16 = true : 2 x 2 v_mfma_f32_16x16x32_f16 tiles - rows 16 rr + (l & 15), quads 16 cc + 4 (l >> 4)      (f32x4 acc[TM][TN][4], [2 rr + cc])
// Round 5: on operands that toggle like real data the matrix pipes are power-bound, and the 16x16x32 instruction sustains 1.95 PFLOP/s
// (1.84 GHz) where 32x32x16 sustains 1.62 (1.52 GHz) - tools/mfma_rate.py, profiles/r05_mfma_rate_by_shape.txt; same FLOPs, fragment
// bytes and LDS reads per 32x32 block either way.
template <bool L16>
struct FragLayout;
template <>
struct FragLayout<false> {
    static constexpr int NR = 1, NC = 4;
    static __device__ __forceinline__ int row(int lane, int) { return lane & 31; }
    static __device__ __forceinline__ int col(int lane, int cc) { return 8 * cc + 4 * (lane >> 5); }
    static __device__ __forceinline__ bool leader(int lane) { return lane < 32; }                    // one lane per row
    static __device__ __forceinline__ float row_sum(float v) { return v + __shfl_xor(v, 32); }       // over the lanes that share a row
};
template <>
struct FragLayout<true> {
    static constexpr int NR = 2, NC = 2;
    static __device__ __forceinline__ int row(int lane, int rr) { return 16 * rr + (lane & 15); }
    static __device__ __forceinline__ int col(int lane, int cc) { return 16 * cc + 4 * (lane >> 4); }
    static __device__ __forceinline__ bool leader(int lane) { return lane < 16; }
    static __device__ __forceinline__ float row_sum(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }
};
template <class ACC>
struct AccTraits;
template <int TM_, int TN_>
struct AccTraits<f32x16[TM_][TN_]> { static constexpr bool L16 = false; static constexpr int TM = TM_, TN = TN_; };
template <int TM_, int TN_>
struct AccTraits<f32x4[TM_][TN_][4]> { static constexpr bool L16 = true; static constexpr int TM = TM_, TN = TN_; };
template <int TM, int TN>
__device__ __forceinline__ float acc_get(const f32x16 (&a)[TM][TN], int p, int j, int r) { return a[p][j][r]; }
template <int TM, int TN>
__device__ __forceinline__ float acc_get(const f32x4 (&a)[TM][TN][4], int p, int j, int r) { return a[p][j][r >> 2][r & 3]; }

// ---- math-first epilogue for fp16 outputs ----------------------------------------------------------------------------------------------
// The fp32-staged epilogue below costs a 256x256 tile ~12 us, a third of a K = 1024 GEMM: the accumulators go through LDS as fp32 in up to
// four passes in which only the waves of one or two wave-rows write while the others wait, and every thread then walks 16 dependent rounds of
// {LDS read, bias / activation / residual arithmetic, convert, store}.  Here every wave applies the epilogue to its OWN accumulators in
// registers (lane l of a 32x32 tile owns row l & 31 and, per register quad, four consecutive columns: bias and residual come in as 16- and
// 8-byte loads of exactly those columns), rounds to fp16 and stages 8 bytes per quad; the tile then sits in LDS in its final form at half the
// size - one pass for every tile but 256x320 - and the second phase is a pure copy: all of a thread's 16-byte LDS reads are issued back to
// back, then its global stores.  Element for element the arithmetic is that of epi_fast8 in the same order: results are bit-identical
// (tools/epi_ab.py compares the two forms; ODISE_EPI_OLD=1 in the tools build keeps the fp32-staged form).
// Staging rows are (BN + 8) halves: 16-byte aligned for the copy's ds_read_b128; the quad writes of 16 consecutive rows land 2-way on the
// banks (row pitch = 4 banks mod 32), which stays below the write instruction's own issue cost.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EW, bool HALO, bool STATS, bool GEGLU, class ACC>
__device__ __forceinline__ void gemm_epilogue_f16(const GemmArgs& g, ACC& acc, char* smem, int m0, int n0, int zb) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(AccTraits<ACC>::TM == TM && AccTraits<ACC>::TN == TN, "accumulator array does not match the wave tile");
    using FL = FragLayout<AccTraits<ACC>::L16>;
    constexpr int NR = FL::NR, NC = FL::NC, TR = TM * NR;   // rows per lane: TR
    constexpr int ROWS = EW * WTM;                 // tile rows per pass
    constexpr int BNO = GEGLU ? BN / 2 : BN;       // output columns of the tile
    constexpr int PITCH = BN + 8;                  // halves per staging row
    constexpr int CH = BNO / 8;                    // 16-byte chunks per output row
    constexpr int TOTAL = ROWS * CH, ITERS = (TOTAL + NT - 1) / NT;
    const GemmEpi& e = g.epi;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    f16* stg = reinterpret_cast<f16*>(smem);
    const int NO = GEGLU ? (g.N >> 1) : g.N;       // output columns of the problem
    const int no0 = GEGLU ? (n0 >> 1) : n0;
    // tile row -> output row (the halo kernel owns a 16x16 pixel patch: m0 / BM = patch index)
    auto row_to_m = [&](int rt) -> int {
        if (HALO) {
            const int patch = m0 / BM;
            const int per_img = g.cg.halo_tx * g.cg.halo_ty;
            const int img = patch / per_img, pr = patch - img * per_img;
            const int oy = (pr / g.cg.halo_tx) * 16 + (rt >> 4), ox = (pr % g.cg.halo_tx) * 16 + (rt & 15);
            return (oy < g.cg.OH && ox < g.cg.OW) ? (img * g.cg.OH + oy) * g.cg.OW + ox : g.M;
        }
        return m0 + rt;
    };
    const bool stats = STATS && (NT % CH == 0) && e.gn_stats != nullptr;
    // folded-LayerNorm terms (dense GEMMs only: compiled out of the conv / GEGLU instances)
    constexpr bool LN_OK = !HALO && !STATS && !GEGLU;
    const float* const ln_part = LN_OK ? e.ln_part : nullptr;
    const float* const ln_colsum = LN_OK ? e.ln_colsum : nullptr;
    float* const ln_final_out = LN_OK ? e.ln_final_out : nullptr;
    const float* const ln_final = LN_OK ? e.ln_final : nullptr;
    const float* const ln_rowsum = LN_OK ? e.ln_rowsum : nullptr;
    float* const ln_stats_out = LN_OK ? e.ln_stats_out : nullptr;
    float s8[8], q8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s8[i] = q8[i] = 0.f;
#pragma unroll
    for (int gp = 0; gp < WAVES_M / EW; ++gp) {
        if (gp > 0) lds_barrier();  // the copy reads of the previous pass are done
        if (wm / EW == gp) {
            int mr[TR];
            float alpha_r[TR], bm_r[TR];
            unsigned grp_r[TR];
            // LayerNorm folded into this GEMM (gemm_ln).  With W' = W diag(gamma), b' = b + W beta and cs[n] = sum_k W'[n,k]:
            //   LN(x) W^T + b = rstd_m (x W'^T - mean_m cs) + b'  ->  v * rstd_m + (b'[n] + r1_m cs[n]),  r1_m = -mean_m rstd_m,
            // the row statistics coming as partial (sum, sum of squares) from the epilogue of the GEMM that produced x (ln_stats_out below).
            // In the swapped form (rows of W are the normalised tokens) the same with rows and columns exchanged, from finished (r1, rstd).
            constexpr int NPW = (WTN % kLnPartCols == 0) ? WTN / kLnPartCols : 1;   // statistics parts (kLnPartCols columns each) per wave column
            float r1_r[TR], rs_r[TR], ps_r[TR][NPW], pq_r[TR][NPW];
#pragma unroll
            for (int p = 0; p < TR; ++p) {   // p = NR * (32-row block) + rr
                const int m = row_to_m(wm * WTM + (p / NR) * 32 + FL::row(lane, p % NR));
                mr[p] = m;
                const bool ok = m < g.M;
                float alpha = e.alpha;
                if (e.scale_m && ok) alpha *= e.scale_m[m];
                r1_r[p] = rs_r[p] = 0.f;
#pragma unroll
                for (int u = 0; u < NPW; ++u) ps_r[p][u] = pq_r[p][u] = 0.f;
                if (ln_part && ok) {
                    const float* pp = ln_part + (int64_t)m * e.ln_P * 2;
                    float s1 = 0.f, s2 = 0.f;
                    for (int i = 0; i < e.ln_P; ++i) { s1 += pp[2 * i]; s2 += pp[2 * i + 1]; }
                    const float mean = s1 * e.ln_inv_c;
                    const float var = fmaxf(s2 * e.ln_inv_c - mean * mean, 0.f);
                    const float rstd = rsqrtf(var + e.ln_eps);
                    alpha *= rstd;
                    r1_r[p] = -mean * rstd;
                    if (ln_final_out && n0 == 0 && wn == 0 && FL::leader(lane)) {
                        ln_final_out[2 * (int64_t)m] = r1_r[p];
                        ln_final_out[2 * (int64_t)m + 1] = rstd;
                    }
                }
                if (ln_rowsum && ok) rs_r[p] = ln_rowsum[m];
                alpha_r[p] = alpha;
                bm_r[p] = (e.bias_m && ok) ? e.bias_m[m] : 0.f;
                grp_r[p] = (e.rowgroup_add && ok) ? (unsigned)m / (unsigned)e.rows_per_group : 0u;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const int cl = wn * WTN + j * 32 + FL::col(lane, q);   // tile column of this lane's four values
                    const int n = n0 + cl;
                    const bool nok = n < g.N;                            // N % 8 == 0 on this path: the four columns are in or out together
                    float4 bn = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e.bias_n && nok) bn = *reinterpret_cast<const float4*>(e.bias_n + n);
                    float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f), fa = cs4, fb = cs4;
                    if (ln_colsum && nok) cs4 = *reinterpret_cast<const float4*>(ln_colsum + n);
                    if (ln_final && nok) {   // (r1, rstd) of columns n, n + 1 | n + 2, n + 3
                        fa = *reinterpret_cast<const float4*>(ln_final + 2 * (int64_t)n);
                        fb = *reinterpret_cast<const float4*>(ln_final + 2 * (int64_t)n + 4);
                    }
                    const float csv[4] = {cs4.x, cs4.y, cs4.z, cs4.w};
                    const float r1c[4] = {fa.x, fa.z, fb.x, fb.z}, rsc[4] = {fa.y, fa.w, fb.y, fb.w};
#pragma unroll
                    for (int p = 0; p < TR; ++p) {
                        const bool ok = nok && mr[p] < g.M;
                        const int r0 = 4 * ((p % NR) * NC + q);   // first register of quad (rr, q) of block (p / NR, j)
                        float v[4] = {acc_get(acc, p / NR, j, r0), acc_get(acc, p / NR, j, r0 + 1), acc_get(acc, p / NR, j, r0 + 2), acc_get(acc, p / NR, j, r0 + 3)};
                        float b[4] = {0.f, 0.f, 0.f, 0.f};
                        if (e.bias_n) { b[0] = bn.x; b[1] = bn.y; b[2] = bn.z; b[3] = bn.w; }
                        if (e.rowgroup_add && ok) {
                            const float4 r = *reinterpret_cast<const float4*>(e.rowgroup_add + (int64_t)grp_r[p] * e.ldg + n);
                            b[0] += r.x; b[1] += r.y; b[2] += r.z; b[3] += r.w;
                        }
                        if (e.bias_m) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) b[i] += bm_r[p];
                        }
                        if (ln_colsum) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) b[i] += r1_r[p] * csv[i];
                        }
                        if (ln_final) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = v[i] * (alpha_r[p] * rsc[i]) + (b[i] + rs_r[p] * r1c[i]);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = v[i] * alpha_r[p] + b[i];
                        }
                        const int rl = (wm % EW) * WTM + (p / NR) * 32 + FL::row(lane, p % NR);   // staging row
                        if (GEGLU) {
                            // columns are (a, gate) pairs; the output has N/2 columns (no activation / residual on this path)
                            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                            f16x2 t;
                            t[0] = (f16)(v[0] * gelu_erf(v[1]));
                            t[1] = (f16)(v[2] * gelu_erf(v[3]));
                            *reinterpret_cast<f16x2*>(&stg[rl * PITCH + (cl >> 1)]) = t;
                        } else {
                            f16x4 rr = {0, 0, 0, 0};
                            if (e.residual && ok) rr = *reinterpret_cast<const f16x4*>(e.residual + (int64_t)zb * e.strideR + (int64_t)mr[p] * e.ldr + n);
                            if (e.act == ODISE_ACT_SILU) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] = mul_sigmoid(v[i], v[i]);
                            } else if (e.act == ODISE_ACT_RELU) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
                            } else if (e.act == ODISE_ACT_QUICKGELU) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] = mul_sigmoid(v[i], 1.702f * v[i]);
                            } else if (e.act == ODISE_ACT_GELU) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
                            }
                            f16x4 t;
#pragma unroll
                            for (int i = 0; i < 4; ++i) t[i] = (f16)(v[i] + (float)rr[i]);
                            *reinterpret_cast<f16x4*>(&stg[rl * PITCH + cl]) = t;
                            if (ln_stats_out && ok) {
                                constexpr int TPP = kLnPartCols / 32;   // 32-column tiles per part
#pragma unroll
                                for (int i = 0; i < 4; ++i) { const float r = (float)t[i]; ps_r[p][(j / TPP) % NPW] += r; pq_r[p][(j / TPP) % NPW] += r * r; }
                            }
                        }
                    }
                }
            if (ln_stats_out) {   // this wave's WTN columns of its rows, in parts of kLnPartCols (the same partition whatever the kernel's wave layout):
                                  // the two lane halves hold the two 4-column halves of every 8
                const int part0 = (n0 + wn * WTN) / kLnPartCols, parts = (g.N + kLnPartCols - 1) / kLnPartCols;
#pragma unroll
                for (int p = 0; p < TR; ++p)
#pragma unroll
                    for (int u = 0; u < NPW; ++u) {
                        const float s1 = FL::row_sum(ps_r[p][u]), s2 = FL::row_sum(pq_r[p][u]);
                        if (FL::leader(lane) && mr[p] < g.M && part0 + u < parts) {
                            float* o = ln_stats_out + ((int64_t)mr[p] * parts + part0 + u) * 2;
                            o[0] = s1;
                            o[1] = s2;
                        }
                    }
            }
        }
        lds_barrier();
        // ---- copy phase: the tile is final; 16 bytes per thread and round, in batches of CB rounds: the batch's LDS reads are issued back to
        // back, then its stores (CB is kept small while later passes' accumulators are still live in registers)
        constexpr int CB = (EW == WAVES_M) ? (ITERS < 8 ? ITERS : 8) : (ITERS < 4 ? ITERS : 4);
#pragma unroll
        for (int it0 = 0; it0 < ITERS; it0 += CB) {
            f16x8 buf[CB];
            int mm[CB], nn[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int c = tid + (it0 + u) * NT;
                const bool valid = (it0 + u < ITERS) && ((TOTAL % NT == 0) || c < TOTAL);
                const int row = c / CH, c8 = c - row * CH;
                buf[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (valid) buf[u] = *reinterpret_cast<const f16x8*>(&stg[row * PITCH + c8 * 8]);
                mm[u] = valid ? row_to_m(gp * ROWS + row) : g.M;
                nn[u] = no0 + c8 * 8;
            }
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                if (mm[u] < g.M && nn[u] + 8 <= NO) {
                    *reinterpret_cast<f16x8*>((f16*)e.C + (int64_t)zb * e.strideC + (int64_t)mm[u] * e.ldc + nn[u]) = buf[u];
                    if (stats) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { const float r = (float)buf[u][i]; s8[i] += r; q8[i] += r * r; }
                    }
                }
            }
        }
    }
    if (stats) {
        constexpr int RL = NT / CH;  // row lanes per column chunk
        lds_barrier();                // the last pass is done with the staging buffer
        float* red = reinterpret_cast<float*>(smem);  // [RL][BN][2]
        const int c8 = tid % CH, rl = tid / CH;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            red[((rl * BN) + c8 * 8 + i) * 2 + 0] = s8[i];
            red[((rl * BN) + c8 * 8 + i) * 2 + 1] = q8[i];
        }
        lds_barrier();
        for (int c = tid; c < BN; c += NT) {
            float a = 0.f, b = 0.f;
            for (int r = 0; r < RL; ++r) { a += red[(r * BN + c) * 2]; b += red[(r * BN + c) * 2 + 1]; }
            if (n0 + c < g.N) {
                float* o = e.gn_stats + ((int64_t)(m0 / BM) * g.N + n0 + c) * 2;
                o[0] = a;
                o[1] = b;
            }
        }
    }
}

// ---- wave-private epilogue (round 5) -------------------------------------------------------------------------------------------------
// Measured (tools/g8_ablate.py, profiles/r05_epilogue_share.txt): the two block-wide epilogues below cost a 256x256 tile 18-23 us per residency
// round - 40 of 110 us on the CLIP c_fc GEMM (K = 1024), 16-20 of 115 us at 4096^3 - where the yardstick kernel of csrc/gemm8p.hip spends
// 3-4 us: they stage the whole tile behind block barriers (every wave waits for the slowest), the math-first form reads the residual and
// writes LDS in fragment layout (8-byte pieces of 16 rows per instruction), and with one block per CU nothing else runs meanwhile.
// Here a wave needs nobody after the barrier that ends the main loop: it stages R rows x WTN columns of its OWN accumulators as fp32 in its
// private slice of the block's LDS (float4 writes of a 16-lane group cover the 64 banks once: pitch WTN + 4), reads them back as 8
// consecutive columns of a row per lane, applies the epilogue (same operations per element in the same order as epi_fast8 /
// gemm_epilogue_f16: the same bits) and stores 16 bytes per lane - residual reads and output writes are whole 128-byte row segments.  No
// block barrier, no wave waits for another.  Covers every `fast` (aligned) case without split-K: bias / per-image vector / per-row terms /
// activation / GEGLU / residual / fp16 or fp32 output, the folded-LayerNorm terms and the fused GroupNorm statistics; the rest stays below.
constexpr int wave_epi_rows(int WTM, int WTN, int waves, int lds_bytes, int rmin) {
    const int chw = WTN / 8;
    int best = 0;
    for (int r = rmin; r <= WTM; r += rmin)
        if (WTM % r == 0 && (r * chw) % 64 == 0 && r * (WTN + 4) * 4 <= lds_bytes / waves) best = r;
    return best;
}
template <int BM, int BN, int WAVES_M, int WAVES_N, bool HALO, bool STATS, bool GEGLU_OK, int LDSB, class ACC>
__device__ __forceinline__ void gemm_epilogue_wave(const GemmArgs& g, ACC& acc, char* smem, int m0, int n0, int zb) {
    constexpr bool L16 = AccTraits<ACC>::L16;
    using FL = FragLayout<L16>;
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TN = WTN / 32;
    constexpr int R = wave_epi_rows(WTM, WTN, NWAVES, LDSB, L16 ? 16 : 32);
    static_assert(R > 0, "no wave-private staging fits");
    constexpr int PITCH = WTN + 4;             // floats
    constexpr int CHW = WTN / 8;               // 8-column chunks per row of the wave tile
    constexpr int ITEMS = R * CHW / 64;        // (row, chunk) items per lane and pass
    constexpr bool FIXED = (64 % CHW) == 0;    // a lane keeps ONE column chunk over all its items: per-column terms are loaded once
    constexpr int RSUB = L16 ? 16 : 32;        // rows a lane group covers per fragment row index
    const GemmEpi& e = g.epi;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    float* stg = reinterpret_cast<float*>(smem) + (size_t)wave * R * PITCH;
    auto row_to_m = [&](int rt) -> int {
        if (HALO) {
            const int patch = m0 / BM;
            const int per_img = g.cg.halo_tx * g.cg.halo_ty;
            const int img = patch / per_img, pr = patch - img * per_img;
            const int oy = (pr / g.cg.halo_tx) * 16 + (rt >> 4), ox = (pr % g.cg.halo_tx) * 16 + (rt & 15);
            return (oy < g.cg.OH && ox < g.cg.OW) ? (img * g.cg.OH + oy) * g.cg.OW + ox : g.M;
        }
        return m0 + rt;
    };
    const bool stats = STATS && FIXED && e.gn_stats != nullptr;
    constexpr bool LN_OK = !HALO && !STATS;
    const float* const ln_part = LN_OK ? e.ln_part : nullptr;
    const float* const ln_colsum = LN_OK ? e.ln_colsum : nullptr;
    float* const ln_final_out = LN_OK ? e.ln_final_out : nullptr;
    const float* const ln_final = LN_OK ? e.ln_final : nullptr;
    const float* const ln_rowsum = LN_OK ? e.ln_rowsum : nullptr;
    float* const ln_stats_out = (LN_OK && CHW % 8 == 0) ? e.ln_stats_out : nullptr;
    const bool geglu = GEGLU_OK && e.geglu;
    // nothing per row but the residual, fp16 output: the lean item loop
    // (round 6: the producers of the LayerNorm row statistics - out-proj / c_proj with ln_stats_out - stay in the lean loop too)
    const bool plain = !geglu && e.c_dtype == ODISE_F16 && !e.scale_m && !e.bias_m && !e.rowgroup_add && !ln_part && !ln_colsum && !ln_final && !ln_rowsum &&
                       !ln_final_out;
    float s8[8], q8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s8[i] = q8[i] = 0.f;
    // per-column terms of the lane's chunk (FIXED): bias_n, LayerNorm column sums / finished column statistics
    float bn8[8], cs8[8], r1c8[8], rsc8[8];
    auto load_cols = [&](int n) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { bn8[i] = 0.f; cs8[i] = 0.f; r1c8[i] = 0.f; rsc8[i] = 1.f; }
        if (n + 8 > g.N) return;
        if (e.bias_n) {
            const float4 b0 = *reinterpret_cast<const float4*>(e.bias_n + n), b1 = *reinterpret_cast<const float4*>(e.bias_n + n + 4);
            bn8[0] = b0.x; bn8[1] = b0.y; bn8[2] = b0.z; bn8[3] = b0.w; bn8[4] = b1.x; bn8[5] = b1.y; bn8[6] = b1.z; bn8[7] = b1.w;
        }
        if (ln_colsum) {
            const float4 c0 = *reinterpret_cast<const float4*>(ln_colsum + n), c1 = *reinterpret_cast<const float4*>(ln_colsum + n + 4);
            cs8[0] = c0.x; cs8[1] = c0.y; cs8[2] = c0.z; cs8[3] = c0.w; cs8[4] = c1.x; cs8[5] = c1.y; cs8[6] = c1.z; cs8[7] = c1.w;
        }
        if (ln_final) {   // (r1, rstd) per column
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 f = *reinterpret_cast<const float4*>(ln_final + 2 * (int64_t)(n + 2 * i));
                r1c8[2 * i] = f.x; rsc8[2 * i] = f.y; r1c8[2 * i + 1] = f.z; rsc8[2 * i + 1] = f.w;
            }
        }
    };
    if (FIXED) load_cols(n0 + wn * WTN + (lane % CHW) * 8);
    // Folded LayerNorm, consumer side: finish the statistics of the wave tile's rows ONCE, before the passes - lane l takes rows l, l + 64, ... and
    // walks the producer's P partial pairs of its row in the fixed order of gemm_epilogue_f16 (the same bits in both forms and in every wave
    // and block that covers the row).  The item loop fetches a row's pair from the lane that holds it.  (Finished per item - two dependent
    // global loads in a rolled loop, 16 times per wave tile - this was 35-50 us of the 110-150 us CLIP GEMMs that read LN(x).)
    constexpr int NH = (WTM + 63) / 64;
    float ln_rs[NH], ln_r1[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) ln_rs[h] = ln_r1[h] = 0.f;
    if (ln_part) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int rw = h * 64 + lane;
            const int m = m0 + wm * WTM + rw;          // LN_OK: never a halo tile
            if (rw < WTM && m < g.M) {
                const float2* pp = reinterpret_cast<const float2*>(ln_part) + (int64_t)m * e.ln_P;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
                for (int i = 0; i < e.ln_P; ++i) { const float2 t = pp[i]; s1 += t.x; s2 += t.y; }
                const float mean = s1 * e.ln_inv_c;
                const float var = fmaxf(s2 * e.ln_inv_c - mean * mean, 0.f);
                const float rstd = rsqrtf(var + e.ln_eps);
                ln_rs[h] = rstd;
                ln_r1[h] = -mean * rstd;
                if (ln_final_out && n0 + wn * WTN == 0) {
                    ln_final_out[2 * (int64_t)m] = ln_r1[h];
                    ln_final_out[2 * (int64_t)m + 1] = rstd;
                }
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < WTM / R; ++ps) {
        // ---- stage rows [ps R, ps R + R) of the wave tile (the previous pass's reads are retired: same wave, LDS operations complete in order)
#pragma unroll
        for (int p = 0; p < WTM / 32; ++p)
#pragma unroll
            for (int rr = 0; rr < FL::NR; ++rr) {
                const int rbase = p * 32 + rr * RSUB;          // first tile row of this fragment row group (compile-time after unrolling)
                if (rbase < ps * R || rbase >= (ps + 1) * R) continue;
                const int row = rbase - ps * R + (L16 ? (lane & 15) : (lane & 31));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int cc = 0; cc < FL::NC; ++cc) {
                        const int r0 = 4 * (rr * FL::NC + cc);
                        float4 q = make_float4(acc_get(acc, p, j, r0), acc_get(acc, p, j, r0 + 1), acc_get(acc, p, j, r0 + 2), acc_get(acc, p, j, r0 + 3));
                        *reinterpret_cast<float4*>(&stg[row * PITCH + j * 32 + FL::col(lane, cc)]) = q;
                    }
            }
        // the staged rows are complete before any lane reads them back (the store's data moves from the registers to the LDS asynchronously)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // ---- read back 8 consecutive columns of a row per lane, finish and store.  The item loops are ROLLED (a dozen items per pass; unrolled they
        // were ~800 instructions each with every feature test repeated) and everything that does not change from item to item is set up before them
        if (FIXED && plain) {
            // lean form: bias / activation / residual / fp16 store (+ the fused GroupNorm sums): what the VAE, most UNet and head layers need
            const int c8 = lane % CHW;
            const int n = n0 + wn * WTN + c8 * 8;
            const bool nok = n + 8 <= g.N;
            const float alpha = e.alpha;
            const int ln_parts = (g.N + kLnPartCols - 1) / kLnPartCols;
#pragma unroll 1
            for (int it = 0; it < ITEMS; ++it) {
                const int row = it * (64 / CHW) + lane / CHW;
                const int m = row_to_m(wm * WTM + ps * R + row);
                const bool ok = nok && m < g.M;
                const float4 t0 = *reinterpret_cast<const float4*>(&stg[row * PITCH + c8 * 8]);
                const float4 t1 = *reinterpret_cast<const float4*>(&stg[row * PITCH + c8 * 8 + 4]);
                f16x8 rr8 = {0, 0, 0, 0, 0, 0, 0, 0};
                if (e.residual && ok) rr8 = *reinterpret_cast<const f16x8*>(e.residual + (int64_t)zb * e.strideR + (int64_t)m * e.ldr + n);
                float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] * alpha + bn8[i];
                if (e.act == ODISE_ACT_SILU) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = mul_sigmoid(v[i], v[i]);
                } else if (e.act == ODISE_ACT_RELU) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
                } else if (e.act == ODISE_ACT_QUICKGELU) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = mul_sigmoid(v[i], 1.702f * v[i]);
                } else if (e.act == ODISE_ACT_GELU) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = gelu_erf(v[i]);
                }
                f16x8 t;
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = (f16)(v[i] + (float)rr8[i]);
                if (ok) {
                    *reinterpret_cast<f16x8*>((f16*)e.C + (int64_t)zb * e.strideC + (int64_t)m * e.ldc + n) = t;
                    if (stats) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { const float r = (float)t[i]; s8[i] += r; q8[i] += r * r; }
                    }
                }
                if (ln_stats_out) {   // partial (sum, sum of squares) of this row over the 64 columns of its part = the 8 lanes (chunks) of the part
                    float ps1 = 0.f, pq1 = 0.f;
                    if (ok) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { const float r = (float)t[i]; ps1 += r; pq1 += r * r; }
                    }
#pragma unroll
                    for (int sft = 1; sft < 8; sft <<= 1) { ps1 += __shfl_xor(ps1, sft); pq1 += __shfl_xor(pq1, sft); }
                    if ((c8 & 7) == 0 && ok) {
                        float* o = ln_stats_out + ((int64_t)m * ln_parts + n / kLnPartCols) * 2;
                        o[0] = ps1;
                        o[1] = pq1;
                    }
                }
            }
            continue;   // next pass
        }
#pragma unroll 1
        for (int it = 0; it < ITEMS; ++it) {
            const int idx = it * 64 + lane;
            const int row = idx / CHW, c8 = idx - row * CHW;
            const int rt = wm * WTM + ps * R + row;            // block tile row
            const int m = row_to_m(rt);
            const int n = n0 + wn * WTN + c8 * 8;
            const float4 t0 = *reinterpret_cast<const float4*>(&stg[row * PITCH + c8 * 8]);
            const float4 t1 = *reinterpret_cast<const float4*>(&stg[row * PITCH + c8 * 8 + 4]);
            const bool ok = m < g.M && n + 8 <= g.N;
            if (!FIXED) load_cols(n);
            float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
            float b[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) b[i] = bn8[i];
            f16x8 rr8 = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok && e.residual && !geglu) rr8 = *reinterpret_cast<const f16x8*>(e.residual + (int64_t)zb * e.strideR + (int64_t)m * e.ldr + n);
            float alpha = e.alpha, r1 = 0.f, rs = 0.f;
            if (ln_part) {
                // this row's (rstd, -mean rstd) sit in the lane that finished them before the first pass (ln_rs / ln_r1 above)
                const int rw = ps * R + row;
                float rstd = 0.f;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const float t = __shfl(ln_rs[h], rw & 63), u = __shfl(ln_r1[h], rw & 63);
                    if ((rw >> 6) == h) { rstd = t; r1 = u; }
                }
                alpha *= rstd;
            }
            if (ok) {
                if (e.scale_m) alpha *= e.scale_m[m];
                if (ln_rowsum) rs = ln_rowsum[m];
                if (e.rowgroup_add) {
                    const float* rg = e.rowgroup_add + (int64_t)((unsigned)m / (unsigned)e.rows_per_group) * e.ldg + n;
                    const float4 g0 = *reinterpret_cast<const float4*>(rg), g1 = *reinterpret_cast<const float4*>(rg + 4);
                    b[0] += g0.x; b[1] += g0.y; b[2] += g0.z; b[3] += g0.w; b[4] += g1.x; b[5] += g1.y; b[6] += g1.z; b[7] += g1.w;
                }
                if (e.bias_m) {
                    const float bm = e.bias_m[m];
#pragma unroll
                    for (int i = 0; i < 8; ++i) b[i] += bm;
                }
            }
            if (ln_colsum) {
#pragma unroll
                for (int i = 0; i < 8; ++i) b[i] += r1 * cs8[i];
            }
            if (ln_final) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] * (alpha * rsc8[i]) + (b[i] + rs * r1c8[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] * alpha + b[i];
            }
            if (geglu) {   // columns are (a, gate) pairs; the output has N/2 columns (no activation / residual on this path)
                if (ok) {
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = v[2 * i] * gelu_erf(v[2 * i + 1]);
                    const int64_t off = (int64_t)zb * e.strideC + (int64_t)m * e.ldc + (n >> 1);
                    if (e.c_dtype == ODISE_F16) {
                        f16x4 t;
#pragma unroll
                        for (int i = 0; i < 4; ++i) t[i] = (f16)o[i];
                        *reinterpret_cast<f16x4*>((f16*)e.C + off) = t;
                    } else {
                        *reinterpret_cast<float4*>((float*)e.C + off) = make_float4(o[0], o[1], o[2], o[3]);
                    }
                }
                continue;
            }
            if (e.act == ODISE_ACT_SILU) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = mul_sigmoid(v[i], v[i]);
            } else if (e.act == ODISE_ACT_RELU) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
            } else if (e.act == ODISE_ACT_QUICKGELU) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = mul_sigmoid(v[i], 1.702f * v[i]);
            } else if (e.act == ODISE_ACT_GELU) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = gelu_erf(v[i]);
            }
            const int64_t off = (int64_t)zb * e.strideC + (int64_t)m * e.ldc + n;
            float ps1 = 0.f, pq1 = 0.f;
            if (e.c_dtype == ODISE_F16) {
                f16x8 t;
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = (f16)(v[i] + (float)rr8[i]);
                if (ok) {
                    *reinterpret_cast<f16x8*>((f16*)e.C + off) = t;
                    if (stats) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { const float r = (float)t[i]; s8[i] += r; q8[i] += r * r; }
                    }
                    if (ln_stats_out) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { const float r = (float)t[i]; ps1 += r; pq1 += r * r; }
                    }
                }
            } else if (ok) {
                float* c = (float*)e.C + off;
                *reinterpret_cast<float4*>(c) = make_float4(v[0] + (float)rr8[0], v[1] + (float)rr8[1], v[2] + (float)rr8[2], v[3] + (float)rr8[3]);
                *reinterpret_cast<float4*>(c + 4) = make_float4(v[4] + (float)rr8[4], v[5] + (float)rr8[5], v[6] + (float)rr8[6], v[7] + (float)rr8[7]);
            }
            if (ln_stats_out) {   // partial (sum, sum of squares) of this row over the 64 columns of its part: the 8 lanes (chunks) of the part
#pragma unroll
                for (int sft = 1; sft < 8; sft <<= 1) { ps1 += __shfl_xor(ps1, sft); pq1 += __shfl_xor(pq1, sft); }
                if ((c8 & 7) == 0 && ok) {
                    const int parts = (g.N + kLnPartCols - 1) / kLnPartCols;
                    float* o = ln_stats_out + ((int64_t)m * parts + n / kLnPartCols) * 2;
                    o[0] = ps1;
                    o[1] = pq1;
                }
            }
        }
    }
    if (STATS) {
        // fused GroupNorm statistics: per-channel (sum, sum of squares) of the block's rows.  A lane holds its chunk's sums over its rows; fold the
        // lanes of the wave that share a chunk (fixed order), then the WAVES_M waves of a wave column through LDS
        if (stats) {
#pragma unroll
            for (int sft = CHW; sft < 64; sft <<= 1)
#pragma unroll
                for (int i = 0; i < 8; ++i) { s8[i] += __shfl_xor(s8[i], sft); q8[i] += __shfl_xor(q8[i], sft); }
        }
        __syncthreads();   // every wave is done with its staging slice (block-uniform branch: `stats` does not depend on the thread)
        if (stats) {
            float* red = reinterpret_cast<float*>(smem);   // [WAVES_M][BN][2]
            if (lane < CHW) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    red[((wm * BN) + wn * WTN + lane * 8 + i) * 2 + 0] = s8[i];
                    red[((wm * BN) + wn * WTN + lane * 8 + i) * 2 + 1] = q8[i];
                }
            }
            __syncthreads();
            for (int c = threadIdx.x; c < BN; c += 64 * NWAVES) {
                float a = 0.f, bb = 0.f;
                for (int r = 0; r < WAVES_M; ++r) { a += red[(r * BN + c) * 2]; bb += red[(r * BN + c) * 2 + 1]; }
                if (n0 + c < g.N) {
                    float* o = e.gn_stats + ((int64_t)(m0 / BM) * g.N + n0 + c) * 2;
                    o[0] = a;
                    o[1] = bb;
                }
            }
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int WG, bool HALO = false, bool STATS = false, bool PAIRS = true, int EW16 = 0, bool GEGLU_OK = false, int LDSB = 0,
          class ACC>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, ACC& acc, char* smem, int m0, int n0, int z, int zb, bool split) {
    using FL = FragLayout<AccTraits<ACC>::L16>;
    // LDSB = the kernel's LDS request: the wave-private form wherever it applies (tools: ODISE_EPI_OLD=1 keeps the block-wide forms).  8-wave kernels
    // only: in the 4-wave kernels (64x64 .. 128x128 tiles, several blocks per CU) it returned rare wrong elements on the hardware - one dword of a
    // staged row read as zero in lanes 48-63, not cured by a barrier or a full LDS wait between staging and read-back, never seen with 8 waves
    // (tools/epi_debug.py; unexplained, so those kernels keep the block-wide forms)
    // Where it measured faster (tools/g8_shapes.py, profiles/r05_epilogue_forms.txt): wave tiles whose 8-column chunks divide the wavefront (a lane keeps one
    // chunk: everything but the 256x320 tile), without GEGLU (its half-width rows leave the lean loop; the block-wide form is ~15 % ahead there).
    // ODISE_GEMM_FLAGS 32768 keeps the block-wide forms everywhere (A/B of whole steps: bench.py --gemm-flags).
    if constexpr (LDSB > 0 && WAVES_M * WAVES_N == 8 && (64 % ((BN / WAVES_N) / 8)) == 0) {
        if (g.epi.fast && !split && !g.epi.geglu && !g.epi_block && !ODISE_ABLATE(g, 8 | 32)) {
            gemm_epilogue_wave<BM, BN, WAVES_M, WAVES_N, HALO, STATS, GEGLU_OK, LDSB>(g, acc, smem, m0, n0, zb);
            return;
        }
    }
    if constexpr (EW16 > 0) {
        if (g.epi.f16path && !split && !ODISE_ABLATE(g, 8 | 32)) {   // tools: ODISE_EPI_OLD=1 (bit 32) keeps the fp32-staged form for A/B runs
            if constexpr (GEGLU_OK) {
                if (g.epi.geglu) { gemm_epilogue_f16<BM, BN, WAVES_M, WAVES_N, EW16, HALO, false, true>(g, acc, smem, m0, n0, zb); return; }
            }
            if (!g.epi.geglu) { gemm_epilogue_f16<BM, BN, WAVES_M, WAVES_N, EW16, HALO, STATS, false>(g, acc, smem, m0, n0, zb); return; }
        }
    }
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int ROWS = WG * WTM;  // rows per pass
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    constexpr int LDS_LD = BN + 4;
    constexpr int CH = BN / 8;
    float* stg = reinterpret_cast<float*>(smem);
    // fused GroupNorm statistics: with NT % CH == 0 a thread keeps the same 8 columns over the whole loop, so it accumulates their
    // (sum, sum of squares) over its rows in registers; the block folds the NT / CH row lanes in LDS in a fixed order afterwards
    // (STATS is only instantiated for the conv kernels that feed GroupNorms: the 16 accumulators cost registers in a 128-accumulator epilogue)
    const bool stats = STATS && (NT % CH == 0) && g.epi.gn_stats != nullptr && !split;
    float s8[8], q8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s8[i] = q8[i] = 0.f;
    // Residual prefetch.  The read phase below fetches the residual tile row by row, one HBM round trip per row group: measured, a conv
    // with a residual runs 10-21 % longer than the same conv without (K = 1152: 1.85 -> 2.25 ms), i.e. those round trips are exposed.
    // Touch every 128-byte line of the block's residual tile NOW (one dword per lane and line, 2-3 loads per thread); the lines travel to
    // the L2 while the accumulators are staged, and the fake use after the first staging barrier retires the loads before the read phase.
    // Measured (tools/halo512_probe.py, same box): -5..9 % on the VAE convolutions that carry a residual.
    constexpr int LPR = (BN * 2 + 127) / 128;                 // lines per tile row
    constexpr int NPF = (BM * LPR + NT - 1) / NT;             // prefetch loads per thread
    unsigned pf[NPF];
    const bool prefetch = g.epi.residual != nullptr && !split && !ODISE_ABLATE(g, 64);   // tools: ODISE_NO_RES_PREFETCH=1 (bit 64) for A/B runs
    if (prefetch) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int li = tid + i * NT;
            const int rt = li / LPR, seg = li - rt * LPR;
            int m = m0 + rt;
            if (HALO) {
                const int patch = m0 / BM;
                const int per_img = g.cg.halo_tx * g.cg.halo_ty;
                const int img = patch / per_img, pr = patch - img * per_img;
                const int oy = (pr / g.cg.halo_tx) * 16 + (rt >> 4), ox = (pr % g.cg.halo_tx) * 16 + (rt & 15);
                m = (oy < g.cg.OH && ox < g.cg.OW) ? (img * g.cg.OH + oy) * g.cg.OW + ox : g.M;
            }
            pf[i] = 0u;
            if (rt < BM && m < g.M && n0 + seg * 64 + 2 <= g.N)
                pf[i] = *reinterpret_cast<const unsigned*>(g.epi.residual + (int64_t)zb * g.epi.strideR + (int64_t)m * g.epi.ldr + n0 + seg * 64);
        }
        asm volatile("" ::: "memory");  // keep the loads ahead of the staging writes
    }
#pragma unroll
    for (int gp = 0; gp < WAVES_M / WG; ++gp) {
        if (gp > 0) lds_barrier();  // staging reads of the previous pass are done (global stores may still fly)
        if (wm / WG == gp) {
#pragma unroll
            for (int p = 0; p < TM; ++p)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // The kernels multiply with the operands swapped - acc = mfma(b_frag, a_frag, acc), same products and k order, i.e. the same
                    // bits - so the 32x32 tile sits TRANSPOSED in the registers: lane l owns row l & 31 and, per register quad q, the four
                    // consecutive columns 8q + 4(l >> 5) + (0..3).  A quad is one 16-byte staging write (32 ds_write_b128 per wave and tile
                    // instead of 128 ds_write_b32); eight consecutive rows of the padded staging image hit all 32 banks once ((BN + 4) % 32 == 4).
#pragma unroll
                    for (int rr = 0; rr < FL::NR; ++rr) {
                        const int row = (wm % WG) * WTM + p * 32 + FL::row(lane, rr);
#pragma unroll
                        for (int q = 0; q < FL::NC; ++q) {
                            const int col = wn * WTN + j * 32 + FL::col(lane, q);
                            const int r0 = 4 * (rr * FL::NC + q);
                            *reinterpret_cast<float4*>(&stg[row * LDS_LD + col]) =
                                make_float4(acc_get(acc, p, j, r0), acc_get(acc, p, j, r0 + 1), acc_get(acc, p, j, r0 + 2), acc_get(acc, p, j, r0 + 3));
                        }
                    }
                }
        }
        lds_barrier();
        if (gp == 0 && prefetch) {
#pragma unroll
            for (int i = 0; i < NPF; ++i) asm volatile("" ::"v"(pf[i]));  // the prefetched lines have landed (in the L2, too); registers free again
        }
        // ---- read phase.  Common case (lean epilogue, whole column chunks, NT % CH == 0): a thread keeps ONE 8-column chunk over the pass
        // and walks rows row0, row0 + NT/CH, ... - the trip count is a compile-time constant, so the rows are processed in groups of U with
        // every load of the group (staged accumulators, residual, per-image vector) issued before the first dependent instruction; bias_n is
        // loaded once per kernel.  The runtime-flag-per-element form below serialised a global-load round trip per row (8-16 per tile).
        // Same operations in the same order per element as epi_fast8: results are bit-identical.
        if constexpr ((NT % CH == 0) && ((ROWS * CH) % NT == 0)) {
            if (g.epi.fast && !split && n0 + BN <= g.N && !ODISE_ABLATE(g, 8 | 32)) {   // tools: ODISE_EPI_OLD=1 (bit 32) keeps the previous form for A/B runs
                constexpr int IT = (ROWS * CH) / NT, RSTEP = NT / CH;
                // rows in pairs where the registers allow it: kernels that also carry the GroupNorm statistics (16 accumulators) or the plain
                // kernel's wider address state would spill to scratch (measured as +144 MB of HBM writes per launch on the dominant conv)
                constexpr int U = (IT % 2 == 0 && PAIRS && !STATS && !HALO) ? 2 : 1;
                const GemmEpi& e = g.epi;
                const int c8 = tid % CH, row0 = tid / CH;
                const int n = n0 + c8 * 8;
                // bias_n plus the per-image vector (time-embedding term) of the pass's first row - almost always the group of every row of the
                // pass - kept across the rows where the registers allow it (HOIST); otherwise every row fetches its own (L1-resident) copy
                constexpr bool HOIST = (U == 2);
                float bg8[8];
                unsigned grp0 = 0xffffffffu;
                auto load_bias_group = [&](float (&b)[8], unsigned grp, bool with_group) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) b[i] = 0.f;
                    if (e.bias_n) {
                        const float4 b0 = *reinterpret_cast<const float4*>(e.bias_n + n);
                        const float4 b1 = *reinterpret_cast<const float4*>(e.bias_n + n + 4);
                        b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
                    }
                    if (with_group) {
                        const float* rg = e.rowgroup_add + (int64_t)grp * e.ldg + n;
                        const float4 r0 = *reinterpret_cast<const float4*>(rg);
                        const float4 r1 = *reinterpret_cast<const float4*>(rg + 4);
                        b[0] += r0.x; b[1] += r0.y; b[2] += r0.z; b[3] += r0.w; b[4] += r1.x; b[5] += r1.y; b[6] += r1.z; b[7] += r1.w;
                    }
                };
                if (HOIST) {
                    const int mf = m0 + gp * ROWS + row0;
                    const bool have = e.rowgroup_add && mf < g.M;
                    if (have) grp0 = (unsigned)mf / (unsigned)e.rows_per_group;
                    load_bias_group(bg8, grp0, have);
                }
#pragma unroll
                for (int it0 = 0; it0 < IT; it0 += U) {
                    int mm[U];
                    float4 t0[U], t1[U];
                    f16x8 rr[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int row = row0 + (it0 + u) * RSTEP;
                        int m = m0 + gp * ROWS + row;
                        if (HALO) {
                            const int rt = gp * ROWS + row;
                            const int patch = m0 / BM;
                            const int per_img = g.cg.halo_tx * g.cg.halo_ty;
                            const int img = patch / per_img, pr = patch - img * per_img;
                            const int oy = (pr / g.cg.halo_tx) * 16 + (rt >> 4), ox = (pr % g.cg.halo_tx) * 16 + (rt & 15);
                            m = (oy < g.cg.OH && ox < g.cg.OW) ? (img * g.cg.OH + oy) * g.cg.OW + ox : g.M;
                        }
                        mm[u] = m;
                        t0[u] = *reinterpret_cast<const float4*>(&stg[row * LDS_LD + c8 * 8]);
                        t1[u] = *reinterpret_cast<const float4*>(&stg[row * LDS_LD + c8 * 8 + 4]);
                        rr[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                        if (m < g.M && e.residual) rr[u] = *reinterpret_cast<const f16x8*>(e.residual + (int64_t)zb * e.strideR + (int64_t)m * e.ldr + n);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int m = mm[u];
                        if (m >= g.M) continue;
                        float v[8] = {t0[u].x, t0[u].y, t0[u].z, t0[u].w, t1[u].x, t1[u].y, t1[u].z, t1[u].w};
                        float b[8];
                        if (HOIST) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) b[i] = bg8[i];
                            if (e.rowgroup_add) {
                                const unsigned grp = (unsigned)m / (unsigned)e.rows_per_group;
                                if (grp != grp0) load_bias_group(b, grp, true);   // a pass that straddles two images: this row's own vector, same order of additions
                            }
                        } else {
                            load_bias_group(b, e.rowgroup_add ? (unsigned)m / (unsigned)e.rows_per_group : 0u, e.rowgroup_add != nullptr);
                        }
                        float alpha = e.alpha;
                        if (e.scale_m) alpha *= e.scale_m[m];
                        if (e.bias_m) {
                            const float bm = e.bias_m[m];
#pragma unroll
                            for (int i = 0; i < 8; ++i) b[i] += bm;
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = v[i] * alpha + b[i];
                        if (e.geglu) {
                            float o[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) o[i] = v[2 * i] * gelu_erf(v[2 * i + 1]);
                            const int64_t off = (int64_t)zb * e.strideC + (int64_t)m * e.ldc + (n >> 1);
                            if (e.c_dtype == ODISE_F16) {
                                f16x4 t;
#pragma unroll
                                for (int i = 0; i < 4; ++i) t[i] = (f16)o[i];
                                *reinterpret_cast<f16x4*>((f16*)e.C + off) = t;
                            } else {
                                *reinterpret_cast<float4*>((float*)e.C + off) = make_float4(o[0], o[1], o[2], o[3]);
                            }
                            continue;
                        }
                        if (e.act == ODISE_ACT_SILU) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = mul_sigmoid(v[i], v[i]);
                        } else if (e.act == ODISE_ACT_RELU) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
                        } else if (e.act == ODISE_ACT_QUICKGELU) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = mul_sigmoid(v[i], 1.702f * v[i]);
                        } else if (e.act == ODISE_ACT_GELU) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = gelu_erf(v[i]);
                        }
                        const int64_t off = (int64_t)zb * e.strideC + (int64_t)m * e.ldc + n;
                        if (e.c_dtype == ODISE_F16) {
                            f16x8 t;
#pragma unroll
                            for (int i = 0; i < 8; ++i) t[i] = (f16)(v[i] + (float)rr[u][i]);
                            *reinterpret_cast<f16x8*>((f16*)e.C + off) = t;
                            if (stats) {
#pragma unroll
                                for (int i = 0; i < 8; ++i) { const float r = (float)t[i]; s8[i] += r; q8[i] += r * r; }
                            }
                        } else {
                            float* c = (float*)e.C + off;
                            *reinterpret_cast<float4*>(c) = make_float4(v[0] + (float)rr[u][0], v[1] + (float)rr[u][1], v[2] + (float)rr[u][2], v[3] + (float)rr[u][3]);
                            *reinterpret_cast<float4*>(c + 4) = make_float4(v[4] + (float)rr[u][4], v[5] + (float)rr[u][5], v[6] + (float)rr[u][6], v[7] + (float)rr[u][7]);
                        }
                    }
                }
                continue;   // next pass
            }
        }
        for (int c = tid; c < ROWS * CH; c += NT) {
            const int row = c / CH;
            const int c8 = c - row * CH;
            int m = m0 + gp * ROWS + row;
            if (HALO) {
                // block tile row -> pixel of the block's 16x16 output patch (m0 / BM = patch index: image, patch row, patch column)
                const int rt = gp * ROWS + row;
                const int patch = m0 / BM;
                const int per_img = g.cg.halo_tx * g.cg.halo_ty;
                const int img = patch / per_img, pr = patch - img * per_img;
                const int oy = (pr / g.cg.halo_tx) * 16 + (rt >> 4), ox = (pr % g.cg.halo_tx) * 16 + (rt & 15);
                m = (oy < g.cg.OH && ox < g.cg.OW) ? (img * g.cg.OH + oy) * g.cg.OW + ox : g.M;
            }
            const int n = n0 + c8 * 8;
            if (m < g.M && n < g.N) {
                float v[8];
                const float4 t0 = *reinterpret_cast<const float4*>(&stg[row * LDS_LD + c8 * 8]);
                const float4 t1 = *reinterpret_cast<const float4*>(&stg[row * LDS_LD + c8 * 8 + 4]);
                v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w;
                v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
                if (split) {
                    float* w = g.ws + ((int64_t)z * g.M + m) * g.N + n;
                    const int nv = (g.N - n) < 8 ? (g.N - n) : 8;
                    if (nv == 8 && (g.N & 3) == 0) {
                        *reinterpret_cast<float4*>(w) = t0;
                        *reinterpret_cast<float4*>(w + 4) = t1;
                    } else {
                        for (int i = 0; i < nv; ++i) w[i] = v[i];
                    }
                } else if (!(ODISE_ABLATE(g, 8)) || v[0] == 12345.678f) {  // dbg 8: ablate the global store + epilogue math
                    if (g.epi.fast && n + 8 <= g.N) {
                        if (stats) {
                            float r8[8];
                            epi_fast8(g.epi, v, m, n, zb, r8);
#pragma unroll
                            for (int i = 0; i < 8; ++i) { s8[i] += r8[i]; q8[i] += r8[i] * r8[i]; }
                        } else {
                            epi_fast8(g.epi, v, m, n, zb);
                        }
                    } else {
                        epi_store8(g.epi, v, m, n, g.N, zb);
                    }
                }
            }
        }
    }
    if (stats) {
        constexpr int RL = NT / CH;  // row lanes per column chunk
        lds_barrier();                // the last pass is done with the staging buffer
        float* red = reinterpret_cast<float*>(smem);  // [RL][BN][2]
        const int c8 = tid % CH, rl = tid / CH;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            red[((rl * BN) + c8 * 8 + i) * 2 + 0] = s8[i];
            red[((rl * BN) + c8 * 8 + i) * 2 + 1] = q8[i];
        }
        lds_barrier();
        for (int c = tid; c < BN; c += NT) {
            float a = 0.f, b = 0.f;
            for (int r = 0; r < RL; ++r) { a += red[(r * BN + c) * 2]; b += red[(r * BN + c) * 2 + 1]; }
            if (n0 + c < g.N) {
                float* o = g.epi.gn_stats + ((int64_t)(m0 / BM) * g.N + n0 + c) * 2;
                o[0] = a;
                o[1] = b;
            }
        }
    }
}

// ---- MFMA shape of the main loops --------------------------------------------------------------------------------------------------
// Every kernel below walks a 64-deep K-tile of a 32x32 output block in Frag::NSTEP steps; a step consumes Frag::PER fragments (16 bytes
// per lane each) of the block's A rows and of its B rows:
//   32x32x16: 4 steps of one MFMA; fragment = rows l & 31, 16-byte k-slot 2 st + (l >> 5)
//   16x16x32: 2 steps of 2 x 2 MFMAs; fragment u = rows 16 u + (l & 15), k-slot 4 st + (l >> 4)
// Same fragment bytes, LDS reads and FLOPs per block either way.  kL16 selects the shape for all of them (round 5: the 16x16x32
// instruction sustains 1.95 PFLOP/s on operands that toggle like real data, 32x32x16 is power-bound at 1.62 - FragLayout above); the
// k order inside a K-tile (steps ascending) is the same in every kernel, so kernels still agree bit for bit on a given K walk.
#ifdef ODISE_MFMA32   // A/B build only (python -m odise_amd.build --m32): the MFMA shape of rounds 1-4
constexpr bool kL16 = false;
#else
constexpr bool kL16 = true;
#endif
template <bool L16>
struct Frag {
    static constexpr int NSTEP = L16 ? 2 : 4;
    static constexpr int PER = L16 ? 2 : 1;
    static __device__ __forceinline__ int lrow(int lane) { return L16 ? (lane & 15) : (lane & 31); }
    static __device__ __forceinline__ int lk(int lane) { return L16 ? (lane >> 4) : (lane >> 5); }
    static __device__ __forceinline__ int kslot(int st, int lk) { return L16 ? st * 4 + lk : st * 2 + lk; }
};
template <bool L16> struct AccBlockT { typedef f32x16 type; };
template <> struct AccBlockT<true> { typedef f32x4 type[4]; };
template <bool L16> using AccBlock = typename AccBlockT<L16>::type;
__device__ __forceinline__ void acc_zero(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
__device__ __forceinline__ void acc_zero(f32x4 (&a)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// one step of a 32x32 block; operands swapped (transposed tile in the registers, see gemm_epilogue)
__device__ __forceinline__ void mma_step(f32x16& acc, const f16x8 (&b)[1], const f16x8 (&a)[1]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[0], a[0], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(f32x4 (&acc)[4], const f16x8 (&b)[2], const f16x8 (&a)[2]) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt * 2 + ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ct], a[rt], acc[rt * 2 + ct], 0, 0, 0);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool CONV, bool INTERLEAVE>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) gemm_kernel(GemmArgs g) {
    constexpr int BK = 64;
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;  // wave sub-tile
    constexpr int TM = WTM / 32, TN = WTN / 32;            // 32x32 MFMA tiles per wave
    constexpr int RPI = NT / 8;                            // operand rows staged per load instruction sweep
    constexpr int JA = BM / RPI, JB = BN / RPI;            // 16-byte LDS-DMA loads per thread per K-tile
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BM % RPI == 0 && BN % RPI == 0 && RPI % 16 == 0, "bad tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int hi = lane >> 5, l31 = lane & 31;
    // XCD-aware tile order: the dispatcher round-robins workgroups over the 8 XCDs (private 4 MiB L2 each); remap so that
    // each XCD walks a contiguous run of tiles (n fastest): its co-resident blocks then share A row-panels and W
    // column-panels in L2 instead of every XCD streaming every panel from HBM.  Bijective for any grid size.
    int bx, by;
    {
        const int nbx = gridDim.x, nb = gridDim.x * gridDim.y;
        const int bid = blockIdx.y * nbx + blockIdx.x;
        const int q = nb >> 3, r = nb & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        by = logical / nbx;
        bx = logical - by * nbx;
    }
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int z = blockIdx.z;
    const bool split = g.splitk > 1;
    const int zb = split ? 0 : z;  // batch index

    const int nk_total = (g.K + BK - 1) / BK;
    int kt_begin = 0, kt_end = nk_total;
    if (split) {
        kt_begin = z * g.ktiles_per_split;
        kt_end = kt_begin + g.ktiles_per_split;
        if (kt_end > nk_total) kt_end = nk_total;
    }

    const f16* Ab = g.A + (int64_t)zb * g.strideA;
    const f16* Wb = g.W + (int64_t)zb * g.strideW;

    const int slot = tid & 7;
    const int rbase = tid >> 3;  // 0..RPI-1

    // per-thread A row descriptors
    int64_t a_off[JA];
    int a_iy0[JA], a_ix0[JA];
    bool a_ok[JA];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int m = m0 + rbase + RPI * j;
        a_ok[j] = m < g.M;
        if (CONV) {
            const int ohw = g.cg.OH * g.cg.OW;
            const int mm = a_ok[j] ? m : 0;
            const int img = mm / ohw;
            const int rem = mm - img * ohw;
            const int oy = rem / g.cg.OW;
            const int ox = rem - oy * g.cg.OW;
            a_iy0[j] = oy * g.cg.stride - g.cg.pad_t;
            a_ix0[j] = ox * g.cg.stride - g.cg.pad_l;
            a_off[j] = (int64_t)img * g.cg.H * g.cg.W * g.cg.Cin;
        } else {
            a_iy0[j] = a_ix0[j] = 0;
            a_off[j] = (int64_t)m * g.lda;
        }
    }
    int64_t b_off[JB];
    bool b_ok[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int n = n0 + rbase + RPI * j;
        b_ok[j] = n < g.N;
        b_off[j] = (int64_t)n * g.ldw;
    }

    // ---- global -> LDS direct (LDS-DMA, 16 B per lane).  The LDS image of one instruction is lane-linear (wave base +
    // lane*16 = 8 rows x 128 B), so the bank swizzle is applied to the SOURCE: the lane that fills physical slot p of row r
    // fetches logical k-slot p ^ ((r>>1)&7); (r>>1)&7 does not depend on j because rows advance by RPI (a multiple of 16).
    // Rows / k beyond the problem and conv padding read from a zero line instead of being predicated (every lane must
    // write its slot).
    const int ls = slot ^ ((rbase >> 1) & 7);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // per-K-tile decode shared by this lane's loads
    int tk_k = 0, tk_ky = 0, tk_kx = 0, tk_c = 0;
    bool tk_ok = false;
    auto prep_tile = [&](int kt) {
        tk_k = kt * BK + ls * 8;
        tk_ok = tk_k < g.K;
        if (CONV && g.cg.chunk_major) {
            // Cin % 64 == 0: visit the KH*KW taps of one 64-channel chunk back to back.  Consecutive K-tiles then read the
            // same input pixels shifted by one column (or one row every KW tiles), so the shifted re-reads hit the XCD's L2
            // instead of coming back from the Infinity Cache after Cin/64 K-tiles of other traffic (4 MiB L2 holds ~4 K-tile
            // steps of the 32 co-resident blocks).  Only the fp32 summation order changes.
            const int taps = g.cg.KH * g.cg.KW;
            const int chunk = kt / taps;
            const int tap = kt - chunk * taps;
            tk_c = chunk * BK + ls * 8;
            tk_ky = tap / g.cg.KW;
            tk_kx = tap - tk_ky * g.cg.KW;
            tk_k = tap * g.cg.Cin + tk_c;
            tk_ok = true;
        } else if (CONV) {
            tk_ky = tk_kx = tk_c = 0;
            if (tk_ok) {
                const int tap = tk_k / g.cg.Cin;
                tk_c = tk_k - tap * g.cg.Cin;
                tk_ky = tap / g.cg.KW;
                tk_kx = tap - tk_ky * g.cg.KW;
            }
        }
    };
    // load #l of a tile: l < JA -> A rows sweep l, else W rows sweep l-JA
    auto issue_load = [&](int l, int stage) {
        char* sa = smem + stage * STAGE_BYTES;
        char* sb = sa + BM * BK * 2;
        if (l < JA) {
            const int j = l;
            const f16* src;
            if (CONV) {
                int iy = a_iy0[j] + tk_ky, ix = a_ix0[j] + tk_kx;
                bool ok = a_ok[j] && tk_ok;
                if (g.cg.ups) {
                    ok = ok && iy >= 0 && ix >= 0 && iy < 2 * g.cg.H && ix < 2 * g.cg.W;
                    iy >>= 1;
                    ix >>= 1;
                } else {
                    ok = ok && iy >= 0 && ix >= 0 && iy < g.cg.H && ix < g.cg.W;
                }
                src = ok ? Ab + a_off[j] + ((int64_t)iy * g.cg.W + ix) * g.cg.Cin + tk_c : g.zeros;
            } else {
                src = (a_ok[j] && tk_ok) ? Ab + a_off[j] + tk_k : g.zeros;
            }
            glds16(src, sa + (j * NT + wave_u * 64) * 16);
        } else {
            const int j = l - JA;
            const f16* src = (b_ok[j] && tk_ok) ? Wb + b_off[j] + tk_k : g.zeros;
            glds16(src, sb + (j * NT + wave_u * 64) * 16);
        }
    };
    constexpr int NL = JA + JB;
    auto issue_tile = [&](int kt, int stage) {
        prep_tile(kt);
#pragma unroll
        for (int l = 0; l < NL; ++l) issue_load(l, stage);
    };

    AccBlock<kL16> acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc_zero(acc[i][j]);

    // The swizzle key (r>>1)&7 of a fragment row r = wave_base + 32*tile + (lane&31) only depends on the lane (bases are
    // multiples of 32), so the four k-step slot offsets are shared by every A and B fragment of this lane.
    using FR = Frag<kL16>;
    const int lrow = FR::lrow(lane), lkq = FR::lk(lane);
    int koff[FR::NSTEP];   // 16-byte slot of step st, swizzled with the lane's row key (tile bases are multiples of 16 rows); fragment u: + u * 16 rows
#pragma unroll
    for (int s = 0; s < FR::NSTEP; ++s) koff[s] = (FR::kslot(s, lkq) ^ ((lrow >> 1) & 7)) << 4;
    const int a_lane_off = (wm * WTM + lrow) * 128;
    const int b_lane_off = (wn * WTN + lrow) * 128;

    if (kt_begin < kt_end) issue_tile(kt_begin, 0);

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const bool more = (kt + 1) < kt_end;
        // tile kt has landed (this wave's DMA) and, after the barrier, everybody's; all waves are also done reading
        // the other buffer (they finished compute(kt-1) before arriving here), so it can be refilled.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool dma = more && !(ODISE_ABLATE(g, 1));
        if (dma) {
            if (INTERLEAVE) prep_tile(kt + 1);
            else issue_tile(kt + 1, cur ^ 1);
        }
        // fragment reads: per-lane base + per-k-step swizzled slot offset (loop invariant) + compile-time tile offset
        const char* fa = smem + cur * STAGE_BYTES + a_lane_off;
        const char* fb = smem + cur * STAGE_BYTES + BM * BK * 2 + b_lane_off;
        f16x8 af[TM][FR::PER], bf[TN][FR::PER];
#pragma unroll
        for (int s = 0; s < FR::NSTEP; ++s) {
            if (!(ODISE_ABLATE(g, 2)) || (kt == kt_begin && s == 0)) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int u = 0; u < FR::PER; ++u) af[i][u] = *reinterpret_cast<const f16x8*>(fa + koff[s] + i * 4096 + u * 2048);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int u = 0; u < FR::PER; ++u) bf[j][u] = *reinterpret_cast<const f16x8*>(fb + koff[s] + j * 4096 + u * 2048);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma_step(acc[i][j], bf[j], af[i]);  // operands swapped: transposed tile (see gemm_epilogue)
            if (INTERLEAVE) {
                // next tile's LDS-DMA loads are issued in the shadow of this k-step's MFMAs (the matrix pipe keeps
                // draining the queued MFMAs while the wave issues address math + global_load_lds)
                if (dma) {
#pragma unroll
                    for (int l = 0; l < NL; ++l)
                        if ((l * FR::NSTEP) / NL == s) issue_load(l, cur ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();  // every wave is done with the operand tiles before the staging buffer is reused
    if (ODISE_ABLATE(g, 4)) return;  // ablation: main loop only

    gemm_epilogue<BM, BN, WAVES_M, WAVES_N, epi_wave_rows(BM, BN, WAVES_M, plain_lds_bytes(BM, BN, WAVES_M)), false, false, (BM * BN < 256 * 256),
                  epi16_rows_if_enabled(BM, BN, WAVES_M, plain_lds_bytes(BM, BN, WAVES_M)), !CONV, plain_lds_bytes(BM, BN, WAVES_M)>(g, acc, smem, m0, n0, z, zb, split);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- Ping-pong pipelined variant of the 256 x BN tile (BN = 256 / 320, 8 waves as 4(M) x 2(N), K % 64 == 0) ---------------------
// The plain kernel above drains every LDS-DMA at each K-tile boundary (vmcnt(0) + barrier) and all 8 waves read fragments,
// multiply and wait in lockstep: the matrix pipe idles while operands are fetched and vice versa (measured: 22 % of the time is
// DMA wait, 8 % fragment reads).  Here
//   * a K-tile is consumed in NP phases of two 32-column N-tiles of each wave (A fragments of the whole K-tile are read in phase
//     0 and kept in VGPRs, B fragments per phase), so the A rows and the B rows of a stage are released progressively and refilled
//     with K-tile t+2 while K-tile t+1 is multiplied: 1 to 1.5 K-tiles of LDS-DMA stay in flight ACROSS the barriers, retired by
//     counted `s_waitcnt vmcnt(N)` (never 0 in steady state);
//   * the waves form two groups (wm 0-1 / wm 2-3 = the two waves of every SIMD) staggered by one barrier: while one group issues its
//     16 MFMAs (s_setprio 1), the other reads its next fragments and issues its share of the DMA, then they swap.
// Every phase is  {ds_read fragments, issue DMA group, vmcnt(allowed), lgkmcnt(0)} barrier {MFMA} barrier.
// Hazards: (RAW) the vmcnt before barrier A of phase q covers what phase q+1 reads, so both groups have waited for their share and
// passed a barrier before anyone reads it; (WAR) fragment reads are retired (lgkmcnt(0)) before barrier A, refills of those rows
// are issued one phase later at the earliest, i.e. after a barrier both groups passed.
// DMA groups of the tile sequence (per thread: 4 A loads, TN B loads; B piece j = the rows of N-tile j of both wave columns):
//   NP = 2: (t,0) issues B0..B3 of tile t+1, (t,1) issues A0..A3 of tile t+2
//   NP = 3: (t,0) issues B2,B3,B4 of tile t+1, (t,1) issues A0,A1,A2 of tile t+2, (t,2) issues A3,B0,B1 of tile t+2
template <int BM, int BN, int WAVES_N, int PT, bool CONV>
__global__ void __launch_bounds__(512) gemm_pp_kernel(GemmArgs g) {
    // (measured and rejected: issuing the DMA before the fragment reads -17 %, dropping s_setprio +-1 %; a persistent form - 256 resident
    // blocks walking the tile list - is bit-identical but 5-9 % SLOWER: vmcnt also counts stores on this part, so the next tile's first
    // counted wait drains the previous tile's global stores, which a retiring block leaves to the memory system while the dispatcher
    // already starts its successor)
    constexpr int BK = 64, WAVES_M = 8 / WAVES_N;
    constexpr int WTN = BN / WAVES_N;
    constexpr int TM = 2, TN = WTN / 32;
    constexpr int NP = (TN + PT - 1) / PT;   // phases per K-tile (PT N-tiles of the wave each)
    constexpr int JA = BM / 64, JB = BN / 64;  // 64-row DMA pieces (one 16-byte load per thread each)
    constexpr int TPB = TN / JB;             // N-tiles of a wave covered by one B piece (the piece spans every wave column)
    constexpr int NL = JA + JB;              // LDS-DMA loads per thread per K-tile
    constexpr int LPP = (NL + NP - 1) / NP;  // loads per phase
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    static_assert(PT == 1 || PT == 2, "one or two N-tiles per phase");
    static_assert(BM / WAVES_M == 64 && WTN % 32 == 0 && TN % JB == 0, "bad tile");
    // WAR rule of the schedule: B piece j (read in phase j*TPB/PT) is refilled in slot (JA+j)/LPP, i.e. one phase later at the earliest
    static_assert((JA + JB - 1) / LPP >= ((JB - 1) * TPB) / PT && JA / LPP >= 0, "refill would overtake the fragment reads");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int grp = wave >> 2;  // waves w and w+4 share a SIMD
    const int hi = lane >> 5, l31 = lane & 31;
    int bx, by;
    {
        const int nbx = gridDim.x, nb = gridDim.x * gridDim.y;
        const int bid = blockIdx.y * nbx + blockIdx.x;
        const int q = nb >> 3, r = nb & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        by = logical / nbx;
        bx = logical - by * nbx;
    }
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int z = blockIdx.z;
    const bool split = g.splitk > 1;
    const int zb = split ? 0 : z;
    const int nk_total = g.K / BK;
    int kt_begin = 0, kt_end = nk_total;
    if (split) {
        kt_begin = z * g.ktiles_per_split;
        kt_end = kt_begin + g.ktiles_per_split;
        if (kt_end > nk_total) kt_end = nk_total;
    }
    const f16* Ab = g.A + (int64_t)zb * g.strideA;
    const f16* Wb = g.W + (int64_t)zb * g.strideW;

    // ---- per-lane DMA descriptors: lane fills physical 16-byte slot (lane & 7) of row (8*wave + lane/8) of every 64-row piece
    // and fetches logical slot ls (XOR swizzle on the source, see gemm_kernel)
    const int rbase = wave * 8 + (lane >> 3);
    const int ls = (lane & 7) ^ ((rbase >> 1) & 7);
    int64_t a_off[JA];
    int a_iy0[JA], a_ix0[JA];  // conv: top-left input coordinate of the row's window; rows >= M get iy0 far out of range
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int m = m0 + rbase + 64 * j;
        const bool ok = m < g.M;
        if (CONV) {
            const int ohw = g.cg.OH * g.cg.OW;
            const int mm = ok ? m : 0;
            const int img = mm / ohw;
            const int rem = mm - img * ohw;
            const int oy = rem / g.cg.OW;
            const int ox = rem - oy * g.cg.OW;
            a_iy0[j] = ok ? oy * g.cg.stride - g.cg.pad_t : -(1 << 28);
            a_ix0[j] = ox * g.cg.stride - g.cg.pad_l;
            a_off[j] = ((int64_t)img * g.cg.H * g.cg.W + (int64_t)a_iy0[j] * g.cg.W + a_ix0[j]) * g.cg.Cin + ls * 8;
        } else {
            a_iy0[j] = ok ? 0 : -1;
            a_ix0[j] = 0;
            a_off[j] = (int64_t)m * g.lda + ls * 8;
        }
    }
    // B piece j = rows [j*RB, (j+1)*RB) of EVERY wave column (RB = 64 / WAVES_N), i.e. exactly the rows the waves read for N-tiles
    // j*TPB .. of theirs; this wave fills 8 of them
    constexpr int RB = 64 / WAVES_N;
    const int b_row0 = (WAVES_N == 2) ? (wave >> 2) * WTN + (wave & 3) * 8 : wave * 8;
    const int nb0 = n0 + b_row0 + (lane >> 3);
    const int64_t b_off0 = (int64_t)nb0 * g.ldw + ls * 8;
    const int b_lds0 = b_row0 * 128;  // LDS byte offset of this wave's 8 rows inside piece 0

    // K-tile position (wave-uniform, advanced incrementally: no divisions in the loop).  Conv taps are whole 64-channel chunks
    // (Cin % 64 == 0), walked chunk-major or tap-major (see gemm_kernel::prep_tile).
    struct TileK {
        int ky, kx, c0;
        int64_t a_delta;  // element offset added to a_off
        int kw;           // element offset inside a weight row
    };
    auto finish = [&](TileK& t) {
        if (CONV) {
            t.a_delta = ((int64_t)t.ky * g.cg.W + t.kx) * g.cg.Cin + t.c0;
            t.kw = (t.ky * g.cg.KW + t.kx) * g.cg.Cin + t.c0;
        } else {
            t.a_delta = t.c0;
            t.kw = t.c0;
        }
    };
    auto decode = [&](int kt) {
        TileK t;
        t.ky = t.kx = 0;
        t.c0 = kt * BK;
        if (CONV) {
            int tap;
            if (g.cg.chunk_major) {
                const int taps = g.cg.KH * g.cg.KW;
                const int chunk = kt / taps;
                tap = kt - chunk * taps;
                t.c0 = chunk * BK;
            } else {
                tap = (kt * BK) / g.cg.Cin;
                t.c0 = kt * BK - tap * g.cg.Cin;
            }
            t.ky = tap / g.cg.KW;
            t.kx = tap - t.ky * g.cg.KW;
        }
        finish(t);
        return t;
    };
    auto advance = [&](TileK& t) {
        if (ODISE_ABLATE(g, 16)) return;  // timing experiment (ODISE_GEMM_FREEZE_K): every K-tile re-reads the first one - hot lines, wrong results
        if (CONV) {
            if (g.cg.chunk_major) {
                if (++t.kx == g.cg.KW) {
                    t.kx = 0;
                    if (++t.ky == g.cg.KH) { t.ky = 0; t.c0 += BK; }
                }
            } else {
                t.c0 += BK;
                if (t.c0 == g.cg.Cin) {
                    t.c0 = 0;
                    if (++t.kx == g.cg.KW) { t.kx = 0; ++t.ky; }
                }
            }
        } else {
            t.c0 += BK;
        }
        finish(t);
    };
    auto issue_A = [&](int j, int stage, const TileK& t) {
        bool ok;
        if (CONV) {
            const int iy = a_iy0[j] + t.ky, ix = a_ix0[j] + t.kx;
            ok = (unsigned)iy < (unsigned)g.cg.H && (unsigned)ix < (unsigned)g.cg.W;
        } else {
            ok = a_iy0[j] >= 0;
        }
        const f16* src = ok ? Ab + a_off[j] + t.a_delta : g.zeros;
        glds16(src, smem + stage * STAGE_BYTES + (j * 64 + wave * 8) * 128);
    };
    auto issue_B = [&](int j, int stage, const TileK& t) {
        const bool ok = (nb0 + j * RB) < g.N;
        const f16* src = ok ? Wb + b_off0 + (int64_t)(j * RB) * g.ldw + t.kw : g.zeros;
        glds16(src, smem + stage * STAGE_BYTES + A_BYTES + b_lds0 + j * RB * 128);
    };

    AccBlock<kL16> acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc_zero(acc[i][j]);

    using FR = Frag<kL16>;
    const int lrow = FR::lrow(lane), lkq = FR::lk(lane);
    int koff[FR::NSTEP];   // 16-byte slot of step st, swizzled with the lane's row key (tile bases are multiples of 16 rows); fragment u: + u * 16 rows
#pragma unroll
    for (int s = 0; s < FR::NSTEP; ++s) koff[s] = (FR::kslot(s, lkq) ^ ((lrow >> 1) & 7)) << 4;
    const int a_lane_off = (wm * 64 + lrow) * 128;
    const int b_lane_off = A_BYTES + (wn * WTN + lrow) * 128;

    // load l of a K-tile: l < JA -> A piece l, else B piece l-JA; issued in slot l / LPP of the tile (see the schedule above)
    auto issue = [&](int l, int stage, const TileK& t) {
        if (l < JA) issue_A(l, stage, t);
        else issue_B(l - JA, stage, t);
    };
    // number of loads (from A0 of a tile) that must have landed before its N-tiles [0, nt) can be read
    auto loads_for_tiles = [](int nt) { return JA + (nt + TPB - 1) / TPB; };
    // ---- prologue: all of tile 0, and the slots of tile 1 that the steady state would have issued before phase (0,0)
    TileK t1 = decode(kt_begin), t2;  // positions of K-tiles kt+1 and kt+2 while tile kt is multiplied
    if (kt_begin < kt_end) {
        const TileK t0 = t1;
        advance(t1);
#pragma unroll
        for (int l = 0; l < NL; ++l) issue(l, 0, t0);
        constexpr int pro1 = (NP - 1) * LPP < NL ? (NP - 1) * LPP : NL;
        if (kt_begin + 1 < kt_end) {
#pragma unroll
            for (int l = 0; l < pro1; ++l) issue(l, 1, t1);
            wait_vmcnt<NL + pro1 - (JA + (PT + TPB - 1) / TPB)>();  // phase (0,0) needs A and the first PT N-tiles of B of tile 0
        } else {
            wait_vmcnt<NL - (JA + (PT + TPB - 1) / TPB)>();
        }
    }
    t2 = t1;
    advance(t2);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger: group 1 runs one barrier interval behind group 0

    f16x8 af[TM][FR::NSTEP][FR::PER], bf[PT][FR::NSTEP][FR::PER];
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const bool has1 = (kt + 1) < kt_end && !(ODISE_ABLATE(g, 1)), has2 = (kt + 2) < kt_end && !(ODISE_ABLATE(g, 1));  // dbg 1: ablate the DMA
        const char* fa = smem + cur * STAGE_BYTES + a_lane_off;
        const char* fb = smem + cur * STAGE_BYTES + b_lane_off;
        const bool rd = !(ODISE_ABLATE(g, 2)) || kt == kt_begin;  // dbg 2: ablate the fragment reads
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int j0 = PT * p;
            const int nj = (j0 + PT <= TN) ? PT : (TN - j0);
            // -------- load segment: fragments of this phase, one slot of DMA, counted wait for what the NEXT phase reads
            auto read_frags = [&]() {
                if (p == 0 && rd) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int s = 0; s < FR::NSTEP; ++s)
#pragma unroll
                            for (int u = 0; u < FR::PER; ++u) af[i][s][u] = *reinterpret_cast<const f16x8*>(fa + koff[s] + i * 4096 + u * 2048);
                }
#pragma unroll
                for (int jj = 0; jj < PT; ++jj)
                    if (jj < nj && rd) {
#pragma unroll
                        for (int s = 0; s < FR::NSTEP; ++s)
#pragma unroll
                            for (int u = 0; u < FR::PER; ++u) bf[jj][s][u] = *reinterpret_cast<const f16x8*>(fb + koff[s] + (j0 + jj) * 4096 + u * 2048);
                    }
            };
            read_frags();
            // loads through index `need` (counted from A0 of tile kt) must have landed before the next phase reads
            const int need = (p + 1 < NP) ? loads_for_tiles((p + 2) * PT < TN ? (p + 2) * PT : TN) : NL + loads_for_tiles(PT);
            if (p == 0) {
                if (has1) {
#pragma unroll
                    for (int l = 0; l < NL; ++l)
                        if (l / LPP == NP - 1) issue(l, cur ^ 1, t1);
                    wait_vmcnt<2 * NL - (JA + ((2 * PT < TN ? 2 * PT : TN) + TPB - 1) / TPB)>();
                } else {
                    wait_vmcnt<0>();
                }
            } else {
                const int issued2 = (p * LPP < NL) ? p * LPP : NL;  // loads of tile kt+2 issued once this phase's slot is out
                if (has2) {
#pragma unroll
                    for (int l = 0; l < NL; ++l)
                        if (l / LPP == p - 1) issue(l, cur, t2);
                    switch (2 * NL + issued2 - need) {  // compile-time after unrolling
                        case 4: wait_vmcnt<4>(); break;
                        case 5: wait_vmcnt<5>(); break;
                        case 6: wait_vmcnt<6>(); break;
                        case 7: wait_vmcnt<7>(); break;
                        case 8: wait_vmcnt<8>(); break;
                        case 9: wait_vmcnt<9>(); break;
                        case 10: wait_vmcnt<10>(); break;
                        case 11: wait_vmcnt<11>(); break;
                        case 12: wait_vmcnt<12>(); break;
                        case 13: wait_vmcnt<13>(); break;
                        case 14: wait_vmcnt<14>(); break;
                        case 15: wait_vmcnt<15>(); break;
                        default: wait_vmcnt<0>(); break;
                    }
                } else if (has1) {
                    switch (2 * NL - need) {
                        case 1: wait_vmcnt<1>(); break;
                        case 2: wait_vmcnt<2>(); break;
                        case 3: wait_vmcnt<3>(); break;
                        case 4: wait_vmcnt<4>(); break;
                        case 8: wait_vmcnt<8>(); break;
                        case 9: wait_vmcnt<9>(); break;
                        case 10: wait_vmcnt<10>(); break;
                        case 11: wait_vmcnt<11>(); break;
                        default: wait_vmcnt<0>(); break;
                    }
                } else {
                    wait_vmcnt<0>();
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // -------- MFMA segment
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < FR::NSTEP; ++s)
#pragma unroll
                for (int jj = 0; jj < PT; ++jj)
                    if (jj < nj) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) mma_step(acc[i][j0 + jj], bf[jj][s], af[i][s]);  // transposed tile (see gemm_epilogue)
                    }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        t1 = t2;
        advance(t2);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // re-align the two groups
    __syncthreads();
    if (ODISE_ABLATE(g, 4)) return;
    gemm_epilogue<BM, BN, WAVES_M, WAVES_N, epi_wave_rows(BM, BN, WAVES_M, pp_lds_bytes(BM, BN, WAVES_M)), false, (CONV && BM == 512), true,
                  epi16_rows_if_enabled(BM, BN, WAVES_M, pp_lds_bytes(BM, BN, WAVES_M)), !CONV, pp_lds_bytes(BM, BN, WAVES_M)>(g, acc, smem, m0, n0, z, zb, split);
}


// ---- Ping-pong kernel, second generation: fragment reads moved into the MFMA segment -----------------------------------------------
// PMC of gemm_pp_kernel (SQ_WAVE_CYCLES vs MFMA cycles): a barrier interval lasts ~950 cycles for 512 cycles of MFMA - the critical
// path is the OTHER group's load segment (16 ds_read_b128, their latency, then 4 LDS-DMA issues).  Here every k-step of the MFMA
// segment is followed by the ds_reads that refill exactly the fragment registers it just consumed with the NEXT phase's data, so the
// reads fly under the remaining MFMAs and the load segment shrinks to {DMA issue, counted vmcnt, lgkmcnt(0)}.
//   * RAW: the reads of phase q+1 are issued after barrier A(q); the vmcnt at the end of load segment q-1 therefore covers what phase
//     q+1 reads (one phase further ahead than gemm_pp_kernel), for both groups before a barrier the reader has passed.
//   * WAR: those reads are retired by the lgkmcnt(0) of load segment q+1 before barrier A(q+1); the refill of their rows is issued in
//     load segment q+2 at the earliest (slot = read phase + 1 of the tile two K-tiles ahead).
//   * DMA slots of tile T (phases of tile T-2, slot NP = phase 0 of tile T-1): B piece read in phase r -> slot r+1; A pieces spread
//     over slots 1..NP-1.  The allowed-outstanding counts are computed by pp2_allowed() from this table.
namespace pp2 {
constexpr int read_phase(int l, int JA, int TPB, int PT) { return l < JA ? 0 : ((l - JA) * TPB) / PT; }
constexpr int slot(int l, int JA, int NP, int TPB, int PT) {
    return l < JA ? (NP >= 3 ? 1 + (l * (NP - 1)) / JA : 1) : read_phase(l, JA, TPB, PT) + 1;
}
// loads issued up to and including phase p of the current tile (tiles 0 = current, 1, 2 = next ones; nf = how many of those exist)
// that come after the last load phase p+2 needs -> s_waitcnt vmcnt(that)
constexpr int allowed(int p, int nf, int JA, int JB, int NP, int TPB, int PT) {
    const int NL = JA + JB;
    const int np = p + 2, Tn = np / NP, pn = np % NP;
    // key = (tile * (NP + 1) + slot) * 64 + l  (issue order)
    int last = -1;
    for (int T = 0; T <= Tn && T <= nf; ++T)
        for (int l = 0; l < NL; ++l)
            if (T < Tn || read_phase(l, JA, TPB, PT) <= pn) {
                const int key = (T * (NP + 1) + slot(l, JA, NP, TPB, PT)) * 64 + l;
                if (key > last) last = key;
            }
    int cnt = 0;
    for (int T = 0; T <= 2 && T <= nf; ++T)
        for (int l = 0; l < NL; ++l) {
            const int sl = slot(l, JA, NP, TPB, PT);
            const int time = (T - 2) * NP + sl;  // relative to phase 0 of the current tile
            const int key = (T * (NP + 1) + sl) * 64 + l;
            if (time <= p && key > last) ++cnt;
        }
    return cnt;
}
template <int NF, int JA, int JB, int NP, int TPB, int PT>
__device__ __forceinline__ void wait_phase(int p) {  // p is a compile-time constant after unrolling: the switch folds
    switch (p) {
        case 0: wait_vmcnt<allowed(0, NF, JA, JB, NP, TPB, PT)>(); break;
        case 1: wait_vmcnt<allowed(1, NF, JA, JB, NP, TPB, PT)>(); break;
        case 2: wait_vmcnt<allowed(NP > 2 ? 2 : 0, NF, JA, JB, NP, TPB, PT)>(); break;
        case 3: wait_vmcnt<allowed(NP > 3 ? 3 : 0, NF, JA, JB, NP, TPB, PT)>(); break;
        default: wait_vmcnt<allowed(NP > 4 ? 4 : 0, NF, JA, JB, NP, TPB, PT)>(); break;
    }
}
}  // namespace pp2

template <int BM, int BN, int WAVES_N, int PT, bool CONV>
__global__ void __launch_bounds__(512) gemm_pp2_kernel(GemmArgs g) {
    constexpr int BK = 64, WAVES_M = 8 / WAVES_N;
    constexpr int WTN = BN / WAVES_N;
    constexpr int TM = 2, TN = WTN / 32;
    constexpr int NP = (TN + PT - 1) / PT;   // phases per K-tile (PT N-tiles of the wave each)
    constexpr int JA = BM / 64, JB = BN / 64;  // 64-row DMA pieces (one 16-byte load per thread each)
    constexpr int TPB = TN / JB;             // N-tiles of a wave covered by one B piece (the piece spans every wave column)
    constexpr int NL = JA + JB;              // LDS-DMA loads per thread per K-tile
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    static_assert(PT == 1 || PT == 2, "one or two N-tiles per phase");
    static_assert(BM / WAVES_M == 64 && WTN % 32 == 0 && TN % JB == 0, "bad tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int grp = wave >> 2;  // waves w and w+4 share a SIMD
    const int hi = lane >> 5, l31 = lane & 31;
    int bx, by;
    {
        const int nbx = gridDim.x, nb = gridDim.x * gridDim.y;
        const int bid = blockIdx.y * nbx + blockIdx.x;
        const int q = nb >> 3, r = nb & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        by = logical / nbx;
        bx = logical - by * nbx;
    }
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int z = blockIdx.z;
    const bool split = g.splitk > 1;
    const int zb = split ? 0 : z;
    const int nk_total = g.K / BK;
    int kt_begin = 0, kt_end = nk_total;
    if (split) {
        kt_begin = z * g.ktiles_per_split;
        kt_end = kt_begin + g.ktiles_per_split;
        if (kt_end > nk_total) kt_end = nk_total;
    }
    const f16* Ab = g.A + (int64_t)zb * g.strideA;
    const f16* Wb = g.W + (int64_t)zb * g.strideW;

    // ---- per-lane DMA descriptors: lane fills physical 16-byte slot (lane & 7) of row (8*wave + lane/8) of every 64-row piece
    // and fetches logical slot ls (XOR swizzle on the source, see gemm_kernel)
    const int rbase = wave * 8 + (lane >> 3);
    const int ls = (lane & 7) ^ ((rbase >> 1) & 7);
    int64_t a_off[JA];
    int a_iy0[JA], a_ix0[JA];  // conv: top-left input coordinate of the row's window; rows >= M get iy0 far out of range
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int m = m0 + rbase + 64 * j;
        const bool ok = m < g.M;
        if (CONV) {
            const int ohw = g.cg.OH * g.cg.OW;
            const int mm = ok ? m : 0;
            const int img = mm / ohw;
            const int rem = mm - img * ohw;
            const int oy = rem / g.cg.OW;
            const int ox = rem - oy * g.cg.OW;
            a_iy0[j] = ok ? oy * g.cg.stride - g.cg.pad_t : -(1 << 28);
            a_ix0[j] = ox * g.cg.stride - g.cg.pad_l;
            a_off[j] = ((int64_t)img * g.cg.H * g.cg.W + (int64_t)a_iy0[j] * g.cg.W + a_ix0[j]) * g.cg.Cin + ls * 8;
        } else {
            a_iy0[j] = ok ? 0 : -1;
            a_ix0[j] = 0;
            a_off[j] = (int64_t)m * g.lda + ls * 8;
        }
    }
    // B piece j = rows [j*RB, (j+1)*RB) of EVERY wave column (RB = 64 / WAVES_N), i.e. exactly the rows the waves read for N-tiles
    // j*TPB .. of theirs; this wave fills 8 of them
    constexpr int RB = 64 / WAVES_N;
    const int b_row0 = (WAVES_N == 2) ? (wave >> 2) * WTN + (wave & 3) * 8 : wave * 8;
    const int nb0 = n0 + b_row0 + (lane >> 3);
    const int64_t b_off0 = (int64_t)nb0 * g.ldw + ls * 8;
    const int b_lds0 = b_row0 * 128;  // LDS byte offset of this wave's 8 rows inside piece 0

    // K-tile position (wave-uniform, advanced incrementally: no divisions in the loop).  Conv taps are whole 64-channel chunks
    // (Cin % 64 == 0), walked chunk-major or tap-major (see gemm_kernel::prep_tile).
    struct TileK {
        int ky, kx, c0;
        int64_t a_delta;  // element offset added to a_off
        int kw;           // element offset inside a weight row
    };
    auto finish = [&](TileK& t) {
        if (CONV) {
            t.a_delta = ((int64_t)t.ky * g.cg.W + t.kx) * g.cg.Cin + t.c0;
            t.kw = (t.ky * g.cg.KW + t.kx) * g.cg.Cin + t.c0;
        } else {
            t.a_delta = t.c0;
            t.kw = t.c0;
        }
    };
    auto decode = [&](int kt) {
        TileK t;
        t.ky = t.kx = 0;
        t.c0 = kt * BK;
        if (CONV) {
            int tap;
            if (g.cg.chunk_major) {
                const int taps = g.cg.KH * g.cg.KW;
                const int chunk = kt / taps;
                tap = kt - chunk * taps;
                t.c0 = chunk * BK;
            } else {
                tap = (kt * BK) / g.cg.Cin;
                t.c0 = kt * BK - tap * g.cg.Cin;
            }
            t.ky = tap / g.cg.KW;
            t.kx = tap - t.ky * g.cg.KW;
        }
        finish(t);
        return t;
    };
    auto advance = [&](TileK& t) {
        if (CONV) {
            if (g.cg.chunk_major) {
                if (++t.kx == g.cg.KW) {
                    t.kx = 0;
                    if (++t.ky == g.cg.KH) { t.ky = 0; t.c0 += BK; }
                }
            } else {
                t.c0 += BK;
                if (t.c0 == g.cg.Cin) {
                    t.c0 = 0;
                    if (++t.kx == g.cg.KW) { t.kx = 0; ++t.ky; }
                }
            }
        } else {
            t.c0 += BK;
        }
        finish(t);
    };
    auto issue_A = [&](int j, int stage, const TileK& t) {
        bool ok;
        if (CONV) {
            const int iy = a_iy0[j] + t.ky, ix = a_ix0[j] + t.kx;
            ok = (unsigned)iy < (unsigned)g.cg.H && (unsigned)ix < (unsigned)g.cg.W;
        } else {
            ok = a_iy0[j] >= 0;
        }
        const f16* src = ok ? Ab + a_off[j] + t.a_delta : g.zeros;
        glds16(src, smem + stage * STAGE_BYTES + (j * 64 + wave * 8) * 128);
    };
    auto issue_B = [&](int j, int stage, const TileK& t) {
        const bool ok = (nb0 + j * RB) < g.N;
        const f16* src = ok ? Wb + b_off0 + (int64_t)(j * RB) * g.ldw + t.kw : g.zeros;
        glds16(src, smem + stage * STAGE_BYTES + A_BYTES + b_lds0 + j * RB * 128);
    };

    AccBlock<kL16> acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc_zero(acc[i][j]);

    using FR = Frag<kL16>;
    const int lrow = FR::lrow(lane), lkq = FR::lk(lane);
    int koff[FR::NSTEP];   // 16-byte slot of step st, swizzled with the lane's row key (tile bases are multiples of 16 rows); fragment u: + u * 16 rows
#pragma unroll
    for (int s = 0; s < FR::NSTEP; ++s) koff[s] = (FR::kslot(s, lkq) ^ ((lrow >> 1) & 7)) << 4;
    const int a_lane_off = (wm * 64 + lrow) * 128;
    const int b_lane_off = A_BYTES + (wn * WTN + lrow) * 128;

    // issue every load of tile T whose slot is `sl` (program order = ascending l)
    auto issue_slot = [&](int sl, int stage, const TileK& t) {
#pragma unroll
        for (int l = 0; l < NL; ++l)
            if (pp2::slot(l, JA, NP, TPB, PT) == sl) {
                if (l < JA) issue_A(l, stage, t);
                else issue_B(l - JA, stage, t);
            }
    };
    constexpr int pro1 = []() { int c = 0; for (int l = 0; l < JA + JB; ++l) c += pp2::slot(l, JA, NP, TPB, PT) < NP ? 1 : 0; return c; }();
    // ---- prologue: all of tile 0, the slots of tile 1 that precede phase (0,0); tile 0 must have landed before the first fragment reads
    TileK t1 = decode(kt_begin), t2;
    if (kt_begin < kt_end) {
        const TileK t0 = t1;
        advance(t1);
#pragma unroll
        for (int sl = 1; sl <= NP; ++sl) issue_slot(sl, 0, t0);
        if (kt_begin + 1 < kt_end) {
#pragma unroll
            for (int sl = 1; sl < NP; ++sl) issue_slot(sl, 1, t1);
            wait_vmcnt<pro1>();
        } else {
            wait_vmcnt<0>();
        }
    }
    t2 = t1;
    advance(t2);
    __builtin_amdgcn_s_barrier();
    f16x8 af[TM][FR::NSTEP][FR::PER], bf[PT][FR::NSTEP][FR::PER];
    // fragments of phase (0,0)
    {
        const char* fa = smem + a_lane_off;
        const char* fb = smem + b_lane_off;
#pragma unroll
        for (int s = 0; s < FR::NSTEP; ++s)
#pragma unroll
            for (int u = 0; u < FR::PER; ++u) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i][s][u] = *reinterpret_cast<const f16x8*>(fa + koff[s] + i * 4096 + u * 2048);
#pragma unroll
                for (int jj = 0; jj < PT; ++jj) bf[jj][s][u] = *reinterpret_cast<const f16x8*>(fb + koff[s] + jj * 4096 + u * 2048);
            }
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger: group 1 runs one barrier interval behind group 0

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const bool has1 = (kt + 1) < kt_end, has2 = (kt + 2) < kt_end;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int j0 = PT * p;
            // -------- load segment: one slot of DMA, counted wait for what phase q+2 reads, retire the reads issued in the last MFMA segment
            if (p == 0) {
                if (has1) issue_slot(NP, cur ^ 1, t1);
            } else {
                if (has2) issue_slot(p, cur, t2);
            }
            if (has2) pp2::wait_phase<2, JA, JB, NP, TPB, PT>(p);
            else if (has1) pp2::wait_phase<1, JA, JB, NP, TPB, PT>(p);
            else pp2::wait_phase<0, JA, JB, NP, TPB, PT>(p);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // -------- MFMA segment; after each k-step its fragment registers are refilled with the next phase's data
            const bool next_in_tile = (p + 1 < NP);
            const bool have_next = next_in_tile || has1;
            const char* nfa = smem + (cur ^ 1) * STAGE_BYTES + a_lane_off;                                   // A of tile kt+1
            const char* nfb = smem + (next_in_tile ? cur : (cur ^ 1)) * STAGE_BYTES + b_lane_off + (next_in_tile ? (j0 + PT) * 4096 : 0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < FR::NSTEP; ++s) {
#pragma unroll
                for (int jj = 0; jj < PT; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mma_step(acc[i][j0 + jj], bf[jj][s], af[i][s]);  // transposed tile (see gemm_epilogue)
                if (have_next) {
#pragma unroll
                    for (int u = 0; u < FR::PER; ++u) {
                        if (!next_in_tile) {
#pragma unroll
                            for (int i = 0; i < TM; ++i) af[i][s][u] = *reinterpret_cast<const f16x8*>(nfa + koff[s] + i * 4096 + u * 2048);
                        }
#pragma unroll
                        for (int jj = 0; jj < PT; ++jj) bf[jj][s][u] = *reinterpret_cast<const f16x8*>(nfb + koff[s] + jj * 4096 + u * 2048);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        t1 = t2;
        advance(t2);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // re-align the two groups
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (ODISE_ABLATE(g, 4)) return;
    gemm_epilogue<BM, BN, WAVES_M, WAVES_N, epi_wave_rows(BM, BN, WAVES_M, pp_lds_bytes(BM, BN, WAVES_M)), false, CONV, true,
                  epi16_rows_if_enabled(BM, BN, WAVES_M, pp_lds_bytes(BM, BN, WAVES_M)), !CONV, pp_lds_bytes(BM, BN, WAVES_M)>(g, acc, smem, m0, n0, z, zb, split);
}

// ---- 8-phase pipelined 256x256 tile (round 5) ---------------------------------------------------------------------------------------
// The schedule of /opt/skills/guides/cdna_hip_programming.md section 5 ("The 256^2 8-phase template"), first built as a yardstick
// (csrc/gemm8p.hip, tools/gemm8p_bench.py: 1320-1360 TFLOP/s on uniform random operands at 4096^3 / 8192^3 / 65536x1024x4096 against
// 1030-1110 for the two ping-pong kernels above on the same box, same process, same data: profiles/r05_8phase_vs_pp2.txt) and then moved
// onto the product's operands, MFMA shape and epilogue:
//   * 8 waves as 2(M) x 4(N), 128x64 per wave = 4 x 2 tiles of v_mfma_f32_32x32x16_f16 (the shape every kernel of this file uses: with the
//     k-steps of a K-tile taken in the same order the fp32 sums are the same bits as the other kernels');
//   * LDS: 2 K-tile buffers x {A0, A1, B0, B1} half-tiles (128 rows x 128 B each; half X0 / X1 = the first / second 64 rows (32 columns) of
//     every wave row (wave column)), each two global_load_lds_dwordx4 per thread;
//   * a K-tile is four phases, each one 64x32 quadrant of the wave tile over the whole K-tile (8 MFMAs = 512 matrix-pipe cycles):
//       phase 1: read B0, A0; stage A1(t+1) -> C[0][0]     phase 3: read A1; stage A0(t+2)  -> C[1][1]
//       phase 2: read B1;     stage B0(t+2) -> C[0][1]     phase 4: stage B1(t+2); vmcnt(6) -> C[1][0]
//     {reads, 2 DMA, [wait]} s_barrier lgkmcnt(0) {MFMA} s_barrier; wave row 1 runs one barrier behind wave row 0 (one wave of each per
//     SIMD), so one group multiplies while the other reads and stages;
//   * three half-tiles of DMA stay in flight across every barrier; the one vector-memory wait per K-tile (phase 4) retires K-tile t+1,
//     which is read from the next phase on; a buffer is restaged two phases after its last read (B0: one phase, its reads are retired by
//     lgkmcnt(8) before phase 1's first barrier).
// Rows beyond M / N and padded convolution taps read a zero line (no predication of the DMA, uniform vmcnt accounting).
template <int BM, int BN, bool CONV, bool L16>
__global__ void __launch_bounds__(512) gemm8_kernel(GemmArgs g) {
    constexpr int BK = 64, WAVES_M = BM / 128, WAVES_N = BN / 64;
    static_assert(WAVES_M * WAVES_N == 8 && BM % 128 == 0 && BN % 128 == 0, "8 waves of 128x64");
    constexpr int JA = BM / 128, JB = BN / 128;                 // 64-row DMA pieces (one 16-byte load per thread) per A / B half-tile
    constexpr int HALF_A = (BM / 2) * BK * 2, HALF_B = (BN / 2) * BK * 2, STAGE = 2 * HALF_A + 2 * HALF_B;
    constexpr int OFF_A0 = 0, OFF_A1 = HALF_A, OFF_B0 = 2 * HALF_A, OFF_B1 = 2 * HALF_A + HALF_B;
    constexpr int INFLIGHT = JA + 2 * JB;                       // loads issued after A1(t+1): B0, A0, B1 of K-tile t+2
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int grp = wave >> 2;   // waves w and w + 4 share a SIMD: the two groups run one barrier apart
    int bx, by;
    {
        const int nbx = gridDim.x, nb = gridDim.x * gridDim.y;
        const int bid = blockIdx.y * nbx + blockIdx.x;
        const int q = nb >> 3, r = nb & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        by = logical / nbx;
        bx = logical - by * nbx;
    }
    const int m0 = by * BM;
    const int n0 = bx * BN;
    const int z = blockIdx.z;
    const bool split = g.splitk > 1;
    const int zb = split ? 0 : z;
    const int nk_total = g.K / BK;
    int kt_begin = 0, kt_end = nk_total;
    if (split) {
        kt_begin = z * g.ktiles_per_split;
        kt_end = kt_begin + g.ktiles_per_split;
        if (kt_end > nk_total) kt_end = nk_total;
    }
    const int nk = kt_end - kt_begin;
    const f16* Ab = g.A + (int64_t)zb * g.strideA;
    const f16* Wb = g.W + (int64_t)zb * g.strideW;

    // ---- staging descriptors: the thread fills physical 16-byte slot (lane & 7) of half-tile row i*64 + srow (piece i) and fetches
    // logical slot ls (XOR swizzle on the source: the LDS image of a DMA instruction is lane-linear)
    const int srow = wave * 8 + (lane >> 3);
    const int ls = (lane & 7) ^ ((srow >> 1) & 7);
    // A half h, piece i: tile row i*128 + h*64 + srow.  Dense: one base pointer, row validity by compare.  Conv: per row the window's
    // top-left input coordinate (packed y << 16 | x & 0xffff; rows >= M get y far out of range) and a 32-bit element offset.
    int a_yx[2][JA], a_eoff[2][JA];
    bool a_ok[2][JA];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < JA; ++i) {
            const int m = m0 + i * 128 + h * 64 + srow;
            const bool ok = m < g.M;
            a_ok[h][i] = ok;
            a_yx[h][i] = 0;
            a_eoff[h][i] = 0;
            if (CONV) {
                const int ohw = g.cg.OH * g.cg.OW;
                const int mm = ok ? m : 0;
                const int img = mm / ohw;
                const int rem = mm - img * ohw;
                const int oy = rem / g.cg.OW;
                const int ox = rem - oy * g.cg.OW;
                const int iy0 = ok ? oy * g.cg.stride - g.cg.pad_t : -30000;
                const int ix0 = ox * g.cg.stride - g.cg.pad_l;
                a_yx[h][i] = (int)(((unsigned)iy0 << 16) | ((unsigned)ix0 & 0xffffu));
                a_eoff[h][i] = ok ? (int)((((int64_t)img * g.cg.H + iy0) * g.cg.W + ix0) * g.cg.Cin) + ls * 8 : 0;
            }
        }
    const f16* const a_src = Ab + (int64_t)(m0 + srow) * g.lda + ls * 8;   // dense only
    // B half h, piece i: tile row (i*2 + wave/4)*64 + h*32 + (wave%4)*8 + lane/8
    const int b_row = (wave >> 2) * 64 + (wave & 3) * 8 + (lane >> 3);
    const f16* const b_src = Wb + (int64_t)(n0 + b_row) * g.ldw + ls * 8;
    bool b_ok[2][JB];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < JB; ++i) b_ok[h][i] = (n0 + b_row + i * 128 + h * 32) < g.N;
    char* const lds_w = smem + wave * 1024;   // this wave's 8 rows inside a 64-row piece

    // K-tile position (wave-uniform, advanced incrementally; conv taps are whole 64-channel chunks walked chunk-major or tap-major: see gemm_kernel::prep_tile)
    struct TileK {
        int ky, kx, c0;
        int64_t a_delta;  // element offset added to the A row offset
        int kw;           // element offset inside a weight row
    };
    auto finish = [&](TileK& t) {
        if (CONV) {
            t.a_delta = ((int64_t)t.ky * g.cg.W + t.kx) * g.cg.Cin + t.c0;
            t.kw = (t.ky * g.cg.KW + t.kx) * g.cg.Cin + t.c0;
        } else {
            t.a_delta = t.c0;
            t.kw = t.c0;
        }
    };
    auto decode = [&](int kt) {
        TileK t;
        t.ky = t.kx = 0;
        t.c0 = kt * BK;
        if (CONV) {
            int tap;
            if (g.cg.chunk_major) {
                const int taps = g.cg.KH * g.cg.KW;
                const int chunk = kt / taps;
                tap = kt - chunk * taps;
                t.c0 = chunk * BK;
            } else {
                tap = (kt * BK) / g.cg.Cin;
                t.c0 = kt * BK - tap * g.cg.Cin;
            }
            t.ky = tap / g.cg.KW;
            t.kx = tap - t.ky * g.cg.KW;
        }
        finish(t);
        return t;
    };
    auto advance = [&](TileK& t) {
        if (CONV) {
            if (g.cg.chunk_major) {
                if (++t.kx == g.cg.KW) {
                    t.kx = 0;
                    if (++t.ky == g.cg.KH) { t.ky = 0; t.c0 += BK; }
                }
            } else {
                t.c0 += BK;
                if (t.c0 == g.cg.Cin) {
                    t.c0 = 0;
                    if (++t.kx == g.cg.KW) { t.kx = 0; ++t.ky; }
                }
            }
        } else {
            t.c0 += BK;
        }
        finish(t);
    };
    auto stage_A = [&](int h, int buf, const TileK& t) {
        char* d = lds_w + buf * STAGE + (h ? OFF_A1 : OFF_A0);
#pragma unroll
        for (int i = 0; i < JA; ++i) {
            const f16* src;
            if (CONV) {
                const int iy = (a_yx[h][i] >> 16) + t.ky, ix = (int)(short)(a_yx[h][i] & 0xffff) + t.kx;
                const bool ok = (unsigned)iy < (unsigned)g.cg.H && (unsigned)ix < (unsigned)g.cg.W;
                src = ok ? Ab + ((int64_t)a_eoff[h][i] + t.a_delta) : g.zeros;
            } else {
                src = a_ok[h][i] ? a_src + (int64_t)(i * 128 + h * 64) * g.lda + t.a_delta : g.zeros;
            }
            glds16(src, d + i * 8192);
        }
    };
    auto stage_B = [&](int h, int buf, const TileK& t) {
        char* d = lds_w + buf * STAGE + (h ? OFF_B1 : OFF_B0);
#pragma unroll
        for (int i = 0; i < JB; ++i) {
            const f16* src = b_ok[h][i] ? b_src + (int64_t)(i * 128 + h * 32) * g.ldw + t.kw : g.zeros;
            glds16(src, d + i * 8192);
        }
    };

    // ---- accumulators and fragments of the wave's 128x64 tile (FragLayout).  L16: 16-row tiles i = 0..3 of the 64-row sub-tile (block
    // p = 2 ih + i / 2, row half i % 2), 16-column tiles jj = 0, 1 of the 32-column sub-tile, k-steps of 32; L32: 32-row tiles i = 0, 1, k-steps of 16
    typedef typename std::conditional<L16, f32x4[4][2][4], f32x16[4][2]>::type AccT;
    AccT acc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (L16) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[i][j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        }
    // lane reads row lrow of a tile, 16-byte slot kslot(s) of k-step s -> physical slot ^ ((row >> 1) & 7) (tile bases are multiples of 16 rows)
    const int lrow = L16 ? (lane & 15) : (lane & 31);
    const int lk = L16 ? (lane >> 4) : (lane >> 5);
    constexpr int KS = L16 ? 2 : 4, KSLOTS = L16 ? 4 : 2;       // k-steps per K-tile, 16-byte slots per lane group and k-step
    constexpr int MT = L16 ? 4 : 2, NT_ = L16 ? 2 : 1;          // row / column tiles per sub-tile
    constexpr int TROWS = L16 ? 16 : 32;
    int koff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) koff[s] = ((s * KSLOTS + lk) ^ ((lrow >> 1) & 7)) << 4;
    const int a_lane = (wm * 64 + lrow) * 128;
    const int b_lane = (wn * 32 + lrow) * 128;
    f16x8 af[MT][KS], b0f[NT_][KS], b1f[NT_][KS];
    auto read_A = [&](int h, int buf) {
        const char* p = smem + buf * STAGE + (h ? OFF_A1 : OFF_A0) + a_lane;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int s = 0; s < KS; ++s) af[i][s] = *reinterpret_cast<const f16x8*>(p + i * TROWS * 128 + koff[s]);
    };
    auto read_B = [&](f16x8 (&bf)[NT_][KS], int h, int buf) {
        const char* p = smem + buf * STAGE + (h ? OFF_B1 : OFF_B0) + b_lane;
#pragma unroll
        for (int j = 0; j < NT_; ++j)
#pragma unroll
            for (int s = 0; s < KS; ++s) bf[j][s] = *reinterpret_cast<const f16x8*>(p + j * TROWS * 128 + koff[s]);
    };
#define G8_BAR()                               \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
    // the MFMAs of quadrant (ih, jh): 64 rows x 32 columns x the whole K-tile; operands swapped (transposed tile in the registers, see gemm_epilogue)
#define G8_MMA(ih, bfr, jh)                                                                                                               \
    do {                                                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                                                    \
        if constexpr (L16) {                                                                                                              \
            _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                                 \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                             \
                    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                      \
                        acc[(ih) * 2 + i / 2][jh][(i % 2) * 2 + jj] =                                                                     \
                            __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[jj][s], af[i][s], acc[(ih) * 2 + i / 2][jh][(i % 2) * 2 + jj], 0, 0, 0); \
        } else {                                                                                                                          \
            _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                                 \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                             \
                    acc[(ih) * 2 + i][jh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bfr[0][s], af[i][s], acc[(ih) * 2 + i][jh], 0, 0, 0);  \
        }                                                                                                                                 \
        __builtin_amdgcn_s_setprio(0);                                                                                                    \
        G8_BAR();                                                                                                                         \
    } while (0)

    TileK t1 = decode(kt_begin), t2;   // t1 / t2: K-tiles t+1 / t+2 of the tile being multiplied
    // one K-tile (4 phases) on buffer `buf`; has1 / has2: K-tiles t+1 / t+2 exist (compile-time constants in the steady loop, block-uniform
    // run-time flags in the tail of at most three K-tiles)
    auto ktile = [&](const int buf, const bool has1, const bool has2) __attribute__((always_inline)) {
        // phase 1
        read_B(b0f, 0, buf);
        __builtin_amdgcn_sched_barrier(0);
        read_A(0, buf);
        if (has1) stage_A(1, buf ^ 1, t1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // retires the four B0 reads (issued first): B0 may be restaged in the next phase
        G8_BAR();
        G8_MMA(0, b0f, 0);
        // phase 2
        read_B(b1f, 1, buf);
        if (has2) stage_B(0, buf, t2);
        G8_BAR();
        G8_MMA(0, b1f, 1);
        // phase 3
        read_A(1, buf);
        if (has2) stage_A(0, buf, t2);
        G8_BAR();
        G8_MMA(1, b1f, 1);
        // phase 4
        if (has2) { stage_B(1, buf, t2); wait_vmcnt<INFLIGHT>(); }
        else if (has1) wait_vmcnt<0>();
        G8_BAR();
        G8_MMA(1, b0f, 0);
        t1 = t2;
        advance(t2);
    };

    // ---- prologue: K-tile 0 (B0, A0, B1, A1) and B0, A0, B1 of K-tile 1
    if (nk > 0) {
        const TileK t0 = t1;
        advance(t1);
        stage_B(0, 0, t0); stage_A(0, 0, t0); stage_B(1, 0, t0); stage_A(1, 0, t0);
        if (nk > 1) {
            stage_B(0, 1, t1); stage_A(0, 1, t1); stage_B(1, 1, t1);
            wait_vmcnt<INFLIGHT>();
        } else {
            wait_vmcnt<0>();
        }
    }
    t2 = t1;
    advance(t2);
    G8_BAR();
    if (grp == 1) G8_BAR();   // stagger: the second wave group runs one barrier behind the first

    int kt = 0;
    for (; kt + 3 < nk; kt += 2) {   // steady state: two K-tiles per iteration, buffers and flags compile-time constants
        ktile(0, true, true);
        ktile(1, true, true);
    }
    for (; kt < nk; ++kt) ktile(kt & 1, kt + 1 < nk, kt + 2 < nk);   // the last one to three K-tiles (kt is even here)
    if (grp == 0) G8_BAR();   // re-align the two groups
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
#undef G8_MMA
#undef G8_BAR
    if (ODISE_ABLATE(g, 4)) return;   // tools: main loop only
    constexpr int LDS = pp_lds_bytes(BM, BN, WAVES_M);
    static_assert(LDS >= 2 * STAGE, "operand stages exceed the LDS request");
    gemm_epilogue<BM, BN, WAVES_M, WAVES_N, epi_wave_rows(BM, BN, WAVES_M, LDS), false, CONV, true, epi16_rows_if_enabled(BM, BN, WAVES_M, LDS), !CONV, LDS>(
        g, acc, smem, m0, n0, z, zb, split);
}

// (A third structure - every wave free-running through the four k-steps with register double-buffered fragments and ONE barrier per
// K-tile, i.e. the classic software-pipelined GEMM - was built and measured too: bit-identical results, 5-12 % SLOWER than the
// ping-pong kernels on the large shapes (898 vs 1012 TFLOP/s at 65536x512x4096), so the barrier count is not what bounds them.)

// ---- 3x3 / stride 1 / pad 1 convolution with the A operand reused from an LDS-resident input patch ("halo") ------------------------
// The im2col view re-fetches every input pixel 9 times (once per tap) through the LDS-DMA path, which is what bounds the ping-pong
// kernel on the 3x3 layers (ablation: 23 % of the loop is operand delivery).  Here a block owns a 16x16 OUTPUT patch of one image:
// per 64-channel chunk it fetches the 18x18 input patch once (41 KB instead of 9 x 32 KB) and the nine K-tiles of the chunk read
// their A fragments from it at the tap's pixel offset; only the weights stream per K-tile.  Same ping-pong phase structure, MFMA
// order and fp32 summation order (chunk-major) as gemm_pp_kernel, so results are identical to it.
//   LDS: [B stage 0][B stage 1][halo 0][halo 1]; halo pixel hp (row-major 18x18) at hp*128 B, 16-byte slots XOR-swizzled with
//   (halo column >> 1) & 7 on the DMA source and on the fragment reads: a ds_read_b128 lane group covers 8 columns of one patch row
//   and the 8 complementary columns of the next, i.e. 16 consecutive columns = 16 distinct 16-byte units (keying on the pixel index
//   instead costs +3.7 LDS cycles per A read: measured SQ_LDS_BANK_CONFLICT 20x).  DMA groups of 8 pixels (1 KiB per wave-instruction): 41 real groups + dummy ones so that every wave
//   issues the same H = 6 loads per chunk (uniform vmcnt accounting).
// Schedule per K-tile t (LB = B loads per thread per phase): (t,0) issues [halo of the NEXT chunk when t opens a chunk] and the
// second half of B(t+1); (t,1) issues the first half of B(t+2).  Split-K ranges are whole chunks.
template <int BN, int PT>
__global__ void __launch_bounds__(512) conv3_halo_kernel(GemmArgs g) {
    constexpr int BM = 256, BK = 64, WAVES_N = 2, WAVES_M = 4;
    constexpr int WTN = BN / WAVES_N;
    constexpr int TM = 2, TN = WTN / 32;
    constexpr int NP = 2;
    constexpr int JB = BN / 64;       // B loads per thread per K-tile
    constexpr int LB = JB / 2;        // per phase
    constexpr int RB = 32;            // rows of a B piece per wave column
    constexpr int HW_ = 18, HPIX = HW_ * HW_;
    constexpr int HGROUPS = (HPIX + 7) / 8;  // 41 real 8-pixel groups
    constexpr int H = 6;                     // halo loads per thread per chunk (48 groups over 8 waves; groups >= HGROUPS are dummies)
    constexpr int B_BYTES = BN * BK * 2;
    constexpr int HALO_BYTES = (HGROUPS + 1) * 1024;  // + one dummy group
    constexpr int HALO0 = 2 * B_BYTES;
    static_assert(TN == PT * NP && JB == 2 * LB, "two phases of PT N-tiles, one B piece per N-tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int grp = wave >> 2;
    const int hi = lane >> 5, l31 = lane & 31;
    int bx, by;
    {
        const int nbx = gridDim.x, nb = gridDim.x * gridDim.y;
        const int bid = blockIdx.y * nbx + blockIdx.x;
        const int q = nb >> 3, r = nb & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        by = logical / nbx;
        bx = logical - by * nbx;
    }
    const int m0 = by * BM;  // patch index * 256 (the epilogue maps tile rows to pixels)
    const int n0 = bx * BN;
    const int z = blockIdx.z;
    const bool split = g.splitk > 1;
    const int zb = 0;
    const int nk_total = g.K / BK;
    int kt_begin = 0, kt_end = nk_total;
    if (split) {
        kt_begin = z * g.ktiles_per_split;  // a multiple of 9: whole chunks
        kt_end = kt_begin + g.ktiles_per_split;
        if (kt_end > nk_total) kt_end = nk_total;
    }
    const f16* Ab = g.A;
    const f16* Wb = g.W;
    const int per_img = g.cg.halo_tx * g.cg.halo_ty;
    const int img = by / per_img, pr = by - img * per_img;
    const int py0 = (pr / g.cg.halo_tx) * 16 - 1, px0 = (pr % g.cg.halo_tx) * 16 - 1;  // input coordinate of halo pixel (0,0)

    // ---- halo DMA descriptors: load h of this wave fills group h*8 + wave; lane -> pixel 8*group + lane/8, physical slot lane & 7
    int64_t h_off[H];  // element offset of the lane's 16 bytes for chunk 0, or -1 (out of the image / dummy -> zero line)
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const int group = h * 8 + wave;
        const int hp = group * 8 + (lane >> 3);
        h_off[h] = -1;
        if (group < HGROUPS && hp < HPIX) {
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int iy = py0 + hy, ix = px0 + hx;
            // fused nearest-2x upsample (round 6): the patch is the UPSAMPLED input's - pixel (iy, ix) of it is source pixel (iy / 2, ix / 2); the
            // duplicates come out of the L2 and the nine taps read the patch exactly as for a plain 3x3 convolution
            const int up = g.cg.ups ? 1 : 0;
            if ((unsigned)iy < (unsigned)(g.cg.H << up) && (unsigned)ix < (unsigned)(g.cg.W << up))
                h_off[h] = (((int64_t)img * g.cg.H + (iy >> up)) * g.cg.W + (ix >> up)) * g.cg.Cin + (((lane & 7) ^ ((hx >> 1) & 7)) << 3);
        }
    }
    auto issue_halo = [&](int chunk) {
        char* dst = smem + HALO0 + (chunk & 1) * HALO_BYTES;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int group = h * 8 + wave;
            const f16* src = h_off[h] >= 0 ? Ab + h_off[h] + chunk * BK : g.zeros;
            glds16(src, dst + (group < HGROUPS ? group : HGROUPS) * 1024);
        }
    };
    // ---- B (weight) DMA: piece j = rows [32j, 32j+32) of both wave columns
    const int rbase = wave * 8 + (lane >> 3);
    const int ls = (lane & 7) ^ ((rbase >> 1) & 7);
    const int b_row0 = (wave >> 2) * WTN + (wave & 3) * 8;
    const int nb0 = n0 + b_row0 + (lane >> 3);
    const int64_t b_off0 = (int64_t)nb0 * g.ldw + ls * 8;
    const int b_lds0 = b_row0 * 128;
    auto issue_B = [&](int j, int stage, int kw) {
        const bool ok = (nb0 + j * RB) < g.N;
        const f16* src = ok ? Wb + b_off0 + (int64_t)(j * RB) * g.ldw + kw : g.zeros;
        glds16(src, smem + stage * B_BYTES + b_lds0 + j * RB * 128);
    };
    // K-tile position: chunk-major (tap inner).  kw = element offset inside a weight row [Cout][ky][kx][Cin]
    struct TileK {
        int ky, kx, chunk, kw;
    };
    auto finish = [&](TileK& t) { t.kw = (t.ky * 3 + t.kx) * g.cg.Cin + t.chunk * BK; };
    auto advance = [&](TileK& t) {
        if (++t.kx == 3) {
            t.kx = 0;
            if (++t.ky == 3) { t.ky = 0; ++t.chunk; }
        }
        finish(t);
    };

    AccBlock<kL16> acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc_zero(acc[i][j]);

    using FR = Frag<kL16>;
    const int lrow = FR::lrow(lane), lkq = FR::lk(lane);
    int koff[FR::NSTEP];   // 16-byte slot of step st, swizzled with the lane's row key (tile bases are multiples of 16 rows); fragment u: + u * 16 rows
#pragma unroll
    for (int s = 0; s < FR::NSTEP; ++s) koff[s] = (FR::kslot(s, lkq) ^ ((lrow >> 1) & 7)) << 4;
    const int b_lane_off = (wn * WTN + lrow) * 128;
    const int a_pix0 = (wm * 4 + (lrow >> 4)) * HW_ + (lrow & 15);  // halo pixel of this lane's row of A tile i = 0, fragment 0 at tap (0,0); tile 1: +2 rows, fragment u: + u rows

    // A fragments of the K-tile at position t (chunk, tap) for k-step s, from the chunk's halo buffer
    auto read_a = [&](const TileK& t, int s, f16x8 (&dst)[TM][FR::NSTEP][FR::PER]) {
        const char* ha = smem + HALO0 + (t.chunk & 1) * HALO_BYTES;
        const int pix = a_pix0 + t.ky * HW_ + t.kx;
        const int key = (((lrow & 15) + t.kx) >> 1) & 7;  // swizzle key = halo COLUMN / 2 (see the header)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int u = 0; u < FR::PER; ++u)
                dst[i][s][u] = *reinterpret_cast<const f16x8*>(ha + (pix + (i * 2 + u) * HW_) * 128 + ((FR::kslot(s, lkq) ^ key) << 4));
    };
    // ---- prologue: halo of the first chunk, B of tile 0, first half of B of tile 1; everything of tile 0 lands before the first reads
    TileK t0;
    t0.ky = t0.kx = 0;
    t0.chunk = kt_begin / 9;
    finish(t0);
    TileK t1 = t0, t2;
    if (kt_begin < kt_end) {
        advance(t1);
        issue_halo(t0.chunk);
#pragma unroll
        for (int j = 0; j < JB; ++j) issue_B(j, 0, t0.kw);
        if (kt_begin + 1 < kt_end) {
#pragma unroll
            for (int j = 0; j < LB; ++j) issue_B(j, 1, t1.kw);
            wait_vmcnt<LB>();
        } else {
            wait_vmcnt<0>();
        }
    }
    t2 = t1;
    advance(t2);
    __builtin_amdgcn_s_barrier();
    TileK tc = t0;  // position of the K-tile being multiplied
    f16x8 af[TM][FR::NSTEP][FR::PER], bf[PT][FR::NSTEP][FR::PER];
    {   // fragments of phase (0,0)
        const char* fb = smem + b_lane_off;
#pragma unroll
        for (int s = 0; s < FR::NSTEP; ++s) {
            read_a(tc, s, af);
#pragma unroll
            for (int jj = 0; jj < PT; ++jj)
#pragma unroll
                for (int u = 0; u < FR::PER; ++u) bf[jj][s][u] = *reinterpret_cast<const f16x8*>(fb + koff[s] + jj * 4096 + u * 2048);
        }
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger: group 1 runs one barrier interval behind group 0

    // Fragment reads live in the MFMA segment (see gemm_pp2_kernel): after k-step s its registers are refilled with the next phase's
    // data, so the counted waits cover what phase q+2 reads.  (t,0) issues the second half of B(t+1) and then [the halo of the next
    // chunk when t opens a chunk]; (t,1) issues the first half of B(t+2).
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const bool has1 = (kt + 1) < kt_end, has2 = (kt + 2) < kt_end;
        const bool opens = tc.ky == 0 && tc.kx == 0;                       // first tap of a chunk
        const bool halo_next = opens && (kt + 9) < kt_end;                 // the next chunk is inside this block's K range
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int j0 = PT * p;
            // -------- load segment
            if (p == 0) {
                if (has1) {
#pragma unroll
                    for (int j = LB; j < JB; ++j) issue_B(j, cur ^ 1, t1.kw);
                }
                if (halo_next) issue_halo(tc.chunk + 1);
                // phase (t+1,0) reads the first half of B(t+1) and a halo fetched a chunk ago: younger = this segment's loads
                if (halo_next) { if (has1) wait_vmcnt<LB + H>(); else wait_vmcnt<H>(); }
                else { if (has1) wait_vmcnt<LB>(); else wait_vmcnt<0>(); }
            } else {
                if (has2) {
#pragma unroll
                    for (int j = 0; j < LB; ++j) issue_B(j, cur, t2.kw);
                }
                // phase (t+1,1) reads the second half of B(t+1) (issued in (t,0) BEFORE the halo): younger = [halo], first half of B(t+2)
                if (halo_next) { if (has2) wait_vmcnt<LB + H>(); else wait_vmcnt<H>(); }
                else { if (has2) wait_vmcnt<LB>(); else wait_vmcnt<0>(); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // -------- MFMA segment with the next phase's fragment reads
            const bool next_in_tile = (p + 1 < NP);
            const bool have_next = next_in_tile || has1;
            const char* nfb = smem + (next_in_tile ? cur : (cur ^ 1)) * B_BYTES + b_lane_off + (next_in_tile ? (j0 + PT) * 4096 : 0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < FR::NSTEP; ++s) {
#pragma unroll
                for (int jj = 0; jj < PT; ++jj)
#pragma unroll
                    for (int i = 0; i < TM; ++i) mma_step(acc[i][j0 + jj], bf[jj][s], af[i][s]);  // transposed tile (see gemm_epilogue)
                if (have_next) {
                    if (!next_in_tile) read_a(t1, s, af);
#pragma unroll
                    for (int jj = 0; jj < PT; ++jj)
#pragma unroll
                        for (int u = 0; u < FR::PER; ++u) bf[jj][s][u] = *reinterpret_cast<const f16x8*>(nfb + koff[s] + jj * 4096 + u * 2048);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        tc = t1;
        t1 = t2;
        advance(t2);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // re-align the two groups
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    gemm_epilogue<BM, BN, WAVES_M, WAVES_N, epi_wave_rows(BM, BN, WAVES_M, halo_lds_bytes(BN)), true, true, true,
                  epi16_rows_if_enabled(BM, BN, WAVES_M, halo_lds_bytes(BN)), false, halo_lds_bytes(BN)>(g, acc, smem, m0, n0, z, zb, split);
}

// ---- 3x3 halo convolution, 4-wave form: TWO co-resident blocks per CU --------------------------------------------------------------------
// The 8-wave halo kernel owns a CU alone, and the rounds of a launch stay in step across the chip: every block loads its first halo and
// weights, multiplies, then writes its tile at the same time as all the others - memory and matrix pipes take turns instead of overlapping.
// On the 128-channel VAE level (K = 1152: 18 K-tiles per block) the two phases are nearly equal, which is why three different 8-wave
// structures all land at 630-650 TFLOP/s (profiles/r02_vae128_level_experiments.txt): 3.5 GB of compulsory traffic at ~5 TB/s is 0.7 ms, the
// MFMA work at the ~1.1 PFLOP/s these loops sustain is 1.1 ms, and the launch takes their SUM, 1.8 ms.
// Here a block is four waves (one per SIMD) on the same 16x16-pixel x BN-channel tile - a wave owns 4 patch rows x all BN channels, the
// accumulator shape of the 256-channel kernel's waves - with ONE halo buffer and two weight stages: 74 KB of LDS and <= 256 VGPRs, so two
// blocks share a CU.  They drift apart within a few tiles (16 384 blocks per launch), and the prologue / halo reload / epilogue of one runs
// under the K-tiles of the other: the hardware interleaves the two instruction streams per SIMD, no cross-block protocol.  Inside a
// block the loop is the plain one (weights of K-tile t+1 in flight during K-tile t, one barrier per K-tile; the halo of the next chunk is
// fetched at the chunk boundary, once the last tap has been read).  Same chunk-major fp32 summation order as conv3_halo_kernel: bit-identical.
template <int BN>
__global__ void __launch_bounds__(256, 2) conv3_halo4_kernel(GemmArgs g) {
    constexpr int BM = 256, BK = 64, WAVES_M = 4, WAVES_N = 1;
    constexpr int TM = 2, TN = BN / 32;
    constexpr int JB = BN / 32;              // weight loads per thread per K-tile (BN rows x 8 slots over 256 threads)
    constexpr int HW_ = 18, HPIX = HW_ * HW_;
    constexpr int HGROUPS = (HPIX + 7) / 8;  // 41 real 8-pixel groups
    constexpr int H = 11;                    // halo loads per thread per chunk (44 group slots over 4 waves; slots >= HGROUPS are dummies)
    constexpr int B_BYTES = BN * BK * 2;
    constexpr int HALO0 = 2 * B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave;
    const int hi = lane >> 5, l31 = lane & 31;
    int bx, by;
    {
        const int nbx = gridDim.x, nb = gridDim.x * gridDim.y;
        const int bid = blockIdx.y * nbx + blockIdx.x;
        const int q = nb >> 3, r = nb & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        by = logical / nbx;
        bx = logical - by * nbx;
    }
    const int m0 = by * BM;  // patch index * 256 (the epilogue maps tile rows to pixels)
    const int n0 = bx * BN;
    const int z = blockIdx.z;
    const bool split = g.splitk > 1;
    const int zb = 0;
    const int nk_total = g.K / BK;
    int kt_begin = 0, kt_end = nk_total;
    if (split) {
        kt_begin = z * g.ktiles_per_split;  // a multiple of 9: whole chunks
        kt_end = kt_begin + g.ktiles_per_split;
        if (kt_end > nk_total) kt_end = nk_total;
    }
    const f16* Wb = g.W;
    const int per_img = g.cg.halo_tx * g.cg.halo_ty;
    const int img = by / per_img, pr = by - img * per_img;
    const int py0 = (pr / g.cg.halo_tx) * 16 - 1, px0 = (pr % g.cg.halo_tx) * 16 - 1;  // input coordinate of halo pixel (0,0)
    const f16* Ab = g.A + (int64_t)img * g.cg.H * g.cg.W * g.cg.Cin;                    // this image's plane: 32-bit offsets below

    // ---- halo DMA descriptors: load h of this wave fills group h*4 + wave; lane -> pixel 8*group + lane/8, physical slot lane & 7
    int h_off[H];  // element offset inside the image of the lane's 16 bytes for chunk 0, or -1 (out of the image / dummy -> zero line)
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const int group = h * 4 + wave;
        const int hp = group * 8 + (lane >> 3);
        h_off[h] = -1;
        if (group < HGROUPS && hp < HPIX) {
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int iy = py0 + hy, ix = px0 + hx;
            const int up = g.cg.ups ? 1 : 0;   // fused nearest-2x upsample: see conv3_halo_kernel
            if ((unsigned)iy < (unsigned)(g.cg.H << up) && (unsigned)ix < (unsigned)(g.cg.W << up))
                h_off[h] = ((iy >> up) * g.cg.W + (ix >> up)) * g.cg.Cin + (((lane & 7) ^ ((hx >> 1) & 7)) << 3);
        }
    }
    auto issue_halo = [&](int chunk) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int group = h * 4 + wave;
            const f16* src = h_off[h] >= 0 ? Ab + h_off[h] + chunk * BK : g.zeros;
            glds16(src, smem + HALO0 + (group < HGROUPS ? group : HGROUPS) * 1024);
        }
    };
    // ---- weight DMA: load j fills rows [32j + 8*wave, +8) of the stage; lane -> row +lane/8, physical slot lane & 7, logical slot swizzled
    const int rb = wave * 8 + (lane >> 3);
    const int ls = (lane & 7) ^ ((rb >> 1) & 7);
    const int64_t b_off0 = (int64_t)(n0 + rb) * g.ldw + ls * 8;
    auto issue_B = [&](int stage, int kw) {
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const bool ok = (n0 + rb + 32 * j) < g.N;
            const f16* src = ok ? Wb + b_off0 + (int64_t)(32 * j) * g.ldw + kw : g.zeros;
            glds16(src, smem + stage * B_BYTES + (32 * j + wave * 8) * 128);
        }
    };

    AccBlock<kL16> acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc_zero(acc[i][j]);

    using FR = Frag<kL16>;
    const int lrow = FR::lrow(lane), lkq = FR::lk(lane);
    int koff[FR::NSTEP];   // 16-byte slot of step st, swizzled with the lane's row key (tile bases are multiples of 16 rows); fragment u: + u * 16 rows
#pragma unroll
    for (int s = 0; s < FR::NSTEP; ++s) koff[s] = (FR::kslot(s, lkq) ^ ((lrow >> 1) & 7)) << 4;
    const int b_lane_off = lrow * 128;
    const int a_pix0 = (wm * 4 + (lrow >> 4)) * HW_ + (lrow & 15);  // halo pixel of this lane's row of A tile 0, fragment 0 at tap (0,0); tile 1: +2 patch rows, fragment u: + u

    // K-tile position: chunk-major (tap inner); kw = element offset inside a weight row [Cout][ky][kx][Cin]
    int ky = 0, kx = 0, chunk = kt_begin / 9;
    if (kt_begin < kt_end) {
        issue_halo(chunk);
        issue_B(0, chunk * BK);
    }
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        // K-tile kt's weights (and, at a chunk boundary, the halo) have landed - this wave's share, then everybody's; every wave is also done
        // with the other weight stage (it finished K-tile kt-1 before arriving here)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int nky = ky, nkx = kx + 1, nchunk = chunk;
        if (nkx == 3) { nkx = 0; if (++nky == 3) { nky = 0; ++nchunk; } }
        const bool more = kt + 1 < kt_end;
        if (more) issue_B(cur ^ 1, (nky * 3 + nkx) * g.cg.Cin + nchunk * BK);
        const char* ha = smem + HALO0;
        const char* fb = smem + cur * B_BYTES + b_lane_off;
        const int pix = a_pix0 + ky * HW_ + kx;
        const int key = (((lrow & 15) + kx) >> 1) & 7;  // swizzle key = halo COLUMN / 2 (see conv3_halo_kernel)
        // fragments double-buffered in registers: the reads of k-step s + 1 are issued before the MFMAs of k-step s (counted lgkmcnt), so a
        // lone wave does not sit out an LDS round trip per k-step while its SIMD partner (the other block's wave) is in a barrier or a wait
        f16x8 af[2][TM][FR::PER], bf[2][TN][FR::PER];
        auto read_frags = [&](int s, int b) {
#pragma unroll
            for (int u = 0; u < FR::PER; ++u) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[b][i][u] = *reinterpret_cast<const f16x8*>(ha + (pix + (i * 2 + u) * HW_) * 128 + ((FR::kslot(s, lkq) ^ key) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[b][j][u] = *reinterpret_cast<const f16x8*>(fb + koff[s] + j * 4096 + u * 2048);
            }
        };
        read_frags(0, 0);
#pragma unroll
        for (int s = 0; s < FR::NSTEP; ++s) {
            if (s + 1 < FR::NSTEP) read_frags(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs (the scheduler otherwise sinks it to save registers)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) mma_step(acc[i][j], bf[s & 1][j], af[s & 1][i]);  // transposed tile (see gemm_epilogue)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more && nchunk != chunk) {
            // chunk boundary: the single halo buffer is refilled once every wave has read the last tap (LDS reads retired, then a barrier);
            // the co-resident block covers the round trip
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue_halo(nchunk);
        }
        ky = nky; kx = nkx; chunk = nchunk;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();  // every wave is done with the operand buffers before the staging image reuses them
    gemm_epilogue<BM, BN, WAVES_M, WAVES_N, epi_wave_rows(BM, BN, WAVES_M, halo4_lds_bytes(BN)), true, true, true,
                  epi16_rows_if_enabled(BM, BN, WAVES_M, halo4_lds_bytes(BN)), false, halo4_lds_bytes(BN)>(g, acc, smem, m0, n0, z, zb, split);
}

__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, int splitk, int M, int N, GemmEpi e) {
    const int CH = (N + 7) / 8;
    const int64_t total = (int64_t)M * CH;
    const bool vec = (N & 3) == 0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / CH);
        const int n = (int)(idx - (int64_t)m * CH) * 8;
        const int nv = (N - n) < 8 ? (N - n) : 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        if (vec && nv == 8) {
            for (int s = 0; s < splitk; ++s) {
                const float4* w = reinterpret_cast<const float4*>(ws + ((int64_t)s * M + m) * N + n);
                const float4 a = w[0], b = w[1];
                v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
                v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
            }
        } else {
            for (int s = 0; s < splitk; ++s) {
                const float* w = ws + ((int64_t)s * M + m) * N + n;
                for (int i = 0; i < nv; ++i) v[i] += w[i];
            }
        }
        if (e.fast && n + 8 <= N) epi_fast8(e, v, m, n, 0);
        else epi_store8(e, v, m, n, N, 0);
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool CONV, bool INTERLEAVE = true>
static int launch_gemm_t(odise_hip_ctx* ctx, GemmArgs& g, int batch) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int lds = plain_lds_bytes(BM, BN, WAVES_M);
    static_assert(epi_lds_bytes(BM, BN, WAVES_M, epi_wave_rows(BM, BN, WAVES_M, lds)) <= lds, "epilogue staging exceeds the LDS request");
    auto kern = gemm_kernel<BM, BN, WAVES_M, WAVES_N, CONV, INTERLEAVE>;
    if (lds > 65536) {
        static LdsAttrOnce once;  // per instantiation; tracked per device inside
        ODISE_TRY(ensure_dyn_lds(ctx, once, (const void*)kern, lds));
    }
    dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)(g.splitk > 1 ? g.splitk : batch));
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, ctx->stream, g);
    ODISE_CHECK_HIP(hipGetLastError());
    if (g.splitk > 1) {
        const int64_t total = (int64_t)g.M * ceil_div(g.N, 8);
        const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, ctx->stream, g.ws, g.splitk, g.M, g.N, g.epi);
        ODISE_CHECK_HIP(hipGetLastError());
    }
    return ODISE_OK;
}

template <int BM, int BN, int WAVES_N, int PT, bool CONV>
static int launch_gemm_pp(odise_hip_ctx* ctx, GemmArgs& g, int batch) {
    constexpr int lds = pp_lds_bytes(BM, BN, 8 / WAVES_N);
    static_assert(epi_lds_bytes(BM, BN, 8 / WAVES_N, epi_wave_rows(BM, BN, 8 / WAVES_N, lds)) <= lds, "epilogue staging exceeds the LDS request");
    auto kern = gemm_pp_kernel<BM, BN, WAVES_N, PT, CONV>;
    static LdsAttrOnce once;  // per instantiation; tracked per device inside
    ODISE_TRY(ensure_dyn_lds(ctx, once, (const void*)kern, lds));
    dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)(g.splitk > 1 ? g.splitk : batch));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, ctx->stream, g);
    ODISE_CHECK_HIP(hipGetLastError());
    if (g.splitk > 1) {
        const int64_t total = (int64_t)g.M * ceil_div(g.N, 8);
        const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, ctx->stream, g.ws, g.splitk, g.M, g.N, g.epi);
        ODISE_CHECK_HIP(hipGetLastError());
    }
    return ODISE_OK;
}

template <int BM, int BN, int WAVES_N, int PT, bool CONV>
static int launch_gemm_pp2(odise_hip_ctx* ctx, GemmArgs& g, int batch) {
    static_assert((BN / WAVES_N / 32) % PT == 0, "whole phases");
    constexpr int lds = pp_lds_bytes(BM, BN, 8 / WAVES_N);
    auto kern = gemm_pp2_kernel<BM, BN, WAVES_N, PT, CONV>;
    static LdsAttrOnce once;  // per instantiation; tracked per device inside
    ODISE_TRY(ensure_dyn_lds(ctx, once, (const void*)kern, lds));
    dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)(g.splitk > 1 ? g.splitk : batch));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, ctx->stream, g);
    ODISE_CHECK_HIP(hipGetLastError());
    if (g.splitk > 1) {
        const int64_t total = (int64_t)g.M * ceil_div(g.N, 8);
        const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, ctx->stream, g.ws, g.splitk, g.M, g.N, g.epi);
        ODISE_CHECK_HIP(hipGetLastError());
    }
    return ODISE_OK;
}

template <int BM, int BN, bool CONV, bool L16>
static int launch_gemm8(odise_hip_ctx* ctx, GemmArgs& g, int batch) {
    constexpr int lds = pp_lds_bytes(BM, BN, BM / 128);
    static_assert(epi_lds_bytes(BM, BN, BM / 128, epi_wave_rows(BM, BN, BM / 128, lds)) <= lds, "epilogue staging exceeds the LDS request");
    auto kern = gemm8_kernel<BM, BN, CONV, L16>;
    static LdsAttrOnce once;  // per instantiation; tracked per device inside
    ODISE_TRY(ensure_dyn_lds(ctx, once, (const void*)kern, lds));
    dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)ceil_div(g.M, BM), (unsigned)(g.splitk > 1 ? g.splitk : batch));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, ctx->stream, g);
    ODISE_CHECK_HIP(hipGetLastError());
    if (g.splitk > 1) {
        const int64_t total = (int64_t)g.M * ceil_div(g.N, 8);
        const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, ctx->stream, g.ws, g.splitk, g.M, g.N, g.epi);
        ODISE_CHECK_HIP(hipGetLastError());
    }
    return ODISE_OK;
}

template <int BN, int PT>
static int launch_conv3_halo(odise_hip_ctx* ctx, GemmArgs& g) {
    constexpr int lds = halo_lds_bytes(BN);
    static_assert(epi_lds_bytes(256, BN, 4, epi_wave_rows(256, BN, 4, lds)) <= lds, "epilogue staging exceeds the LDS request");
    auto kern = conv3_halo_kernel<BN, PT>;
    static LdsAttrOnce once;  // per instantiation; tracked per device inside
    ODISE_TRY(ensure_dyn_lds(ctx, once, (const void*)kern, lds));
    const int n_img = g.M / (g.cg.OH * g.cg.OW);
    dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)(n_img * g.cg.halo_tx * g.cg.halo_ty), (unsigned)(g.splitk > 1 ? g.splitk : 1));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, ctx->stream, g);
    ODISE_CHECK_HIP(hipGetLastError());
    if (g.splitk > 1) {
        const int64_t total = (int64_t)g.M * ceil_div(g.N, 8);
        const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, ctx->stream, g.ws, g.splitk, g.M, g.N, g.epi);
        ODISE_CHECK_HIP(hipGetLastError());
    }
    return ODISE_OK;
}

template <int BN>
static int launch_conv3_halo4(odise_hip_ctx* ctx, GemmArgs& g) {
    constexpr int lds = halo4_lds_bytes(BN);
    static_assert(2 * lds <= 160 * 1024, "two blocks must fit a CU's LDS");
    static_assert(epi_lds_bytes(256, BN, 4, epi_wave_rows(256, BN, 4, lds)) <= lds, "epilogue staging exceeds the LDS request");
    auto kern = conv3_halo4_kernel<BN>;
    static LdsAttrOnce once;  // per instantiation; tracked per device inside
    ODISE_TRY(ensure_dyn_lds(ctx, once, (const void*)kern, lds));
    const int n_img = g.M / (g.cg.OH * g.cg.OW);
    dim3 grid((unsigned)ceil_div(g.N, BN), (unsigned)(n_img * g.cg.halo_tx * g.cg.halo_ty), (unsigned)(g.splitk > 1 ? g.splitk : 1));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, g);
    ODISE_CHECK_HIP(hipGetLastError());
    if (g.splitk > 1) {
        const int64_t total = (int64_t)g.M * ceil_div(g.N, 8);
        const int blocks = (int)std::min<int64_t>(ceil_div(total, 256), 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, ctx->stream, g.ws, g.splitk, g.M, g.N, g.epi);
        ODISE_CHECK_HIP(hipGetLastError());
    }
    return ODISE_OK;
}

static int g_gemm_debug = 0;  // see GemmArgs::dbg
static int g_epi_old = 0;     // tools only: 1 = keep the fp32-staged epilogue (odise_hip_gemm_debug bit 1 << 24), for same-process A/B runs
static int g_conv_flags = 0;  // tools only: 1 = tap-major K order even when Cin % 64 == 0, 2 = never use the ping-pong kernel, 4 = one N-tile per phase at BN = 256

// Tile ids: 0:128x128 1:64x128 2:64x64 (4 waves)  3:256x320 4:256x256 5:256x128 (8 waves)  6:512x128 (8 waves, ping-pong only)
//           7: 16x16-pixel patch x 256 channels, 8: 16x16-pixel patch x 128 channels (conv3_halo_kernel: 3x3 / stride 1 / pad 1 only)
//           9: 16x16-pixel patch x 128 channels, 4 waves, two blocks per CU (conv3_halo4_kernel)
// (The dense counterpart of tile 9 - four waves on 128x256 / 256x128 with a three-slot ring of 32-deep K-tiles, two blocks per CU - was built,
//  bit-identical, and measured 5-20 % SLOWER than the 8-wave ping-pong kernels on 19 of the step's 22 dense shapes, never more than 9 % faster
//  (profiles/r03_gemm4_two_blocks_dense.txt): two operand streams through LDS-DMA at half the tile size cost more than the overlap of
//  neighbouring blocks returns, where the convolution's input patch is fetched once for nine K-tiles.  Not kept.)
static const int kNumTiles = 10;
static const int kTileBM[kNumTiles] = {128, 64, 64, 256, 256, 256, 512, 256, 256, 256};
static const int kTileBN[kNumTiles] = {128, 128, 64, 320, 256, 128, 128, 256, 128, 128};

// Tile / split-K selection by a small cost model (times in microseconds, calibrated on MI355X with tools/gemm_bench.py):
//   t = rounds * (k_tiles_per_split * t_ktile + t_fixed) + t_reduce,   rounds = ceil(blocks * split / resident slots)
// t_ktile is the measured steady-state time of one 64-deep K-tile of a resident block, t_fixed the prologue + LDS-staged
// epilogue, t_reduce the fp32 partial write + read of split-K.  Large tiles win whenever they can fill the CUs; split-K
// recovers parallelism for the small-M (weight-streaming) layers; short-K layers avoid split-K because the partials
// would cost more than the idle CUs.
struct TileCost {
    double t_ktile, t_fixed;
    int slots_per_cu;
};
static const TileCost kTileCost[kNumTiles] = {
    // fitted on MI355X with tools/tile_calib.py (M = 131072, N = 512 | 640, K = 320 / 1152 / 4096, all CUs busy)
    {1.45, 6.0, 2},   // 128x128
    {1.25, 3.5, 3},   // 64x128
    {1.06, 2.6, 4},   // 64x64
    {2.65, 15.0, 1},  // 256x320 (plain kernel: K % 64 != 0, ragged conv channels, fused upsample)
    {2.10, 15.0, 1},  // 256x256 (plain kernel)
    {1.42, 6.0, 1},   // 256x128
    {2.25, 12.0, 1},  // 512x128 (ping-pong kernel only)
    {1.47, 22.5, 1},  // halo 256 pixels x 256 channels (round 5, 16x16x32 MFMAs + wave-private epilogue: the dominant convolution 1027 us = 8 rounds of 72
                      // K-tiles, 256 -> 256 at 256^2 1208 us = 16 rounds of 36 - 2-5 % ahead of every other tile on all convolutions with 256 or more
                      // output channels, profiles/r05_conv_tiles_final.txt)
    {1.12, 9.0, 1},   // halo 256 pixels x 128 channels (re-fitted in round 2: 28 us per block of 18 K-tiles on the 128-channel VAE level, where
                      // the im2col 512x128 tile takes 65 us per block of twice the size: 1.80 vs 2.07 ms per launch, tools/halo512_probe.py)
    {1.70, 12.0, 2},  // the same tile as four waves, two blocks per CU (round 5 refit: 128 -> 128 at 512^2 1365 us = 32 rounds of 18; only offered for
                      // N <= 128 now - with more column blocks every block re-fetches the patch and the 256-channel halo tile is 4-8 % ahead)
};
// (Round 3: on launches of at least two full rounds a refit of this tile, {1.78, 12.0}, also takes the 256- / 512-channel VAE layers from the
// 256-channel halo tile - isolated they run 2-6 % faster on it although every patch's halo is then fetched by Cout / 128 column blocks
// (tools/conv_tiles.py) - but the two-lane step did not get faster, 87.0 / 87.4 ms against 86.9 / 86.8 ms: not adopted.)
// the 256-row tiles as run by the ping-pong kernel (K % 64 == 0; conv: Cin % 64 == 0, no fused upsample)
static const TileCost kTileCostPP[2] = {
    {2.00, 16.0, 1},  // 256x320 (round 5 refit: 9344x4096x1024 93 us = 1.9 rounds of 16, conv 320 -> 320 at 64^2 108 us = one round of 45)
    {1.56, 10.0, 1},  // 256x256 (round 5 refit, 16x16x32 MFMAs + wave-private epilogue: 4096^3 110 us = one round of 64, 9472x4096x1024 97 us = 2.3 rounds of 16)
};
// the 512x128 tile on implicit-GEMM convolutions: every input pixel passes the LDS-DMA path nine times (measured 61-65 us per block of
// 18 K-tiles on the 128-channel VAE level; the dense fit above says 52 us).  Still ahead of the plain 256x128 tile on the stride-2 conv
// of that level (622 vs 651 us), behind the halo tile on the stride-1 ones.
static const TileCost kTileCostConv512 = {2.5, 16.0, 1};
// the 256x256 ping-pong tile on implicit-GEMM convolutions (round 5: the dominant convolution 1063 us = 8 rounds of 72, 256 -> 256 at 256^2 1267 us = 16 of 36)
static const TileCost kTileCostPPConv256 = {1.49, 25.5, 1};
// the 256x256 tile as run by the 8-phase kernel on v_mfma_f32_16x16x32_f16 (round 5; fitted on tools/g8_shapes.py, profiles/r05_gemm8_by_shape.txt:
// 9344x1024x4096 93.5 us = one 58 % round of 64 K-tiles, 9344x4096x1024 122.7 us = 2.3 rounds of 16, the dominant convolution 989 us = 8 rounds of 72)
static const TileCost kTileCost8 = {1.21, 24.0, 1};
// ... and on implicit-GEMM convolutions (every input pixel passes the LDS-DMA path nine times; the dominant convolution 1037 us = 8 rounds of 72,
// 256 -> 256 at 256^2 1263 us = 16 rounds of 36: profiles/r05_conv_tiles_l16.txt)
static const TileCost kTileCost8Conv = {1.41, 28.0, 1};
// previous fit (before the lean epilogue / ping-pong kernel), kept selectable for A/B runs: ODISE_GEMM_FLAGS=8
static const TileCost kTileCostOld[kNumTiles] = {{1.68, 9.5, 2}, {1.58, 4.1, 3}, {1.28, 2.4, 4}, {3.04, 29.0, 1}, {2.58, 22.0, 1},
                                                 {1.75, 10.6, 1}, {2.25, 12.0, 1}, {1.92, 12.0, 1}, {1.32, 10.0, 1}, {2.64, 10.0, 2}};
static thread_local int g_last_tile = -1;   // odise_hip_last_tile: tile | split-K << 8 of this thread's last GEMM / conv launch
static int env_gemm_flags() {
#ifdef ODISE_TOOLS
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ODISE_GEMM_FLAGS");  // developer switch: same bits as odise_hip_gemm_debug(flags) >> 4
        v = e ? atoi(e) : 0;
    }
    return v;
#else
    return 0;
#endif
}

template <bool CONV>
static int launch_gemm_select(odise_hip_ctx* ctx, GemmArgs& g, int batch, int force_tile, int force_split, unsigned tile_mask);

template <bool CONV>
static int launch_gemm(odise_hip_ctx* ctx, GemmArgs& g, int batch, int force_tile, int force_split, unsigned tile_mask = ~0u) {
    LaunchProbe* pr = probe_match(ctx, CONV, g.M, g.N, g.K);
    if (pr && CONV && g.cg.ups) pr = nullptr;   // a convolution with the fused nearest-2x upsample has the same (M, N, K) and runs another kernel: not the probed shape
    if (pr) {   // a measurement hook must never fail a model call: not inside a stream capture (event timing of captured nodes is invalid) ...
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(ctx->stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) pr = nullptr;
    }
    if (!pr) return launch_gemm_select<CONV>(ctx, g, batch, force_tile, force_split, tile_mask);
    const int i = pr->n;
    const bool started = hipEventRecord(pr->ev[2 * i], ctx->stream) == hipSuccess;
    const int rc = launch_gemm_select<CONV>(ctx, g, batch, force_tile, force_split, tile_mask);
    if (started && hipEventRecord(pr->ev[2 * i + 1], ctx->stream) == hipSuccess) pr->n = i + 1;   // ... counted only when both records succeeded
    else { (void)hipGetLastError(); pr->armed = false; }                                           // ... and disarmed, not fatal, when one did not
    return rc;
}

static int flags_early() { return g_conv_flags | env_gemm_flags(); }   // (ODISE_GEMM_FLAGS 131072: no halo tile for the fused-upsample convolutions, A/B)
template <bool CONV>
static int launch_gemm_select(odise_hip_ctx* ctx, GemmArgs& g, int batch, int force_tile, int force_split, unsigned tile_mask) {
    const int64_t cus = ctx->cu_count;
    const int nk = (int)ceil_div(g.K, 64);
    // the halo kernel owns 16x16 output patches: 3x3 / stride 1 / pad 1 convs over whole 64-channel chunks
    bool halo_ok = false;
    int64_t halo_patches = 0;
    if (CONV) {
        // (round 6: also with the fused nearest-2x upsample - the halo patch is gathered from the half-size source, everything after it is the
        // plain kernel; the decoder's 512 -> 512 upsampling convolution ran 2.4 ms on the un-pipelined implicit GEMM, the UNet's three likewise)
        const int up = g.cg.ups ? 1 : 0;
        halo_ok = g.cg.KH == 3 && g.cg.KW == 3 && g.cg.stride == 1 && g.cg.pad_t == 1 && g.cg.pad_l == 1 && g.cg.Cin % 64 == 0 &&
                  g.cg.OH == (g.cg.H << up) && g.cg.OW == (g.cg.W << up) && batch == 1 && !(up && (flags_early() & 131072));
        g.cg.halo_tx = (int)ceil_div(g.cg.OW, 16);
        g.cg.halo_ty = (int)ceil_div(g.cg.OH, 16);
        halo_patches = (int64_t)(g.M / (g.cg.OH * g.cg.OW)) * g.cg.halo_tx * g.cg.halo_ty;
    }
    auto blocks = [&](int t) {
        return ((t >= 7 && t <= 9) ? halo_patches : ceil_div(g.M, kTileBM[t])) * ceil_div(g.N, kTileBN[t]) * (int64_t)batch;
    };
    const bool no_interleave = force_tile >= 16;  // test hook: tile + 16 selects the non-interleaved issue order
    if (no_interleave) force_tile -= 16;
    int tile = 2, best_split = 1;
    double best = 1e30;
    const int flags = g_conv_flags | env_gemm_flags();
    const bool pp_base = !no_interleave && !(flags & 2) && g.K % 64 == 0 && (!CONV || g.cg.Cin % 64 == 0);   // what the halo tiles need (they gather the fused upsample themselves)
    const bool pp_ok = pp_base && !(CONV && g.cg.ups);
    // the 8-phase kernels (gemm8_kernel) run the 256x256 / 512x128 tiles wherever the ping-pong kernels could (the convolution form keeps 32-bit
    // element offsets) - on request only, ODISE_GEMM_FLAGS 16384: built as the guide's yardstick schedule and kept for A/B runs; once every main loop
    // multiplied with 16x16x32 MFMAs and wrote its tile through the wave-private epilogue, the ping-pong kernels measured 3-7 % ahead of it on the
    // same tiles (tools/g8_ablate.py: main loops 72.7 vs 76.3 us at 9472x4096x1024, full kernels 97 vs 104; profiles/r05_epilogue_forms.txt)
    const bool g8_ok = pp_ok && (flags & 16384) && !(flags & (4096 | 512 | 1024)) &&
                       (!CONV || (int64_t)(g.M / (g.cg.OH * g.cg.OW)) * g.cg.H * g.cg.W * g.cg.Cin < ((int64_t)1 << 31));
    for (int t = 0; t < kNumTiles; ++t) {
        if (force_tile >= 0 && force_tile < kNumTiles && t != force_tile) continue;
        if (!((tile_mask >> t) & 1)) continue;
        if (t == 6 && (!pp_ok || (flags & 16))) continue;
        if (t >= 7 && t <= 9 && (!halo_ok || !pp_base || (flags & 64))) continue;  // ODISE_GEMM_FLAGS=64: never use the halo kernels
        if (t == 9 && (flags & 2048)) continue;                        // ODISE_GEMM_FLAGS=2048: never use the two-blocks-per-CU halo kernel
        if (t == 9 && force_tile < 0 && g.N > 128) continue;           // (see kTileCost[9])
        const TileCost& tc = (flags & 8) ? kTileCostOld[t] : (g8_ok && t == 4) ? (CONV ? kTileCost8Conv : kTileCost8) : (CONV && pp_ok && t == 4) ? kTileCostPPConv256 :
                             (pp_ok && (t == 3 || t == 4)) ? kTileCostPP[t - 3] :
                             (CONV && t == 6) ? kTileCostConv512 : kTileCost[t];
        if (force_tile < 0) {
            if (kTileBM[t] > 64 && g.M <= kTileBM[t] / 2) continue;            // mostly-empty row tiles
            if (kTileBN[t] > 64 && g.N <= kTileBN[t] / 2 && t != 2) continue;  // mostly-empty column tiles
        }
        const int64_t nb = blocks(t);
        const int64_t slots = cus * tc.slots_per_cu;
        const int max_split = (batch == 1) ? std::max(1, std::min(nk / 2, 32)) : 1;
        for (int sp = 1; sp <= max_split; ++sp) {
            if (force_split > 0 && sp != std::min(force_split, std::max(1, nk))) continue;
            if (sp > 1 && (size_t)sp * g.M * g.N * sizeof(float) > ctx->ws_bytes) break;
            // the halo kernel splits K in whole 64-channel chunks (9 K-tiles each)
            const int per = (t >= 7 && t <= 9) ? (int)ceil_div(nk / 9, sp) * 9 : (int)ceil_div(nk, sp);
            const int eff_sp = (int)ceil_div(nk, per);
            // Full residency rounds run at the calibrated K-tile time; the last (or only) partial round still costs most of a
            // block's time: an under-filled chip delivers operands only a little faster per block (measured with
            // tools/shape_sweep.py: ~0.8x at 60 % fill), so idle CUs are better bought back with split-K than left idle.
            const int64_t nblk = nb * eff_sp;
            const int64_t rounds = ceil_div(nblk, slots);
            const double fill_last = (double)(nblk - (rounds - 1) * slots) / (double)slots;
            double t_us = per * tc.t_ktile * ((double)(rounds - 1) + 0.55 + 0.45 * fill_last) + (double)rounds * tc.t_fixed;
            // split-K partials: write + read fp32 per split, mostly Infinity-Cache resident at these sizes
            if (eff_sp > 1) t_us += 4.0 + ((2.0 * eff_sp * 4.0 + 2.0) * (double)g.M * g.N) / 6.0e6;
            if (t_us < best) { best = t_us; tile = t; best_split = eff_sp; }
        }
    }
    if (force_tile >= 0 && force_tile < kNumTiles && !(force_tile == 6 && !pp_ok) && !(force_tile >= 7 && force_tile <= 9 && !(halo_ok && pp_base)))
        tile = force_tile;
    g.splitk = 1;
    g.ktiles_per_split = nk;
    if (force_split > 0 && batch == 1) best_split = std::min(force_split, nk);
    if (best_split > 1 && batch == 1) {
        g.ktiles_per_split = (tile >= 7 && tile <= 9) ? (int)ceil_div(nk / 9, best_split) * 9 : (int)ceil_div(nk, best_split);
        g.splitk = (int)ceil_div(nk, g.ktiles_per_split);
        ODISE_REQUIRE((size_t)g.splitk * g.M * g.N * sizeof(float) <= ctx->ws_bytes, "gemm: split-K workspace too small");
    }
    g_last_tile = tile | (best_split << 8);
    if (ctx->launch_log) launch_log_push(ctx, LaunchRec{(int)CONV, g.M, g.N, g.K, tile, g.splitk});
    if (flags & 32) {  // ODISE_GEMM_FLAGS=32: log every launch (tools/gemm_eff.py joins the log with a rocprofv3 kernel trace: per-shape TFLOP/s)
        fprintf(stderr, "GEMMLOG conv=%d M=%d N=%d K=%d batch=%d cin=%d kh=%d h=%d w=%d stride=%d ups=%d tile=%d split=%d pp=%d\n", (int)CONV, g.M, g.N,
                g.K, batch, g.cg.Cin, g.cg.KH, g.cg.H, g.cg.W, g.cg.stride, g.cg.ups, tile, best_split, (int)pp_ok);
    }
    g.ws = (float*)ctx->ws;
    {
        // epi_fast8 preconditions: every vector access of a full 8-column chunk is naturally aligned
        const GemmEpi& e = g.epi;
        const int esz = e.c_dtype == ODISE_F16 ? 2 : 4;
        const int out_vec = (e.geglu ? 4 : 8) * esz;  // bytes stored per chunk
        auto aligned = [](const void* p, int64_t ld_elems, int64_t stride_elems, int elem, int bytes) {
            return ((uintptr_t)p % bytes) == 0 && (ld_elems * elem) % bytes == 0 && (stride_elems * elem) % bytes == 0;
        };
        bool ok = aligned(e.C, e.ldc, e.strideC, esz, out_vec);
        ok = ok && (!e.residual || (!e.geglu && aligned(e.residual, e.ldr, e.strideR, 2, 16)));
        ok = ok && (!e.bias_n || ((uintptr_t)e.bias_n & 15) == 0);
        ok = ok && (!e.rowgroup_add || (((uintptr_t)e.rowgroup_add & 15) == 0 && e.ldg % 4 == 0));
        ok = ok && (!e.geglu || e.act == ODISE_ACT_NONE);
        g.epi.fast = ok ? 1 : 0;
        // math-first epilogue: fp16 output in whole, 16-byte aligned 8-column chunks (GEGLU: of the N/2-wide output)
        const int n_out = e.geglu ? g.N / 2 : g.N;
        g.epi.f16path = (ok && e.c_dtype == ODISE_F16 && g.N % 8 == 0 && n_out % 8 == 0 && aligned(e.C, e.ldc, e.strideC, 2, 16) &&
                         (!e.residual || aligned(e.residual, e.ldr, e.strideR, 2, 8))) ? 1 : 0;
    }
    if (g.epi.ln_part || g.epi.ln_final || g.epi.ln_stats_out) {
        // the LayerNorm terms only exist in the math-first epilogue (256x256 / 256x128 tiles); the producer's partial sums are per 128 columns =
        // the wave columns of the 256x256 tile
        ODISE_REQUIRE(!CONV && (tile == 4 || tile == 5) && g.splitk == 1 && g.epi.f16path && !g.epi.geglu && batch == 1 &&
                          (!g.epi.ln_stats_out || (tile == 4 && g.N % kLnPartCols == 0)),
                      "gemm_ln: M=%d N=%d K=%d cannot take the folded-LayerNorm path", g.M, g.N, g.K);
    }
    g.stats_blocks = 0;
    if (g.epi.gn_stats) {
        // fused GroupNorm statistics need: the lean epilogue on whole 8-column chunks, no split-K (the reduce kernel would own the
        // epilogue), a column-chunk count that divides the thread count, and row blocks that never straddle two images
        const int ohw = CONV ? g.cg.OH * g.cg.OW : 0;
        // kernels instantiated with the statistics epilogue: halo tiles, pp2 conv (256x256) and the 512x128 conv tile
        const bool pp2_used = (tile == 4) && pp_ok && (g8_ok || (flags & 512) || (!(flags & 1024) && CONV));   // gemm8_kernel<.., CONV = true> carries the statistics epilogue too
        bool ok = CONV && g.splitk == 1 && g.epi.fast && g.N % 8 == 0 && ((tile >= 7 && tile <= 9) || (tile == 6 && pp_ok && !(flags & 512)) || pp2_used);
        if (ok && tile >= 7) g.stats_blocks = g.cg.halo_tx * g.cg.halo_ty;
        else if (ok && ohw % kTileBM[tile] == 0) g.stats_blocks = ohw / kTileBM[tile];
        else ok = false;
        if (!ok) { g.epi.gn_stats = nullptr; g.stats_blocks = 0; }
    }
    g.zeros = (const f16*)ctx->zeros;
    g.epi_block = (flags & 32768) ? 1 : 0;
#ifdef ODISE_TOOLS
    static const int freeze_k = (getenv("ODISE_GEMM_FREEZE_K") ? 16 : 0) | (getenv("ODISE_EPI_OLD") ? 32 : 0) | (getenv("ODISE_NO_RES_PREFETCH") ? 64 : 0);
    g.dbg = g_gemm_debug | freeze_k | (g_epi_old ? 32 : 0);
#else
    g.dbg = 0;
#endif
    // the 256-row tiles run the ping-pong pipelined kernel whenever its preconditions hold
    if (tile >= 7) {
        g.cg.chunk_major = 1;
        if (tile == 9) return launch_conv3_halo4<128>(ctx, g);
        return tile == 7 ? launch_conv3_halo<256, 2>(ctx, g) : launch_conv3_halo<128, 1>(ctx, g);
    }
    g.cg.halo_tx = g.cg.halo_ty = 0;
    // second-generation ping-pong kernel (fragment reads under the MFMAs): measured +3..18 % on the implicit-GEMM convs and on
    // dense problems that do not fill the chip twice; the large dense GEMMs keep the first generation (-5..12 % there).
    // ODISE_GEMM_FLAGS: 512 forces it, 1024 forbids it.
    if ((tile == 4 || tile == 6) && g8_ok) {   // ODISE_GEMM_FLAGS 8192: the 8-phase 256x256 kernel on the OTHER MFMA shape (32x32x16 in the product build)
        if (tile == 6) return launch_gemm8<512, 128, CONV, kL16>(ctx, g, batch);
        if (flags & 8192) return launch_gemm8<256, 256, CONV, !kL16>(ctx, g, batch);
        return launch_gemm8<256, 256, CONV, kL16>(ctx, g, batch);
    }
    const bool pp2_auto = !(flags & 1024) && ((CONV && tile != 6) || (!CONV && blocks(tile) * (g.splitk > 1 ? g.splitk : 1) <= 2 * cus));
    if ((tile == 3 || tile == 4 || tile == 6) && pp_ok && ((flags & 512) || pp2_auto)) {
        if (tile == 6) return launch_gemm_pp2<512, 128, 1, 2, CONV>(ctx, g, batch);
        if (tile == 3) return launch_gemm_pp2<256, 320, 2, 1, CONV>(ctx, g, batch);
        return launch_gemm_pp2<256, 256, 2, 2, CONV>(ctx, g, batch);
    }
    if ((tile == 3 || tile == 4 || tile == 6) && pp_ok) {
        if (tile == 6) return launch_gemm_pp<512, 128, 1, 2, CONV>(ctx, g, batch);
        if (tile == 3) return launch_gemm_pp<256, 320, 2, 1, CONV>(ctx, g, batch);
        if (flags & 4) return launch_gemm_pp<256, 256, 2, 1, CONV>(ctx, g, batch);
        return launch_gemm_pp<256, 256, 2, 2, CONV>(ctx, g, batch);
    }
    if (no_interleave) {
        if (tile == 3) return launch_gemm_t<256, 320, 4, 2, CONV, false>(ctx, g, batch);
        if (tile == 4) return launch_gemm_t<256, 256, 4, 2, CONV, false>(ctx, g, batch);
        if (tile == 0) return launch_gemm_t<128, 128, 2, 2, CONV, false>(ctx, g, batch);
    }
    switch (tile) {
        case 0: return launch_gemm_t<128, 128, 2, 2, CONV>(ctx, g, batch);
        case 1: return launch_gemm_t<64, 128, 2, 2, CONV>(ctx, g, batch);
        case 3: return launch_gemm_t<256, 320, 4, 2, CONV>(ctx, g, batch);
        case 4: return launch_gemm_t<256, 256, 4, 2, CONV>(ctx, g, batch);
        case 5: return launch_gemm_t<256, 128, 4, 2, CONV>(ctx, g, batch);
        default: return launch_gemm_t<64, 64, 2, 2, CONV>(ctx, g, batch);
    }
}

int conv_forced(odise_hip_ctx* ctx, const odise_conv_desc* d, int force_tile, int force_split, float* gn_stats = nullptr, int* stats_blocks = nullptr);

int gemm_forced(odise_hip_ctx* ctx, const odise_gemm_desc* d, int force_tile, int force_split, const LnEpi* ln) {
    ODISE_REQUIRE(ctx && d, "gemm: null argument");
    ODISE_REQUIRE(d->M >= 0 && d->N >= 0 && d->K > 0, "gemm: bad dims M=%d N=%d K=%d", d->M, d->N, d->K);
    if (d->M == 0 || d->N == 0) return ODISE_OK;
    ODISE_REQUIRE(d->K % 8 == 0, "gemm: K=%d must be a multiple of 8", d->K);
    ODISE_REQUIRE(d->lda % 8 == 0 && d->ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8 (16-byte rows)");
    ODISE_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->W & 15) == 0, "gemm: A/W must be 16-byte aligned");
    ODISE_REQUIRE(d->A && d->W && d->C, "gemm: null device pointer");
    ODISE_REQUIRE(d->c_dtype == ODISE_F16 || d->c_dtype == ODISE_F32, "gemm: bad c_dtype");
    ODISE_REQUIRE(!d->geglu || (d->N % 2 == 0), "gemm: geglu needs even N");
    ODISE_REQUIRE(!d->rowgroup_add || d->rows_per_group > 0, "gemm: rows_per_group must be > 0");
    const int batch = d->batch < 1 ? 1 : d->batch;
    ODISE_REQUIRE(batch == 1 || ((d->strideA % 8 == 0) && (d->strideW % 8 == 0)), "gemm: batch strides must keep 16-byte alignment");
    GemmArgs g;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.A = (const f16*)d->A; g.lda = d->lda; g.strideA = d->strideA;
    g.W = (const f16*)d->W; g.ldw = d->ldw; g.strideW = d->strideW;
    g.epi.C = d->C; g.epi.ldc = d->ldc; g.epi.c_dtype = d->c_dtype;
    g.epi.bias_n = d->bias_n; g.epi.bias_m = d->bias_m; g.epi.scale_m = d->scale_m;
    g.epi.residual = (const f16*)d->residual; g.epi.ldr = d->ldr;
    g.epi.rowgroup_add = d->rowgroup_add; g.epi.rows_per_group = d->rows_per_group > 0 ? d->rows_per_group : 1;
    g.epi.ldg = d->ldg > 0 ? d->ldg : d->N;
    g.epi.act = d->act; g.epi.geglu = d->geglu; g.epi.alpha = d->alpha;
    g.epi.strideC = d->strideC; g.epi.strideR = d->strideR;
    g.epi.gn_stats = nullptr;
    g.epi.ln_part = nullptr; g.epi.ln_P = 0; g.epi.ln_inv_c = 0.f; g.epi.ln_eps = 0.f; g.epi.ln_colsum = nullptr; g.epi.ln_final_out = nullptr;
    g.epi.ln_final = nullptr; g.epi.ln_rowsum = nullptr; g.epi.ln_stats_out = nullptr;
    if (ln) {
        g.epi.ln_part = ln->part; g.epi.ln_P = ln->P; g.epi.ln_inv_c = ln->inv_c; g.epi.ln_eps = ln->eps; g.epi.ln_colsum = ln->colsum;
        g.epi.ln_final_out = ln->final_out; g.epi.ln_final = ln->fin; g.epi.ln_rowsum = ln->rowsum; g.epi.ln_stats_out = ln->stats_out;
        ODISE_REQUIRE((!ln->part || (ln->P > 0 && ln->colsum)) && (!ln->fin || ln->rowsum), "gemm_ln: incomplete LayerNorm terms");
    }
    g.cg = ConvGeom{};
    // folded LayerNorm: no split-K, a tile with the math-first epilogue (the producer of the statistics: the 256x256 one)
    if (ln) return launch_gemm<false>(ctx, g, batch, force_tile, 1, ln->stats_out ? (1u << 4) : (1u << 4) | (1u << 5));
    return launch_gemm<false>(ctx, g, batch, force_tile, force_split);
}

int gemm_ln(odise_hip_ctx* ctx, const odise_gemm_desc* d, const LnEpi& ln) { return gemm_forced(ctx, d, -1, 1, &ln); }

int conv_forced(odise_hip_ctx* ctx, const odise_conv_desc* d, int force_tile, int force_split, float* gn_stats, int* stats_blocks) {
    ODISE_REQUIRE(ctx && d, "conv2d: null argument");
    ODISE_REQUIRE(d->N >= 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv2d: bad dims");
    ODISE_REQUIRE(d->Cin % 8 == 0, "conv2d: Cin=%d must be a multiple of 8 (pad the input channels)", d->Cin);
    ODISE_REQUIRE(d->KH >= 1 && d->KW >= 1 && d->stride >= 1 && d->OH > 0 && d->OW > 0, "conv2d: bad kernel geometry");
    ODISE_REQUIRE(d->X && d->Wt && d->Y, "conv2d: null device pointer");
    ODISE_REQUIRE(((uintptr_t)d->X & 15) == 0 && ((uintptr_t)d->Wt & 15) == 0, "conv2d: X/Wt must be 16-byte aligned");
    if (d->N == 0) return ODISE_OK;
    // the 8-channel 3x3 convolution (AutoencoderKL's conv_in) has a kernel of its own (conv_c8.hip); a forced tile / split keeps the implicit GEMM
    if (force_tile < 0 && force_split <= 0 && conv3_c8_ok(d)) return launch_conv3_c8(ctx, d, gn_stats, stats_blocks);
    GemmArgs g;
    g.M = d->N * d->OH * d->OW;
    g.N = d->Cout;
    g.K = d->KH * d->KW * d->Cin;
    g.A = (const f16*)d->X; g.lda = 0; g.strideA = 0;
    g.W = (const f16*)d->Wt; g.ldw = g.K; g.strideW = 0;
    g.epi.C = d->Y; g.epi.ldc = d->Cout; g.epi.c_dtype = d->y_dtype;
    g.epi.bias_n = d->bias; g.epi.bias_m = nullptr; g.epi.scale_m = nullptr;
    g.epi.residual = (const f16*)d->residual; g.epi.ldr = d->Cout;
    g.epi.rowgroup_add = d->per_image_add; g.epi.rows_per_group = d->OH * d->OW;
    g.epi.ldg = d->per_image_add_ld > 0 ? d->per_image_add_ld : d->Cout;
    g.epi.act = d->act; g.epi.geglu = 0; g.epi.alpha = 1.0f;
    g.epi.strideC = 0; g.epi.strideR = 0;
    g.epi.gn_stats = gn_stats;
    g.epi.ln_part = nullptr; g.epi.ln_P = 0; g.epi.ln_inv_c = 0.f; g.epi.ln_eps = 0.f; g.epi.ln_colsum = nullptr; g.epi.ln_final_out = nullptr;
    g.epi.ln_final = nullptr; g.epi.ln_rowsum = nullptr; g.epi.ln_stats_out = nullptr;
    if (stats_blocks) *stats_blocks = 0;
    g.cg.H = d->H; g.cg.W = d->W; g.cg.Cin = d->Cin; g.cg.KH = d->KH; g.cg.KW = d->KW;
    g.cg.stride = d->stride; g.cg.pad_t = d->pad_t; g.cg.pad_l = d->pad_l; g.cg.OH = d->OH; g.cg.OW = d->OW;
    g.cg.ups = d->upsample2x;
    // chunk-major K order pays when the input stays in the 256 MiB Infinity Cache (shifted tap re-reads hit L2: +4..10 % on the
    // 64x64 / 32x32 latents); a streamed input prefers tap-major (whole 1-2 KiB pixel vectors in DRAM-page order: +10 % at 128^2+)
    const int64_t in_bytes = (int64_t)d->N * d->H * d->W * d->Cin * 2;
    g.cg.chunk_major = (d->Cin % 64 == 0 && d->KH * d->KW > 1 && in_bytes <= (128ll << 20) && !((g_conv_flags | env_gemm_flags()) & 1)) ? 1 : 0;
    // a 1x1 stride-1 unpadded conv is a plain GEMM over pixels
    if (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0 && !d->upsample2x && d->OH == d->H &&
        d->OW == d->W) {
        g.lda = d->Cin;
        return launch_gemm<false>(ctx, g, 1, force_tile, force_split);  // (statistics fusion is declined there: *stats_blocks stays 0)
    }
    const int rc = launch_gemm<true>(ctx, g, 1, force_tile, force_split);
    if (stats_blocks) *stats_blocks = (rc == ODISE_OK) ? g.stats_blocks : 0;
    return rc;
}

// norm.hip
int group_norm_from_colpart(odise_hip_ctx* ctx, const void* x, void* y, const float* gamma, const float* beta, int N, int HW, int C, int groups,
                            float eps, int act, const float* colpart, int nblk);

}  // namespace odise

extern "C" int odise_hip_gemm_debug(int flags) {
    odise::g_gemm_debug = flags & 15;
    odise::g_conv_flags = (flags >> 4) & 0xfffff;
    odise::g_epi_old = (flags >> 24) & 1;
    return 0;
}
extern "C" int odise_hip_gemm(odise_hip_ctx* ctx, const odise_gemm_desc* d) { return odise::gemm_forced(ctx, d, -1, 0); }
extern "C" int odise_hip_conv2d(odise_hip_ctx* ctx, const odise_conv_desc* d) { return odise::conv_forced(ctx, d, -1, 0); }
// test hooks: force a tile shape (0:128x128, 1:64x128, 2:64x64) and/or a split-K factor
extern "C" int odise_hip_gemm_forced(odise_hip_ctx* ctx, const odise_gemm_desc* d, int tile, int splitk) {
    return odise::gemm_forced(ctx, d, tile, splitk);
}
extern "C" int odise_hip_last_tile(void) { return odise::g_last_tile; }
extern "C" int odise_hip_gemm_ln(odise_hip_ctx* ctx, const odise_gemm_desc* d, const float* part, int parts, float inv_c, float eps,
                                 const float* colsum, float* final_out, const float* fin, const float* rowsum, float* stats_out) {
    odise::LnEpi ln;
    ln.part = part; ln.P = parts; ln.inv_c = inv_c; ln.eps = eps; ln.colsum = colsum; ln.final_out = final_out;
    ln.fin = fin; ln.rowsum = rowsum; ln.stats_out = stats_out;
    return odise::gemm_ln(ctx, d, ln);
}
extern "C" int odise_hip_conv2d_forced(odise_hip_ctx* ctx, const odise_conv_desc* d, int tile, int splitk) {
    return odise::conv_forced(ctx, d, tile, splitk);
}
// conv (forced tile) whose epilogue also reduces the GroupNorm statistics, then the GroupNorm that consumes them: the pair the VAE / UNet
// ResBlocks run (engine.cpp Exec::conv -> Exec::group_norm).  *stats_blocks = 0 reports that the chosen kernel declined the fusion and
// the stand-alone GroupNorm ran instead.  stats_scratch: N * ceil(OH*OW / 64) * Cout * 2 floats.
extern "C" int odise_hip_conv2d_gn_forced(odise_hip_ctx* ctx, const odise_conv_desc* d, int tile, int splitk, const float* gamma, const float* beta,
                                          int groups, float eps, int act, void* y_norm, float* stats_scratch, int* stats_blocks) {
    ODISE_REQUIRE(ctx && d && gamma && beta && y_norm && stats_scratch && stats_blocks, "conv2d_gn_forced: null argument");
    ODISE_REQUIRE(d->y_dtype == ODISE_F16, "conv2d_gn_forced: f16 output only");
    if (int rc = odise::conv_forced(ctx, d, tile, splitk, stats_scratch, stats_blocks)) return rc;
    if (*stats_blocks > 0)
        return odise::group_norm_from_colpart(ctx, d->Y, y_norm, gamma, beta, d->N, d->OH * d->OW, d->Cout, groups, eps, act, stats_scratch, *stats_blocks);
    return odise_hip_group_norm(ctx, d->Y, y_norm, gamma, beta, d->N, d->OH * d->OW, d->Cout, groups, eps, act);
}
