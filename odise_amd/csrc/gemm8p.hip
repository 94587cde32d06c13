// gemm8p.hip — the 256x256 "8-phase" GEMM schedule of /opt/skills/guides/cdna_hip_programming.md section 5 ("The 256^2 8-phase template"),
// written from that section's geometry / phase list / vmcnt rules for fp16 operands, as the yardstick VERDICT r04 asks the repo's own
// main loops (gemm_pp2_kernel, conv3_halo_kernel) to be measured against.  MEASUREMENT BUILD ONLY (-DODISE_TOOLS, libodise_hip_tools.so):
// nothing on the product path calls it; tools/gemm8p_bench.py does.
//
//   C[M,N] (fp16) = A[M,K] * W[N,K]^T, fp32 accumulation;  M % 256 == 0, N % 256 == 0, K % 128 == 0.
//
// Geometry (the guide's table): tile 256x256, BK = 64, 8 waves as 2(M) x 4(N), 128x64 per wave = 8 x 4 tiles of v_mfma_f32_16x16x32_f16
// (64 MFMAs per K-tile and wave), 128 KiB of LDS = 2 K-tile buffers x {A0, A1, B0, B1} half-tiles of 128 rows x 128 B, each half-tile two
// global_load_lds_dwordx4 per thread.  Half-tile X0 / X1 holds, for every wave row (column), the first / second 64 (32) of its 128 (64)
// rows (columns): a phase multiplies one 64x32 quadrant of the wave tile over the whole K-tile (16 MFMAs) and needs at most one new A
// sub-tile (8 ds_read_b128) and one new B sub-tile (4).
//   phase 1: read B0, A0; stage A1(t+1)      -> C[0][0]        phase 3: read A1; stage A0(t+2)   -> C[1][1]
//   phase 2: read B1;     stage B0(t+2)      -> C[0][1]        phase 4: stage B1(t+2); vmcnt(6)  -> C[1][0]
// Each phase: {ds_reads, 2 glds, [counted wait]} s_barrier, lgkmcnt(0), setprio 1, 16 MFMA, setprio 0, s_barrier.  The waves of wave row 1
// run one barrier behind those of wave row 0 (one wave of each group per SIMD): one group multiplies while the other reads and stages.
// Three half-tiles stay in flight across every barrier; the only vector-memory wait of a K-tile is phase 4's vmcnt(6), which retires
// the K-tile read from the NEXT phase on (RAW rule of the guide), and a buffer is restaged two phases after its last read (one phase
// after for B0, whose reads are retired by lgkmcnt(8) before phase 1's first barrier).
// LDS image: rows of 128 B; 16-byte slot s of half-tile row r sits at physical slot s ^ ((r >> 1) & 7) (applied to the DMA SOURCE
// address, the LDS image of one DMA instruction being lane-linear): the 16 lanes of a ds_read_b128 group (16 rows, one k-slot) hit 16
// distinct 16-byte units of the 256-B bank row.
#ifdef ODISE_TOOLS
#include "common.h"
#include <type_traits>

namespace odise {
namespace g8p {

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int HALF = 16 * 1024;       // one half-tile
constexpr int STAGE = 4 * HALF;       // A0, A1, B0, B1
constexpr int OFF_A0 = 0, OFF_A1 = HALF, OFF_B0 = 2 * HALF, OFF_B1 = 3 * HALF;

// VARIANT bit 0: no s_setprio; bit 1: wave groups not staggered (lockstep); bit 2: v_mfma_f32_32x32x16_f16 (4 x 2 tiles per wave, the
// product kernels' shape) instead of 16x16x32 (8 x 4 tiles) in the same schedule: same LDS image, DMA, barriers and read counts
template <int VARIANT>
__global__ void __launch_bounds__(512) gemm8p_kernel(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool PRIO = !(VARIANT & 1), STAGGER = !(VARIANT & 2), M32 = (VARIANT & 4) != 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
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
    const int m0 = by * 256, n0 = bx * 256;
    const int nk = K >> 6;

    // ---- staging: thread fills physical slot (lane & 7) of half-tile row i*64 + wave*8 + lane/8 (i = 0, 1) and fetches logical slot ls
    const int srow = wave * 8 + (lane >> 3);
    const int ls = (lane & 7) ^ ((srow >> 1) & 7);
    // A half h, piece i: row m0 + i*128 + h*64 + srow;  B half h, piece i: row n0 + (i*2 + wave/4)*64 + h*32 + (wave%4)*8 + lane/8
    const f16* a_src = A + (int64_t)(m0 + srow) * K + ls * 8;
    const f16* b_src = W + (int64_t)(n0 + (wave >> 2) * 64 + (wave & 3) * 8 + (lane >> 3)) * K + ls * 8;
    const int64_t a_piece = (int64_t)128 * K, a_half = (int64_t)64 * K;
    const int64_t b_piece = (int64_t)128 * K, b_half = (int64_t)32 * K;
    char* const lds_w = smem + wave * 1024;   // this wave's 8 rows inside a 64-row piece
    auto stage_A = [&](int h, int buf, int kt) {
        const f16* s = a_src + h * a_half + (int64_t)kt * 64;
        char* d = lds_w + buf * STAGE + (h ? OFF_A1 : OFF_A0);
        glds16(s, d);
        glds16(s + a_piece, d + 8192);
    };
    auto stage_B = [&](int h, int buf, int kt) {
        const f16* s = b_src + h * b_half + (int64_t)kt * 64;
        char* d = lds_w + buf * STAGE + (h ? OFF_B1 : OFF_B0);
        glds16(s, d);
        glds16(s + b_piece, d + 8192);
    };

    // ---- fragment addresses: lane reads row (l & 15) of a 16-row tile, logical slot s*4 + (l >> 4) -> physical ^ ((row >> 1) & 7)
    const int l15 = lane & 15, lq = lane >> 4;
    const int l31 = lane & 31, hi = lane >> 5;
    int koff[2], koff32[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) koff[s] = ((s * 4 + lq) ^ ((l15 >> 1) & 7)) << 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) koff32[s] = ((s * 2 + hi) ^ ((l31 >> 1) & 7)) << 4;   // 32x32x16: row l & 31, logical slot 2s + (l >> 5)
    const int a_lane = (wr * 64 + (M32 ? l31 : l15)) * 128;   // + i*2048 per 16-row m-tile (i*4096 per 32-row tile)
    const int b_lane = (wc * 32 + (M32 ? l31 : l15)) * 128;   // + j*2048 per 16-row n-tile

    f32x4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    f16x8 af[4][2], b0f[2][2], b1f[2][2];
    f32x16 acc32[4][2];                     // M32: rows wr*128 + p*32, columns wc*64 + j*32
    f16x8 af32[2][4], b0f32[4], b1f32[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;

    auto read_A = [&](int h, int buf) {
        const char* p = smem + buf * STAGE + (h ? OFF_A1 : OFF_A0) + a_lane;
        if constexpr (M32) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) af32[i][s] = *reinterpret_cast<const f16x8*>(p + i * 4096 + koff32[s]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const f16x8*>(p + i * 2048 + koff[s]);
        }
    };
    auto read_B0 = [&](int buf) {
        const char* p = smem + buf * STAGE + OFF_B0 + b_lane;
        if constexpr (M32) {
#pragma unroll
            for (int s = 0; s < 4; ++s) b0f32[s] = *reinterpret_cast<const f16x8*>(p + koff32[s]);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) b0f[j][s] = *reinterpret_cast<const f16x8*>(p + j * 2048 + koff[s]);
        }
    };
    auto read_B1 = [&](int buf) {
        const char* p = smem + buf * STAGE + OFF_B1 + b_lane;
        if constexpr (M32) {
#pragma unroll
            for (int s = 0; s < 4; ++s) b1f32[s] = *reinterpret_cast<const f16x8*>(p + koff32[s]);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) b1f[j][s] = *reinterpret_cast<const f16x8*>(p + j * 2048 + koff[s]);
        }
    };
    // 16 MFMAs of quadrant (ih, jh); operands swapped (transposed 16x16 result: a lane owns 4 consecutive columns of row l & 15)
#define G8P_MMA(ih, bfr, jh)                                                                                                   \
    do {                                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                               \
        if constexpr (M32) {                                                                                                   \
            _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                      \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                  \
                    acc32[(ih) * 2 + i][jh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bfr##32[s], af32[i][s], acc32[(ih) * 2 + i][jh], 0, 0, 0); \
        } else {                                                                                                               \
            _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                      \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                  \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
                        acc[(ih) * 4 + i][(jh) * 2 + j] =                                                                      \
                            __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j][s], af[i][s], acc[(ih) * 4 + i][(jh) * 2 + j], 0, 0, 0); \
        }                                                                                                                      \
        if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
        __builtin_amdgcn_s_barrier();                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                     \
    } while (0)
#define G8P_BAR()                              \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

    // one K-tile (4 phases) on buffer `buf`; MODE 0: steady state, 1: second-to-last tile (only A1(t+1) left to stage), 2: last tile
    auto ktile = [&](auto bufc, auto modec, int kt) {
        constexpr int buf = decltype(bufc)::value, MODE = decltype(modec)::value;
        // phase 1
        read_B0(buf);
        __builtin_amdgcn_sched_barrier(0);
        read_A(0, buf);
        if (MODE <= 1) stage_A(1, buf ^ 1, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        G8P_BAR();
        G8P_MMA(0, b0f, 0);
        // phase 2
        read_B1(buf);
        if (MODE == 0) stage_B(0, buf, kt + 2);
        G8P_BAR();
        G8P_MMA(0, b1f, 1);
        // phase 3
        read_A(1, buf);
        if (MODE == 0) stage_A(0, buf, kt + 2);
        G8P_BAR();
        G8P_MMA(1, b1f, 1);
        // phase 4
        if (MODE == 0) { stage_B(1, buf, kt + 2); wait_vm<6>(); }
        if (MODE == 1) wait_vm<0>();
        G8P_BAR();
        G8P_MMA(1, b0f, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ---- prologue: tile 0 (B0, A0, B1, A1) and B0, A0, B1 of tile 1
    stage_B(0, 0, 0); stage_A(0, 0, 0); stage_B(1, 0, 0); stage_A(1, 0, 0);
    stage_B(0, 1, 1); stage_A(0, 1, 1); stage_B(1, 1, 1);
    wait_vm<6>();
    G8P_BAR();
    if (STAGGER && wr == 1) G8P_BAR();

    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        ktile(I0{}, I0{}, kt);
        ktile(I1{}, I0{}, kt + 1);   // nk is even: tile kt + 3 exists
    }
    ktile(I0{}, I1{}, kt);
    ktile(I1{}, I2{}, kt + 1);
    if (STAGGER && wr == 0) G8P_BAR();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    G8P_BAR();

    // ---- epilogue: each wave stages its 128x64 tile as fp16 in its own 16 KiB (rows of 128 B, 16-byte slots XOR (row & 7)) and copies it out
    char* est = smem + wave * 16384;
    if constexpr (M32) {   // transposed 32x32 tile: lane owns row l & 31 and, per register quad q, columns 8q + 4(l >> 5) + (0..3)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = p * 32 + l31;
                    const int slot = (j * 4 + q) ^ (row & 7);
                    f16x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (f16)acc32[p][j][4 * q + r];
                    *reinterpret_cast<f16x4*>(est + row * 128 + slot * 16 + hi * 8) = v;
                }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = i * 16 + l15;
                const int slot = (j * 2 + (lq >> 1)) ^ (row & 7);
                f16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (f16)acc[i][j][r];
                *reinterpret_cast<f16x4*>(est + row * 128 + slot * 16 + (lq & 1) * 8) = v;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f16* cb = C + (int64_t)(m0 + wr * 128) * N + n0 + wc * 64 + (lane & 7) * 8;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int row = it * 8 + (lane >> 3);
        const f16x8 v = *reinterpret_cast<const f16x8*>(est + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
        *reinterpret_cast<f16x8*>(cb + (int64_t)row * N) = v;
    }
#undef G8P_MMA
#undef G8P_BAR
}

template <int V>
static int launch(odise_hip_ctx* ctx, const f16* A, const f16* W, f16* C, int M, int N, int K) {
    ODISE_CHECK_HIP(hipFuncSetAttribute((const void*)gemm8p_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    hipLaunchKernelGGL(gemm8p_kernel<V>, dim3(N / 256, M / 256), dim3(512), 2 * STAGE, ctx->stream, A, W, C, M, N, K);
    ODISE_CHECK_HIP(hipGetLastError());
    return ODISE_OK;
}
}  // namespace g8p
}  // namespace odise

extern "C" int odise_hip_gemm8p(odise_hip_ctx* ctx, const void* A, const void* W, void* C, int M, int N, int K, int variant) {
    using namespace odise;
    ODISE_REQUIRE(ctx && A && W && C, "gemm8p: null argument");
    ODISE_REQUIRE(M > 0 && N > 0 && M % 256 == 0 && N % 256 == 0 && K >= 256 && K % 128 == 0, "gemm8p: M, N multiples of 256, K a multiple of 128 (>= 256)");
    switch (variant) {
        case 0: return g8p::launch<0>(ctx, (const f16*)A, (const f16*)W, (f16*)C, M, N, K);
        case 1: return g8p::launch<1>(ctx, (const f16*)A, (const f16*)W, (f16*)C, M, N, K);
        case 2: return g8p::launch<2>(ctx, (const f16*)A, (const f16*)W, (f16*)C, M, N, K);
        case 4: return g8p::launch<4>(ctx, (const f16*)A, (const f16*)W, (f16*)C, M, N, K);
        default: ODISE_REQUIRE(false, "gemm8p: variant 0, 1, 2 or 4"); return ODISE_ERR_ARG;
    }
}
#endif  // ODISE_TOOLS
