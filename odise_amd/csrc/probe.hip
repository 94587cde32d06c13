// probe.hip — MFMA fragment-layout probe (test hook).  Runs v_mfma_f32_32x32x16_f16 on one-hot style operands and
// returns the raw per-lane accumulator registers so a test can verify the lane->(row,col) maps every kernel here
// relies on:  A: lane l holds A[l&31][8*(l>>5)+e],  B: lane l holds B[8*(l>>5)+e][l&31],
//             C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
#include "common.h"

namespace odise {
__global__ void mfma_probe_kernel(float* out) {
    const int lane = threadIdx.x;
    const int hi = lane >> 5, l31 = lane & 31;
    f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 c;
    // test 0: A[i][0] = i+1, B[0][j] = 1  ->  C[i][j] = i+1 (reveals the row of each accumulator register)
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    if (hi == 0) { a[0] = (f16)(float)(l31 + 1); b[0] = (f16)1.f; }
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[(0 * 64 + lane) * 16 + r] = c[r];
    // test 1: A[i][0] = 1, B[0][j] = j+1  ->  C[i][j] = j+1 (reveals the column)
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    a[0] = (f16)0.f; b[0] = (f16)0.f;
    if (hi == 0) { a[0] = (f16)1.f; b[0] = (f16)(float)(l31 + 1); }
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[(1 * 64 + lane) * 16 + r] = c[r];
    // test 2: k pairing: A[i][k] = (k == 8*hi+e ? 2^e : 0) only in lane group hi, B[k][j] = k+1 in the assumed k order.
    // C[i][j] = sum_k A[i][k] B[k][j] = sum_e 2^e * (8*hi_a + e + 1) summed over both lane groups
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    for (int e = 0; e < 8; ++e) { a[e] = (f16)(float)(1 << e); b[e] = (f16)(float)(8 * hi + e + 1); }
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[(2 * 64 + lane) * 16 + r] = c[r];
}
}  // namespace odise

extern "C" int odise_hip_mfma_probe(odise_hip_ctx* ctx, float* host_out /* [3][64][16] */) {
    using namespace odise;
    ODISE_REQUIRE(ctx && host_out, "mfma_probe: null argument");
    float* d = nullptr;
    ODISE_CHECK_HIP(hipMalloc((void**)&d, 3 * 64 * 16 * sizeof(float)));
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, ctx->stream, d);
    ODISE_CHECK_HIP(hipGetLastError());
    ODISE_CHECK_HIP(hipMemcpyAsync(host_out, d, 3 * 64 * 16 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    ODISE_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ODISE_CHECK_HIP(hipFree(d));
    return ODISE_OK;
}

#ifdef ODISE_TOOLS   // the rate probes below exist only in the measurement build (libodise_hip_tools.so; include/odise_hip_lab.h)
// ---- MFMA issue-rate probe (tools/mfma_rate.py): what the matrix pipes sustain on this part without any operand traffic -----------
// Every wave runs `iters` rounds of 16 v_mfma_f32_32x32x16_f16 over CHAINS independent accumulator tiles.  MODE selects the
// synchronisation skeleton around each round: 0 none (free running), 1 one workgroup barrier per round, 2 the ping-pong skeleton of
// gemm_pp_kernel (two barriers per round, wave group 1 staggered by one barrier), 3 one barrier per two rounds.
namespace odise {
// RANDOM: four distinct A and B fragments of pseudo-random fp16 values in [-1, 1) (what a real GEMM feeds the multipliers: the
// power drawn by the matrix pipes depends on how many operand bits toggle between consecutive instructions) instead of one constant pair.
template <int CHAINS, int MODE, int THREADS, bool RANDOM>
__global__ void __launch_bounds__(THREADS) mfma_rate_kernel(float* out, int iters, float seed, unsigned long long* clocks) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int grp = wave >> 2;
    unsigned long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }
    f16x8 af[4], bf[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (RANDOM) {
                unsigned h = (unsigned)(threadIdx.x * 8 + e) * 2654435761u + (unsigned)q * 40503u + blockIdx.x * 97u;
                h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                af[q][e] = (f16)(seed * ((float)(h & 0xffff) / 32768.f - 1.f));
                bf[q][e] = (f16)(seed * ((float)(h >> 16) / 32768.f - 1.f));
            } else {
                af[q][e] = (f16)(seed * (float)((lane + e) & 3));
                bf[q][e] = (f16)(seed * (float)((lane * 3 + e) & 1));
            }
        }
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    if (MODE == 2 && grp == 1) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 16 / CHAINS; ++s)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[RANDOM ? (c & 1) * 2 + (s & 1) : 0], bf[RANDOM ? (c >> 1 & 1) * 2 + (s >> 1 & 1) : 0], acc[c], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (MODE == 1 || MODE == 2 || (MODE == 3 && (it & 1))) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (MODE == 2 && grp == 0) __builtin_amdgcn_s_barrier();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[threadIdx.x] = s;  // keeps the accumulators alive; never true for the seeds used
    if (blockIdx.x == 0 && threadIdx.x == 0) {     // shader cycles and 100 MHz reference ticks of this workgroup: the clock it ran at
        clocks[0] = __builtin_readcyclecounter() - c0;
        clocks[1] = wall_clock64() - r0;
    }
}
}  // namespace odise

// The same probe on v_mfma_f32_16x16x32_f16 (round 5: the MFMA shape of the guide's 8-phase GEMM template): 32 instructions per round over
// 4 * CHAINS independent 16x16 accumulators = the same FLOPs per round as the 16 32x32x16 instructions above.
namespace odise {
template <int CHAINS, int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) mfma_rate16_kernel(float* out, int iters, float seed, unsigned long long* clocks) {
    const int wave = threadIdx.x >> 6;
    const int grp = wave >> 2;
    unsigned long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }
    f16x8 af[4], bf[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned h = (unsigned)(threadIdx.x * 8 + e) * 2654435761u + (unsigned)q * 40503u + blockIdx.x * 97u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            af[q][e] = (f16)(seed * ((float)(h & 0xffff) / 32768.f - 1.f));
            bf[q][e] = (f16)(seed * ((float)(h >> 16) / 32768.f - 1.f));
        }
    constexpr int NA = 4 * CHAINS;
    f32x4 acc[NA];
#pragma unroll
    for (int c = 0; c < NA; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE == 2 && grp == 1) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 32 / NA; ++s)
#pragma unroll
            for (int c = 0; c < NA; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[(c & 1) * 2 + (s & 1)], bf[(c >> 1 & 1) * 2 + (s >> 1 & 1)], acc[c], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (MODE == 2) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    }
    if (MODE == 2 && grp == 0) __builtin_amdgcn_s_barrier();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NA; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[c][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = __builtin_readcyclecounter() - c0;
        clocks[1] = wall_clock64() - r0;
    }
}
}  // namespace odise

// variant: 0 free/2 waves per SIMD/4 chains, 1 free/1 wave per SIMD/4 chains, 2 ping-pong skeleton, 3 barrier per round,
// 4 barrier per two rounds, 5 free/2 waves/8 chains, 6 free/1 wave/8 chains, 7 = 0 with random operands, 8 = 2 with random operands,
// 9 = 6 with random operands; 10 / 11 / 12 = 7 / 8 / 9 on v_mfma_f32_16x16x32_f16 (mfma_rate16_kernel).  Returns the average launch time of `reps` launches and the shader clock (MHz) of the last one.
extern "C" int odise_hip_mfma_rate(odise_hip_ctx* ctx, int variant, int iters, int blocks, int reps, float* ms_out, double* flops_out, double* mhz_out) {
    using namespace odise;
    ODISE_REQUIRE(ctx && ms_out && flops_out && iters > 0 && blocks > 0 && reps > 0, "mfma_rate: bad argument");
    float* d = (float*)ctx->ws;
    unsigned long long* clk = (unsigned long long*)((char*)ctx->ws + 65536);
    int threads = 512;
    auto launch = [&]() {
        switch (variant) {
            case 0: hipLaunchKernelGGL((mfma_rate_kernel<4, 0, 512, false>), dim3(blocks), dim3(512), 0, ctx->stream, d, iters, 0.5f, clk); break;
            case 1: threads = 256; hipLaunchKernelGGL((mfma_rate_kernel<4, 0, 256, false>), dim3(blocks), dim3(256), 0, ctx->stream, d, iters, 0.5f, clk); break;
            case 2: hipLaunchKernelGGL((mfma_rate_kernel<4, 2, 512, false>), dim3(blocks), dim3(512), 0, ctx->stream, d, iters, 0.5f, clk); break;
            case 3: hipLaunchKernelGGL((mfma_rate_kernel<4, 1, 512, false>), dim3(blocks), dim3(512), 0, ctx->stream, d, iters, 0.5f, clk); break;
            case 4: hipLaunchKernelGGL((mfma_rate_kernel<4, 3, 512, false>), dim3(blocks), dim3(512), 0, ctx->stream, d, iters, 0.5f, clk); break;
            case 5: hipLaunchKernelGGL((mfma_rate_kernel<8, 0, 512, false>), dim3(blocks), dim3(512), 0, ctx->stream, d, iters, 0.5f, clk); break;
            case 6: threads = 256; hipLaunchKernelGGL((mfma_rate_kernel<8, 0, 256, false>), dim3(blocks), dim3(256), 0, ctx->stream, d, iters, 0.5f, clk); break;
            case 7: hipLaunchKernelGGL((mfma_rate_kernel<4, 0, 512, true>), dim3(blocks), dim3(512), 0, ctx->stream, d, iters, 0.5f, clk); break;
            case 8: hipLaunchKernelGGL((mfma_rate_kernel<4, 2, 512, true>), dim3(blocks), dim3(512), 0, ctx->stream, d, iters, 0.5f, clk); break;
            case 10: hipLaunchKernelGGL((mfma_rate16_kernel<4, 0, 512>), dim3(blocks), dim3(512), 0, ctx->stream, d, iters, 0.5f, clk); break;
            case 11: hipLaunchKernelGGL((mfma_rate16_kernel<4, 2, 512>), dim3(blocks), dim3(512), 0, ctx->stream, d, iters, 0.5f, clk); break;
            case 12: threads = 256; hipLaunchKernelGGL((mfma_rate16_kernel<8, 0, 256>), dim3(blocks), dim3(256), 0, ctx->stream, d, iters, 0.5f, clk); break;
            default: threads = 256; hipLaunchKernelGGL((mfma_rate_kernel<8, 0, 256, true>), dim3(blocks), dim3(256), 0, ctx->stream, d, iters, 0.5f, clk); break;
        }
    };
    launch();  // warm-up
    ODISE_CHECK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < reps; ++r) launch();
    ODISE_CHECK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    ODISE_CHECK_HIP(hipEventSynchronize(ctx->ev1));
    ODISE_CHECK_HIP(hipGetLastError());
    float ms = 0.f;
    ODISE_CHECK_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *ms_out = ms / reps;
    *flops_out = (double)blocks * (threads / 64) * (double)iters * 16.0 * 32768.0;
    unsigned long long hc[2] = {0, 0};
    ODISE_CHECK_HIP(hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost));
    if (mhz_out) *mhz_out = hc[1] ? 100.0 * (double)hc[0] / (double)hc[1] : 0.0;
    return ODISE_OK;
}

// ---- LDS port probe (EXPERIMENT, tools/lds_rate.py): how fast LDS-DMA data lands in LDS, alone and next to fragment reads ------------------
// One workgroup of 8 waves per CU, 128 KiB of LDS.  Per round every thread issues LOADS global_load_lds_dwordx4 (8 = the 64 KiB per
// K-tile of a 256x256x64 GEMM step) from a 64 KiB source that stays hot in L2, and / or READS ds_read_b128 (24 = the fragment reads of
// that step), then waits for both and passes a barrier.  Reported: clocks per round on the device's own cycle counter.
namespace odise {
template <int LOADS, int READS>
__global__ void __launch_bounds__(512) lds_rate_kernel(const uint4* __restrict__ src, float* out, int rounds, unsigned long long* clocks) {
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long c0 = 0;
    if (blockIdx.x == 0 && tid == 0) c0 = __builtin_readcyclecounter();
    uint4 acc = {0u, 0u, 0u, 0u};
    for (int r = 0; r < rounds; ++r) {
        const int stage = (r & 1) * 65536;
#pragma unroll
        for (int j = 0; j < LOADS; ++j)  // wave-instruction j fills 1 KiB at lsm + stage + (j * 8 + wave) * 1024, lane-linear
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (j * 8 + wave) * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(lsm + stage + (j * 8 + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int k = 0; k < READS; ++k) {  // conflict-free 16-byte reads of the other stage
            const uint4 v = *reinterpret_cast<const uint4*>(lsm + (stage ^ 65536) + ((k * 512 + tid) & 4095) * 16);
            acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[tid] = 1.f;  // keeps the reads alive
    if (blockIdx.x == 0 && tid == 0) clocks[0] = __builtin_readcyclecounter() - c0;
}
}  // namespace odise

// variant: 0 = 8 DMA loads per round, 1 = 24 reads per round, 2 = both, 3 = 4 DMA loads + 24 reads, 4 = 16 DMA loads.  clocks_per_round: device cycles.
extern "C" int odise_hip_lds_rate(odise_hip_ctx* ctx, int variant, int rounds, int blocks, double* clocks_per_round, float* ms_out) {
    using namespace odise;
    ODISE_REQUIRE(ctx && clocks_per_round && ms_out && rounds > 0 && blocks > 0, "lds_rate: bad argument");
    const uint4* src = (const uint4*)ctx->ws;  // 64 KiB .. 128 KiB of whatever the workspace holds
    float* out = (float*)((char*)ctx->ws + (1 << 20));
    unsigned long long* clk = (unsigned long long*)((char*)ctx->ws + (2 << 20));
    const size_t lds = 131072;
    auto launch = [&]() -> hipError_t {
        switch (variant) {
            case 0: { auto k = lds_rate_kernel<8, 0>; (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, ctx->stream, src, out, rounds, clk); break; }
            case 1: { auto k = lds_rate_kernel<0, 24>; (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, ctx->stream, src, out, rounds, clk); break; }
            case 2: { auto k = lds_rate_kernel<8, 24>; (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, ctx->stream, src, out, rounds, clk); break; }
            case 3: { auto k = lds_rate_kernel<4, 24>; (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, ctx->stream, src, out, rounds, clk); break; }
            default: { auto k = lds_rate_kernel<16, 0>; (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, ctx->stream, src, out, rounds, clk); break; }
        }
        return hipGetLastError();
    };
    ODISE_CHECK_HIP(launch());
    ODISE_CHECK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    ODISE_CHECK_HIP(launch());
    ODISE_CHECK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    ODISE_CHECK_HIP(hipEventSynchronize(ctx->ev1));
    ODISE_CHECK_HIP(hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
    unsigned long long hc = 0;
    ODISE_CHECK_HIP(hipMemcpy(&hc, clk, sizeof(hc), hipMemcpyDeviceToHost));
    *clocks_per_round = (double)hc / rounds;
    return ODISE_OK;
}
#endif  // ODISE_TOOLS
