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
