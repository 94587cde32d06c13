/*
 * odise_hip_lab.h — hooks that exist ONLY in the measurement build of the library (`python -m odise_amd.build --tools` ->
 * libodise_hip_tools.so, -DODISE_TOOLS).  The product library (libodise_hip.so) does not export them and tests/test_lib_abi.py asserts
 * that; tools/ scripts select the measurement build with ODISE_HIP_LIB.
 */
#ifndef ODISE_HIP_LAB_H
#define ODISE_HIP_LAB_H

#include "odise_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* probe.hip: sustained MFMA rate on register-resident operands, LDS port rates (tools/mfma_rate.py, tools/lds_rate.py) */
int odise_hip_mfma_rate(odise_hip_ctx* ctx, int variant, int iters, int blocks, int reps, float* ms_out, double* flops_out, double* mhz_out);
int odise_hip_lds_rate(odise_hip_ctx* ctx, int variant, int rounds, int blocks, double* clocks_per_round, float* ms_out);

/* gemm8p.hip: the 256x256 8-phase GEMM schedule of /opt/skills/guides/cdna_hip_programming.md section 5, fp16 (the yardstick the
 * repo's main loops are measured against, tools/gemm8p_bench.py).  C[M,N] (f16, ld N) = A[M,K] (ld K) * W[N,K]^T (ld K);
 * M, N multiples of 256, K a multiple of 128.  variant: bit 0 = no s_setprio, bit 1 = wave groups in lockstep */
int odise_hip_gemm8p(odise_hip_ctx* ctx, const void* A, const void* W, void* C, int M, int N, int K, int variant);

#ifdef __cplusplus
}
#endif
#endif /* ODISE_HIP_LAB_H */
