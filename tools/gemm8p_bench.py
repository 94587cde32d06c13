"""A/B of the guide's 256x256 8-phase GEMM schedule (csrc/gemm8p.hip, written from /opt/skills/guides/cdna_hip_programming.md section 5) against
the repo's main loops on the SAME uniform-random [-1, 1) fp16 operands, same process, interleaved rounds (guide section 5.4 rules 24 / 25):

    8p        gemm8p_kernel<0>  (16x16x32 MFMA, 2M x 4N waves, 3 half-tiles of LDS-DMA in flight, vmcnt(6) once per K-tile, setprio, stagger)
    8p-noprio the same without s_setprio;  8p-lock: wave groups in lockstep
    8p-m32    the same schedule, LDS image, DMA and read counts on v_mfma_f32_32x32x16_f16 (4 x 2 tiles per wave: the product kernels' MFMA shape)
    pp2       gemm_pp2_kernel<256,256,2,2>  (the repo's ping-pong loop, 32x32x16 MFMA, fragment reads under the MFMAs; math-first epilogue)
    pp        gemm_pp_kernel<256,256,2,2>

Measurement build only:  ODISE_HIP_LIB=odise_amd/lib/libodise_hip_tools.so python tools/gemm8p_bench.py [zero]
Prints min / median time over the rounds, TFLOP/s of the min, and max |difference| against an fp32 numpy product of a 256-row slice
(transpose-detecting: A and W are independent random matrices)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

ZERO = "zero" in sys.argv[1:]
SHORT = "short" in sys.argv[1:]
ctx = Context(0)
assert hasattr(ctx.lib, "odise_hip_gemm8p"), "needs the measurement build: ODISE_HIP_LIB=odise_amd/lib/libodise_hip_tools.so"
rng = np.random.default_rng(0)
PP2, NO_HALO, PP1 = 512 << 4, 64 << 4, 1024 << 4


def uni(shape):
    if ZERO:
        return ctx.zeros(shape, np.float16)
    return ctx.to_device(rng.uniform(-1.0, 1.0, size=shape).astype(np.float16))


def run8p(A, W, O, M, N, K, v):
    rc = ctx.lib.odise_hip_gemm8p(ctx.h, A, W, O, M, N, K, v)
    assert rc == 0, ctx.lib.odise_hip_last_error()
    return O


def main():
    print(f"# operands: {'ZERO-filled (for the DVFS effect of rule 25 only)' if ZERO else 'uniform random [-1, 1) fp16'}; fp16 output; times in us", flush=True)
    shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (65536, 1024, 4096), (65536, 512, 4608), (9472, 4096, 1024), (9472, 1024, 4096)]
    for (M, N, K) in ([shapes[0], shapes[3], shapes[5]] if SHORT else shapes):
        A, W = uni((M, K)), uni((N, K))
        O = ctx.empty((M, N), np.float16)
        variants = [
            ("8p", lambda: run8p(A, W, O, M, N, K, 0)),
            ("8p-noprio", lambda: run8p(A, W, O, M, N, K, 1)),
            ("8p-lock", lambda: run8p(A, W, O, M, N, K, 2)),
            ("8p-m32", lambda: run8p(A, W, O, M, N, K, 4)),
            ("pp2", lambda: (ctx.lib.odise_hip_gemm_debug(PP2 | NO_HALO), ctx.gemm(A, W, force_tile=4, out=O))[1]),
            ("pp", lambda: (ctx.lib.odise_hip_gemm_debug(PP1 | NO_HALO), ctx.gemm(A, W, force_tile=4, out=O))[1]),
        ]
        flop = 2.0 * M * N * K
        it = max(3, int(2e13 / flop))
        times = {n: [] for n, _ in variants}
        err = {}
        ref = None
        if not ZERO:
            a = A.view((256, K)).numpy().astype(np.float32)
            w = W.numpy().astype(np.float32)
            ref = a @ w.T
        for r in range(6):
            for name, fn in variants:
                fn()
                ctx.sync()
                ctx.timer_start()
                for _ in range(it):
                    fn()
                ms = ctx.timer_stop() / it
                if r > 0:
                    times[name].append(ms)
                if r == 0 and ref is not None:
                    got = O.view((256, N)).numpy().astype(np.float32)
                    err[name] = float(np.abs(got - ref).max() / np.abs(ref).max())
        ctx.lib.odise_hip_gemm_debug(0)
        for name, _ in variants:
            t = np.array(times[name])
            e = f"rel err {err[name]:.2e}" if name in err else ""
            print(f"M={M:6d} N={N:5d} K={K:5d} {name:10s} min {t.min()*1e3:8.1f}  median {np.median(t)*1e3:8.1f}  {flop/(t.min()*1e-3)/1e12:7.1f} TFLOP/s (median {flop/(np.median(t)*1e-3)/1e12:7.1f})  {e}", flush=True)
        A.free(); W.free(); O.free()


if __name__ == "__main__":
    main()
