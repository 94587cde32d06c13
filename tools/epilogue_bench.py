"""EXPERIMENT (branch epilogue-direct): transposed accumulators + direct epilogue (flag 4096) against the LDS-staged epilogue, on
gemm_pp_kernel: results must be bit-identical for every epilogue variant; time per launch by K (the epilogue is 36 % of a K = 640 GEMM)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd._lib import ACT_GELU, ACT_NONE, ACT_SILU  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)
BASE = (512 if os.environ.get("PP2") else 1024) << 4   # gemm_pp_kernel everywhere (PP2=1: gemm_pp2_kernel everywhere)


def dev(a):
    return ctx.to_device(np.ascontiguousarray(a))


def f16(shape, s=1.0):
    return dev((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))


def f32(shape, s=1.0):
    return dev(rng.standard_normal(shape, dtype=np.float32) * s)


def bench(fn, out, label, flop):
    res, t = {}, {}
    for rnd in range(3):
        for name, flags in (("staged", BASE), ("direct", BASE | (4096 << 4))):
            ctx.lib.odise_hip_gemm_debug(flags)
            for _ in range(2):
                fn()
            ctx.sync()
            ctx.timer_start()
            for _ in range(10):
                fn()
            t[name] = min(t.get(name, 1e9), ctx.timer_stop() / 10)
            res[name] = out.numpy().copy()
    ctx.lib.odise_hip_gemm_debug(0)
    d = np.abs(res["direct"].astype(np.float32) - res["staged"].astype(np.float32)).max()
    a, b = t["staged"], t["direct"]
    print(f"{label:58s}: {a*1e3:8.1f} us -> {b*1e3:8.1f} us  ({(a/b-1)*100:+5.1f} %)  {flop/(b*1e-3)/1e12:7.1f} TFLOP/s  max|d|={d:g}", flush=True)


# ---- every epilogue variant once (correctness first) on a ragged M
M, N, K = 9232, 1024, 640
A, W = f16((M, K)), f16((N, K), K ** -0.5)
variants = {
    "plain": {}, "bias_n": dict(bias_n=f32((N,))), "bias_n + SiLU": dict(bias_n=f32((N,)), act=ACT_SILU),
    "bias_n + residual": dict(bias_n=f32((N,)), residual=f16((M, N))), "GELU + residual": dict(act=ACT_GELU, residual=f16((M, N))),
    "GEGLU": dict(bias_n=f32((N,)), geglu=True), "row group + bias": dict(bias_n=f32((N,)), rowgroup_add=f32((-(-M // 577), N)), rows_per_group=577),
    "bias_m + scale_m": dict(bias_m=f32((M,)), scale_m=f32((M,))), "fp32 out": dict(bias_n=f32((N,)), out_dtype=np.float32), "alpha": dict(alpha=0.125),
}
for name, kw in variants.items():
    O = ctx.empty((M, N // 2 if kw.get("geglu") else N), kw.get("out_dtype", np.float16))
    bench(lambda: ctx.gemm(A, W, force_tile=4, force_split=1, out=O, **kw), O, f"{name} (M={M} N={N} K={K})", 2.0 * M * N * K)
O = ctx.empty((M, N), np.float16)
bench(lambda: ctx.gemm(A, W, force_tile=4, force_split=2, out=O), O, "split-K 2 (fp32 partials)", 2.0 * M * N * K)
Ab, Wb, Ob = f16((3, 4096, 320)), f16((3, 768, 320), 320 ** -0.5), ctx.empty((3, 4096, 768), np.float16)
bench(lambda: ctx.gemm(Ab, Wb, force_tile=4, force_split=1, out=Ob), Ob, "batched 3 x (4096 x 768 x 320)", 2.0 * 3 * 4096 * 768 * 320)
# ---- time by K and tile
for (M, N, K, tile) in [(65536, 1024, 320, 4), (65536, 1024, 640, 4), (65536, 1024, 1280, 4), (65536, 1024, 4096, 4), (65536, 1280, 640, 3),
                        (36928, 1024, 1024, 4)]:
    A, W, O = f16((M, K)), f16((N, K), K ** -0.5), ctx.empty((M, N), np.float16)
    b = f32((N,))
    bench(lambda: ctx.gemm(A, W, bias_n=b, force_tile=tile, force_split=1, out=O), O, f"gemm M={M} N={N} K={K} tile {tile}", 2.0 * M * N * K)
    for a in (A, W, O):
        a.free()
for (B, H, Wd, Cin, Cout, tile) in [(16, 64, 64, 320, 320, 3), (16, 128, 128, 256, 256, 4), (4, 512, 512, 128, 128, 6)]:
    X, Wt, O = f16((B, H, Wd, Cin)), f16((Cout, 3, 3, Cin), (9 * Cin) ** -0.5), ctx.empty((B, H, Wd, Cout), np.float16)
    ctx.lib.odise_hip_gemm_debug(BASE | (64 << 4))
    bench(lambda: ctx.conv2d(X, Wt, force_tile=tile, out=O), O, f"conv {B}x{H}x{Wd} {Cin}->{Cout} tile {tile}", 2.0 * B * H * Wd * Cout * 9 * Cin)
