"""A/B of the GEMM epilogue's read phase (unrolled, loads hoisted) against the previous per-row form: run once with ODISE_EPI_OLD=1 and once
without (measurement build: ODISE_HIP_LIB=.../libodise_hip_tools.so); prints time per launch and a checksum of every output - the two runs
must print identical checksums (same operations per element in the same order)."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd._lib import ACT_GELU, ACT_NONE, ACT_QUICKGELU, ACT_SILU  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)
tag = "old" if os.environ.get("ODISE_EPI_OLD") else "new"


def f16(shape, s=1.0):
    return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))


def f32(shape, s=1.0):
    return ctx.to_device(rng.standard_normal(shape, dtype=np.float32) * s)


def run(label, fn, out, flop):
    for _ in range(3):
        fn()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(10):
            fn()
        best = min(best, ctx.timer_stop() / 10)
    h = hashlib.sha1(out.numpy().tobytes()).hexdigest()[:12]
    print(f"{tag} {label:64s} {best*1e3:9.1f} us {flop/(best*1e-3)/1e12:8.1f} TFLOP/s  sha {h}", flush=True)


M, N, K = 9232, 1024, 640
A, W = f16((M, K)), f16((N, K), K ** -0.5)
for name, kw in {"plain": {}, "bias": dict(bias_n=f32((N,))), "bias+SiLU": dict(bias_n=f32((N,)), act=ACT_SILU),
                 "bias+residual": dict(bias_n=f32((N,)), residual=f16((M, N))), "GELU+residual": dict(act=ACT_GELU, residual=f16((M, N))),
                 "QuickGELU": dict(bias_n=f32((N,)), act=ACT_QUICKGELU), "GEGLU": dict(bias_n=f32((N,)), geglu=True),
                 "rowgroup+bias": dict(bias_n=f32((N,)), rowgroup_add=f32((-(-M // 577), N)), rows_per_group=577),
                 "bias_m+scale_m": dict(bias_m=f32((M,)), scale_m=f32((M,))), "fp32 out": dict(bias_n=f32((N,)), out_dtype=np.float32)}.items():
    O = ctx.empty((M, N // 2 if kw.get("geglu") else N), kw.get("out_dtype", np.float16))
    run(f"{name} M={M} N={N} K={K}", lambda: ctx.gemm(A, W, out=O, **kw), O, 2.0 * M * N * K)
for (M, N, K) in [(65536, 1024, 320), (65536, 1024, 640), (65536, 1280, 640), (65536, 1024, 1280), (65536, 512, 4096), (65536, 1024, 4096), (36928, 1024, 1024),
                  (9232, 3072, 1024), (9232, 4096, 1024), (9232, 1024, 4096)]:
    A, W, O = f16((M, K)), f16((N, K), K ** -0.5), ctx.empty((M, N), np.float16)
    b, r = f32((N,)), f16((M, N))
    run(f"gemm bias+residual M={M} N={N} K={K}", lambda: ctx.gemm(A, W, bias_n=b, residual=r, out=O), O, 2.0 * M * N * K)
    for a in (A, W, O, r):
        a.free()
for (B, H, Wd, Cin, Cout) in [(16, 64, 64, 320, 320), (16, 32, 32, 640, 640), (16, 128, 128, 512, 512), (16, 128, 128, 256, 256), (4, 512, 512, 128, 128), (16, 256, 256, 256, 256)]:
    X, Wt, O = f16((B, H, Wd, Cin)), f16((Cout, 3, 3, Cin), (9 * Cin) ** -0.5), ctx.empty((B, H, Wd, Cout), np.float16)
    b, pia = f32((Cout,)), f32((B, Cout))
    run(f"conv3x3 {B}x{H}x{Wd} {Cin}->{Cout} bias+per-image", lambda: ctx.conv2d(X, Wt, bias=b, per_image_add=pia, out=O), O, 2.0 * B * H * Wd * Cout * 9 * Cin)
    for a in (X, Wt, O):
        a.free()
