"""How much of each short-K GEMM of the step is epilogue?  Full kernel / accumulators staged but neither epilogue math nor global stores
(dbg 8) / main loop only (dbg 4), with the cost model's own tile choice.  Timing only (the ablated runs write nothing).  Needs the tools
build: ODISE_HIP_LIB=odise_amd/lib/libodise_hip_tools.so python tools/epi_share.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd._lib import ACT_NONE, ACT_QUICKGELU  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)


def f16(shape, s=1.0):
    return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))


def t(fn):
    best = 1e9
    for _ in range(3):
        for _ in range(2):
            fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(20):
            fn()
        best = min(best, ctx.timer_stop() / 20)
    return best * 1e3


shapes = [(9344, 1024, 1024, True, ACT_NONE, "CLIP attention out-projection (+residual)"), (9344, 4096, 1024, False, ACT_QUICKGELU, "CLIP fc1 (QuickGELU)"),
          (9344, 1024, 4096, True, ACT_NONE, "CLIP fc2 (+residual)"), (9344, 2048, 1024, False, ACT_NONE, "CLIP q,k projection"),
          (65536, 320, 320, False, ACT_NONE, "UNet 64x64 attention projection"), (16384, 640, 640, False, ACT_NONE, "UNet 32x32 attention projection"),
          (4096, 1280, 1280, True, ACT_NONE, "UNet 16x16 projection (+residual)"), (86016, 256, 256, False, ACT_NONE, "pixel decoder projection"),
          (65536, 1024, 4096, True, ACT_NONE, "large reference GEMM")]
for (M, N, K, res, act, what) in shapes:
    A, W, O = f16((M, K)), f16((N, K), K ** -0.5), ctx.empty((M, N), np.float16)
    b = ctx.to_device(rng.standard_normal(N, dtype=np.float32))
    r = f16((M, N)) if res else None
    row = []
    for dbg in (0, 8, 4):
        ctx.lib.odise_hip_gemm_debug(dbg)
        row.append(t(lambda: ctx.gemm(A, W, bias_n=b, residual=r, act=act, out=O)))
    ctx.lib.odise_hip_gemm_debug(0)
    full, staged, loop = row
    print(f"M={M:6d} N={N:5d} K={K:5d} {what:42s}: full {full:7.1f} us ({2.0 * M * N * K / full / 1e6:5.0f} TF/s)  staging only {staged:7.1f}  loop only {loop:7.1f}"
          f"  -> epilogue {100 * (full - loop) / full:4.1f} % of the kernel (math + stores {100 * (full - staged) / full:4.1f} %)", flush=True)
    for a in (A, W, O, r):
        if a is not None:
            a.free()
