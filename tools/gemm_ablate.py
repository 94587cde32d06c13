"""Ablation of the GEMM main loop (guide §5.4 rule: ablate before optimising): full / no operand DMA / no DMA + no fragment reads."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context
ctx = Context(0)
rng = np.random.default_rng(0)
def rand(shape, s=1.0): return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))
def timeit(fn, it=10):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(it): fn()
    return ctx.timer_stop() / it
for (M, N, K) in [(65536, 640, 2880), (65536, 640, 320)]:
    A, W, O = rand((M, K)), rand((N, K), K ** -0.5), ctx.empty((M, N), np.float16)
    for t in ([0, 4, 5] if N % 320 else [0, 3, 4]):
        for dbg in (0, 8, 4):
            ctx.lib.odise_hip_gemm_debug(dbg)
            ms = timeit(lambda: ctx.gemm(A, W, force_tile=t, out=O))
            print(f"M={M} N={N} K={K} tile {t} dbg {dbg}: {ms*1e3:8.1f} us {2.0*M*N*K/(ms*1e-3)/1e12:7.1f} TF/s", flush=True)
    ctx.lib.odise_hip_gemm_debug(0)
    A.free(); W.free(); O.free()
