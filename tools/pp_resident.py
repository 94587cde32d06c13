"""Is the pipelined GEMM bound by operand latency (HBM first touch) or by the LDS-DMA path itself?  Same GEMM with A rows at the
normal stride (streamed from HBM), overlapping at 128 B (A is L2-resident) and at stride 0 (L1-resident)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context
ctx = Context(0)
rng = np.random.default_rng(0)
def rand(shape, s=1.0): return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))
def timeit(fn, it=10):
    for _ in range(5): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(it): fn()
    return ctx.timer_stop() / it
M, N, K = 65536, 512, 4096
A, W, O = rand((M, K)), rand((N, K), K ** -0.5), ctx.empty((M, N), np.float16)
for r in range(2):
    for name, lda in (("hbm-streamed A", K), ("L2-resident A", 64), ("L1-resident A", 0)):
        for dbg in (4, 5):
            ctx.lib.odise_hip_gemm_debug(dbg)
            ms = timeit(lambda: ctx.gemm(A, W, force_tile=4, out=O, lda=lda))
            print(f"{name:16s} {'loop only' if dbg == 4 else 'loop, no dma'}: {ms*1e3:8.1f} us {2.0*M*N*K/(ms*1e-3)/1e12:7.1f} TF/s", flush=True)
ctx.lib.odise_hip_gemm_debug(0)
