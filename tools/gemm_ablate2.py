"""Ablation of gemm_pp_kernel's main loop on the whole chip and on 32 CUs: loop only / without LDS-DMA / without fragment reads /
without both (timing only; results are wrong by construction)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)
N, K = 1024, int(os.environ.get('K', 4096))
W = ctx.to_device((rng.standard_normal((N, K), dtype=np.float32) * K ** -0.5).astype(np.float16))
for M in (2048, 65536):
    A = ctx.to_device(rng.standard_normal((M, K), dtype=np.float32).astype(np.float16))
    O = ctx.empty((M, N), np.float16)
    for name, dbg in (("full", 0), ("LDS staging, no stores", 8), ("loop only (no epilogue)", 4), ("loop, no DMA", 5), ("loop, no fragment reads", 6), ("loop, no DMA, no reads", 7)):
        ctx.lib.odise_hip_gemm_debug((1024 << 4) | dbg)
        best = 1e9
        for rnd in range(3):
            for _ in range(2):
                ctx.gemm(A, W, force_tile=4, force_split=1, out=O)
            ctx.sync()
            ctx.timer_start()
            for _ in range(10):
                ctx.gemm(A, W, force_tile=4, force_split=1, out=O)
            best = min(best, ctx.timer_stop() / 10)
        tiles = (M // 256) * (N // 256)
        print(f"M={M:6d} ({tiles:4d} tiles) {name:26s}: {best*1e3:8.1f} us  {2.0*M*N*K/(best*1e-3)/1e12:7.1f} TFLOP/s", flush=True)
    ctx.lib.odise_hip_gemm_debug(0)
    A.free()
    O.free()
