"""(ODISE_GEMM_FREEZE_K=1: every K-tile re-reads the first one = hot lines, timing only.)
Per-CU throughput of the ping-pong GEMM as the grid fills more of the chip (N=1024, K=4096, 256x256 tiles, one tile per CU up to
256 tiles, then whole rounds): separates what the instruction schedule can do from what the chip sustains (clock, fabric)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
ctx.lib.odise_hip_gemm_debug(1024 << 4)   # gemm_pp_kernel for every size (no pp2), so that ODISE_GEMM_FREEZE_K=1 applies throughout
rng = np.random.default_rng(0)
N, K = 1024, 4096
W = ctx.to_device((rng.standard_normal((N, K), dtype=np.float32) * K ** -0.5).astype(np.float16))
for M in (2048, 4096, 8192, 12288, 16384, 32768, 65536, 131072):
    A = ctx.to_device(rng.standard_normal((M, K), dtype=np.float32).astype(np.float16))
    O = ctx.empty((M, N), np.float16)
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
    tf = 2.0 * M * N * K / (best * 1e-3) / 1e12
    cus = min(tiles, 256)
    print(f"M={M:6d}: {tiles:4d} tiles  {best*1e3:8.1f} us  {tf:7.1f} TFLOP/s  {tf / cus:5.2f} TFLOP/s per busy CU", flush=True)
    A.free()
    O.free()
