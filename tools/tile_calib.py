"""Calibrate the per-K-tile cost of every GEMM tile configuration (feeds kTileCost in gemm.hip)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context
ctx = Context(0)
rng = np.random.default_rng(0)
BM = [128, 64, 64, 256, 256, 256]; BN = [128, 128, 64, 320, 256, 128]; SLOTS = [2, 3, 4, 1, 1, 1]
def rand(shape, s=1.0): return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))
def timeit(fn, it=10):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(it): fn()
    return ctx.timer_stop() / it
for (M, K) in [(131072, 1152), (131072, 320), (16384, 5760)]:
    for t in range(6):
        N = 640 if t == 3 else 512
        A, W, O = rand((M, K)), rand((N, K), K ** -0.5), ctx.empty((M, N), np.float16)
        ms = timeit(lambda: ctx.gemm(A, W, force_tile=t, out=O))
        nb = -(-M // BM[t]) * -(-N // BN[t]); rounds = -(-nb // (256 * SLOTS[t])); nk = -(-K // 64)
        print(f"M={M} N={N} K={K} tile {t} ({BM[t]}x{BN[t]}): {ms*1e3:8.1f} us {2.0*M*N*K/(ms*1e-3)/1e12:7.1f} TF/s  blocks {nb} rounds {rounds} -> {ms*1e3/rounds:7.1f} us/round, {ms*1e3/rounds/nk:6.2f} us/ktile (incl. fixed)", flush=True)
        A.free(); W.free(); O.free()
