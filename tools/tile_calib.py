"""Calibrate the per-K-tile cost of every GEMM tile configuration (feeds kTileCost / kTileCostPP in gemm.hip).
Two K values per tile give the slope (t_ktile) and the intercept (t_fixed) of one residency round."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context
ctx = Context(0)
rng = np.random.default_rng(0)
BM = [128, 64, 64, 256, 256, 256, 512]; BN = [128, 128, 64, 320, 256, 128, 128]; SLOTS = [2, 3, 4, 1, 1, 1, 1]
def rand(shape, s=1.0): return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))
def timeit(fn, it=10, rounds=3):
    best = 1e9
    for r in range(rounds):
        for _ in range(3): fn()
        ctx.sync(); ctx.timer_start()
        for _ in range(it): fn()
        best = min(best, ctx.timer_stop() / it)
    return best
M = 131072
for t in range(7):
    N = 640 if t == 3 else 512
    res = {}
    for K in (320, 1152, 4096):
        A, W, O = rand((M, K)), rand((N, K), K ** -0.5), ctx.empty((M, N), np.float16)
        ms = timeit(lambda: ctx.gemm(A, W, force_tile=t, out=O))
        nb = -(-M // BM[t]) * -(-N // BN[t]); rounds = nb / (256 * SLOTS[t]); nk = K // 64
        res[K] = (ms * 1e3 / rounds, nk)
        print(f"N={N} K={K} tile {t} ({BM[t]}x{BN[t]}): {ms*1e3:8.1f} us {2.0*M*N*K/(ms*1e-3)/1e12:7.1f} TF/s  rounds {rounds:.2f} -> {ms*1e3/rounds:7.1f} us/round", flush=True)
        A.free(); W.free(); O.free()
    (u0, k0), (u1, k1), (u2, k2) = res[320], res[1152], res[4096]
    s01 = (u1 - u0) / (k1 - k0); s12 = (u2 - u1) / (k2 - k1)
    print(f"   tile {t}: t_ktile {s01:.2f} (short K) / {s12:.2f} (long K) us, t_fixed {u0 - s01 * k0:.1f} us", flush=True)
# the N = 128 convolution of the VAE encoder's first stage
X = rand((8, 512, 512, 128)); Wt = rand((128, 3, 3, 128), 1152 ** -0.5); O = ctx.empty((8, 512, 512, 128), np.float16)
for t in (0, 5, 6):
    ms = timeit(lambda: ctx.conv2d(X, Wt, force_tile=t, out=O), it=5)
    print(f"conv 8x512x512 128->128 tile {t}: {ms*1e3:8.1f} us {2.0*8*512*512*128*1152/(ms*1e-3)/1e12:7.1f} TF/s", flush=True)
