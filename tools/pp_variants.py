"""A/B of tuning variants of the 256x256 ping-pong tile (ODISE debug flags bits 7-8): 0 baseline, 1 DMA before fragment reads,
2 no s_setprio, 3 both."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context
ctx = Context(0)
rng = np.random.default_rng(0)
def rand(shape, s=1.0): return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))
def bench(label, fn, flop, it=8, rounds=3):
    best = {}
    for r in range(rounds + 1):
        for v in range(4):
            ctx.lib.odise_hip_gemm_debug((v << 7) << 4 | (64 << 4))  # variant + no halo kernel
            fn(); ctx.sync(); ctx.timer_start()
            for _ in range(it): fn()
            ms = ctx.timer_stop() / it
            if r > 0: best[v] = min(best.get(v, 1e9), ms)
    ctx.lib.odise_hip_gemm_debug(0)
    print(label + "  " + "  ".join(f"v{v}: {best[v]*1e3:7.1f} us {flop/(best[v]*1e-3)/1e12:6.1f} TF/s" for v in range(4)), flush=True)
for (M, N, K) in [(65536, 512, 4096), (65536, 512, 640), (16384, 1024, 1024)]:
    A, W, O = rand((M, K)), rand((N, K), K ** -0.5), ctx.empty((M, N), np.float16)
    bench(f"gemm {M}x{N}x{K}", lambda: ctx.gemm(A, W, force_tile=4, out=O), 2.0 * M * N * K)
    A.free(); W.free(); O.free()
X = rand((16, 64, 64, 512)); Wt = rand((512, 3, 3, 512), 4608 ** -0.5); O = ctx.empty((16, 64, 64, 512), np.float16)
bench("conv 16x64x64 512->512", lambda: ctx.conv2d(X, Wt, force_tile=4, out=O), 2.0 * 16 * 64 * 64 * 512 * 4608, it=5)
