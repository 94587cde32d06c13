"""The two ping-pong kernels against each other on the same tile (gemm_pp_kernel: fragment reads in the load segment; gemm_pp2_kernel: under the
MFMAs): time (min over interleaved rounds) and max |difference| of the outputs.  usage: pp_bench.py [tile=4]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = Context(0)
rng = np.random.default_rng(0)


def rand(shape, s=1.0):
    return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))


def bench(variants, fn, flop, label, it=10, rounds=3):
    best = {}
    outs = {}
    for r in range(rounds + 1):  # round 0 = warm-up (clocks, caches)
        for name, flags in variants:
            ctx.lib.odise_hip_gemm_debug(flags)
            fn()
            ctx.sync()
            ctx.timer_start()
            for _ in range(it):
                out = fn()
            ms = ctx.timer_stop() / it
            if r > 0:
                best[name] = min(best.get(name, 1e9), ms)
            if r == rounds:
                outs[name] = out.numpy().astype(np.float32)
    ctx.lib.odise_hip_gemm_debug(0)
    ref = outs[variants[0][0]]
    for name, _ in variants:
        print(f"{label} {name:8s}: {best[name]*1e3:8.1f} us {flop/(best[name]*1e-3)/1e12:7.1f} TF/s   max|d|={np.abs(outs[name]-ref).max():.3g}", flush=True)


NO_PP, PT1, PP2, NO_HALO = 2 << 4, 4 << 4, 512 << 4, 64 << 4
variants = [("pp", NO_HALO | (1024 << 4)), ("pp2", PP2 | NO_HALO)]
bn = 320 if tile == 3 else 128 if tile == 6 else 256
for (M, N, K) in [(65536, 2 * bn, 4096), (65536, 2 * bn, 320), (65536, 2 * bn, 640), (65536, 4 * bn, 1024), (4096, 4 * bn, 4096), (16384, bn, 8192), (1000, bn + 8, 192)]:
    A, W, O = rand((M, K)), rand((N, K), K ** -0.5), ctx.empty((M, N), np.float16)
    bench(variants, lambda: ctx.gemm(A, W, force_tile=tile, out=O), 2.0 * M * N * K, f"gemm M={M} N={N} K={K} tile {tile}")
    A.free(); W.free(); O.free()

for (B, H, W_, Cin, Cout) in [(16, 128, 128, 512, 2 * bn), (16, 64, 64, 512, 2 * bn), (16, 256, 256, 256, bn), (16, 64, 64, bn, bn), (16, 32, 32, 2 * bn, 2 * bn), (2, 37, 41, 128, bn)]:
    X = rand((B, H, W_, Cin))
    Wt = rand((Cout, 3, 3, Cin), (9 * Cin) ** -0.5)
    O = ctx.empty((B, H, W_, Cout), np.float16)
    bench(variants, lambda: ctx.conv2d(X, Wt, force_tile=tile, out=O), 2.0 * B * H * W_ * Cout * 9 * Cin, f"conv {B}x{H}x{W_} {Cin}->{Cout} tile {tile}", it=5)
    X.free(); Wt.free(); O.free()
