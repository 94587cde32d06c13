"""Measured time of every (tile, split-K) choice on the under-filled GEMM / conv shapes of the pipeline, next to the automatic choice."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context
ctx = Context(0)
rng = np.random.default_rng(0)
def rand(shape, s=1.0): return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))
def timeit(fn, it=6, rounds=2):
    best = 1e9
    for r in range(rounds):
        for _ in range(2): fn()
        ctx.sync(); ctx.timer_start()
        for _ in range(it): fn()
        best = min(best, ctx.timer_stop() / it)
    return best * 1e3
TILES = ["128x128", "64x128", "64x64", "256x320", "256x256", "256x128", "512x128", "halo256", "halo128"]
def sweep(label, fn, splits=(1, 2, 3, 4, 6, 8)):
    auto = timeit(lambda: fn(-1, 0))
    res = []
    for t in range(9):
        for sp in splits:
            try:
                res.append((timeit(lambda: fn(t, sp)), t, sp))
            except Exception as e:  # workspace too small etc.
                pass
    res.sort()
    print(f"{label}: auto {auto:7.1f} us | best " + ", ".join(f"{TILES[t]}/s{sp} {us:.1f}" for us, t, sp in res[:5]), flush=True)
for (M, N, K) in [(9472, 1024, 1024), (9472, 4096, 1024), (9472, 1024, 4096), (2752, 1024, 1024), (2752, 4096, 1024), (2752, 1024, 4096),
                  (4096, 1280, 1280), (4096, 1280, 5120), (4096, 10240, 1280), (1024, 1280, 1280), (16384, 640, 640), (16384, 640, 2560), (21504, 256, 256), (21504, 1024, 256)]:
    A, W, O = rand((M, K)), rand((N, K), K ** -0.5), ctx.empty((M, N), np.float16)
    sweep(f"gemm M={M} N={N} K={K}", lambda t, sp: ctx.gemm(A, W, force_tile=t, force_split=sp, out=O))
    A.free(); W.free(); O.free()
for (B, H, Cin, Cout) in [(16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 8, 1280, 1280), (16, 32, 640, 640), (16, 32, 1280, 640), (16, 32, 1920, 640)]:
    X = rand((B, H, H, Cin)); Wt = rand((Cout, 3, 3, Cin), (9 * Cin) ** -0.5); O = ctx.empty((B, H, H, Cout), np.float16)
    sweep(f"conv {B}x{H}x{H} {Cin}->{Cout}", lambda t, sp: ctx.conv2d(X, Wt, force_tile=t, force_split=sp, out=O))
    X.free(); Wt.free(); O.free()
