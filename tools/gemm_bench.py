"""Micro-benchmark of the GEMM / conv kernels through the C ABI. usage: python tools/gemm_bench.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)


def rand(shape, scale=1.0):
    return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * scale).astype(np.float16))


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(iters):
        fn()
    return ctx.timer_stop() / iters


print("dense GEMM  M x N x K  tile split -> us, TF/s")
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (65536, 320, 2880), (65536, 320, 320), (16384, 640, 5760), (16384, 640, 640),
                  (4096, 1280, 11520), (4096, 1280, 1280), (65536, 2560, 320), (65536, 320, 1280)]:
    A, W = rand((M, K)), rand((N, K), K ** -0.5)
    O = ctx.empty((M, N), np.float16)
    for tile, split in [(0, 0), (3, 0), (4, 0), (5, 0), (-1, 0)]:
        if tile in (3,) and N % 320:
            continue
        try:
            ms = timeit(lambda: ctx.gemm(A, W, force_tile=tile, force_split=split, out=O))
        except RuntimeError as e:
            print("  fail", tile, e)
            continue
        print(f"  {M:6d} x {N:5d} x {K:6d} tile {tile:2d}: {ms * 1e3:9.1f} us  {2.0 * M * N * K / (ms * 1e-3) / 1e12:8.1f} TF/s", flush=True)
    A.free(); W.free(); O.free()

print("conv3x3 NHWC  N,H,W,Cin->Cout")
for (N, H, W_, Cin, Cout) in [(16, 64, 64, 320, 320), (16, 32, 32, 640, 640), (16, 16, 16, 1280, 1280), (16, 64, 64, 960, 320),
                              (1, 64, 64, 320, 320), (4, 64, 64, 320, 320)]:
    X, Wt = rand((N, H, W_, Cin)), rand((Cout, 3, 3, Cin), (9 * Cin) ** -0.5)
    O = ctx.empty((N, H, W_, Cout), np.float16)
    for tile in (0, 3, 4, 5, -1):
        if tile == 3 and Cout % 320:
            continue
        ms = timeit(lambda: ctx.conv2d(X, Wt, force_tile=tile, out=O))
        fl = 2.0 * N * H * W_ * Cout * 9 * Cin
        print(f"  {N}x{H}x{W_}x{Cin}->{Cout} tile {tile:2d}: {ms * 1e3:9.1f} us  {fl / (ms * 1e-3) / 1e12:8.1f} TF/s", flush=True)
    X.free(); Wt.free(); O.free()
