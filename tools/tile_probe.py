"""Which tile / split-K wins on the under-filled GEMMs of the step (CLIP tower at M = 16 x 577 and 4 x 677 tokens, UNet projections)?
Times every (tile, split) against the cost model's own choice.  Tile ids: 0 128x128, 1 64x128, 2 64x64, 3 256x320, 4 256x256, 5 256x128."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd._lib import ACT_QUICKGELU  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)


def f16(shape, s=1.0):
    return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))


def t(fn):
    for _ in range(3):
        fn()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(20):
            fn()
        best = min(best, ctx.timer_stop() / 20)
    return best * 1e3


shapes = [(9232, 1024, 1024), (9232, 3072, 1024), (9232, 4096, 1024), (9232, 1024, 4096), (2708, 1024, 1024), (2708, 3072, 1024), (2708, 4096, 1024),
          (2708, 1024, 4096), (65536, 320, 320), (65536, 960, 320), (16384, 640, 640), (16384, 1920, 640), (4096, 1280, 1280), (4096, 3840, 1280),
          (21504, 256, 256), (21504, 1024, 256), (21504, 256, 1024)]
if len(sys.argv) > 1:  # tile_probe.py M,N,K [M,N,K ...]
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (M, N, K) in shapes:
    A, W, O = f16((M, K)), f16((N, K), K ** -0.5), ctx.empty((M, N), np.float16)
    b = ctx.to_device(rng.standard_normal(N, dtype=np.float32))
    auto = t(lambda: ctx.gemm(A, W, bias_n=b, out=O))
    res = []
    for tile in (0, 1, 2, 3, 4, 5):
        for split in (1, 2, 4):
            if split > 1 and K < 512:
                continue
            try:
                res.append((t(lambda: ctx.gemm(A, W, bias_n=b, out=O, force_tile=tile, force_split=split)), tile, split))
            except RuntimeError:
                pass
    res.sort()
    flop = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d}: auto {auto:7.1f} us ({flop / auto / 1e6:6.1f} TF/s)   best " + "  ".join(f"t{tl}/s{sp} {us:6.1f}" for us, tl, sp in res[:4]), flush=True)
    for a in (A, W, O):
        a.free()
