"""The four GEMMs of one block of MaskCLIP's mask-token pass (4 pictures x 100 mask tokens = 400 rows, width 1024; extractor.cpp
maskclip_mask_pass), every tile / split-K of the library against the cost model's choice.

    python tools/maskclip_gemm_bench.py [rounds=3] [M=400]          (GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from odise_amd import _lib  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    ctx = Context(0)
    rng = np.random.default_rng(0)
    f16 = lambda *s, sc=1.0: ctx.to_device((rng.standard_normal(s, dtype=np.float32) * sc).astype(np.float16))   # noqa: E731
    f32 = lambda *s, sc=1.0: ctx.to_device((rng.standard_normal(s, dtype=np.float32) * sc).astype(np.float32))   # noqa: E731
    x, h = f16(M, 1024), f16(M, 4096)
    cases = [
        ("q / out-proj (+res)  N=1024 K=1024", x, f16(1024, 1024, sc=1 / 32), dict(bias_n=f32(1024), residual=x)),
        ("c_fc + QuickGELU     N=4096 K=1024", x, f16(4096, 1024, sc=1 / 32), dict(bias_n=f32(4096), act=_lib.ACT_QUICKGELU)),
        ("c_proj (+res)        N=1024 K=4096", h, f16(1024, 4096, sc=1 / 64), dict(bias_n=f32(1024), residual=x)),
    ]

    def timed(fn, it=20):
        fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(it):
            fn()
        return ctx.timer_stop() / it * 1e3

    print(f"M = {M} rows; us per call (GEMM + its split-K reduce), median of {rounds} rounds of 20 back-to-back calls; 'auto' = the cost model")
    for name, A, W, kw in cases:
        N, K = W.shape
        out = ctx.empty((M, N), np.float16)
        t_auto = np.median([timed(lambda: ctx.gemm(A, W, out=out, **kw)) for _ in range(rounds)])
        print(f"  {name}: auto {t_auto:6.1f}")
        for tile in (0, 1, 2, 5):
            cells = []
            for split in (1, 2, 4, 8):
                try:
                    ts = [timed(lambda: ctx.gemm(A, W, out=out, force_tile=tile, force_split=split, **kw)) for _ in range(rounds)]
                    cells.append(f"x{split}:{np.median(ts):6.1f}")
                except Exception:
                    cells.append(f"x{split}:   n/a")
            print(f"      tile {tile}  " + "  ".join(cells))
    ctx.close()


if __name__ == "__main__":
    main()
