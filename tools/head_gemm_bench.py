"""The pixel decoder's GEMMs at 4 x 1024^2 (M = 86 016 token rows of the three encoder levels, msdeformattn.py:92-131), every tile of the library
against the cost model's choice: these shapes are bound by their operands' bytes, not by MFMA time, which is what the model was fitted on.

    python tools/head_gemm_bench.py [rounds=3]          (GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from odise_amd import _lib  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    M = 86016
    ctx = Context(0)
    rng = np.random.default_rng(0)
    f16 = lambda *s, sc=1.0: ctx.to_device((rng.standard_normal(s, dtype=np.float32) * sc).astype(np.float16))   # noqa: E731
    f32 = lambda *s, sc=1.0: ctx.to_device((rng.standard_normal(s, dtype=np.float32) * sc).astype(np.float32))   # noqa: E731
    x256, x1024 = f16(M, 256), f16(M, 1024)
    cases = [
        ("offsets | logits  N=288 K=256 -> f32", x256, f16(288, 256, sc=1 / 16), dict(bias_n=f32(288), out_dtype=np.float32)),
        ("offsets           N=192 K=256 -> f32", x256, f16(192, 256, sc=1 / 16), dict(bias_n=f32(192), out_dtype=np.float32)),
        ("logits            N= 96 K=256 -> f32", x256, f16(96, 256, sc=1 / 16), dict(bias_n=f32(96), out_dtype=np.float32)),
        ("value / out-proj  N=256 K=256 (+res)", x256, f16(256, 256, sc=1 / 16), dict(bias_n=f32(256), residual=x256)),
        ("linear1 + ReLU    N=1024 K=256", x256, f16(1024, 256, sc=1 / 16), dict(bias_n=f32(1024), act=_lib.ACT_RELU)),
        ("linear2 (+res)    N=256 K=1024", x1024, f16(256, 1024, sc=1 / 32), dict(bias_n=f32(256), residual=x256)),
    ]

    def timed(fn, it=10):
        fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(it):
            fn()
        return ctx.timer_stop() / it * 1e3

    print(f"M = {M}; us per launch, median of {rounds} rounds of 10; tile -1 = the cost model's choice")
    for name, A, W, kw in cases:
        N, K = W.shape
        out = ctx.empty((M, N), kw.get("out_dtype", np.float16))
        cells = []
        for tile in (-1, 0, 1, 2, 3, 4, 5, 6):
            ts = []
            try:
                for _ in range(rounds):
                    ts.append(timed(lambda: ctx.gemm(A, W, out=out, force_tile=tile, **kw)))
                cells.append(f"t{tile}:{np.median(ts):6.1f}")
            except Exception:   # a tile whose preconditions do not hold for the shape
                cells.append(f"t{tile}:   n/a")
        byts = M * K * 2 + N * K * 2 + M * N * (4 if kw.get("out_dtype") is np.float32 else 2) + (M * N * 2 if "residual" in kw else 0)
        print(f"  {name:40s} {byts / 1e6:6.0f} MB  " + "  ".join(cells))
    ctx.close()


if __name__ == "__main__":
    main()
