"""AutoencoderKL's encoder.conv_in at the benchmarked shape (16 crops x 512^2, 8 -> 128 channels): the kernel of its own (csrc/conv_c8.hip)
against the implicit GEMM it replaces, with and without the GroupNorm that follows (statistics from the epilogue vs a pass of their own).

    python tools/conv_in_bench.py [reps=10]          (GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from odise_amd.runtime import Context  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ctx = Context(0)
    rng = np.random.default_rng(0)
    N, H, W = 16, 512, 512
    x = np.zeros((N, H, W, 8), np.float16)
    x[..., :3] = rng.standard_normal((N, H, W, 3), dtype=np.float32).astype(np.float16)
    w = np.zeros((128, 3, 3, 8), np.float16)
    w[..., :3] = (rng.standard_normal((128, 3, 3, 3), dtype=np.float32) / 27 ** 0.5).astype(np.float16)
    dx, dw = ctx.to_device(x), ctx.to_device(w)
    db = ctx.to_device(rng.standard_normal(128, dtype=np.float32))
    out = ctx.empty((N, H, W, 128), np.float16)

    def timed(fn):
        fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        return ctx.timer_stop() / reps * 1e3

    a = timed(lambda: ctx.conv2d(dx, dw, bias=db, out=out))
    b = timed(lambda: ctx.conv2d(dx, dw, bias=db, out=out, force_tile=1))
    gb = out.nbytes / 1e9
    print(f"conv_in 16 x 512^2, 8 -> 128 channels ({gb:.2f} GB written): own kernel {a:7.1f} us ({gb / a * 1e3:.2f} TB/s), implicit GEMM (64 x 128 tile) {b:7.1f} us")
    g = ctx.to_device(np.ones(128, np.float32))
    t = timed(lambda: ctx.group_norm(out, g, g, 32, 1e-6, 1))
    print(f"GroupNorm + SiLU of the output with a statistics pass of its own: {t:7.1f} us")
    ctx.close()


if __name__ == "__main__":
    main()
