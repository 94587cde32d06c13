"""3x3 conv K-tile order: chunk-major (taps of one 64-channel chunk back to back) vs tap-major, on the VAE / UNet shapes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)


def timeit(fn, it=5):
    for _ in range(2):
        fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(it):
        fn()
    return ctx.timer_stop() / it


SHAPES = [(16, 128, 128, 512, 512), (16, 256, 256, 256, 256), (8, 512, 512, 128, 128), (16, 64, 64, 512, 512),
          (16, 64, 64, 320, 320), (16, 32, 32, 640, 640), (16, 16, 16, 1280, 1280), (16, 64, 64, 960, 320)]
for (N, H, W_, Cin, Cout) in SHAPES:
    X = ctx.to_device((rng.standard_normal((N, H, W_, Cin), dtype=np.float32)).astype(np.float16))
    Wt = ctx.to_device((rng.standard_normal((Cout, 3, 3, Cin), dtype=np.float32) * (9 * Cin) ** -0.5).astype(np.float16))
    O = ctx.empty((N, H, W_, Cout), np.float16)
    for flags in (0, 16):
        ctx.lib.odise_hip_gemm_debug(flags)
        ms = timeit(lambda: ctx.conv2d(X, Wt, out=O))
        print(f"conv {N}x{H}x{W_} {Cin}->{Cout} {'tap-major' if flags else 'chunk-major'}: {ms*1e3:8.1f} us "
              f"{2.0*N*H*W_*Cout*9*Cin/(ms*1e-3)/1e12:7.1f} TF/s", flush=True)
    ctx.lib.odise_hip_gemm_debug(0)
    X.free(); Wt.free(); O.free()
