"""3x3 convolutions of the step by shape: the cost model's choice against forced tiles (gemm.hip kTileBM / kTileBN: 3 = 256x320 im2col,
4 = 256x256 im2col, 7 = halo 256 px x 256 ch, 8 = halo x 128 ch, 9 = halo x 128 ch as four waves, two blocks per CU), split-K left to the model.
usage: conv_tiles.py  (prints one line per shape; outputs of every tile must agree bit for bit at split 1)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)


def rand(shape, s=1.0):
    return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))


SHAPES = [(16, 16, 1280, 1280), (16, 16, 1920, 1280), (16, 16, 2560, 1280), (16, 32, 960, 640), (16, 32, 1920, 640), (16, 32, 1280, 640), (16, 32, 640, 640),
          (16, 32, 320, 640), (16, 64, 320, 320), (16, 64, 640, 320), (16, 64, 960, 320), (16, 8, 1280, 1280), (16, 8, 2560, 1280),
          (16, 128, 512, 512), (16, 64, 512, 512), (16, 256, 256, 256), (16, 256, 128, 256), (16, 128, 256, 512), (4, 128, 128, 128)]
for (B, H, Cin, Cout) in SHAPES:
    X = rand((B, H, H, Cin))
    Wt = rand((Cout, 3, 3, Cin), (9 * Cin) ** -0.5)
    bias = ctx.to_device(rng.standard_normal(Cout).astype(np.float32))
    O = ctx.empty((B, H, H, Cout), np.float16)
    flop = 2.0 * B * H * H * Cout * 9 * Cin
    best, chosen = {}, None
    for r in range(3):
        for tile in (-1, 3, 4, 7, 9):
            try:
                ctx.conv2d(X, Wt, bias=bias, force_tile=tile, out=O)
            except RuntimeError:
                continue
            if tile < 0:
                chosen = ctx.lib.odise_hip_last_tile()
            ctx.sync()
            ctx.timer_start()
            for _ in range(5):
                ctx.conv2d(X, Wt, bias=bias, force_tile=tile, out=O)
            ms = ctx.timer_stop() / 5
            if r > 0:
                best[tile] = min(best.get(tile, 1e9), ms)
    auto = best[-1]
    others = {t: v for t, v in best.items() if t >= 0}
    tb = min(others, key=others.get)
    print(f"conv3x3 {B}x{H}x{H} {Cin}->{Cout}: auto tile {chosen & 255} split {chosen >> 8} {auto*1e3:7.1f} us {flop/(auto*1e-3)/1e12:6.0f} TF/s | "
          + "  ".join(f"t{t} {v*1e3:7.1f}" for t, v in sorted(others.items())) + f" | best forced t{tb} {'(auto is within 2 %)' if auto <= others[tb] * 1.02 else '<== auto loses %.1f %%' % ((auto / others[tb] - 1) * 100)}",
          flush=True)
    for a in (X, Wt, bias, O):
        a.free()
