"""The 128-channel 512x512 VAE level (3x3 conv 128 -> 128, 16 crops: 8 ms of the step): the im2col 512x128 ping-pong tile (6) against the
16x16-pixel halo tile (8), plus the neighbouring shapes; with the tools build, ODISE_NO_RES_PREFETCH=1 switches the epilogue's residual
prefetch off for an A/B run."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)


def f16(shape, s=1.0):
    return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))


def t(fn, out):
    for _ in range(2):
        fn()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(5):
            fn()
        best = min(best, ctx.timer_stop() / 5)
    return best * 1e3, hashlib.sha1(out.numpy().tobytes()).hexdigest()[:10]


for (B, H, W, Cin, Cout, tiles) in [(16, 512, 512, 128, 128, (6, 8, -1)), (4, 512, 512, 128, 128, (6, 8, -1)), (16, 256, 256, 128, 256, (7, 4, -1)),
                                    (16, 128, 128, 128, 128, (6, 8, -1)), (16, 256, 256, 256, 256, (7, -1)), (16, 64, 64, 128, 128, (5, 8, -1))]:
    X, Wt, O = f16((B, H, W, Cin)), f16((Cout, 3, 3, Cin), (9 * Cin) ** -0.5), ctx.empty((B, H, W, Cout), np.float16)
    b, r = ctx.to_device(rng.standard_normal(Cout, dtype=np.float32)), f16((B, H, W, Cout))
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    line = f"conv3x3 {B}x{H}x{W} {Cin}->{Cout} (+bias +residual):"
    shas = {}
    for tl in tiles:
        us, sha = t(lambda: ctx.conv2d(X, Wt, bias=b, residual=r, out=O, force_tile=tl), O)
        shas[tl] = sha
        line += f"  tile {tl:2d} {us:8.1f} us {fl / us / 1e6:6.0f} TF/s"
    line += "   bit-identical: " + str(len(set(shas.values())) == 1)
    print(line, flush=True)
    for a in (X, Wt, O, r):
        a.free()
