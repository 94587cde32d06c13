"""The 128-channel 3x3 convolutions: 8-wave halo tile (8: one block per CU) against the 4-wave form (9: two co-resident blocks per CU), same
box, interleaved rounds; outputs must be bit-identical.  Also the fused GroupNorm-statistics pair.  Shapes: the VAE encoder's 512^2 level
(the step runs it 4 x per 16 crops), smaller batches, and the Cout = 128 layers of other resolutions."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)


def rand(shape, s=1.0):
    return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))


for (B, H, W_, Cin, Cout, res) in [(16, 512, 512, 128, 128, True), (16, 512, 512, 128, 128, False), (4, 512, 512, 128, 128, True), (16, 256, 256, 128, 128, True),
                                   (16, 64, 64, 256, 128, False), (16, 128, 128, 128, 128, False), (1, 512, 512, 128, 128, True)]:
    X = rand((B, H, W_, Cin))
    Wt = rand((Cout, 3, 3, Cin), (9 * Cin) ** -0.5)
    bias = ctx.to_device(rng.standard_normal(Cout).astype(np.float32))
    R = rand((B, H, W_, Cout)) if res else None
    O = ctx.empty((B, H, W_, Cout), np.float16)
    flop = 2.0 * B * H * W_ * Cout * 9 * Cin
    best, outs = {}, {}
    for r in range(4):
        for name, tile in (("8-wave", 8), ("4-wave x2", 9), ("im2col 512x128", 6)):
            ctx.conv2d(X, Wt, bias=bias, residual=R, force_tile=tile, force_split=1, out=O)
            ctx.sync()
            ctx.timer_start()
            for _ in range(5):
                ctx.conv2d(X, Wt, bias=bias, residual=R, force_tile=tile, force_split=1, out=O)
            ms = ctx.timer_stop() / 5
            if r > 0:
                best[name] = min(best.get(name, 1e9), ms)
            if r == 3:
                outs[name] = O.numpy().tobytes()
    same = outs["8-wave"] == outs["4-wave x2"]
    print(f"conv3x3 {B}x{H}x{W_} {Cin}->{Cout} res={int(res)}: " + "  ".join(f"{k} {v*1e3:8.1f} us {flop/(v*1e-3)/1e12:6.1f} TF/s" for k, v in best.items())
          + f"  4-wave vs 8-wave bits {'identical' if same else 'DIFFERENT'}", flush=True)
    for a in (X, Wt, bias, R, O):
        if a is not None:
            a.free()

# the wide layers (Cout = 256 / 512 / 640): the 256-channel halo tile (7, 8 waves, one block per CU, the halo fetched once per two column
# blocks) against the 128-channel tiles - every patch's halo is then fetched by Cout / 128 column blocks, but two blocks share a CU
print("# wide layers: tile 7 (halo 256 px x 256 ch) vs 8 (8-wave x 128 ch) vs 9 (4-wave x 128 ch, two per CU)")
for (B, H, W_, Cin, Cout, res) in [(16, 64, 64, 512, 512, True), (4, 128, 128, 512, 512, True), (16, 128, 128, 256, 256, True), (4, 256, 256, 256, 256, True),
                                   (4, 256, 256, 128, 256, False), (16, 32, 32, 640, 640, False)]:
    X = rand((B, H, W_, Cin))
    Wt = rand((Cout, 3, 3, Cin), (9 * Cin) ** -0.5)
    bias = ctx.to_device(rng.standard_normal(Cout).astype(np.float32))
    R = rand((B, H, W_, Cout)) if res else None
    O = ctx.empty((B, H, W_, Cout), np.float16)
    flop = 2.0 * B * H * W_ * Cout * 9 * Cin
    best, outs = {}, {}
    for r in range(4):
        for name, tile in (("halo256", 7), ("8-wave128", 8), ("4-wave128 x2", 9)):
            ctx.conv2d(X, Wt, bias=bias, residual=R, force_tile=tile, force_split=1, out=O)
            ctx.sync()
            ctx.timer_start()
            for _ in range(5):
                ctx.conv2d(X, Wt, bias=bias, residual=R, force_tile=tile, force_split=1, out=O)
            ms = ctx.timer_stop() / 5
            if r > 0:
                best[name] = min(best.get(name, 1e9), ms)
            if r == 3:
                outs[name] = O.numpy().tobytes()
    same = outs["halo256"] == outs["4-wave128 x2"]
    print(f"conv3x3 {B}x{H}x{W_} {Cin}->{Cout} res={int(res)}: " + "  ".join(f"{k} {v*1e3:8.1f} us {flop/(v*1e-3)/1e12:6.1f} TF/s" for k, v in best.items())
          + f"  bits {'identical' if same else 'DIFFERENT'}", flush=True)
    for a in (X, Wt, bias, R, O):
        if a is not None:
            a.free()
