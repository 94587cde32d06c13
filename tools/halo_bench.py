"""3x3 conv: LDS-resident input patch (halo) kernel vs the im2col ping-pong kernel: time (min over rounds) and max |difference|."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)


def rand(shape, s=1.0):
    return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))


def run(label, fn, variants, flop, it=5, rounds=3):
    best, outs = {}, {}
    for r in range(rounds + 1):
        for name, tile, split in variants:
            fn(tile, split)
            ctx.sync()
            ctx.timer_start()
            for _ in range(it):
                o = fn(tile, split)
            ms = ctx.timer_stop() / it
            if r > 0:
                best[name] = min(best.get(name, 1e9), ms)
            if r == rounds:
                outs[name] = o.numpy().astype(np.float32)
    ref = outs[variants[0][0]]
    for name, _, _ in variants:
        print(f"{label} {name:10s}: {best[name]*1e3:8.1f} us {flop/(best[name]*1e-3)/1e12:7.1f} TF/s   max|d|={np.abs(outs[name]-ref).max():.3g} (scale {np.abs(ref).max():.3g})", flush=True)


for (B, H, W_, Cin, Cout, quick) in [(3, 37, 41, 128, 256, 1), (16, 64, 64, 512, 512, 0), (16, 128, 128, 512, 512, 0),
                                     (4, 256, 256, 256, 256, 0), (8, 512, 512, 128, 128, 0), (16, 16, 16, 1280, 1280, 0), (16, 64, 64, 256, 128, 0)]:
    X = rand((B, H, W_, Cin))
    Wt = rand((Cout, 3, 3, Cin), (9 * Cin) ** -0.5)
    bias = ctx.to_device(rng.standard_normal(Cout).astype(np.float32))
    R = rand((B, H, W_, Cout))
    O = ctx.empty((B, H, W_, Cout), np.float16)
    halo = 7 if Cout > 128 else 8
    base = 4 if Cout > 128 else 6
    variants = [("pp", base, 0), ("halo", halo, 0), ("auto", -1, 0)] + ([("pp/s2", base, 2), ("halo/s2", halo, 2)] if Cin >= 128 and quick else [])
    run(f"conv {B}x{H}x{W_} {Cin}->{Cout}", lambda t, sp: ctx.conv2d(X, Wt, bias=bias, residual=R, force_tile=t, force_split=sp, out=O), variants,
        2.0 * B * H * W_ * Cout * 9 * Cin, it=2 if quick else 5)
    for a in (X, Wt, bias, R, O):
        a.free()
