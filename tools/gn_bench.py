"""GroupNorm (+SiLU) pass on the VAE tensor shapes: time and effective HBM rate (2 B read twice without fused statistics + 2 B written)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(0)
for (N, HW, C) in [(16, 128 * 128, 512), (16, 256 * 256, 256), (16, 512 * 512, 128), (16, 64 * 64, 512), (16, 64 * 64, 320)]:
    x = ctx.to_device(rng.standard_normal((N, HW, C), dtype=np.float32).astype(np.float16))
    g, b = ctx.to_device(rng.standard_normal(C, dtype=np.float32)), ctx.to_device(rng.standard_normal(C, dtype=np.float32))
    y = None
    best = 1e9
    for rnd in range(3):
        for _ in range(2):
            y = ctx.group_norm(x, g, b, 32, 1e-6, 1)
        ctx.sync()
        ctx.timer_start()
        for _ in range(10):
            y = ctx.group_norm(x, g, b, 32, 1e-6, 1)
        best = min(best, ctx.timer_stop() / 10)
    nbytes = N * HW * C * 2
    print(f"group_norm+silu {N}x{HW}x{C}: {best * 1e3:8.1f} us  statistics + apply = {3 * nbytes / best / 1e9:6.2f} TB/s effective (apply alone would be {2 * nbytes / 1e6:.0f} MB)", flush=True)
    x.free()
