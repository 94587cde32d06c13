"""The 128-channel 3x3 VAE conv at 512x512 (tile 512x128): first- vs second-generation ping-pong kernel, and the 256-pixel halo tile."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context
ctx = Context(0)
rng = np.random.default_rng(0)
X = ctx.to_device(rng.standard_normal((16, 512, 512, 128), dtype=np.float32).astype(np.float16))
Wt = ctx.to_device((rng.standard_normal((128, 3, 3, 128), dtype=np.float32) * 1152 ** -0.5).astype(np.float16))
O = ctx.empty((16, 512, 512, 128), np.float16)
best = {}
for r in range(4):
    for name, tile, flags in (("pp 512x128", 6, 0), ("pp2 512x128", 6, 512 << 4), ("halo128", 8, 0)):
        ctx.lib.odise_hip_gemm_debug(flags)
        ctx.conv2d(X, Wt, force_tile=tile, out=O); ctx.sync(); ctx.timer_start()
        for _ in range(4): ctx.conv2d(X, Wt, force_tile=tile, out=O)
        ms = ctx.timer_stop() / 4
        if r: best[name] = min(best.get(name, 1e9), ms)
ctx.lib.odise_hip_gemm_debug(0)
for k, v in best.items():
    print(f"conv 16x512x512 128->128 {k:12s}: {v*1e3:8.1f} us {2.0*16*512*512*128*1152/(v*1e-3)/1e12:7.1f} TF/s")
