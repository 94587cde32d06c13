"""Run one conv / gemm shape repeatedly (target for rocprofv3 --pmc). usage: one_conv.py tile [iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = Context(0)
rng = np.random.default_rng(0)
N, H, W_, Cin, Cout = 16, 64, 64, 320, 320
X = ctx.to_device((rng.standard_normal((N, H, W_, Cin), dtype=np.float32)).astype(np.float16))
Wt = ctx.to_device((rng.standard_normal((Cout, 3, 3, Cin), dtype=np.float32) * (9 * Cin) ** -0.5).astype(np.float16))
O = ctx.empty((N, H, W_, Cout), np.float16)
for _ in range(iters):
    ctx.conv2d(X, Wt, force_tile=tile, out=O)
ctx.sync()
ctx.timer_start()
for _ in range(iters):
    ctx.conv2d(X, Wt, force_tile=tile, out=O)
ms = ctx.timer_stop() / iters
print(f"conv tile {tile}: {ms*1e3:.1f} us {2.0*N*H*W_*Cout*9*Cin/(ms*1e-3)/1e12:.1f} TF/s")
