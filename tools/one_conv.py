"""Run one conv shape repeatedly (target for rocprofv3 --pmc).  usage: one_conv.py tile [iters] [N H Cin Cout]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, H, Cin, Cout = (int(a) for a in sys.argv[3:7]) if len(sys.argv) > 6 else (16, 64, 320, 320)
ctx = Context(0)
rng = np.random.default_rng(0)
X = ctx.to_device((rng.standard_normal((N, H, H, Cin), dtype=np.float32)).astype(np.float16))
Wt = ctx.to_device((rng.standard_normal((Cout, 3, 3, Cin), dtype=np.float32) * (9 * Cin) ** -0.5).astype(np.float16))
O = ctx.empty((N, H, H, Cout), np.float16)
for _ in range(iters):
    ctx.conv2d(X, Wt, force_tile=tile, out=O)
ctx.sync()
ctx.timer_start()
for _ in range(iters):
    ctx.conv2d(X, Wt, force_tile=tile, out=O)
ms = ctx.timer_stop() / iters
print(f"conv {N}x{H}x{H} {Cin}->{Cout} tile {tile} (ran on tile {ctx.lib.odise_hip_last_tile() & 255}): {ms*1e3:.1f} us {2.0*N*H*H*Cout*9*Cin/(ms*1e-3)/1e12:.1f} TF/s")
