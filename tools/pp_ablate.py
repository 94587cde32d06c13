"""Ablation of the ping-pong GEMM: full / no DMA / no DMA + no fragment reads / main loop only.  usage: pp_ablate.py [tile=4]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bn = 320 if tile == 3 else 256
ctx = Context(0)
rng = np.random.default_rng(0)
def rand(shape, s=1.0): return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))
def timeit(fn, it=8):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(it): fn()
    return ctx.timer_stop() / it
for (M, N, K) in [(65536, 2 * bn, 4096), (65536, 2 * bn, 640)]:
    A, W, O = rand((M, K)), rand((N, K), K ** -0.5), ctx.empty((M, N), np.float16)
    for name, dbg in (("full", 0), ("no store", 8), ("loop only", 4), ("loop, no dma", 5), ("loop, no dma/frag", 7)):
        ctx.lib.odise_hip_gemm_debug(dbg)
        ms = timeit(lambda: ctx.gemm(A, W, force_tile=tile, out=O))
        print(f"M={M} N={N} K={K} tile {tile} {name:18s}: {ms*1e3:8.1f} us {2.0*M*N*K/(ms*1e-3)/1e12:7.1f} TF/s", flush=True)
    ctx.lib.odise_hip_gemm_debug(0)
    A.free(); W.free(); O.free()
X = rand((16, 128, 128, 512)); Wt = rand((2 * bn, 3, 3, 512), 4608 ** -0.5); O = ctx.empty((16, 128, 128, 2 * bn), np.float16)
for name, dbg in (("full", 0), ("loop only", 4), ("loop, no dma", 5), ("loop, no dma/frag", 7), ("tap-major", 16)):
    ctx.lib.odise_hip_gemm_debug(dbg)
    ms = timeit(lambda: ctx.conv2d(X, Wt, force_tile=tile, out=O))
    print(f"conv 16x128x128 512->{2*bn} tile {tile} {name:18s}: {ms*1e3:8.1f} us {2.0*16*128*128*2*bn*4608/(ms*1e-3)/1e12:7.1f} TF/s", flush=True)
ctx.lib.odise_hip_gemm_debug(0)
