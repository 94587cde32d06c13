"""DEBUG: which tile's wave-private epilogue drops elements at 4096x1280x320, and is the pattern deterministic?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context
ctx = Context(0)
rng = np.random.default_rng(0)
import sys as _s
DBG = int(_s.argv[1]) if len(_s.argv) > 1 else 0
MODE = int(_s.argv[2]) if len(_s.argv) > 2 else 0
ctx.lib.odise_hip_gemm_debug((DBG & 3) << 25 | ((DBG >> 2) & 1) << 24)   # bits 0-1: debug variants of the wave epilogue, bit 2: the block-wide epilogues
print('debug bits', DBG, 'mode', MODE)
for (M, N, K) in [(4096, 1280, 320)]:
    A = rng.standard_normal((M, K)).astype(np.float16); W = (rng.standard_normal((N, K)) / K ** 0.5).astype(np.float16)
    ref = A.astype(np.float32) @ W.astype(np.float32).T
    dA, dW = ctx.to_device(A), ctx.to_device(W)
    for tile in (1, 2):
        pats = []
        for rep in range(3):
            O = ctx.empty((M, N), np.float16)
            if MODE == 0:   # background 7.0, sync
                O.copy_from(np.full((M, N), 7.0, np.float16)); ctx.sync()
            elif MODE == 1:  # background 0 (memset), sync
                ctx.lib.odise_hip_memset(ctx.h, O, 0, O.nbytes); ctx.sync()
            elif MODE == 2:  # background 0 (memset), no sync
                ctx.lib.odise_hip_memset(ctx.h, O, 0, O.nbytes)
            elif MODE == 3:  # background 0x4700 (7.0) by memset of bytes 0x47 -> halves 0x4747 = 7.28, no sync
                ctx.lib.odise_hip_memset(ctx.h, O, 0x47, O.nbytes)
            ctx.gemm(dA, dW, force_tile=tile, force_split=1 if tile >= 0 else 0, out=O)
            got = O.numpy().astype(np.float32)
            bad = np.abs(got - ref) > 2e-2 * np.abs(ref).max()
            idx = np.argwhere(bad)
            pats.append((int(bad.sum()), sorted(set((idx[:, 0] % 64).tolist()))[:12], sorted(set((idx[:, 1] % 8).tolist())), sorted(set((idx[:, 0]).tolist()))[:6],
                         sorted(set(np.round(got[bad], 2).tolist()))[:4] if bad.any() else []))
            O.free()
        print(f"{M}x{N}x{K} tile {tile} (ran {ctx.lib.odise_hip_last_tile() & 255}/s{ctx.lib.odise_hip_last_tile() >> 8}):", pats, flush=True)
