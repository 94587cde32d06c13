"""Attention kernel timing on the pipeline's shapes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context
ctx = Context(0)
rng = np.random.default_rng(0)
def rand(shape, s=1.0): return ctx.to_device((rng.standard_normal(shape, dtype=np.float32) * s).astype(np.float16))
def timeit(fn, it=5, rounds=3):
    best = 1e9
    for r in range(rounds):
        for _ in range(2): fn()
        ctx.sync(); ctx.timer_start()
        for _ in range(it): fn()
        best = min(best, ctx.timer_stop() / it)
    return best * 1e3
for (B, H, Lq, Lk, D) in [(16, 8, 4096, 4096, 40), (16, 8, 1024, 1024, 80), (16, 8, 256, 256, 160), (16, 16, 577, 577, 64), (4, 16, 677, 677, 64), (4, 8, 100, 16384, 32), (4, 8, 100, 4096, 32), (16, 8, 4096, 77, 40)]:
    Q, K = rand((B, Lq, H * D)), rand((B, Lk, H * D))
    ldvt = (Lk + 7) // 8 * 8
    Vt = rand((B, H * D, ldvt))
    O = ctx.empty((B, Lq, H * D), np.float16)
    us = timeit(lambda: ctx.attention(Q, K, Vt, H, D ** -0.5, Lk=Lk, out=O))
    fl = 4.0 * B * H * Lq * Lk * D
    print(f"attn B{B} H{H} Lq{Lq} Lk{Lk} D{D}: {us:8.1f} us {fl/us/1e6:7.1f} TF/s", flush=True)
