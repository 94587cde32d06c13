"""The pixel decoder's MSDeformAttn at the benchmarked shape (4 pictures of 1024^2: 21 504 queries each over 32^2 / 64^2 / 128^2, 8 heads x 32 channels, 4 points):
msda_fused_kernel with 8 or 4 lanes per (query, head) against msda_prepare_kernel + the native-op kernel (include/odise_hip_tools.h
odise_hip_msda_fused_forward).  Offsets of a few pixels around the reference points, like a trained model's.
usage: msda_bench.py [reps=20] [offset_px=4]          (GPU)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd._lib import check  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    px = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    ctx = Context(0)
    rng = np.random.default_rng(0)
    M, hs, ws = 8, (32, 64, 128), (32, 64, 128)
    Lq = sum(a * b for a, b in zip(hs, ws))
    hs3, ws3 = (C.c_int * 3)(*hs), (C.c_int * 3)(*ws)
    for B in (4, 1):
        v = ctx.to_device(rng.standard_normal((B, Lq, M, 32), dtype=np.float32).astype(np.float16))
        off = ctx.to_device((rng.standard_normal((B * Lq, M * 24), dtype=np.float32) * px).astype(np.float32))
        aw = ctx.to_device(rng.standard_normal((B * Lq, M * 12), dtype=np.float32))
        loc, w = ctx.empty((B * Lq * M * 24,), np.float32), ctx.empty((B * Lq * M * 12,), np.float32)
        out = ctx.empty((B * Lq, M * 32), np.float16)
        res = {}
        for rnd in range(3):
            for name, mode in (("fused, 4 lanes x 16 B / pair (default)", 2), ("fused, 8 lanes x 8 B / pair", 1), ("prepare + native op", 0)):
                run = lambda: check(ctx.lib.odise_hip_msda_fused_forward(ctx.h, v, off, aw, hs3, ws3, B, M, mode, out, loc, w), "msda")   # noqa: E731
                run()
                ctx.sync()
                ctx.timer_start()
                for _ in range(reps):
                    run()
                res.setdefault(name, []).append(ctx.timer_stop() / reps * 1e3)
        alg = B * Lq * (256 * 2 * 2 + M * 36 * 4)     # value in + out (fp16) + raw offsets / logits (fp32)
        print(f"B={B}: " + "   ".join(f"{k} {np.median(t):6.1f} us ({alg / np.median(t) / 1e6:5.2f} TB/s of algorithmic bytes)" for k, t in res.items()), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
