"""CLIP-tower attention in isolation (577 / 677 queries x 577 keys, d_head 64, 16 heads): the K/V-resident kernel against the tiled one.
usage: attn_bench.py [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    ctx = Context(0)
    rng = np.random.default_rng(0)
    for name, B, Lq, Lk, masked in (("crops' tower, 16 crops", 16, 577, 577, False), ("crops' tower, 32 crops", 32, 577, 577, False),
                                    ("crops' tower, 18 crops (2 x 1280^2)", 18, 577, 577, False), ("MaskCLIP, 4 pictures", 4, 677, 577, True),
                                    ("MaskCLIP, 8 pictures", 8, 677, 577, True),
                                    ("MaskCLIP mask tokens, 4 pictures", 4, 100, 577, True), ("MaskCLIP mask tokens, 8 pictures", 8, 100, 577, True)):
        H, D = 16, 64
        HD = H * D
        q = ctx.to_device(rng.standard_normal((B, Lq, HD), dtype=np.float32).astype(np.float16))
        k = ctx.to_device(rng.standard_normal((B, Lk, HD), dtype=np.float32).astype(np.float16))
        vt = ctx.to_device(rng.standard_normal((B, HD, 584), dtype=np.float32).astype(np.float16))
        m = None
        if masked:
            m8 = np.zeros((B, Lq, 580), np.uint8)
            if Lq > 577:
                m8[:, 577:, :577] = rng.random((B, Lq - 577, 577)) < 0.7
            else:
                m8[:, :, 1:577] = rng.random((B, Lq, 576)) < 0.7
            m = ctx.to_device(m8)
        o = ctx.empty((B, Lq, HD), np.float16)
        flops = 4.0 * B * H * Lq * Lk * D
        line = f"{name:38s} B={B:2d} Lq={Lq} Lk={Lk}:"
        for on in (1, 0):
            ctx.set_option(ctx.OPT_ATTN_KV_RESIDENT, 0 if on else 2)
            for _ in range(5):
                ctx.attention(q, k, vt, H, D ** -0.5, mask=m, Lk=Lk, out=o)
            ctx.sync()
            ctx.timer_start()
            for _ in range(reps):
                ctx.attention(q, k, vt, H, D ** -0.5, mask=m, Lk=Lk, out=o)
            us = ctx.timer_stop() / reps * 1e3
            line += f"  {'library rule (K/V-resident where it applies)' if on else 'tiled'} {us:7.1f} us = {flops / us / 1e6:6.1f} TFLOP/s"
        ctx.set_option(ctx.OPT_ATTN_KV_RESIDENT, 0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
