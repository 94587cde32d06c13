"""The SD UNet's self-attention launches in isolation (8 heads; 64^2 tokens at d_head 40, 32^2 at 80, 16^2 at 160): the software-pipelined kernel
(attn.hip attn_sa_kernel, round 6) against the tiled one (ODISE_OPT_ATTN_KV_RESIDENT = 4), bit-compared, interleaved rounds.
usage: attn_unet_bench.py [reps=20]          (GPU)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = Context(0)
    rng = np.random.default_rng(0)
    H = 8
    for B in (16, 4, 1):
        for L, D in ((4096, 40), (1024, 80), (256, 160)):
            HD = H * D
            q = ctx.to_device(rng.standard_normal((B, L, HD), dtype=np.float32).astype(np.float16))
            k = ctx.to_device(rng.standard_normal((B, L, HD), dtype=np.float32).astype(np.float16))
            vt = ctx.to_device(rng.standard_normal((B, HD, L), dtype=np.float32).astype(np.float16))
            outs, us = {}, {}
            for rnd in range(3):
                for name, opt in (("pipelined", 0), ("tiled", 4)):
                    ctx.set_option(ctx.OPT_ATTN_KV_RESIDENT, opt)
                    o = ctx.attention(q, k, vt, H, D ** -0.5)
                    ctx.sync()
                    ctx.timer_start()
                    for _ in range(reps):
                        ctx.attention(q, k, vt, H, D ** -0.5, out=o)
                    us.setdefault(name, []).append(ctx.timer_stop() / reps * 1e3)
                    outs[name] = o.numpy()
                    o.free()
            ctx.set_option(ctx.OPT_ATTN_KV_RESIDENT, 0)
            flops = 4.0 * B * H * L * L * D
            same = np.array_equal(outs["pipelined"], outs["tiled"])
            a, b = float(np.median(us["pipelined"])), float(np.median(us["tiled"]))
            print(f"B={B:2d} tokens {L:4d} d_head {D:3d}: library rule {a:7.1f} us ({flops / a / 1e6:6.1f} TFLOP/s)   tiled kernel {b:7.1f} us ({flops / b / 1e6:6.1f})   "
                  f"bit-identical {same}   finite {bool(np.isfinite(outs['pipelined'].astype(np.float32)).all())}", flush=True)
            for x in (q, k, vt):
                x.free()
    ctx.close()


if __name__ == "__main__":
    main()
