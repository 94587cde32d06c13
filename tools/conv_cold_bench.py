"""The VAE encoder's stride-2 downsampling convolution (128 -> 128 channels, 16 crops x 512^2 -> 256^2; F.pad (0,1,0,1) + conv3x3 stride 2,
ldm Downsample) with its gigabyte of input COLD (a 2 GB fill runs between the launches, as the ResBlocks before it do in the step) and warm
(back-to-back launches: the input stays in the 256 MB MALL / L2), per tile of the library.

    python tools/conv_cold_bench.py [reps=5]          (GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from odise_amd.runtime import Context  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    ctx = Context(0)
    rng = np.random.default_rng(0)
    N, H, C = 16, 512, 128
    x1 = rng.standard_normal((1, H, H, C), dtype=np.float32).astype(np.float16)
    X = ctx.to_device(np.broadcast_to(x1, (N, H, H, C)).copy())
    Wt = ctx.to_device((rng.standard_normal((C, 3, 3, C), dtype=np.float32) * (9 * C) ** -0.5).astype(np.float16))
    O = ctx.empty((N, H // 2, H // 2, C), np.float16)
    import ctypes as C_
    junk = ctx.empty((1 << 31,), np.uint8)

    def conv(tile):
        ctx.conv2d(X, Wt, stride=2, pad=0, pad_tl=(0, 0), out_hw=(H // 2, H // 2), force_tile=tile, out=O)

    flops = 2.0 * N * (H // 2) ** 2 * C * 9 * C
    print(f"stride-2 conv 16 x 512^2 x 128 -> 256^2 x 128; us per launch (TFLOP/s)")
    for tile in (-1, 0, 3, 4, 5, 6):
        try:
            conv(tile)
            ctx.sync()
        except Exception as e:
            print(f"  tile {tile}: n/a ({str(e)[:60]})")
            continue
        ctx.timer_start()
        for _ in range(reps):
            conv(tile)
        warm = ctx.timer_stop() / reps * 1e3
        cold = []
        for _ in range(reps):
            ctx.lib.odise_hip_memset(ctx.h, C_.c_void_p(junk.ptr), 0, C_.c_size_t(1 << 31))
            ctx.sync()
            ctx.timer_start()
            conv(tile)
            cold.append(ctx.timer_stop() * 1e3)
        print(f"  tile {tile:2d} (ran on {ctx.lib.odise_hip_last_tile() & 255}): warm {warm:7.1f} ({flops / warm / 1e6:5.0f})   cold {np.median(cold):7.1f} ({flops / np.median(cold) / 1e6:5.0f})")
    ctx.close()


if __name__ == "__main__":
    main()
