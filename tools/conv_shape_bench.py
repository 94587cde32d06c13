"""One convolution shape under every tile the library can be forced to (odise_hip_conv2d_forced), next to the cost model's own choice.
usage: conv_shape_bench.py N H W Cin Cout k stride [pad_t pad_l OH OW]          (GPU)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.runtime import Context  # noqa: E402

NAMES = {0: "128x128", 1: "64x128", 2: "64x64", 3: "256x320", 4: "256x256", 5: "256x128", 6: "512x128", 7: "halo 256x256", 8: "halo 256x128", 9: "halo4 256x128"}


def main():
    a = [int(x) for x in sys.argv[1:]]
    N, H, W, Cin, Cout, k, stride = a[:7]
    pad = k // 2
    kw = {}
    if len(a) >= 11:
        kw = dict(pad_tl=(a[7], a[8]), out_hw=(a[9], a[10]))
    ctx = Context(0)
    rng = np.random.default_rng(0)
    X = ctx.to_device(rng.standard_normal((N, H, W, Cin), dtype=np.float32).astype(np.float16))
    Wt = ctx.to_device((rng.standard_normal((Cout, k, k, Cin), dtype=np.float32) * (k * k * Cin) ** -0.5).astype(np.float16))
    ref = None
    for tile in (-1, 0, 1, 3, 4, 5, 6, 7, 8, 9):
        try:
            O = ctx.conv2d(X, Wt, stride=stride, pad=pad, force_tile=tile, **kw)
        except RuntimeError as e:
            print(f"tile {tile:2d} {NAMES.get(tile, 'cost model'):14s}: refused ({str(e)[-60:]})")
            continue
        got = ctx.lib.odise_hip_last_tile()
        ctx.sync()
        ctx.timer_start()
        it = 5
        for _ in range(it):
            ctx.conv2d(X, Wt, stride=stride, pad=pad, force_tile=tile, out=O, **kw)
        us = ctx.timer_stop() / it * 1e3
        o = O.numpy()
        if ref is None:
            ref = o
        flops = 2.0 * O.shape[0] * O.shape[1] * O.shape[2] * Cout * k * k * Cin
        print(f"tile {tile:2d} {NAMES.get(tile, 'cost model'):14s}: ran tile {got & 255} split {got >> 8}: {us:8.1f} us = {flops / us / 1e6:6.0f} TFLOP/s   max diff vs first {float(np.abs(o.astype(np.float32) - ref.astype(np.float32)).max()):.3g}", flush=True)
        O.free()
    ctx.close()


if __name__ == "__main__":
    main()
