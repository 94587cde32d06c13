"""Would the UNet of 16 crops run faster as two halves of 8 crops side by side on two streams?  (The crops are independent; after the VAE
lane is done the main stream idles ~13 ms while the UNet runs alone at ~22 % of the MFMA peak.)  Two contexts, each with its own UNet graph
for 8 crops, replayed concurrently from two host threads, against one context's 16-crop graph.

    python tools/unet_split_bench.py [reps=10]          (GPU)"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from odise_amd.runtime import Context  # noqa: E402
from odise_amd.synthetic import synthetic_state  # noqa: E402
from odise_amd.unet import HipUNet  # noqa: E402


def inputs(ctx, B):
    r = np.random.default_rng(1)
    x = r.standard_normal((B, 4, 64, 64), dtype=np.float32)
    c = (r.standard_normal((1, 77, 768), dtype=np.float32) + 0.1 * r.standard_normal((B, 77, 768), dtype=np.float32)).astype(np.float32)
    e = (0.02 * r.standard_normal((B, 1280), dtype=np.float32)).astype(np.float32)
    return ctx.to_device(x), ctx.to_device(c), ctx.to_device(e)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    state = synthetic_state(["model.diffusion_model."], strip="model.diffusion_model.")
    ctxs = [Context(0), Context(0)]
    nets = [HipUNet(c, state, use_graph=True) for c in ctxs]
    del state

    def timed_one(k, B):
        dx, dc, de = inputs(ctxs[k], B)
        for _ in range(2):
            nets[k].run_nhwc(dx, dc, de, 0)
        ctxs[k].sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            nets[k].run_nhwc(dx, dc, de, 0)
        ctxs[k].sync()
        return (time.perf_counter() - t0) / reps * 1e3

    t16 = timed_one(0, 16)
    t8 = timed_one(0, 8)
    ins = [inputs(ctxs[k], 8) for k in range(2)]
    for k in range(2):
        nets[k].run_nhwc(*ins[k], 0)
        ctxs[k].sync()
    gate = threading.Barrier(3)
    done = [0.0, 0.0]

    def worker(k):
        gate.wait()
        for _ in range(reps):
            nets[k].run_nhwc(*ins[k], 0)
        ctxs[k].sync()
        done[k] = time.perf_counter()

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    t2x8 = (max(done) - t0) / reps * 1e3
    print(f"UNet, 16 crops, one stream: {t16:.2f} ms;  8 crops, one stream: {t8:.2f} ms;  2 x 8 crops on two streams (two weight copies): {t2x8:.2f} ms per pair")
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
