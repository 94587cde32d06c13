"""A/B of the feature extractor's two-lane schedule (CLIP -> UNet on a second stream beside the VAE) on the benchmarked step:
bs = 4 x 1024x1024, full path.  usage: lanes_ab.py [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from odise_amd.pipeline import HipCategoryODISE  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402
from odise_amd.synthetic import synthetic_state, synthetic_vocabulary  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = Context(0)
hip = HipCategoryODISE(ctx, synthetic_state(), overlap_threshold=0.8)
cat, clp, sizes, overlap = synthetic_vocabulary(133, 254, 768)
hip.set_vocabulary(cat, clp, sizes, overlap, set(range(80)), 0.3, 0.7)
imgs = [ctx.to_device(bench.image_u8(1024, b)) for b in range(4)]
hw = [(1024, 1024)] * 4
ref = None
for rnd in range(2):
    for lanes in (1, 2):
        assert ctx.lib.odise_hip_set_lanes(ctx.h, lanes) == 0
        for _ in range(2):
            res = hip.infer_device(imgs, 0, hw, hw, to_host=False)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = hip.infer_device(imgs, 0, hw, hw, to_host=False)
        ctx.sync()
        dt = (time.perf_counter() - t0) / steps
        pan = res[0]["panoptic_seg"][0].numpy()
        sem = res[0]["sem_seg"].view((133, 64, 1024), np.float32).numpy() if False else None
        if ref is None:
            ref = pan
        print(f"lanes {lanes}: {dt * 1e3:7.2f} ms/step  {4 / dt:6.2f} images/s   panoptic map identical to the first run: {bool((pan == ref).all())}", flush=True)
