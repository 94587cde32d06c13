"""Per-pixel post-processing pass (odise.py:326-331 + maskformer_model.py:286-320) of one 1024x1024 image, tiled form against the thread-per-
cell-column form (odise_hip_post_generic(2)) and the generic kernel (1): kernel time and effective HBM rate of the 218 MB pixel-major sigmoid
matrix.  Narrow model (the tests' SMALL configuration) with the REAL head geometry: 100 queries, 256x256 mask logits."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from odise_amd.pipeline import HipCategoryODISE  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402
from odise_amd.synthetic import synthetic_state, synthetic_vocabulary  # noqa: E402

ctx = Context(0)
state = synthetic_state()
hip = HipCategoryODISE(ctx, state, overlap_threshold=0.8)
del state
K = 133
cat, clp, sizes, overlap = synthetic_vocabulary(K, 254, 768)
hip.set_vocabulary(cat, clp, sizes, overlap, set(range(80)), 0.3, 0.7)
S = 1024
img = np.random.default_rng(0).integers(0, 256, (S, S, 3), dtype=np.uint8)
d = ctx.to_device(img)
res = hip.infer_device([d], 0, [(S, S)], [(S, S)], to_host=False)
rng = np.random.default_rng(1)
mc = rng.standard_normal((1, 100, K + 1)).astype(np.float32) * 3
mc = mc - np.log(np.exp(mc).sum(-1, keepdims=True))
dmc = ctx.to_device(mc)
hip.instance_on = False
for mode, name in ((0, "tiled"), (2, "cell-column"), (1, "generic")):
    ctx.lib.odise_hip_post_generic(mode)
    for sem in (True, False):
        hip.semantic_on = sem
        hip.postprocess_batch(dmc, (S, S), (S, S), [(S, S)], to_host=False)
        ctx.sync()
        ctx.timer_start()
        for _ in range(10):
            hip.postprocess_batch(dmc, (S, S), (S, S), [(S, S)], to_host=False)
        ms = ctx.timer_stop() / 10
        print(f"{name:12s} semantic={'on ' if sem else 'off'} post-processing of one {S}x{S} image: {ms*1e3:8.1f} us", flush=True)
ctx.lib.odise_hip_post_generic(0)
hip.semantic_on = True
for tile in (-1, 1, 5, 4):
    ctx.lib.odise_hip_sem_tile(tile if tile >= 0 else -1)
    if tile == 1:
        pass
    hip.postprocess_batch(dmc, (S, S), (S, S), [(S, S)], to_host=False)
    ctx.sync()
    ctx.timer_start()
    for _ in range(10):
        hip.postprocess_batch(dmc, (S, S), (S, S), [(S, S)], to_host=False)
    ms = ctx.timer_stop() / 10
    print(f"semantic GEMM tile {tile:2d} (-1 = rule: 256x128): post-processing of one {S}x{S} image, semantic + panoptic: {ms*1e3:8.1f} us", flush=True)
ctx.lib.odise_hip_sem_tile(-1)
