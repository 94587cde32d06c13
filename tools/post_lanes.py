"""Post-processing of one batch of 4 x 1024x1024 pictures (three heads, decisions on the device) with the per-image decision chain (mask
statistics, segment walk, record write, instance top-k) on the second lane beside the semantic GEMM, against everything on one stream
(odise_hip_set_lanes(ctx, 1)).  Same process, alternating; HIP events on the context's stream (the call joins the second lane before it returns)."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from odise_amd.pipeline import HipCategoryODISE  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402
from odise_amd.synthetic import synthetic_state, synthetic_vocabulary  # noqa: E402

ctx = Context(0)
hip = HipCategoryODISE(ctx, synthetic_state(), overlap_threshold=0.8)
K, B, S = 133, 4, 1024
cat, clp, sizes, overlap = synthetic_vocabulary(K, 254, 768)
hip.set_vocabulary(cat, clp, sizes, overlap, set(range(80)), 0.3, 0.7)
rng = np.random.default_rng(0)
imgs = [ctx.to_device(rng.integers(0, 256, (S, S, 3), dtype=np.uint8)) for _ in range(B)]
hw = [(S, S)] * B
hip.infer_device(imgs, 0, hw, hw, to_host=False)
mc = rng.standard_normal((B, 100, K + 1)).astype(np.float32) * 3
mc = mc - np.log(np.exp(mc).sum(-1, keepdims=True))
dmc = ctx.to_device(mc)
for r in range(3):
    for lanes in (2, 1):
        assert ctx.lib.odise_hip_set_lanes(ctx.h, lanes) == 0
        hip.postprocess_batch(dmc, (S, S), hw, hw, to_host=False)
        ctx.sync()
        ctx.timer_start()
        for _ in range(10):
            hip.postprocess_batch(dmc, (S, S), hw, hw, to_host=False)
        ms = ctx.timer_stop() / 10
        print(f"round {r} lanes {lanes}: post-processing of {B} x {S}x{S}: {ms*1e3:8.1f} us", flush=True)
