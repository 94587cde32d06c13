"""How many panoptic segments / instances the bench's synthetic model keeps per picture as a function of the calibrated fraction of positive mask
logits (bench.py MASK_POSITIVE): the decision kernels should be timed on tables like a trained model's (5-15 segments per picture at overlap
threshold 0.8), not on one-segment tables (What's weak 4)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402

S, B = 1024, 4
u8 = [bench.image_u8(S, b) for b in range(B)]
for frac in [float(a) for a in sys.argv[1:]] or [0.15, 0.06, 0.03, 0.015]:
    ctx = Context(0)
    hip, pos = bench.calibrated_model(ctx, u8[0], S, 133, 254, set(range(80)), None, positive_fraction=frac, anchor_images=([bench.image_u8(S, s) for s in (1, 2, 3)] if os.environ.get('ANCHOR_ALL') else None))
    res = hip.infer_device([ctx.to_device(u) for u in u8], 0, [(S, S)] * B, [(S, S)] * B, to_host=False)
    segs = [len(r["panoptic_seg"][1]) for r in res]
    inst = [int(len(r["instances"]["scores"])) for r in res]
    stuff = [sum(not s["isthing"] for s in r["panoptic_seg"][1]) for r in res]
    print(f"positive fraction {frac:.3f} (measured {pos:.3f}): segments per picture {segs} (stuff {stuff}), instances {inst}", flush=True)
    del hip, res
    ctx.close()
