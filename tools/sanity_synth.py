"""Sanity of the product-side synthetic weights: one full forward, outputs must be finite and non-degenerate."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.pipeline import HipCategoryODISE
from odise_amd.runtime import Context
from odise_amd.synthetic import synthetic_state, synthetic_vocabulary
ctx = Context(0)
hip = HipCategoryODISE(ctx, synthetic_state(), overlap_threshold=0.8)
cat, clp, sizes, ov = synthetic_vocabulary()
hip.set_vocabulary(cat, clp, sizes, ov, set(range(80)), 0.3, 0.7)
H, W, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 1024, 1)
img = (np.random.default_rng(0).random((3, H, W), dtype=np.float32) * 255).astype(np.uint8)
res = hip.forward([{"image": img, "height": H // 2, "width": W // 2} for _ in range(B)])
r = res[-1]
print("image", H, W, "batch", B, "outputs", r["sem_seg"].shape, r["panoptic_seg"][0].shape)
sem = r["sem_seg"]
print("sem_seg finite", np.isfinite(sem).all(), "range", float(sem.min()), float(sem.max()), "classes argmax", len(np.unique(sem.argmax(0))))
print("segments", len(r["panoptic_seg"][1]), "instances", len(r["instances"]["scores"]), "scores finite", np.isfinite(r["instances"]["scores"]).all())
