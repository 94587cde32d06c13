"""Time backbone + head (mask generator) on full-size synthetic weights. usage: pipeline_sweep.py 1,4 [H] [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from odise_amd.pipeline import HipODISE  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402
from oracle.backbone import FeatureExtractorBackbone  # noqa: E402
from oracle.ldm_extractor import ImplicitCaptionerExtractor  # noqa: E402
from oracle.m2f import SemSegHead, init_synthetic_  # noqa: E402

batches = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1,4").split(",")]
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.set_num_threads(16)
ext = ImplicitCaptionerExtractor()
bb = FeatureExtractorBackbone(ext, [512, 512, 2560, 1920, 960, 640, 512, 512])
head = init_synthetic_(SemSegHead())
state = ext.export_state()
state.update({"backbone.feature_projections." + k: v for k, v in bb.feature_projections.state_dict().items()})
state.update({"sem_seg_head." + k: v for k, v in head.state_dict().items()})
ctx = Context(0)
hip = HipODISE(ctx, state)
del ext, bb, head, state
for B in batches:
    img = ctx.to_device(np.random.default_rng(0).random((B, 3, H, H), dtype=np.float32))
    for part in ("backbone", "backbone+head"):
        def run():
            hip.backbone_device(img, want_outputs=False)
            if part != "backbone":
                hip.head_device(None, B, H // 4, H // 4)
        run(); run()
        ctx.sync()
        ctx.timer_start()
        for _ in range(steps):
            run()
        ms = ctx.timer_stop() / steps
        print(f"{part:14s} B={B} {H}x{H}: {ms:9.2f} ms/step  {B / ms * 1e3:7.2f} img/s", flush=True)
