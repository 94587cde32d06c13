"""Golden fixture for the sliding-window backbone, produced by the REFERENCE's own `FeatureExtractorBackbone`
(odise/modeling/backbone/feature_extractor.py: `slide_forward`, `single_forward`, `forward_features`) in the build container.

The diffusion feature extractor it wraps needs the `ldm` package (absent), so a small stand-in with the same interface
(`feature_dims`, `feature_strides`, `grouped_indices`, `__call__(dict(img=...)) -> list of maps`) feeds both the reference class and
the oracle's; what is pinned is the reference's window placement, resize / restore of aspect ratio, per-tap projection and sum, overlap
averaging.  detectron2's `BottleneckBlock` / `ResNet.make_stage` and torchvision's `Resize` are restated in tests/golden/ref_stubs.py.

    python tests/golden/make_golden_backbone.py
"""
import os
import sys

import numpy as np
import torch
from torch import nn

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_stubs  # noqa: E402

ref_stubs.install()
from odise.modeling.backbone.feature_extractor import FeatureExtractorBackbone as RefBackbone  # noqa: E402

ref_stubs.seal()
from oracle.backbone import FeatureExtractorBackbone, TapStandIn  # noqa: E402

DIMS = [16, 16, 32, 24, 16, 16, 16, 16]


def case(name, seed, B, H, W, crop):
    ext = TapStandIn(DIMS, seed)
    oracle = FeatureExtractorBackbone(ext, DIMS, projection_dim=256, backbone_in_size=(crop, crop), seed=seed + 1)
    ref = RefBackbone(ext, out_features=["s2", "s3", "s4", "s5"], backbone_in_size=(crop, crop), slide_training=True, projection_dim=256,
                      num_res_blocks=1).eval()
    print(name, ref.feature_projections.load_state_dict(oracle.feature_projections.state_dict(), strict=True))
    img = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(seed + 2))
    with torch.no_grad():
        out = ref(img)
    print(name, {k: tuple(v.shape) for k, v in out.items()})
    np.savez_compressed(os.path.join(HERE, f"backbone_{name}.npz"), image=img.numpy(), seed=np.int64(seed), crop=np.int64(crop),
                        **{"out_" + k: (v.numpy().astype(np.float16) if k == "s2" else v.numpy()) for k, v in out.items()})   # s2 (the large one) as fp16


if __name__ == "__main__":
    case("a", seed=3, B=1, H=80, W=144, crop=64)     # 2 x 3 windows, the last row / column overlap their neighbours
    case("b", seed=8, B=2, H=64, W=64, crop=64)      # one window, no resize
    case("c", seed=9, B=1, H=48, W=80, crop=64)      # image smaller than the crop: 48-pixel windows resized (bicubic) to 64
