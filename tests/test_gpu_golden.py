"""GPU: the HIP mask generator against golden vectors written by the REFERENCE's own modules (tests/golden/make_golden_m2f.py:
MaskFormerHead = MSDeformAttnPixelDecoder + ODISEMultiScaleMaskedTransformerDecoder from /root/reference, run in the build
container).  fp16 MFMA compute against the reference's fp32 output: same tolerances as tests/test_gpu_maskgen.py."""
import glob
import os

import numpy as np
import pytest

from odise_amd.pipeline import HipODISE
from oracle.m2f import SemSegHead, init_synthetic_
from tests.test_gpu_maskgen import _cmp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "m2f_head_*.npz"))))
def test_head_matches_reference_golden(ctx, path):
    z = np.load(path)
    head = init_synthetic_(SemSegHead(small=True, num_classes=int(z["num_classes"]), in_channels=int(z["in_channels"])), seed=int(z["seed"]))
    hip = HipODISE(ctx, {"sem_seg_head." + k: v for k, v in head.state_dict().items()}, with_extractor=False, with_head=True)
    got = hip.head({k[3:]: z[k] for k in z.files if k.startswith("in_")})
    for k in ("mask_embed", "mask_pooled_features", "pred_masks"):
        _cmp(f"{os.path.basename(path)} {k}", got[k], z["out_" + k], tol=5e-2, cos_min=0.995)
    assert abs(got["logit_scale"] - float(z["out_logit_scale"])) < 1e-4
    gm, rm = got["pred_masks"] > 0, z["out_pred_masks"] > 0
    inter, union = (gm & rm).sum(axis=(2, 3)).astype(np.float64), (gm | rm).sum(axis=(2, 3)).astype(np.float64)
    iou = np.where(union > 0, inter / np.maximum(union, 1), 1.0)
    print("binary mask IoU vs reference: mean", iou.mean(), "min", iou.min())
    assert iou.mean() >= 0.98
