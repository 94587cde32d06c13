"""GPU parity of the mask generator stages against the CPU oracle:
  * FeatureExtractorBackbone (oracle/backbone.py <- odise/modeling/backbone/feature_extractor.py:139-250), incl. slide-window stitching
  * MaskFormerHead: pixel decoder + masked transformer decoder (oracle/m2f.py <- M2F msdeformattn.py:314-358, odise.py:642-776)

Tolerances: backbone maps and continuous head outputs: max|err| <= 2e-2 * max|ref|, cosine >= 0.999; binary masks (pred_masks > 0):
per-query IoU >= 0.98 on average (thresholded logits near zero can legitimately flip under fp16)."""
import numpy as np
import pytest
import torch

from odise_amd._lib import check
from odise_amd.pipeline import HipODISE
from oracle.backbone import FeatureExtractorBackbone, crop_boxes
from oracle.ldm_extractor import ImplicitCaptionerExtractor
from oracle.m2f import SemSegHead, init_synthetic_

pytestmark = pytest.mark.gpu
torch.set_num_threads(min(16, torch.get_num_threads()))

SMALL = dict(unet_div=5, vae_div=4, clip_kw=dict(image_size=336, patch_size=14, width=128, layers=2, heads=2, output_dim=64))


def _image(batch, h, w, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 3, h, w, generator=g)
    x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (4, 4, 4, 4), mode="reflect"), 9, stride=1)
    return (x - x.amin()) / (x.amax() - x.amin())


def _cmp(name, g, r, tol=2e-2, cos_min=0.999):
    r = np.asarray(r, np.float64)
    g = np.asarray(g, np.float64)
    assert g.shape == r.shape, (name, g.shape, r.shape)
    assert np.isfinite(g).all(), name
    scale = np.abs(r).max()
    err = np.abs(g - r).max() / scale
    cos = float((g * r).sum() / (np.linalg.norm(g) * np.linalg.norm(r) + 1e-30))
    print(f"{name:28s} {str(g.shape):22s} max|ref| {scale:9.3f} max-err/scale {err:.3e} cos {cos:.6f}")
    assert err <= tol and cos >= cos_min, (name, err, cos)


def test_crop_boxes_match_reference_windows():
    assert crop_boxes(1024, 1024) == [(0, 0, 512, 512), (0, 512, 512, 1024), (512, 0, 1024, 512), (512, 512, 1024, 1024)]
    starts = sorted({b[0] for b in crop_boxes(1280, 1280)})
    assert starts == [0, 512, 768] and len(crop_boxes(1280, 1280)) == 9   # SURVEY.md §8c golden: 9 boxes, starts 0,512,768


@pytest.fixture(scope="module")
def small_models(ctx):
    ext = ImplicitCaptionerExtractor(**SMALL)
    dims = [128 // 4 * 4, 128 // 4 * 4, 8 * 64, 6 * 64, 3 * 64, 2 * 64, 128 // 4 * 4, 128 // 4 * 4]   # tap channels of the narrow nets
    bb = FeatureExtractorBackbone(ext, dims)
    head = init_synthetic_(SemSegHead(small=True))
    state = ext.export_state()
    state.update({"backbone.feature_projections." + k: v for k, v in bb.feature_projections.state_dict().items()})
    state.update({"sem_seg_head." + k: v for k, v in head.state_dict().items()})
    hip = HipODISE(ctx, state)
    return bb, head, hip


# (1280, 1280): the nine overlapping 512-windows of BASELINE configs[4] (starts 0, 512, 768); the last three: windows below 512 (bicubic)
@pytest.mark.parametrize("h,w", [(512, 512), (1024, 512), (768, 640), (1280, 1280), (320, 448), (256, 256), (384, 1024)])
def test_small_backbone_slide_forward(small_models, h, w):
    bb, _, hip = small_models
    img = _image(1 if (h, w) != (512, 512) else 2, h, w, seed=h + w)
    ref = bb(img)
    got = hip.backbone(img.numpy())
    for k in ("s2", "s3", "s4", "s5"):
        _cmp(f"backbone {h}x{w} {k}", got[k], ref[k].numpy())


def test_small_head_from_given_features(small_models):
    _, head, hip = small_models
    g = torch.Generator().manual_seed(5)
    feats = {k: torch.randn(2, 512, 64 >> i, 64 >> i, generator=g) for i, k in enumerate(("s2", "s3", "s4", "s5"))}
    ref = head(feats)
    got = hip.head({k: v.numpy() for k, v in feats.items()})
    _cmp("head mask_embed", got["mask_embed"], ref["mask_embed"].numpy(), tol=5e-2, cos_min=0.995)
    _cmp("head mask_pooled_features", got["mask_pooled_features"], ref["mask_pooled_features"].numpy(), tol=5e-2, cos_min=0.995)
    _cmp("head pred_masks", got["pred_masks"], ref["pred_masks"].numpy(), tol=5e-2, cos_min=0.995)
    assert abs(got["logit_scale"] - float(ref["logit_scale"])) < 1e-4
    gm, rm = got["pred_masks"] > 0, ref["pred_masks"].numpy() > 0
    inter = (gm & rm).sum(axis=(2, 3)).astype(np.float64)
    union = (gm | rm).sum(axis=(2, 3)).astype(np.float64)
    iou = np.where(union > 0, inter / np.maximum(union, 1), 1.0)
    print("binary mask IoU: mean", iou.mean(), "min", iou.min())
    assert iou.mean() >= 0.98


def test_small_backbone_then_head_end_to_end(small_models):
    bb, head, hip = small_models
    img = _image(1, 512, 1024, seed=9)
    ref = head(bb(img))
    hip.backbone(img.numpy())
    got = hip.head(None, image_hw=(1, 512, 1024))
    _cmp("e2e pred_masks", got["pred_masks"], ref["pred_masks"].numpy(), tol=8e-2, cos_min=0.99)
    _cmp("e2e mask_embed", got["mask_embed"], ref["mask_embed"].numpy(), tol=8e-2, cos_min=0.99)


def test_failed_head_reload_keeps_the_previous_head(small_models):
    """reload_head() with an incomplete checkpoint (a missing key used to leave the context without a head): the library assembles
    the new weights on the side and swaps them in only when every tensor was found, so the previous head keeps answering - bit for bit - and a
    complete reload afterwards works; the same for the tap projections."""
    _, head, hip = small_models
    g = torch.Generator().manual_seed(11)
    feats = {k: torch.randn(1, 512, 64 >> i, 64 >> i, generator=g).numpy() for i, k in enumerate(("s2", "s3", "s4", "s5"))}
    before = hip.head(feats)
    full = {"sem_seg_head." + k: v for k, v in head.state_dict().items()}
    broken = {k: v for k, v in full.items() if "post_mask_embed.logit_scale" not in k}     # the LAST tensor the build asks for: everything else uploads first
    with pytest.raises(RuntimeError, match="logit_scale"):
        hip.reload_head(broken)
    hip.ctx.lib.odise_hip_clear_host_weights(hip.ctx.h)
    after = hip.head(feats)
    for k in ("pred_masks", "mask_embed", "mask_pooled_features"):
        assert np.array_equal(before[k], after[k]), k
    hip.reload_head(full)
    again = hip.head(feats)
    for k in ("pred_masks", "mask_embed", "mask_pooled_features"):
        assert np.array_equal(before[k], again[k]), k
    with pytest.raises(RuntimeError):      # no backbone.feature_projections.* in the host store: the projections in place must survive
        check(hip.ctx.lib.odise_hip_backbone_build(hip.ctx.h), "backbone_build")
    img = _image(1, 256, 256, seed=3)
    assert np.isfinite(hip.backbone(img.numpy())["s2"]).all()
