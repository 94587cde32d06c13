"""The synthetic weights bench.py runs on have the REAL architecture's names and shapes (BASELINE.json: "random-init weights of that
architecture"): odise_amd/weight_spec.json against the full-size oracle modules built on torch's meta device (shapes only, no memory),
and against the published parameter counts of SD v1 (UNet 859.52 M, VAE 83.65 M) and CLIP ViT-L/14@336 (visual tower 304.3 M)."""
import numpy as np
import torch

from odise_amd.synthetic import load_spec, synthetic_state
from oracle.clip_vit import CLIPVisual
from oracle.m2f import SemSegHead
from oracle.sd_unet import UNetModel
from oracle.sd_vae import AutoencoderKL


def _meta_shapes(ctor):
    with torch.device("meta"):
        m = ctor()
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def test_spec_matches_full_size_architecture():
    spec = load_spec()
    groups = {"model.diffusion_model.": _meta_shapes(UNetModel), "first_stage_model.": _meta_shapes(AutoencoderKL),
              "clip.": _meta_shapes(CLIPVisual), "sem_seg_head.": _meta_shapes(lambda: SemSegHead(num_classes=133))}
    for prefix, shapes in groups.items():
        got = {k[len(prefix):]: tuple(v[0]) for k, v in spec.items() if k.startswith(prefix)}
        assert set(got) == set(shapes), (prefix, sorted(set(got) ^ set(shapes))[:5])
        for k in shapes:
            assert got[k] == shapes[k], (prefix + k, got[k], shapes[k])
    params = lambda p: sum(int(np.prod(v[0])) for k, v in spec.items() if k.startswith(p))
    assert abs(params("model.diffusion_model.") / 1e6 - 859.52) < 0.01
    assert abs(params("first_stage_model.") / 1e6 - 83.65) < 0.01
    assert abs(params("clip.visual.") / 1e6 - 304.29) < 0.05
    for k in ("backbone.feature_extractor.clip_project.linear.weight", "backbone.feature_extractor.alpha_cond", "category_head.text_proj.weight",
              "category_head.null_embed", "backbone.feature_extractor.ldm_extractor.shared_noise", "backbone.feature_projections.2.0.conv1.weight"):
        assert k in spec, k
    assert tuple(spec["backbone.feature_projections.2.0.conv1.weight"][0]) == (128, 2560, 1, 1)      # tap u2: 2560 -> 128 -> 512


def test_synthetic_tensors_are_reproducible_and_follow_the_spec():
    a = synthetic_state(["sem_seg_head.predictor.query_feat."])
    b = synthetic_state(["sem_seg_head.predictor."])
    k = "sem_seg_head.predictor.query_feat.weight"
    np.testing.assert_array_equal(a[k], b[k])                             # per-tensor seeds: any subset gives the same values
    shape, mean, std = load_spec()[k]
    assert a[k].shape == tuple(shape) and a[k].dtype == np.float32
    assert abs(a[k].mean() - mean) < 0.05 * max(std, 1e-6) + 1e-3 and abs(a[k].std() - std) < 0.05 * std
