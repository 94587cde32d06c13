"""CPU: the drop-in overlay (odise_amd/dropin) against the reference's OWN files - its LazyConfig model definitions, `instantiate_odise`,
its `OpenPanopticInference` wrapper and its checkpoint key layout - with the third-party packages that are absent here stubbed
(tests/dropin_env.py).  What is checked without a GPU: the dotted paths the configs name resolve to this repository's classes, the
reference's keyword arguments construct them, `backbone.output_shape()` / `size_divisibility` feed `instantiate_odise`
(odise/config/instantiate.py:14-21), the attribute tree answers the wrapper's suffix protocol (pano_wrapper.py:36-52), and the state
dict has exactly the trainable keys of an ODISE(label) checkpoint.  The forward passes are in
tests/test_gpu_dropin.py."""
import os
import sys

import pytest
import torch

REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference checkout (build container only)")


@pytest.fixture(scope="module")
def env():
    import dropin_env
    dropin_env.install()
    return dropin_env


@pytest.fixture(scope="module")
def model(env):
    import importlib
    cfg = importlib.import_module("configs.common.models.odise_with_label").model        # the reference's file, unchanged
    from odise.config.instantiate import instantiate_odise                               # the reference's function
    cfg.criterion = None     # training-only (SetCriterion needs the real detectron2 / scipy); the eval path never touches it
    return instantiate_odise(cfg)


def test_overlay_resolves_dotted_paths_and_chains_the_reference(env):
    import odise
    import mask2former
    from odise_amd.dropin import OVERLAY_DIR
    assert odise.__path__[0].startswith(OVERLAY_DIR) and any(p.startswith(REFERENCE) for p in odise.__path__[1:])
    assert mask2former.__path__[0].startswith(OVERLAY_DIR)
    from odise.modeling.meta_arch.odise import CategoryODISE, MaskPooling, PooledMaskEmbed, CategoryEmbed, PoolingCLIPHead, PseudoClassEmbed, \
        ODISEMultiScaleMaskedTransformerDecoder   # noqa: F401
    from odise.modeling.meta_arch.ldm import LdmImplicitCaptionerExtractor
    from odise.modeling.backbone.feature_extractor import FeatureExtractorBackbone
    from mask2former.modeling.meta_arch.mask_former_head import MaskFormerHead
    from mask2former.modeling.pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder
    import MultiScaleDeformableAttention as MSDA
    for obj in (CategoryODISE, LdmImplicitCaptionerExtractor, FeatureExtractorBackbone, MaskFormerHead, MSDeformAttnPixelDecoder, MSDA.ms_deform_attn_forward):
        assert sys.modules[obj.__module__].__file__.startswith(OVERLAY_DIR), obj
    from odise.data.build import get_openseg_labels           # NOT replaced: comes from the reference through the chained path
    assert sys.modules[get_openseg_labels.__module__].__file__.startswith(REFERENCE)
    from odise.modeling.wrapper.pano_wrapper import OpenPanopticInference
    assert sys.modules[OpenPanopticInference.__module__].__file__.startswith(REFERENCE)


def test_reference_config_instantiates_the_overlay_model(model):
    from odise.modeling.meta_arch.odise import CategoryODISE
    assert type(model) is CategoryODISE
    assert model.backbone.size_divisibility == 64
    shp = model.backbone.output_shape()
    assert list(shp) == ["s2", "s3", "s4", "s5"] and [shp[k].stride for k in shp] == [4, 8, 16, 32] and all(shp[k].channels == 512 for k in shp)
    fe = model.backbone.feature_extractor
    assert fe.feature_dims == [512, 512, 2560, 1920, 960, 640, 512, 512] and fe.feature_strides == [4, 8, 64, 32, 16, 8, 8, 4] and fe.num_groups == 8
    assert model.sem_seg_head.num_classes == 133 and model.num_queries == 100 and model.overlap_threshold == 0.8
    assert model.clip_head.alpha == 0.3 and model.clip_head.beta == 0.7                   # odise_with_label.py:28-29 overrides the class defaults
    assert len(model.category_head.labels) == 133 and model.sem_seg_head.predictor.num_queries == 100
    n = sum(p.numel() for p in model.parameters())
    print("trainable parameters", n)
    assert abs(n / 28.1e6 - 1) < 0.02                                                      # README.md:89: 28.1 M trainable parameters


def test_state_dict_has_the_checkpoint_keys(model):
    """Appendix B of SURVEY.md, spelled out through the oracle modules that load into the reference's own classes with strict=True."""
    from oracle.backbone import FeatureExtractorBackbone as OracleBackbone
    from oracle.m2f import SemSegHead
    keys = set(model.state_dict())
    head = {"sem_seg_head." + k for k in SemSegHead(num_classes=133).state_dict()}
    assert head <= keys
    proj = {k for k in keys if k.startswith("backbone.feature_projections.")}
    assert len(proj) == 8 * 9 + 4 * 3 and "backbone.feature_projections.2.0.shortcut.norm.weight" in proj and "backbone.feature_projections.0.0.shortcut.weight" not in proj
    fe = {"backbone.feature_extractor." + k for k in ("clip_project.linear.weight", "clip_project.linear.bias", "clip_project.positional_embedding", "alpha_cond",
                                                      "time_embed_project.linear.weight", "time_embed_project.linear.bias",
                                                      "time_embed_project.positional_embedding", "alpha_cond_time_embed")}
    cat = {"category_head.text_proj.weight", "category_head.text_proj.bias", "category_head.null_embed"}
    assert keys == head | proj | fe | cat, sorted(keys ^ (head | proj | fe | cat))[:10]
    assert model.state_dict()["backbone.feature_extractor.alpha_cond"].shape == (1, 77, 768)
    assert model.state_dict()["backbone.feature_extractor.time_embed_project.positional_embedding"].shape == (1, 1, 1280)
    assert not any("clip." in k or "ldm" in k for k in keys)                              # frozen networks never enter the state dict
    del OracleBackbone


def test_reference_wrapper_drives_the_open_vocabulary_protocol(model):
    """The reference's own OpenPanopticInference.__init__ / forward bookkeeping (pano_wrapper.py:20-68) against the overlay's attribute tree."""
    from odise.modeling.wrapper.pano_wrapper import OpenPanopticInference
    calls = []
    model.forward = lambda batched_inputs: calls.append({k: v for k, v in model.open_state_dict().items()}) or [{"ok": True}]
    labels = [["cat", "kitten"], ["sky"], ["tree"]]
    md = {"thing_ids": [0]}
    before = dict(model.open_state_dict())
    wrapper = OpenPanopticInference(model=model, labels=labels, metadata=md, semantic_on=False, instance_on=True, panoptic_on=True, test_topk_per_image=50)
    wrapper.eval()
    assert wrapper.forward([{"image": torch.zeros(3, 8, 8)}]) == [{"ok": True}]
    seen = calls[0]
    assert seen["category_head.test_labels"] == labels and seen["clip_head.test_labels"] == labels and seen["metadata"] == md
    assert seen["sem_seg_head.num_classes"] == 3 and seen["semantic_on"] is False and seen["test_topk_per_image"] == 50
    assert dict(model.open_state_dict()) == before                                         # restored afterwards
    del model.forward


def test_default_train_labels_and_overlap_mask_match_the_reference(env, monkeypatch):
    """PoolingCLIPHead's seen / unseen split (odise.py:1446-1447, 1479-1491) for the ADE-150 vocabulary against the fixture written by
    tests/golden/make_golden_overlap.py from the reference's label files."""
    import json
    from odise_amd import checkpoint as ck
    monkeypatch.setenv("ODISE_OPENSEG_LABELS", os.path.join(REFERENCE, "odise", "data", "datasets", "openseg_labels"))
    train = ck.default_train_labels()
    assert len(train) == 133 and sum(len(t) for t in train) == 254
    with open(os.path.join(os.path.dirname(__file__), "golden", "overlap_ade150.json")) as f:
        gold = json.load(f)
    test = ck.read_openseg_labels(os.path.join(REFERENCE, "odise", "data", "datasets", "openseg_labels", "ade20k_150_with_prompt_eng.txt"))
    assert ck.category_overlapping_mask(train, test).tolist() == gold["overlap"]
    assert 0 < sum(gold["overlap"]) < 150                                                  # some ADE classes are novel: beta applies to them
    monkeypatch.delenv("ODISE_OPENSEG_LABELS")
    monkeypatch.setattr(sys, "path", [p for p in sys.path if not p.startswith(REFERENCE)])
    with pytest.raises(FileNotFoundError):
        ck.default_train_labels()


def test_pooled_mask_embed_cache_is_not_copied_and_sees_data_swaps(env):
    """the device-weight cache of the overlay PooledMaskEmbed must not travel with copy.deepcopy / pickle, and its key must change
    when a parameter's storage is replaced without a version bump (`param.data = ...`)."""
    import copy
    import pickle
    from odise.modeling.meta_arch.odise import PooledMaskEmbed
    m = PooledMaskEmbed(hidden_dim=32, mask_dim=32, projection_dim=16)
    key = lambda mod: tuple((p._version, p.data_ptr(), p.dtype, str(p.device)) for p in mod.parameters())
    k0 = key(m)
    m._dev = (k0, object(), {"sentinel": 1})
    c = copy.deepcopy(m)
    assert getattr(c, "_dev", None) is None and m._dev[2] == {"sentinel": 1}
    m2 = pickle.loads(pickle.dumps(m))
    assert getattr(m2, "_dev", None) is None
    p = m.pool_proj[1].weight
    v = p._version
    p.data = p.data.clone()                       # storage swapped, version unchanged
    assert p._version == v and key(m) != k0
