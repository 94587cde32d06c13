"""GPU: the drop-in overlay (odise_amd/dropin) at the benchmarked configuration.  The model is built from this repository's classes at the
reference's dotted paths with the keyword arguments of the reference's model config (configs/common/models/odise_with_label.py +
mask_generator_with_label.py; that those files instantiate exactly this is tests/test_dropin_cpu.py, which needs the reference checkout),
loaded through `load_state_dict` with the reference's key names, driven through the open-vocabulary wrapper protocol
(odise/modeling/wrapper/pano_wrapper.py:36-68), and every replaced class is also called on its own against the oracle.
Shares the oracle pass of tests/test_gpu_fullsize.py (tests/fullsize.py)."""
import operator
import sys

import numpy as np
import pytest
import torch

from fullsize import build_models, export_state, reference
from odise_amd import dropin
from oracle import odise_model as om

pytestmark = pytest.mark.gpu
torch.set_num_threads(min(32, torch.get_num_threads()))
K, K_TOT = 133, 254
THINGS = list(range(80))


def _rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / np.abs(ref).max())


class _BankEncoder:
    """Stands in for tokenizer + CLIP text tower (no BPE merges file in this image): prompt strings -> the rows of the test's text banks."""

    def __init__(self, table):
        self.table = table

    def tokenize(self, strings):
        return list(strings)

    def build_text_embed(self, strings):
        return np.stack([self.table[s] for s in strings]).astype(np.float32)


@pytest.fixture(scope="module")
def env(ctx):
    if dropin.OVERLAY_DIR not in sys.path:
        sys.path.insert(0, dropin.OVERLAY_DIR)
    for m in [m for m in sys.modules if m.split(".")[0] in ("odise", "mask2former", "MultiScaleDeformableAttention")]:
        del sys.modules[m]
    from mask2former.modeling.meta_arch.mask_former_head import MaskFormerHead
    from mask2former.modeling.pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder
    from odise.modeling.backbone.feature_extractor import FeatureExtractorBackbone
    from odise.modeling.meta_arch.ldm import LdmImplicitCaptionerExtractor
    from odise.modeling.meta_arch.odise import (CategoryEmbed, CategoryODISE, ODISEMultiScaleMaskedTransformerDecoder, PooledMaskEmbed, PoolingCLIPHead,
                                                PseudoClassEmbed)
    ext, bb, head = build_models(K)
    img, heads, r = reference(bb, head, ext, 1024, K, K_TOT)
    state = export_state(ext, bb, head, heads)
    dropin.set_context(ctx)
    dropin.set_frozen_state({k: v for k, v in state.items() if k.startswith(("model.diffusion_model.", "first_stage_model.", "clip.", "backbone.feature_extractor.ldm_extractor."))})
    # synthetic label names; prompt strings of both heads map to the rows of the spread banks (tests/fullsize.py)
    labels, table, row = [], {}, 0
    for k, n in enumerate(heads.group_sizes):
        names = [f"class{k}_{j}" for j in range(n)]
        labels.append(names)
        for s in names:
            table[s] = heads.text_embed[row].numpy()
            table[f"a photo of a {s}."] = heads.clip_text_embed[row].numpy()
            row += 1
    enc = _BankEncoder(table)
    dropin.set_text_tools(enc.tokenize, enc)
    # ---- the model exactly as configs/common/models/odise_with_label.py spells it (odise_amd/dropin/zoo.py)
    from odise_amd.dropin import zoo
    model = zoo.category_odise_with_label(labels, THINGS, heads.category_overlapping_mask.tolist())
    zoo.load_flat_state(model, state)
    model.eval()
    return dict(model=model, labels=labels, img=img, heads=heads, r=r, ext=ext, bb=bb, head=head)


def test_model_through_the_wrapper_protocol(env):
    """What OpenPanopticInference does around `self.model(batched_inputs)` (pano_wrapper.py:36-68): enumerate open_state_dict() keys by
    suffix, set them with load_open_state_dict, call the model, restore."""
    model, labels, img, r = env["model"], env["labels"], env["img"], env["r"]
    want = {}
    for k in model.open_state_dict():
        if k.endswith("test_labels"):
            want[k] = labels
        elif k.endswith("metadata"):
            want[k] = {"thing_ids": THINGS}
        elif k.endswith("num_classes"):
            want[k] = len(labels)
        elif k.endswith(("semantic_on", "instance_on", "panoptic_on")):
            want[k] = True
        elif k.endswith("test_topk_per_image"):
            want[k] = 100
    assert {k.rsplit(".", 1)[-1] for k in want} == {"test_labels", "metadata", "num_classes", "semantic_on", "instance_on", "panoptic_on", "test_topk_per_image"}
    saved = model.open_state_dict()
    model.load_open_state_dict(want)
    try:
        with torch.no_grad():
            out = model([{"image": img, "height": 1024, "width": 1024}])[0]
    finally:
        model.load_open_state_dict(saved)
    assert operator.attrgetter("category_head.test_labels")(model) is None                     # restored
    ref = om.postprocess(r["mask_cls"], r["pred_masks"], (1024, 1024), [(1024, 1024)], [(1024, 1024)], K, set(THINGS), 0.8)[0]
    assert out["sem_seg"].dtype == torch.float32 and tuple(out["sem_seg"].shape) == (K, 1024, 1024)
    pan, info = out["panoptic_seg"]
    assert pan.dtype == torch.int32 and info == ref["panoptic_seg"][1]
    agree = float((pan == ref["panoptic_seg"][0]).float().mean())
    serr = _rel(out["sem_seg"].numpy(), ref["sem_seg"].numpy())
    inst = out["instances"]
    print("wrapper call: segments", info, "panoptic agreement", agree, "sem_seg err", serr, "instances", len(inst.scores))
    assert agree > 0.995 and serr < 3e-2
    assert inst.pred_masks.shape[1:] == (1024, 1024) and inst.pred_boxes.shape == (len(inst.scores), 4) and inst.pred_classes.dtype == torch.int64
    assert abs(len(inst.scores) - len(ref["instances"]["scores"])) <= 3


def test_model_on_the_device_writes_its_results_in_place(env):
    """`model.to("cuda")` (what tools/train_net.py does before evaluation): the library writes `sem_seg`, the panoptic map and the instance
    masks into torch tensors on that device (no host round trip), pictures may be device tensors too, and the values are those of the
    host-edge path bit for bit (same kernels, other output buffers)."""
    if not torch.cuda.is_available():
        pytest.skip("torch sees no ROCm device")
    model, labels, img = env["model"], env["labels"], env["img"]
    saved = model.open_state_dict()
    want = {k: (labels if k.endswith("test_labels") else len(labels) if k.endswith("num_classes") else v) for k, v in saved.items()}
    model.load_open_state_dict(want)
    try:
        with torch.no_grad():
            host = model([{"image": img, "height": 1024, "width": 1024}])[0]
            model.to("cuda")
            assert model.device.type == "cuda"
            dev = model([{"image": img, "height": 1024, "width": 1024}])[0]                      # host picture, device results
            dev2 = model([{"image": img.to("cuda"), "height": 1024, "width": 1024}])[0]          # device picture
    finally:
        model.to("cpu")
        model.load_open_state_dict(saved)
    for out in (dev, dev2):
        assert out["sem_seg"].is_cuda and out["panoptic_seg"][0].is_cuda and out["instances"].pred_masks.is_cuda
        assert out["sem_seg"].dtype == torch.float32 and out["panoptic_seg"][0].dtype == torch.int32
        assert out["panoptic_seg"][1] == host["panoptic_seg"][1]
        assert torch.equal(out["sem_seg"].cpu(), host["sem_seg"]) and torch.equal(out["panoptic_seg"][0].cpu(), host["panoptic_seg"][0])
        assert torch.equal(out["instances"].pred_masks.cpu(), host["instances"].pred_masks) and torch.equal(out["instances"].scores.cpu(), host["instances"].scores)
        assert torch.equal(out["instances"].pred_classes.cpu(), host["instances"].pred_classes)


def test_backbone_and_extractor_stand_alone(env):
    model, img, r, ext = env["model"], env["img"], env["r"], env["ext"]
    img01 = img.float()[None] / 255.0
    feats = model.backbone(img01)
    assert list(feats) == ["s2", "s3", "s4", "s5"]
    for k in feats:
        e = _rel(feats[k].numpy(), r[k].numpy())
        print("backbone()", k, tuple(feats[k].shape), e)
        assert e < 1e-2
    crop = img01[:, :, :512, :512]
    taps = model.backbone.feature_extractor(dict(img=crop))
    with torch.no_grad():
        taps_ref = ext(crop)
    assert len(taps) == 8
    for i, (g, t) in enumerate(zip(taps, taps_ref)):
        assert g.shape == t.shape and g.shape[1] == model.backbone.feature_extractor.feature_dims[i]
        assert _rel(g.numpy(), t.numpy()) < 2e-2, i


def _regular_queries(got, ref, tau=2.5e-2):
    """Queries of [1, Q, h, w] mask logits whose worst pixel stays within tau of max|ref|."""
    err = np.abs(got - ref).reshape(got.shape[1], -1).max(1) / np.abs(ref).max()
    return int((err < tau).sum())


def test_head_halves_stand_alone(env):
    model, r, head = env["model"], env["r"], env["head"]
    feats = {k: r[k] for k in ("s2", "s3", "s4", "s5")}
    with torch.no_grad():
        mf_ref, _, ms_ref = head.pixel_decoder.forward_features(feats)
    mf, enc0, ms = model.sem_seg_head.pixel_decoder.forward_features(feats)
    print("pixel decoder: mask_features", _rel(mf.numpy(), mf_ref.numpy()), "multi-scale", [_rel(a.numpy(), b.numpy()) for a, b in zip(ms, ms_ref)])
    assert _rel(mf.numpy(), mf_ref.numpy()) < 1e-2 and all(_rel(a.numpy(), b.numpy()) < 1e-2 for a, b in zip(ms, ms_ref)) and enc0 is ms[0]
    out = model.sem_seg_head.predictor(list(ms_ref), mf_ref)                                 # the predictor alone, on the ORACLE's pixel-decoder outputs
    e = _rel(out["pred_masks"].numpy(), r["pred_masks"].numpy())
    print("predictor alone: pred_masks", e, "mask_embed", _rel(out["mask_embed"].numpy(), r["mask_embed"].numpy()))
    assert _regular_queries(out["pred_masks"].numpy(), r["pred_masks"].numpy()) >= 98 and e < 6e-2 and _rel(out["mask_embed"].numpy(), r["mask_embed"].numpy()) < 5e-2
    assert set(out) >= {"pred_logits", "pred_masks", "aux_outputs", "mask_embed", "mask_pooled_features", "logit_scale"}
    assert tuple(out["pred_logits"].shape) == (1, 100, K + 1) and float(out["logit_scale"]) == pytest.approx(100.0, rel=1e-3)
    whole = model.sem_seg_head(feats)
    # (a query sent the other way at one of the decoder's hard decisions is a draw, tests/test_gpu_fullsize.py: >= 98 regular queries, the rest bounded)
    assert _regular_queries(whole["pred_masks"].numpy(), r["pred_masks"].numpy()) >= 98 and _rel(whole["pred_masks"].numpy(), r["pred_masks"].numpy()) < 6e-2


def test_pooling_modules_stand_alone(env):
    from odise.modeling.meta_arch.odise import MaskPooling
    from oracle.m2f import MaskPooling as OracleMaskPooling, PooledMaskEmbed as OraclePooled
    model, r, head = env["model"], env["r"], env["head"]
    g = torch.Generator().manual_seed(5)
    x, mask = torch.randn(2, 256, 64, 64, generator=g), torch.randn(2, 100, 64, 64, generator=g)
    got = MaskPooling()(x, mask)["mask_pooled_features"]
    ref = OracleMaskPooling()(x, mask)["mask_pooled_features"]
    assert got.shape == ref.shape and _rel(got.numpy(), ref.numpy()) < 2e-3
    pme, pme_ref = model.sem_seg_head.predictor.post_mask_embed, head.predictor.post_mask_embed
    dec = torch.randn(2, 100, 256, generator=g)
    with torch.no_grad():
        want = pme_ref(dec, None, x, None, mask)
    have = pme(dec, None, x, None, mask)
    for k in ("mask_embed", "mask_pooled_features"):
        print("PooledMaskEmbed", k, _rel(have[k].numpy(), want[k].numpy()))
        assert _rel(have[k].numpy(), want[k].numpy()) < 5e-3
    assert float(have["logit_scale"]) == pytest.approx(float(want["logit_scale"]))
    del OraclePooled


def test_text_heads_and_msda_stand_alone(env):
    import MultiScaleDeformableAttention as MSDA
    from oracle.msda import make_inputs, msda_forward_torch
    model, labels, r, heads, img = env["model"], env["labels"], env["r"], env["heads"], env["img"]
    value, shp, start, loc, w = make_inputs(2, 8, 32, 300, [(16, 16), (8, 8), (4, 4)], 4, seed=3, value_scale=1.0)
    got = MSDA.ms_deform_attn_forward(value, shp, start, loc, w, 128)
    assert _rel(got.numpy(), msda_forward_torch(value.double(), shp, start, loc, w).numpy()) < 1e-4
    model.category_head.test_labels = labels
    model.clip_head.test_labels = labels
    try:
        ch = model.category_head({})
        with torch.no_grad():
            assert _rel(ch["text_embed"].numpy(), heads.text_proj(heads.text_embed).numpy()) < 1e-5 and ch["labels"] is labels
            pred_logits = om.cal_pred_logits(r["mask_embed"], heads.text_proj(heads.text_embed), heads.text_proj(heads.null_embed), r["logit_scale"], heads.group_sizes)
            clip_logits = om.mask_clip_pred_logits(r["clip_embed"], heads.clip_text_embed, heads.group_sizes)
            want = om.pooling_clip_head(pred_logits[..., :-1], clip_logits, heads.category_overlapping_mask, 0.3, 0.7)
        outputs = {"images": img.float()[None] / 255.0, "pred_masks": r["pred_masks"], "pred_open_logits": pred_logits[..., :-1].clone()}
        got = model.clip_head(outputs)["pred_open_logits"]
        perr = float((got.softmax(-1) - want.softmax(-1)).abs().max())
        print("PoolingCLIPHead stand-alone: ensemble probability error", perr)
        assert got.shape == want.shape and perr < 3e-2
    finally:
        model.category_head.test_labels = model.clip_head.test_labels = None
