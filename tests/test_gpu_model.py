"""GPU parity of the classification + post-processing stages and of the whole CategoryODISE eval forward against the CPU oracle
(oracle/odise_model.py <- odise.py:181-207, 282-372, 1469-1542; clip.py:252-361; helper.py:79-109; maskformer_model.py:280-380).

Tolerances: class probabilities exp(mask_cls) within 2e-2 absolute; sem_seg within 2e-2 of its max; panoptic maps agree on >= 99.5% of
the pixels with identical segments_info; instances: same (class, query) set for confidently separated scores, mask IoU >= 0.99."""
import numpy as np
import pytest
import torch

from odise_amd.pipeline import HipCategoryODISE
from oracle import odise_model as om
from oracle.backbone import FeatureExtractorBackbone
from oracle.ldm_extractor import ImplicitCaptionerExtractor
from oracle.m2f import SemSegHead, init_synthetic_
from margins import instance_keys

pytestmark = pytest.mark.gpu
torch.set_num_threads(min(16, torch.get_num_threads()))

SMALL = dict(unet_div=5, vae_div=4, clip_kw=dict(image_size=336, patch_size=14, width=128, layers=2, heads=2, output_dim=64))
GROUPS = [1, 2, 1, 3, 1, 1, 2, 1, 1, 2, 1]
THINGS = {0, 2, 3, 5, 8}


def _image_u8(h, w, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(1, 3, h, w, generator=g)
    x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (4, 4, 4, 4), mode="reflect"), 9, stride=1)
    x = (x - x.amin()) / (x.amax() - x.amin())
    return (x[0] * 255).round().to(torch.uint8)


@pytest.fixture(scope="module")
def models(ctx):
    ext = ImplicitCaptionerExtractor(**SMALL)
    bb = FeatureExtractorBackbone(ext, [128, 128, 512, 384, 192, 128, 128, 128])
    head = init_synthetic_(SemSegHead(small=True, num_classes=len(GROUPS)))
    heads = om.OpenVocabHeads(ext.clip, GROUPS, projection_dim=64)
    state = ext.export_state()
    state.update({"backbone.feature_projections." + k: v for k, v in bb.feature_projections.state_dict().items()})
    state.update({"sem_seg_head." + k: v for k, v in head.state_dict().items()})
    state["category_head.text_proj.weight"] = heads.text_proj.weight.detach()
    state["category_head.text_proj.bias"] = heads.text_proj.bias.detach()
    state["category_head.null_embed"] = heads.null_embed.detach()
    hip = HipCategoryODISE(ctx, state, overlap_threshold=0.0)
    hip.set_vocabulary(heads.text_embed.numpy(), heads.clip_text_embed.numpy(), GROUPS, heads.category_overlapping_mask.numpy(), THINGS,
                       heads.alpha, heads.beta)
    return bb, head, heads, hip


def _oracle_forward(bb, head, heads, img_u8, out_hw, overlap_threshold):
    img = img_u8.float()[None] / 255.0
    H, W = img.shape[-2:]
    Hp, Wp = (H + 63) // 64 * 64, (W + 63) // 64 * 64
    padded = torch.zeros(1, 3, Hp, Wp)
    padded[:, :, :H, :W] = img
    outputs = head(bb(padded))
    mask_cls = heads.classify(outputs, img)
    res = om.postprocess(mask_cls, outputs["pred_masks"], (Hp, Wp), [(H, W)], [out_hw], len(GROUPS), THINGS, overlap_threshold)
    return mask_cls, outputs, res[0]


def _oracle_forward_batch(bb, head, heads, imgs_u8, overlap_threshold):
    """The reference's batching: pad to the batch maximum rounded up to 64 for the backbone, to the batch maximum for MaskCLIP."""
    H, W = max(i.shape[-2] for i in imgs_u8), max(i.shape[-1] for i in imgs_u8)
    Hp, Wp = (H + 63) // 64 * 64, (W + 63) // 64 * 64
    padded, den = torch.zeros(len(imgs_u8), 3, Hp, Wp), torch.zeros(len(imgs_u8), 3, H, W)
    for i, im in enumerate(imgs_u8):
        padded[i, :, :im.shape[-2], :im.shape[-1]] = im.float() / 255.0
        den[i, :, :im.shape[-2], :im.shape[-1]] = im.float() / 255.0
    outputs = head(bb(padded))
    mask_cls = heads.classify(outputs, den)
    sizes = [tuple(i.shape[-2:]) for i in imgs_u8]
    return mask_cls, outputs, om.postprocess(mask_cls, outputs["pred_masks"], (Hp, Wp), sizes, sizes, len(GROUPS), THINGS, overlap_threshold)


@pytest.mark.parametrize("h,w,oh,ow", [(512, 512, 512, 512), (512, 704, 256, 352), (300, 400, 300, 400)])   # the last: short side below 512
def test_full_forward_matches_oracle(models, h, w, oh, ow):
    bb, head, heads, hip = models
    img = _image_u8(h, w, seed=h + w)
    ref_cls, ref_out, ref = _oracle_forward(bb, head, heads, img, (oh, ow), 0.0)
    got = hip.forward([{"image": img, "height": oh, "width": ow}])[0]
    # semantic
    sem_ref = ref["sem_seg"].numpy()
    err = np.abs(got["sem_seg"] - sem_ref).max() / np.abs(sem_ref).max()
    print("sem_seg", got["sem_seg"].shape, "max-err/scale", err)
    assert got["sem_seg"].shape == sem_ref.shape and err < 2e-2
    agree = (got["sem_seg"].argmax(0) == sem_ref.argmax(0)).mean()
    print("semantic argmax agreement", agree)
    assert agree > 0.99
    # panoptic
    pan_ref, info_ref = ref["panoptic_seg"]
    pan, info = got["panoptic_seg"]
    print("segments", info, "ref", info_ref)
    assert pan.dtype == np.int32 and pan.shape == tuple(pan_ref.shape)
    assert info == info_ref
    agree = (pan == pan_ref.numpy()).mean()
    print("panoptic pixel agreement", agree)
    assert agree > 0.995
    # instances (maskformer_model.py:344-380) by (query, class) key: the same entries as the reference's top-k wherever the k-th score is
    # separated; an entry may only be missing / extra at the selection boundary; common entries carry the same mask and score
    inst_ref, inst = ref["instances"], got["instances"]
    assert inst["pred_masks"].shape[1:] == (oh, ow)
    K = len(GROUPS)
    key_ref, kth, scores_flat = instance_keys(ref_cls[0], K, THINGS, hip.test_topk_per_image)
    key_got = {(int(q), int(c)): i for i, (q, c) in enumerate(zip(inst["query_index"], inst["pred_classes"]))}
    assert len(key_got) == len(inst["query_index"]), "duplicate (query, class) entries"
    common = sorted(set(key_ref) & set(key_got))
    s_ref = inst_ref["scores"].numpy()
    worst_iou, worst_score = 1.0, 0.0
    for kk in common:
        a, b = inst["pred_masks"][key_got[kk]] > 0.5, inst_ref["pred_masks"][key_ref[kk]].numpy() > 0.5
        worst_iou = min(worst_iou, (a & b).sum() / max((a | b).sum(), 1))
        worst_score = max(worst_score, abs(float(inst["scores"][key_got[kk]]) - float(s_ref[key_ref[kk]])))
    boundary = [abs(float(scores_flat[q * K + c]) - kth) for q, c in set(key_ref) ^ set(key_got)]
    print("instances", len(key_got), "ref", len(key_ref), "in common", len(common), "k-th class score", kth, "entries off the common set", len(boundary),
          "their distance to the k-th score", max(boundary, default=0.0), "worst mask IoU", worst_iou, "worst score diff", worst_score)
    assert all(d < 2e-2 for d in boundary), "an instance entry differs away from the selection boundary of the top-k"
    assert len(common) >= 0.8 * len(key_ref) and worst_iou > 0.97 and worst_score < 2e-2


def test_classification_stage(models, ctx):
    bb, head, heads, hip = models
    img = _image_u8(512, 512, seed=3)
    ref_cls, ref_out, _ = _oracle_forward(bb, head, heads, img, (512, 512), 0.0)
    img01 = (img.float()[None] / 255.0).numpy()
    hip.backbone_device(ctx.to_device(img01), want_outputs=False)
    hip.head_device(None, 1, 128, 128)
    cls_dev, ce = hip.classify_device(ctx.to_device(img01), want_clip_embed=False), None
    got = cls_dev.numpy()
    p_ref, p_got = np.exp(ref_cls.numpy()), np.exp(got)
    print("class prob max abs err", np.abs(p_got - p_ref).max(), "argmax agreement", (p_got.argmax(-1) == p_ref.argmax(-1)).mean())
    assert np.abs(p_got - p_ref).max() < 2e-2
    assert (p_got.argmax(-1) == p_ref.argmax(-1)).mean() >= 0.95


@pytest.mark.parametrize("h,w", [(500, 502), (512, 512), (320, 1100)])   # ragged widths: a partial last tile, a single-cell tail
def test_postprocess_x4_kernel_matches_generic(models, h, w):
    """The x4-upsampling specialisations of the per-pixel pass and of the instance masks (output size = image size) must reproduce
    the generic kernels bit for bit, including ragged widths (ow % 4 != 0) and the clamped border taps."""
    bb, head, heads, hip = models
    img = _image_u8(h, w, seed=11)
    fast = hip.forward([{"image": img}])[0]                 # x4 specialisations, per-pixel pass in its tiled form (256-pixel row tiles through LDS)
    other = {}
    for mode in (1, 2):                                      # 1: generic kernels; 2: x4 specialisations with the thread-per-cell-column per-pixel pass
        hip.ctx.lib.odise_hip_post_generic(mode)
        try:
            other[mode] = hip.forward([{"image": img}])[0]
        finally:
            hip.ctx.lib.odise_hip_post_generic(0)
    np.testing.assert_array_equal(other[1]["sem_seg"], other[2]["sem_seg"])     # the two GEMM-fed forms: bit for bit
    for mode, gen in other.items():
        np.testing.assert_array_equal(fast["panoptic_seg"][0], gen["panoptic_seg"][0])
        assert fast["panoptic_seg"][1] == gen["panoptic_seg"][1]
        # the tiled pass produces the semantic scores itself (round 6: MFMA on the tile's S rows in LDS, v_mfma_f32_32x32x16_f16 chain over the
        # queries); the other two forms run the semantic GEMM (16x16x32 MFMAs): the same fp16 products, summed in fp32 in another grouping
        np.testing.assert_allclose(fast["sem_seg"], gen["sem_seg"], rtol=0, atol=2e-5)
        assert np.array_equal(fast["sem_seg"].argmax(0), gen["sem_seg"].argmax(0)) or (fast["sem_seg"].argmax(0) != gen["sem_seg"].argmax(0)).mean() < 1e-4
        np.testing.assert_array_equal(fast["instances"]["pred_masks"], gen["instances"]["pred_masks"])
        np.testing.assert_array_equal(fast["instances"]["scores"], gen["instances"]["scores"])


@pytest.mark.parametrize("h,w,oh,ow", [(512, 512, 512, 512), (500, 502, 500, 502), (512, 704, 256, 352), (300, 400, 450, 600)])
def test_fused_semantic_argmax_equals_argmax_of_the_score_tensor(models, h, w, oh, ow):
    """`semantic_argmax_kernel` (the [K,h,w]-free form of maskformer_model.py:280-284 + the evaluator's `.argmax(0)`, what
    `bench.py --semantic-only` runs) against the arg-max of the device's own score tensor: the same fp16 operands through the same MFMA
    k order, so every pixel must agree bit for bit - x4 and generic resampling geometries, ragged widths, K not a multiple of 32, and the
    first-maximum tie rule of `argmax`."""
    _, _, _, hip = models
    img = _image_u8(h, w, seed=5 + h)
    batch = [{"image": img, "height": oh, "width": ow}]
    scores = hip.forward(batch)[0]["sem_seg"]
    hip.semantic_argmax = True
    try:
        fused = hip.forward(batch)[0]
    finally:
        hip.semantic_argmax = False
    assert "sem_seg" not in fused and fused["sem_seg_argmax"].shape == (oh, ow) and fused["sem_seg_argmax"].dtype == np.int32
    want = scores.argmax(0)
    print("fused semantic arg-max", (oh, ow), "labels", np.unique(want).tolist(), "mismatches", int((fused["sem_seg_argmax"] != want).sum()))
    np.testing.assert_array_equal(fused["sem_seg_argmax"], want)


def test_panoptic_record_written_into_a_caller_owned_buffer(models):
    """The multi-GPU step (bench.py --gpus N) lets the device write every image's prediction record - panoptic map AND segment table -
    straight into this rank's slice of the all-gather buffer: same map, same segments as the host-returning call."""
    from odise_amd import distributed as D
    _, _, _, hip = models
    img = _image_u8(512, 512, seed=3)
    ref = hip.forward([{"image": img}])[0]
    rec = hip.ctx.zeros((1, D.record_size(512, 512)), np.int32)
    d = hip.ctx.to_device(img.numpy())
    out = hip.infer_device([d], 1, [(512, 512)], [(512, 512)], to_host=False, pan_out=[rec.ptr])[0]
    assert out["panoptic_seg"] == (None, None)
    seg, info = D.unpack_record(torch.from_numpy(rec.numpy()[0]), 512, 512)
    np.testing.assert_array_equal(seg, ref["panoptic_seg"][0])
    assert info == ref["panoptic_seg"][1] and len(info) >= 1


def test_unequal_image_sizes_in_one_batch(models):
    """ImageList.from_tensors pads a batch to its largest image (odise.py:238-244): a mixed batch must reproduce the results of each
    image's own padded call - the network sees the same padded tensors either way."""
    bb, head, heads, hip = models
    a, b = _image_u8(512, 704, seed=21), _image_u8(576, 512, seed=22)
    both = hip.forward([{"image": a}, {"image": b}])
    # reference for image a inside the batch canvas (576 x 704 -> 576 x 704 padded to 576 x 704): run it alone on the same canvas
    for im, got in zip((a, b), both):
        h, w = im.shape[-2:]
        assert got["sem_seg"].shape == (len(GROUPS), h, w) and got["panoptic_seg"][0].shape == (h, w)
        assert got["instances"]["pred_masks"].shape[1:] == (h, w)
    ref_cls, ref_out, ref = _oracle_forward_batch(bb, head, heads, [a, b], 0.0)
    for i, got in enumerate(both):
        sem_ref = ref[i]["sem_seg"].numpy()
        err = np.abs(got["sem_seg"] - sem_ref).max() / np.abs(sem_ref).max()
        agree = (got["panoptic_seg"][0] == ref[i]["panoptic_seg"][0].numpy()).mean()
        print("mixed batch image", i, "sem_seg err", err, "panoptic agreement", agree, got["panoptic_seg"][1], ref[i]["panoptic_seg"][1])
        assert err < 2e-2 and agree > 0.995 and got["panoptic_seg"][1] == ref[i]["panoptic_seg"][1]


def test_caption_variant_matches_oracle(ctx):
    """(Keep this test LAST in the module: it loads another model into the shared context, replacing the weights of the `models` fixture.)
    CaptionODISE eval forward (odise.py:545-619): learned (object, no-object) class head + word bank without a null embedding."""
    from odise_amd.pipeline import HipCaptionODISE
    ext = ImplicitCaptionerExtractor(**SMALL)
    bb = FeatureExtractorBackbone(ext, [128, 128, 512, 384, 192, 128, 128, 128])
    head = init_synthetic_(SemSegHead(small=True, num_classes=1, learned_class_embed=True), seed=778)
    heads = om.OpenVocabHeads(ext.clip, GROUPS, projection_dim=64)
    state = ext.export_state()
    state.update({"backbone.feature_projections." + k: v for k, v in bb.feature_projections.state_dict().items()})
    state.update({"sem_seg_head." + k: v for k, v in head.state_dict().items()})
    assert "sem_seg_head.predictor.class_embed.weight" in state
    state["word_head.text_proj.weight"] = heads.text_proj.weight.detach()
    state["word_head.text_proj.bias"] = heads.text_proj.bias.detach()
    hip = HipCaptionODISE(ctx, state, overlap_threshold=0.0)
    hip.set_vocabulary(heads.text_embed.numpy(), heads.clip_text_embed.numpy(), GROUPS, heads.category_overlapping_mask.numpy(), THINGS,
                       heads.alpha, heads.beta)
    img = _image_u8(512, 512, seed=33)
    imgf = img.float()[None] / 255.0
    outputs = head(bb(imgf))
    assert outputs["pred_logits"].shape[-1] == 2
    ref_cls = om.caption_classify(heads, outputs, imgf)
    ref = om.postprocess(ref_cls, outputs["pred_masks"], (512, 512), [(512, 512)], [(512, 512)], len(GROUPS), THINGS, 0.0)[0]
    # classification stage
    img01 = imgf.numpy()
    hip.backbone_device(ctx.to_device(img01), want_outputs=False)
    hip.head_device(None, 1, 128, 128)
    got_cls = hip.classify_device(ctx.to_device(img01)).numpy()
    p_ref, p_got = np.exp(ref_cls.numpy()), np.exp(got_cls)
    print("caption class prob max abs err", np.abs(p_got - p_ref).max(), "no-object prob range", p_ref[..., -1].min(), p_ref[..., -1].max())
    assert np.abs(p_got - p_ref).max() < 2e-2
    # whole forward
    got = hip.forward([{"image": img}])[0]
    pan_ref, info_ref = ref["panoptic_seg"]
    pan, info = got["panoptic_seg"]
    agree = (pan == pan_ref.numpy()).mean()
    print("segments", info, "ref", info_ref, "agreement", agree)
    assert info == info_ref and agree > 0.995
