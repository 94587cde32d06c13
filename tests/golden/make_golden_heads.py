"""Golden fixtures for the open-vocabulary classification and post-processing stages, produced by the REFERENCE's own
`CategoryODISE.forward` (odise/modeling/meta_arch/odise.py:236-372) in the build container.

What runs is the reference's code: `CategoryODISE.forward` eval branch (ImageList padding, `CategoryEmbed.forward`, `cal_pred_logits`,
`ensemble_logits_with_labels`, `PoolingCLIPHead.forward`, `MaskCLIP.get_mask_embed / encode_image_with_mask / _mask_clip_forward /
pred_logits`, the null-probability merge, mask upsampling, `sem_seg_postprocess`) and Mask2Former's `semantic_inference /
panoptic_inference / instance_inference` (third_party/Mask2Former/mask2former/maskformer_model.py:280-381), on top of
`MaskFormerHead` as in make_golden_m2f.py.  What is substituted, because the packages are absent here:
  * the backbone: a stand-in that returns seeded s2..s5 features (they are stored in the fixture: they are INPUTS of the stages pinned);
  * open_clip's model inside `MaskCLIP`: the oracle's `VisualTransformer` (same attribute names: conv1, class_embedding,
    positional_embedding, ln_pre, transformer(x, attn_mask), ln_post, proj; it is cross-checked against HF separately) - the mask-token
    construction, attention-mask layout, pooling and logits are the reference's code;
  * CLIP text embeddings: seeded tensors placed into the heads' `_test_text_embed_dict` caches under the keys the reference computes
    (`to_tuple(prompt_labels(labels, prompt))`), so `build_clip_text_embed` is never reached;
  * detectron2 helpers: tests/golden/ref_stubs.py.
The heads are created with `__new__` + `nn.Module.__init__` because their constructors download CLIP.

    python tests/golden/make_golden_heads.py
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_stubs  # noqa: E402

ref_stubs.install()
import odise.modeling.meta_arch.odise as ro  # noqa: E402
from odise.modeling.meta_arch.clip import MaskCLIP  # noqa: E402
from odise.data.build import prompt_labels  # noqa: E402
from make_golden_m2f import reference_head  # noqa: E402  (seals the stubs)

from oracle import clip_vit, odise_model as om  # noqa: E402
from oracle.m2f import SemSegHead, init_synthetic_  # noqa: E402


class SeededBackbone(nn.Module):
    size_divisibility = 64

    def __init__(self, feats):
        super().__init__()
        self.feats = feats

    def forward(self, x):
        assert x.shape[-2] // 4 == self.feats["s2"].shape[-2] and x.shape[-1] // 4 == self.feats["s2"].shape[-1]
        return self.feats


class FakeOpenClip(nn.Module):
    """The attributes of open_clip's CLIP that MaskCLIP touches."""

    def __init__(self, visual):
        super().__init__()
        self.visual = visual
        self.logit_scale = nn.Parameter(torch.tensor(float(np.log(100.0))))


def normalize_only(image):   # clip_preprocess on an image that already has the tower's size: Resize and CenterCrop are identities
    mean = torch.tensor(clip_vit.CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(clip_vit.CLIP_STD).view(1, 3, 1, 1)
    return (image - mean) / std


def case(name, seed, sizes, out_sizes, labels, things, train_labels, in_channels=32, topk=30, overlap_threshold=0.8, caption=False, spread=False,
         branch_gain=1.0, null_bias=0.05):
    """caption=True: `CaptionODISE.forward` (odise.py:545-619) with `WordEmbed.forward` eval (1206-1216) and the learned 2-way
    `class_embed` of the decoder (mask_generator_with_caption.py) instead of `CategoryODISE` / `CategoryEmbed` / `PseudoClassEmbed`."""
    K = len(labels)
    group_sizes = [len(l) for l in labels]
    head_o = init_synthetic_(SemSegHead(small=True, num_classes=1 if caption else K, in_channels=in_channels, learned_class_embed=caption), seed=seed,
                             branch_gain=branch_gain)
    clip_o = clip_vit.init_synthetic_(clip_vit.CLIPVisual(image_size=56, patch_size=14, width=128, layers=2, heads=2, output_dim=48), seed=seed + 5).eval()
    overlap = [int(not {s for l in train_labels for s in l}.isdisjoint(set(l))) for l in labels]
    heads_o = om.OpenVocabHeads(clip_o, group_sizes, projection_dim=64, seed=seed + 7, overlap=overlap, alpha=0.35, beta=0.65)

    H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
    Hp, Wp = (H + 63) // 64 * 64, (W + 63) // 64 * 64
    B = len(sizes)
    g = torch.Generator().manual_seed(seed + 1)
    feats = {f"s{i}": torch.randn(B, in_channels, Hp >> i, Wp >> i, generator=g) for i in (2, 3, 4, 5)}
    images = [torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8) for h, w in sizes]

    if spread:
        # text banks that spread the labels over the queries (tests/fullsize.spread_vocabulary): several categories per image, things and
        # stuff, several queries per stuff class (the merging branch of maskformer_model.py:327-333), null-labelled queries.  The banks are
        # INPUTS of the stages pinned; they are stored in the fixture.
        sys.path.insert(0, os.path.join(HERE, ".."))
        from fullsize import spread_vocabulary
        den = torch.zeros(B, 3, H, W)
        for i, im in enumerate(images):
            den[i, :, :im.shape[-2], :im.shape[-1]] = im.float() / 255.0
        with torch.no_grad():
            out_o = head_o(feats)
            ce_o = om.mask_clip_embed(clip_o, den, out_o["pred_masks"])
            head_o.predictor.post_mask_embed.logit_scale.fill_(float(np.log(100.0)))
        spread_vocabulary(heads_o, out_o["mask_embed"][0], ce_o[0], seed=seed + 11, null_queries=3, null_bias=null_bias)

    ref_head = reference_head(1 if caption else K, in_channels, 64, 64, 2, 128, 64, 20, 128, 3, learned_class_embed=caption)
    ref_head.load_state_dict(head_o.state_dict(), strict=True)
    ref_head.num_classes = K     # what OpenPanopticInference writes into `sem_seg_head.num_classes` at test time (pano_wrapper.py:40-41)

    cat = ro.CategoryEmbed.__new__(ro.CategoryEmbed)
    nn.Module.__init__(cat)
    cat.labels, cat.prompt, cat.test_labels = labels, None, labels
    cat.clip = types.SimpleNamespace(device=torch.device("cpu"))
    cat.text_proj = nn.Linear(48, 64)
    cat.text_proj.load_state_dict(heads_o.text_proj.state_dict())
    cat.null_embed = nn.Parameter(heads_o.null_embed.detach().clone())
    cat._test_text_embed_dict = {ro.to_tuple(prompt_labels(labels, None)): heads_o.text_embed.clone()}
    if caption:
        from collections import OrderedDict
        cat = ro.WordEmbed.__new__(ro.WordEmbed)
        nn.Module.__init__(cat)
        cat.prompt, cat.test_labels = "photo", labels
        cat.clip = types.SimpleNamespace(device=torch.device("cpu"))
        cat.text_proj = nn.Linear(48, 64)
        cat.text_proj.load_state_dict(heads_o.text_proj.state_dict())
        cat._test_text_embed_dict = OrderedDict({ro.to_tuple(prompt_labels(labels, "photo")): heads_o.text_embed.clone()})

    mclip = MaskCLIP.__new__(MaskCLIP)
    nn.Module.__init__(mclip)
    mclip.clip, mclip.clip_preprocess, mclip.name, mclip.normalize = FakeOpenClip(clip_o.visual), normalize_only, "synthetic", False
    pool = ro.PoolingCLIPHead.__new__(ro.PoolingCLIPHead)
    nn.Module.__init__(pool)
    pool.clip, pool.alpha, pool.beta, pool.prompt, pool.test_labels = mclip, 0.35, 0.65, "photo", labels
    pool.train_labels, pool.bg_labels, pool.normalize_logits = train_labels, None, True
    pool._test_text_embed_dict = {ro.to_tuple(prompt_labels(labels, "photo")): heads_o.clip_text_embed.clone()}

    meta = types.SimpleNamespace(thing_dataset_id_to_contiguous_id={100 + t: t for t in things})
    common = dict(backbone=SeededBackbone(feats), sem_seg_head=ref_head, criterion=None, num_queries=20, object_mask_threshold=0.0,
                  overlap_threshold=overlap_threshold, metadata=meta, size_divisibility=64, sem_seg_postprocess_before_inference=True,
                  pixel_mean=[0.0, 0.0, 0.0], pixel_std=[255.0, 255.0, 255.0], semantic_on=True, instance_on=True, panoptic_on=True,
                  test_topk_per_image=topk, clip_head=pool)
    model = (ro.CaptionODISE(word_head=cat, grounding_criterion=None, **common) if caption else ro.CategoryODISE(category_head=cat, **common)).eval()
    captured = []
    real_sem = model.semantic_inference
    model.semantic_inference = lambda mask_cls, mask_pred: (captured.append(mask_cls.clone()), real_sem(mask_cls, mask_pred))[1]
    batched = [{"image": im, "height": oh, "width": ow} for im, (oh, ow) in zip(images, out_sizes)]
    with torch.no_grad():
        results = model(batched)

    arrays = {f"feat_{k}": v.numpy() for k, v in feats.items()}
    arrays.update(seed=np.int64(seed), in_channels=np.int64(in_channels), group_sizes=np.array(group_sizes), things=np.array(sorted(things)),
                  overlap=np.array(overlap), topk=np.int64(topk), overlap_threshold=np.float64(overlap_threshold), caption=np.int64(caption),
                  sizes=np.array(sizes), out_sizes=np.array(out_sizes), branch_gain=np.float64(branch_gain))
    if spread:
        arrays.update(text_embed=heads_o.text_embed.numpy(), clip_text_embed=heads_o.clip_text_embed.numpy(), null_embed=heads_o.null_embed.detach().numpy(),
                      logit_scale_param=np.float64(np.log(100.0)))
    for b, (im, r) in enumerate(zip(images, results)):
        pan, info = r["panoptic_seg"]
        inst = r["instances"]
        arrays.update({f"image_{b}": im.numpy(), f"mask_cls_{b}": captured[b].numpy(), f"sem_seg_{b}": r["sem_seg"].numpy().astype(np.float16),
                       f"sem_argmax_{b}": r["sem_seg"].argmax(0).numpy().astype(np.uint8), f"pan_{b}": pan.numpy().astype(np.uint8),
                       f"pan_info_{b}": np.array([[s["id"], int(s["isthing"]), s["category_id"]] for s in info], np.int64).reshape(-1, 3),
                       f"inst_scores_{b}": inst.scores.numpy(), f"inst_classes_{b}": inst.pred_classes.numpy(),
                       f"inst_area_{b}": inst.pred_masks.flatten(1).sum(1).numpy()})
        print(name, b, "segments", info, "instances", len(inst.scores), "sem_seg", tuple(r["sem_seg"].shape))
    np.savez_compressed(os.path.join(HERE, f"heads_{name}.npz"), **arrays)


if __name__ == "__main__":
    LABELS = [["person", "child"], ["sky"], ["tree", "trees", "bush"], ["car"], ["road", "street"], ["building"], ["dog"], ["grass"]]
    TRAIN = [["sky"], ["car", "truck"], ["dog"], ["person"]]
    case("a", seed=77, sizes=[(100, 140)], out_sizes=[(150, 210)], labels=LABELS, things={0, 3, 6}, train_labels=TRAIN)
    case("b", seed=5, sizes=[(128, 128), (128, 128)], out_sizes=[(128, 128), (96, 64)], labels=LABELS[:5], things={0, 3}, train_labels=TRAIN,
         overlap_threshold=0.33, topk=15)
    case("c", seed=55, sizes=[(192, 128)], out_sizes=[(192, 128)], labels=LABELS, things={0, 3, 6}, train_labels=TRAIN, overlap_threshold=0.0, topk=40)
    case("e_caption", seed=55, sizes=[(128, 192)], out_sizes=[(96, 144)], labels=LABELS, things={0, 3, 6}, train_labels=TRAIN, overlap_threshold=0.0,
         topk=25, caption=True)
    case("f_diverse", seed=91, sizes=[(192, 128), (128, 192)], out_sizes=[(192, 128), (96, 144)], labels=LABELS, things={0, 3, 6}, train_labels=TRAIN,
         overlap_threshold=0.0, topk=40, spread=True, branch_gain=0.3)
    case("g_diverse", seed=92, sizes=[(192, 192)], out_sizes=[(192, 192)], labels=LABELS, things={0, 3, 6}, train_labels=TRAIN, overlap_threshold=0.5, topk=30,
         spread=True, branch_gain=0.3, null_bias=0.08)
    case("d", seed=56, sizes=[(192, 128)], out_sizes=[(144, 96)], labels=LABELS, things={0, 3, 6}, train_labels=TRAIN, overlap_threshold=0.0, topk=40)
