"""Oracle: CategoryODISE eval branch after the mask generator (TEST INFRASTRUCTURE — see oracle/__init__.py).

Follows, line by line:
  ensemble_logits_with_labels                odise/modeling/meta_arch/helper.py:79-109
  CategoryODISE.cal_pred_logits / forward    odise/modeling/meta_arch/odise.py:181-207, 282-372
  CategoryEmbed.forward (eval)               odise/modeling/meta_arch/odise.py:1290-1307
  PoolingCLIPHead.forward                    odise/modeling/meta_arch/odise.py:1469-1542
  MaskCLIP.forward / get_mask_embed / encode_image_with_mask / _mask_clip_forward / pred_logits
                                             odise/modeling/meta_arch/clip.py:252-361
  MaskFormer.semantic_/panoptic_/instance_inference   third_party/Mask2Former/mask2former/maskformer_model.py:280-380
  detectron2 sem_seg_postprocess (absent; restated from SURVEY.md Appendix A.4: crop to image_size, bilinear to (h, w))
PINNED: tests/test_oracle_golden.py replays golden vectors written by the reference's own `CategoryODISE.forward`
(tests/golden/make_golden_heads.py).  The CLIP text tower lives in oracle/clip_text.py: text banks enter as precomputed [K_tot, 768] embeddings (the reference caches them per
label set, odise.py:1281-1288); `labels` is the nested synonym list whose group sizes drive the max-ensemble.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import clip_vit


def ensemble_logits_with_labels(logits: torch.Tensor, group_sizes: Sequence[int]) -> torch.Tensor:
    """helper.py:79-109 with ensemble_method='max'; group_sizes = [len(l) for l in labels]."""
    assert logits.shape[-1] == sum(group_sizes)
    out = torch.zeros(*logits.shape[:-1], len(group_sizes), dtype=logits.dtype)
    start = 0
    for i, n in enumerate(group_sizes):
        out[..., i] = logits[..., start:start + n].max(dim=-1).values
        start += n
    return out


def cal_pred_logits(mask_embed, text_embed, null_embed, logit_scale, group_sizes):
    """odise.py:181-207."""
    mask_embed = F.normalize(mask_embed, dim=-1)
    text_embed = F.normalize(text_embed, dim=-1)
    pred = logit_scale * (mask_embed @ text_embed.t())
    pred = ensemble_logits_with_labels(pred, group_sizes)
    null_embed = F.normalize(null_embed, dim=-1)
    null_pred = logit_scale * (mask_embed @ null_embed.t())
    return torch.cat([pred, null_pred], dim=-1)


@torch.no_grad()
def mask_clip_embed(clip: clip_vit.CLIPVisual, image: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """MaskCLIP.get_mask_embed -> encode_image_with_mask -> _mask_clip_forward (clip.py:325-338, 282-323, 252-280).
    image [B,3,H,W] in [0,1]; mask [B,Q,h,w] logits -> [B,Q,output_dim]."""
    v = clip.visual
    size = (v.image_size, v.image_size)
    image = F.interpolate(image, size=size, mode="bilinear", align_corners=False)               # :327-332
    mask = F.interpolate(mask, size=image.shape[-2:], mode="bilinear", align_corners=False)      # :333
    image = clip_vit.clip_preprocess(image, v.image_size)                                        # :284 (resize/crop are identities)
    B, Q = mask.shape[:2]
    mask = mask.sigmoid()
    patch_mask = F.max_pool2d(mask, kernel_size=v.patch_size, stride=v.patch_size)               # :292-296
    mask_token_attn_mask = (patch_mask < 0.5).reshape(B, Q, -1)                                  # :300-302
    num_image_cls = v.positional_embedding.shape[0]
    num_image = num_image_cls - 1
    n_all = Q + num_image_cls
    attn_mask = torch.zeros((n_all, n_all), dtype=torch.bool)
    attn_mask[:, :Q] = True                                                                      # :315
    attn_mask = attn_mask.unsqueeze(0).repeat_interleave(B, dim=0)
    attn_mask[:, :Q, -num_image:] = mask_token_attn_mask                                         # :318
    heads = v.conv1.out_channels // 64                                                           # :319
    attn_mask = attn_mask.unsqueeze(1).expand(-1, heads, -1, -1).reshape(B * heads, n_all, n_all)
    # _mask_clip_forward
    x = v.conv1(image)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    x = torch.cat([v.class_embedding + torch.zeros(x.shape[0], 1, x.shape[-1]), x], dim=1)
    x = x + v.positional_embedding
    x = v.ln_pre(x)
    x = x.permute(1, 0, 2)
    cls_embed = x[0:1].expand(Q, -1, -1)
    x = torch.cat([cls_embed, x], dim=0)
    x = v.transformer(x, attn_mask)
    x = x.permute(1, 0, 2)
    x = v.ln_post(x[:, :Q, :])
    return torch.einsum("nld,dc->nlc", x, v.proj)


def mask_clip_pred_logits(mask_embed, text_embed, group_sizes, logit_scale=100.0):
    """MaskCLIP.pred_logits (clip.py:340-350); logit_scale = clamp(exp(ln 100), max=100) for the OpenAI weights."""
    l = torch.einsum("bqc,nc->bqn", F.normalize(mask_embed, dim=-1), F.normalize(text_embed, dim=-1)) * logit_scale
    return ensemble_logits_with_labels(l, group_sizes)


def pooling_clip_head(pred_open_logits, mask_pred_open_logits, category_overlapping_mask, alpha, beta):
    """PoolingCLIPHead.forward, normalize_logits=True branch (odise.py:1506-1536)."""
    p = pred_open_logits.softmax(dim=-1)
    q = mask_pred_open_logits.softmax(dim=-1)
    m = category_overlapping_mask.to(p.dtype)
    base = (p ** (1 - alpha) * q ** alpha).log() * m
    novel = (p ** (1 - beta) * q ** beta).log() * (1 - m)
    return base + novel


def merge_with_null(pred_logits, pred_open_logits):
    """odise.py:300-323 (with_bg False)."""
    binary_last = F.softmax(pred_logits, dim=-1)[..., -1:]
    probs = F.softmax(pred_open_logits, dim=-1)
    return torch.log(torch.cat([probs * (1 - binary_last), binary_last], dim=-1) + 1e-8)


def sem_seg_postprocess(result, img_size, output_height, output_width):
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


def semantic_inference(mask_cls, mask_pred):                                                     # maskformer_model.py:280-284
    mask_cls = F.softmax(mask_cls, dim=-1)[..., :-1]
    return torch.einsum("qc,qhw->chw", mask_cls, mask_pred.sigmoid())


def panoptic_inference(mask_cls, mask_pred, num_classes, thing_ids, object_mask_threshold=0.0, overlap_threshold=0.8):
    """maskformer_model.py:286-342."""
    scores, labels = F.softmax(mask_cls, dim=-1).max(-1)
    mask_pred = mask_pred.sigmoid()
    keep = labels.ne(num_classes) & (scores > object_mask_threshold)
    cur_scores, cur_classes, cur_masks = scores[keep], labels[keep], mask_pred[keep]
    cur_prob_masks = cur_scores.view(-1, 1, 1) * cur_masks
    h, w = cur_masks.shape[-2:]
    panoptic_seg = torch.zeros((h, w), dtype=torch.int32)
    segments_info = []
    current_segment_id = 0
    if cur_masks.shape[0] == 0:
        return panoptic_seg, segments_info
    cur_mask_ids = cur_prob_masks.argmax(0)
    stuff_memory_list = {}
    for k in range(cur_classes.shape[0]):
        pred_class = cur_classes[k].item()
        isthing = pred_class in thing_ids
        mask_area = (cur_mask_ids == k).sum().item()
        original_area = (cur_masks[k] >= 0.5).sum().item()
        mask = (cur_mask_ids == k) & (cur_masks[k] >= 0.5)
        if mask_area > 0 and original_area > 0 and mask.sum().item() > 0:
            if mask_area / original_area < overlap_threshold:
                continue
            if not isthing:
                if int(pred_class) in stuff_memory_list:
                    panoptic_seg[mask] = stuff_memory_list[int(pred_class)]
                    continue
                stuff_memory_list[int(pred_class)] = current_segment_id + 1
            current_segment_id += 1
            panoptic_seg[mask] = current_segment_id
            segments_info.append({"id": current_segment_id, "isthing": bool(isthing), "category_id": int(pred_class)})
    return panoptic_seg, segments_info


def instance_inference(mask_cls, mask_pred, num_classes, num_queries, thing_ids, topk=100, panoptic_on=True):
    """maskformer_model.py:344-380 (Instances replaced by a dict with the same fields)."""
    scores = F.softmax(mask_cls, dim=-1)[:, :-1]
    labels = torch.arange(num_classes).unsqueeze(0).repeat(num_queries, 1).flatten(0, 1)
    scores_per_image, topk_indices = scores.flatten(0, 1).topk(topk, sorted=False)
    labels_per_image = labels[topk_indices]
    topk_indices = topk_indices // num_classes
    mask_pred = mask_pred[topk_indices]
    if panoptic_on:
        keep = torch.tensor([int(l) in thing_ids for l in labels_per_image], dtype=torch.bool)
        scores_per_image, labels_per_image, mask_pred = scores_per_image[keep], labels_per_image[keep], mask_pred[keep]
    pred_masks = (mask_pred > 0).float()
    mask_scores = (mask_pred.sigmoid().flatten(1) * pred_masks.flatten(1)).sum(1) / (pred_masks.flatten(1).sum(1) + 1e-6)
    return {"pred_masks": pred_masks, "scores": scores_per_image * mask_scores, "pred_classes": labels_per_image}


class OpenVocabHeads(nn.Module):
    """category_head (text_proj + null_embed) and clip_head (MaskCLIP) state for a given vocabulary."""

    def __init__(self, clip: clip_vit.CLIPVisual, group_sizes: Sequence[int], projection_dim=256, seed=31, overlap=None,
                 alpha=0.3, beta=0.7):
        super().__init__()
        self.clip = clip
        dim = clip.visual.proj.shape[1]
        g = torch.Generator().manual_seed(seed)
        self.group_sizes = list(group_sizes)
        k_tot = sum(self.group_sizes)
        self.text_proj = nn.Linear(dim, projection_dim)
        self.null_embed = nn.Parameter(torch.randn(1, dim, generator=g))
        with torch.no_grad():
            self.text_proj.weight.copy_(torch.randn(self.text_proj.weight.shape, generator=g) / dim ** 0.5)
            self.text_proj.bias.copy_(0.02 * torch.randn(self.text_proj.bias.shape, generator=g))
        # synthetic CLIP text embeddings of the two prompt sets (category_head: prompt=None; clip_head: "a photo of a {}.")
        self.register_buffer("text_embed", torch.randn(k_tot, dim, generator=g))
        self.register_buffer("clip_text_embed", torch.randn(k_tot, dim, generator=g))
        if overlap is None:
            overlap = (torch.rand(len(self.group_sizes), generator=g) < 0.6)
        self.register_buffer("category_overlapping_mask", torch.as_tensor(overlap).long())
        self.alpha, self.beta = alpha, beta
        self.eval()

    @torch.no_grad()
    def classify(self, outputs: Dict[str, torch.Tensor], images01: torch.Tensor) -> torch.Tensor:
        """odise.py:285-323: returns mask_cls_results [B,Q,K+1] (log-probabilities)."""
        text_embed = self.text_proj(self.text_embed)                                             # CategoryEmbed.forward :1303-1305
        null_embed = self.text_proj(self.null_embed)
        pred_logits = cal_pred_logits(outputs["mask_embed"], text_embed, null_embed, outputs["logit_scale"], self.group_sizes)
        pred_open_logits = pred_logits[..., :-1]
        clip_embed = mask_clip_embed(self.clip, images01, outputs["pred_masks"])
        mask_pred_open_logits = mask_clip_pred_logits(clip_embed, self.clip_text_embed, self.group_sizes)
        pred_open_logits = pooling_clip_head(pred_open_logits, mask_pred_open_logits, self.category_overlapping_mask, self.alpha, self.beta)
        return merge_with_null(pred_logits, pred_open_logits)


def caption_pred_open_logits(mask_embed, text_embed, logit_scale, group_sizes):
    """CaptionODISE.cal_pred_open_logits (odise.py:432-449): cosine logits against the projected word bank, max over synonyms."""
    me = F.normalize(mask_embed, dim=-1)
    te = F.normalize(text_embed, dim=-1)
    return ensemble_logits_with_labels(logit_scale * (me @ te.t()), group_sizes)


def caption_merge(pred_logits2, pred_open_logits):
    """CaptionODISE.forward eval branch (odise.py:557-569): learned (object, no-object) logits gate the open-vocabulary distribution."""
    binary = F.softmax(pred_logits2, dim=-1)
    probs = F.softmax(pred_open_logits, dim=-1)
    return torch.log(torch.cat([probs * binary[..., 0:1], binary[..., 1:2]], dim=-1) + 1e-8)


@torch.no_grad()
def caption_classify(heads: "OpenVocabHeads", outputs, images01):
    """The classification part of CaptionODISE's eval forward: WordEmbed.forward eval (odise.py:1206-1216: text_proj of the test
    word bank, no null embedding), cal_pred_open_logits, PoolingCLIPHead, merge with the binary class head."""
    text_embed = heads.text_proj(heads.text_embed)
    open_logits = caption_pred_open_logits(outputs["mask_embed"], text_embed, outputs["logit_scale"], heads.group_sizes)
    clip_embed = mask_clip_embed(heads.clip, images01, outputs["pred_masks"])
    clip_logits = mask_clip_pred_logits(clip_embed, heads.clip_text_embed, heads.group_sizes)
    open_logits = pooling_clip_head(open_logits, clip_logits, heads.category_overlapping_mask, heads.alpha, heads.beta)
    return caption_merge(outputs["pred_logits"], open_logits)


@torch.no_grad()
def postprocess(mask_cls_results, pred_masks, padded_hw, image_sizes, out_sizes, num_classes, thing_ids, overlap_threshold=0.8,
                topk=100) -> List[dict]:
    """odise.py:326-370."""
    mask_pred_results = F.interpolate(pred_masks, size=padded_hw, mode="bilinear", align_corners=False)
    results = []
    for mask_cls, mask_pred, image_size, (height, width) in zip(mask_cls_results, mask_pred_results, image_sizes, out_sizes):
        mask_pred = sem_seg_postprocess(mask_pred, image_size, height, width)
        r = {"sem_seg": semantic_inference(mask_cls, mask_pred)}
        r["panoptic_seg"] = panoptic_inference(mask_cls, mask_pred, num_classes, thing_ids, 0.0, overlap_threshold)
        r["instances"] = instance_inference(mask_cls, mask_pred, num_classes, mask_cls.shape[0], thing_ids, topk)
        results.append(r)
    return results
