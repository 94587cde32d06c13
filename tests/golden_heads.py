"""Rebuilds the seeded modules a tests/golden/heads_*.npz fixture was generated with (tests/golden/make_golden_heads.py)."""
import numpy as np
import torch

from oracle import clip_vit, odise_model as om
from oracle.m2f import SemSegHead, init_synthetic_

CLIP_KW = dict(image_size=56, patch_size=14, width=128, layers=2, heads=2, output_dim=48)


def build(z):
    """-> (head, clip, heads, groups, things, caption) with the fixture's weights and text banks."""
    seed, C = int(z["seed"]), int(z["in_channels"])
    groups, things = z["group_sizes"].tolist(), set(z["things"].tolist())
    K = len(groups)
    caption = bool(int(z["caption"]))                                      # CaptionODISE.forward + WordEmbed + the learned 2-way class_embed
    gain = float(z["branch_gain"]) if "branch_gain" in z.files else 1.0
    head = init_synthetic_(SemSegHead(small=True, num_classes=1 if caption else K, in_channels=C, learned_class_embed=caption), seed=seed, branch_gain=gain)
    clip = clip_vit.init_synthetic_(clip_vit.CLIPVisual(**CLIP_KW), seed=seed + 5).eval()
    heads = om.OpenVocabHeads(clip, groups, projection_dim=64, seed=seed + 7, overlap=z["overlap"].tolist(), alpha=0.35, beta=0.65)
    if "text_embed" in z.files:                                            # the "diverse" cases: banks spread over the queries, temperature at its clamp
        with torch.no_grad():
            heads.text_embed.copy_(torch.from_numpy(z["text_embed"]))
            heads.clip_text_embed.copy_(torch.from_numpy(z["clip_text_embed"]))
            heads.null_embed.copy_(torch.from_numpy(z["null_embed"]))
            head.predictor.post_mask_embed.logit_scale.fill_(float(z["logit_scale_param"]))
    return head, clip, heads, groups, things, caption
