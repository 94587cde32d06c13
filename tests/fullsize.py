"""Shared set-up of the full-size parity tests (tests/test_gpu_fullsize.py): the fp32 CPU oracle of the benchmarked configuration
(BASELINE configs[2] shapes) with seeded synthetic weights, its inputs, and a vocabulary that makes the decisions non-degenerate.

Why the vocabulary is built from the oracle's own embeddings: with seeded random weights every query's mask embedding / MaskCLIP
embedding is dominated by a common component, so a random text bank labels all 100 queries with the same class (two segments, one
category - nothing of the stuff-merging / null-dropping logic of maskformer_model.py:286-342 would run).  The text banks are inputs of
the path (`set_vocabulary`), so the tests choose them: every class gets an anchor query and its prompt strings point along that
query's embedding minus the mean embedding (plus seeded noise), mapped back through the pseudo-inverse of `category_head.text_proj`;
the null embedding points at a handful of queries.  Labels then spread over ~65 classes, things and stuff, a third of the queries
labelled null, several queries per stuff class, and the top-2 margins cover 0.01 .. 0.9.  Everything is seeded and a function of the oracle alone.

The oracle results can be cached under tests/.oracle_cache (git-ignored; travels with gpurun snapshots) - the cache only saves CPU
minutes, tests recompute when it is absent."""
import hashlib
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from odise_amd.synthetic import synthetic_vocabulary
from oracle import odise_model as om
from oracle.backbone import FeatureExtractorBackbone
from oracle.ldm_extractor import ImplicitCaptionerExtractor
from oracle.m2f import SemSegHead, init_synthetic_

FEATURE_DIMS = [512, 512, 2560, 1920, 960, 640, 512, 512]   # enc5, enc7, u2, u5, u8, u11, dec2, dec5 (ldm.py:284-346)
CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".oracle_cache")
VERSION = "v6"


def image_u8(h, w, seed=0):
    """SURVEY.md 8d config 1/3 input: seeded uniform uint8 noise smoothed by a 9x9 box filter."""
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.integers(0, 256, size=(1, 3, h, w)).astype(np.float32))
    x = F.avg_pool2d(F.pad(x, (4, 4, 4, 4), mode="reflect"), 9, stride=1)
    x = (x - x.amin()) / (x.amax() - x.amin())
    return (x[0] * 255).round().to(torch.uint8)


_MEMO = {}     # one oracle pass per pytest session: test_gpu_fullsize.py and test_gpu_dropin.py share it


def build_models(num_classes):
    if ("models", num_classes) in _MEMO:
        return _MEMO[("models", num_classes)]
    ext = ImplicitCaptionerExtractor()
    bb = FeatureExtractorBackbone(ext, FEATURE_DIMS)
    head = init_synthetic_(SemSegHead(num_classes=num_classes), branch_gain=0.3)
    with torch.no_grad():   # the learned temperature at its clamp (odise.py:1013 clamps exp(logit_scale) at 100): class distributions as peaked as a trained model's
        head.predictor.post_mask_embed.logit_scale.fill_(math.log(100.0))
    _MEMO[("models", num_classes)] = (ext, bb, head)
    return ext, bb, head


def spread_vocabulary(heads: om.OpenVocabHeads, mask_embed, clip_embed, seed=5, null_queries=6, null_bias=0.07):
    """Overwrite the text banks / null embedding of `heads` (see the module docstring).  mask_embed [Q,256], clip_embed [Q,768]."""
    g = torch.Generator().manual_seed(seed)
    Q = mask_embed.shape[0]
    sizes = heads.group_sizes
    Kc = len(sizes)

    def directions(e):
        e = F.normalize(e.double(), dim=-1)
        mu = e.mean(0, keepdim=True)
        return F.normalize(e - mu, dim=-1), F.normalize(mu, dim=-1)

    def off_mean(t, mu):                                                       # no component along the mean embedding: no class wins everywhere
        return t - (t @ mu.t()) * mu

    (d1, mu1), (d2, mu2) = directions(mask_embed), directions(clip_embed)
    perm = torch.randperm(Q, generator=g)
    anchor1 = torch.tensor([int(perm[(k * 3) % (Q - null_queries)]) for k in range(Kc)])      # the last `null_queries` of perm anchor no class
    other = perm[torch.randint(0, Q - null_queries, (Kc,), generator=g)]
    anchor2 = torch.where(torch.rand(Kc, generator=g) < 0.7, anchor1, other)
    t1, t2 = [], []
    for k, n in enumerate(sizes):
        for _ in range(n):
            t1.append(d1[anchor1[k]] + 0.5 * torch.randn(d1.shape[1], generator=g, dtype=torch.float64) / d1.shape[1] ** 0.5)
            t2.append(d2[anchor2[k]] + 0.5 * torch.randn(d2.shape[1], generator=g, dtype=torch.float64) / d2.shape[1] ** 0.5)
    t1, t2 = off_mean(torch.stack(t1), mu1), off_mean(torch.stack(t2), mu2)
    nullq = perm[-null_queries:]
    tn = off_mean(F.normalize(d1[nullq].sum(0), dim=0)[None], mu1) + null_bias * mu1   # a share of the mean direction lifts the null logit of EVERY query
    W, b = heads.text_proj.weight.detach().double(), heads.text_proj.bias.detach().double()
    pinv = torch.linalg.pinv(W)                                               # [768, 256]: text = pinv (target - b) solves text_proj(text) = target
    with torch.no_grad():
        heads.text_embed.copy_(((t1 - b) @ pinv.t()).float())
        heads.null_embed.copy_(((tn - b) @ pinv.t()).float())
        heads.clip_text_embed.copy_(t2.float())
    return heads


def centre_mask_logits(head, feats, positive_fraction=0.15, rounds=2):
    """With seeded weights the mask logits of every query are positive on most of the image (the queries' common embedding component
    times the mask features), so all masks overlap, nearly nothing passes the 0.8 overlap test and every MaskCLIP mask token sees every
    patch.  A trained model's masks cover a fraction of the image: shift `mask_features.bias` (a per-query constant on the logits,
    me_q . b) so that about `positive_fraction` of the logits are positive.  Deterministic, part of the weights both sides load."""
    grab = {}
    hook = head.predictor.mask_embed.register_forward_hook(lambda m, i, o: grab.__setitem__("me", o.detach()))
    try:
        for _ in range(rounds):
            out = head(feats)
            me = grab["me"][0].double()                                        # [Q, C]: mask embeddings of the final prediction head
            s = torch.quantile(out["pred_masks"].flatten()[::97].double(), 1.0 - positive_fraction)
            delta = -(s * (torch.linalg.pinv(me) @ torch.ones(me.shape[0], 1, dtype=torch.float64)))[:, 0]   # me_q . delta = -s for every q
            with torch.no_grad():
                head.pixel_decoder.mask_features.bias.add_(delta.float())
    finally:
        hook.remove()


def export_state(ext, bb, head, heads):
    state = ext.export_state()
    state.update({"backbone.feature_projections." + k: v for k, v in bb.feature_projections.state_dict().items()})
    state.update({"sem_seg_head." + k: v for k, v in head.state_dict().items()})
    state["category_head.text_proj.weight"] = heads.text_proj.weight.detach()
    state["category_head.text_proj.bias"] = heads.text_proj.bias.detach()
    state["category_head.null_embed"] = heads.null_embed.detach()
    return state


def _cache_path(tag):
    return os.path.join(CACHE, f"{tag}_{VERSION}.npz")


def reference(bb, head, ext, size, num_classes, num_strings, seed=0, use_cache=True):
    """Oracle pass over one size x size image: features, head outputs, MaskCLIP embedding; then the spread vocabulary and mask_cls.
    Returns (img_u8, heads, dict of torch tensors)."""
    key = ("ref", id(head), size, num_classes, num_strings, seed)
    if key in _MEMO:
        return _MEMO[key]
    img = image_u8(size, size, seed)
    cat, clp, sizes, overlap = synthetic_vocabulary(num_classes, num_strings, 768)
    heads = om.OpenVocabHeads(ext.clip, [int(s) for s in sizes], projection_dim=256, overlap=torch.from_numpy(overlap.astype(bool)))
    img01 = img.float()[None] / 255.0

    def cached(tag, fn):
        path = _cache_path(tag)
        if use_cache and os.path.exists(path):
            z = np.load(path)
            return {k: torch.from_numpy(z[k]) for k in z.files}
        out = fn()
        if use_cache and os.environ.get("ODISE_ORACLE_CACHE_WRITE"):
            os.makedirs(CACHE, exist_ok=True)
            np.savez(path, **{k: v.numpy() for k, v in out.items()})
        return out

    feats = cached(f"feats_{size}_{seed}", lambda: bb(img01))                 # size is a multiple of 64: no padding
    centre_mask_logits(head, feats)

    def run_head():
        out = head(feats)
        ce = om.mask_clip_embed(ext.clip, img01, out["pred_masks"])
        return {"pred_masks": out["pred_masks"], "mask_embed": out["mask_embed"], "mask_pooled_features": out["mask_pooled_features"],
                "pred_logits": out["pred_logits"], "logit_scale": torch.as_tensor(float(out["logit_scale"])), "clip_embed": ce}

    r = cached(f"head_{size}_{num_classes}_{seed}", run_head)
    r = {**feats, **r}
    r["logit_scale"] = float(r["logit_scale"])
    spread_vocabulary(heads, r["mask_embed"][0], r["clip_embed"][0])
    out = {"mask_embed": r["mask_embed"], "pred_masks": r["pred_masks"], "logit_scale": r["logit_scale"], "pred_logits": r["pred_logits"]}
    # CategoryODISE.forward after the head (odise.py:285-323) with the MaskCLIP embedding computed above
    with torch.no_grad():
        text_embed = heads.text_proj(heads.text_embed)
        null_embed = heads.text_proj(heads.null_embed)
        pred_logits = om.cal_pred_logits(out["mask_embed"], text_embed, null_embed, out["logit_scale"], heads.group_sizes)
        clip_logits = om.mask_clip_pred_logits(r["clip_embed"], heads.clip_text_embed, heads.group_sizes)
        open_logits = om.pooling_clip_head(pred_logits[..., :-1], clip_logits, heads.category_overlapping_mask, heads.alpha, heads.beta)
        r["mask_cls"] = om.merge_with_null(pred_logits, open_logits)
    _MEMO[key] = (img, heads, r)
    return img, heads, r


def state_digest(state):
    h = hashlib.sha256()
    for k in sorted(state):
        h.update(k.encode())
    return h.hexdigest()[:12]
