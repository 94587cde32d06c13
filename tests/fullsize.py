"""Shared set-up of the full-size parity tests (tests/test_gpu_fullsize.py): the fp32 CPU oracle of the benchmarked configuration
(BASELINE configs[2] shapes) with seeded synthetic weights, its inputs, and a vocabulary that makes the decisions non-degenerate.

Why the vocabulary is built from the oracle's own embeddings: with seeded random weights every query's mask embedding / MaskCLIP
embedding is dominated by a common component, so a random text bank labels all 100 queries with the same class (two segments, one
category - nothing of the stuff-merging / null-dropping logic of maskformer_model.py:286-342 would run).  The text banks are inputs of
the path (`set_vocabulary`), so the tests choose them: every class gets an anchor query and its prompt strings point along that
query's embedding minus the mean embedding (plus seeded noise), mapped back through the pseudo-inverse of `category_head.text_proj`;
the null embedding points at a handful of queries.  Labels then spread over ~65 classes, things and stuff, a third of the queries
labelled null, several queries per stuff class, and the top-2 margins cover 0.01 .. 0.9.  Everything is seeded and a function of the oracle alone.

One head serves every size and vocabulary: its mask logits are centred ONCE, on the 1024x1024 reference image (build_models), so
tests of different sizes / vocabularies may share one device model and no later call mutates the weights.

The oracle passes can be cached under tests/.oracle_cache (a developer convenience: git-ignored AND gpurun-ignored, so the GPU box always
recomputes the oracle itself).  Every cache file carries a digest of the input image and of the weights it was computed with; a file whose
digest does not match the current inputs is ignored and recomputed."""
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
VERSION = "v7"


def image_u8(h, w, seed=0):
    """SURVEY.md 8d config 1/3 input: seeded uniform uint8 noise smoothed by a 9x9 box filter."""
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.integers(0, 256, size=(1, 3, h, w)).astype(np.float32))
    x = F.avg_pool2d(F.pad(x, (4, 4, 4, 4), mode="reflect"), 9, stride=1)
    x = (x - x.amin()) / (x.amax() - x.amin())
    return (x[0] * 255).round().to(torch.uint8)


_MEMO = {}     # one oracle pass per pytest session: test_gpu_fullsize*.py and test_gpu_dropin.py share it


def _sample_digest(h, name, t):
    a = t.detach().reshape(-1)
    h.update(name.encode())
    h.update(str(tuple(t.shape)).encode())
    h.update(a[:: max(1, a.numel() // 4096)].double().numpy().tobytes())


def weights_digest(ext, bb, head):
    """sha256 over (name, shape, ~4096 strided samples) of every tensor the oracle pass depends on."""
    h = hashlib.sha256()
    for k, v in sorted(ext.export_state().items()):
        _sample_digest(h, k, torch.as_tensor(v))
    for k, v in sorted(bb.feature_projections.state_dict().items()):
        _sample_digest(h, "proj." + k, v)
    for k, v in sorted(head.state_dict().items()):
        _sample_digest(h, "head." + k, v)
    return h.hexdigest()


def _cached(tag, digest, fn, use_cache=True):
    """Memo -> (digest-checked) file cache -> compute.  `digest` binds the entry to its inputs and weights."""
    key = ("cache", tag, digest)
    if key in _MEMO:
        return _MEMO[key]
    path = os.path.join(CACHE, f"{tag}_{VERSION}.npz")
    out = None
    if use_cache and os.path.exists(path):
        z = np.load(path)
        if "__digest__" in z.files and str(z["__digest__"]) == digest:
            out = {k: torch.from_numpy(z[k]) for k in z.files if k != "__digest__"}
    if out is None:
        with torch.no_grad():
            out = fn()
        if use_cache and os.environ.get("ODISE_ORACLE_CACHE_WRITE"):
            os.makedirs(CACHE, exist_ok=True)
            np.savez(path, __digest__=np.asarray(digest), **{k: v.numpy() for k, v in out.items()})
    _MEMO[key] = out
    return out


def features(ext, bb, size, seed=0):
    """Oracle backbone features of the seeded size x size image (size is a multiple of 64: no padding)."""
    if ("wd_bb",) not in _MEMO:   # the backbone's weights never change: digest once
        h = hashlib.sha256()
        for k, v in sorted(ext.export_state().items()):
            _sample_digest(h, k, torch.as_tensor(v))
        for k, v in sorted(bb.feature_projections.state_dict().items()):
            _sample_digest(h, "proj." + k, v)
        _MEMO[("wd_bb",)] = h.hexdigest()
    img = image_u8(size, size, seed)
    digest = hashlib.sha256(_MEMO[("wd_bb",)].encode() + img.numpy().tobytes()).hexdigest()
    return img, _cached(f"feats_{size}_{seed}", digest, lambda: bb(img.float()[None] / 255.0))


def build_models(num_classes=133):
    """-> (extractor, backbone, head) of the full-size model, seeded.  `num_classes` is accepted for call-site symmetry only: no weight
    of the label model depends on it (the class head is the parameter-free PseudoClassEmbed, odise.py:1273-1307)."""
    if "models" in _MEMO:
        return _MEMO["models"]
    ext = ImplicitCaptionerExtractor()
    bb = FeatureExtractorBackbone(ext, FEATURE_DIMS)
    head = init_synthetic_(SemSegHead(num_classes=133), branch_gain=0.3)
    with torch.no_grad():   # the learned temperature at its clamp (odise.py:1013 clamps exp(logit_scale) at 100): class distributions as peaked as a trained model's
        head.predictor.post_mask_embed.logit_scale.fill_(math.log(100.0))
    _, feats = features(ext, bb, 1024, 0)
    with torch.no_grad():
        centre_mask_logits(head, feats)
    _MEMO["models"] = (ext, bb, head)
    return ext, bb, head


def spread_vocabulary(heads: om.OpenVocabHeads, mask_embed, clip_embed, seed=5, null_queries=6, null_bias=0.07, anchored=None):
    """Overwrite the text banks / null embedding of `heads` (see the module docstring).  mask_embed [Q,256], clip_embed [Q,768].
    `anchored`: number of classes that point at a query (None = all).  A vocabulary much larger than the number of queries (A-847) would
    otherwise put ~9 near-identical classes on every query - every decision a near-tie; like in a real image, most of its classes are then
    absent: a seeded subset spread over the whole id range is anchored, the other classes' strings are unrelated directions."""
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
    present = torch.ones(Kc, dtype=torch.bool)
    if anchored is not None and anchored < Kc:
        present[:] = False
        chosen = torch.randperm(Kc, generator=g)[:anchored]
        present[chosen] = True
        for j, k in enumerate(sorted(chosen.tolist())):                                       # the present classes share the queries evenly
            anchor1[k] = int(perm[(j * 3) % (Q - null_queries)])
            if anchor2[k] != other[k]:
                anchor2[k] = anchor1[k]
    t1, t2 = [], []
    for k, n in enumerate(sizes):
        for _ in range(n):
            n1 = torch.randn(d1.shape[1], generator=g, dtype=torch.float64) / d1.shape[1] ** 0.5
            n2 = torch.randn(d2.shape[1], generator=g, dtype=torch.float64) / d2.shape[1] ** 0.5
            t1.append(d1[anchor1[k]] + 0.5 * n1 if present[k] else 1.1 * n1)
            t2.append(d2[anchor2[k]] + 0.5 * n2 if present[k] else 1.1 * n2)
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


def category_head_state(heads):
    """The classification head's own weights (odise.py:1236-1241).  The null embedding depends on the image the vocabulary was spread over
    (spread_vocabulary), not on the vocabulary's size."""
    return {"category_head.text_proj.weight": heads.text_proj.weight.detach(), "category_head.text_proj.bias": heads.text_proj.bias.detach(),
            "category_head.null_embed": heads.null_embed.detach()}


def export_state(ext, bb, head, heads):
    state = ext.export_state()
    state.update({"backbone.feature_projections." + k: v for k, v in bb.feature_projections.state_dict().items()})
    state.update({"sem_seg_head." + k: v for k, v in head.state_dict().items()})
    state.update(category_head_state(heads))
    return state


def head_reference(bb, head, ext, size, seed=0, use_cache=True):
    """Vocabulary-independent part of the oracle pass over one size x size image: backbone features, head outputs, MaskCLIP embedding."""
    img, feats = features(ext, bb, size, seed)
    if ("wd",) not in _MEMO:
        _MEMO[("wd",)] = weights_digest(ext, bb, head)
    digest = hashlib.sha256(_MEMO[("wd",)].encode() + img.numpy().tobytes()).hexdigest()
    img01 = img.float()[None] / 255.0

    def run_head():
        out = head(feats)
        ce = om.mask_clip_embed(ext.clip, img01, out["pred_masks"])
        return {"pred_masks": out["pred_masks"], "mask_embed": out["mask_embed"], "mask_pooled_features": out["mask_pooled_features"],
                "pred_logits": out["pred_logits"], "logit_scale": torch.as_tensor(float(out["logit_scale"])), "clip_embed": ce}

    r = dict(_cached(f"head_{size}_{seed}", digest, run_head, use_cache))
    r.update(feats)
    r["logit_scale"] = float(r["logit_scale"])
    return img, r


def classify_reference(heads, r):
    """CategoryODISE.forward after the head (odise.py:285-323) on the oracle's head outputs `r` under the vocabulary `heads` -> mask_cls."""
    with torch.no_grad():
        text_embed = heads.text_proj(heads.text_embed)
        null_embed = heads.text_proj(heads.null_embed)
        pred_logits = om.cal_pred_logits(r["mask_embed"], text_embed, null_embed, r["logit_scale"], heads.group_sizes)
        clip_logits = om.mask_clip_pred_logits(r["clip_embed"], heads.clip_text_embed, heads.group_sizes)
        open_logits = om.pooling_clip_head(pred_logits[..., :-1], clip_logits, heads.category_overlapping_mask, heads.alpha, heads.beta)
        return om.merge_with_null(pred_logits, open_logits)


def reference_with(bb, head, ext, size, heads, seed, use_cache=True):
    """Oracle pass over the seeded size x size image `seed` under a GIVEN vocabulary (the one spread over image 0: a batch shares one text
    bank, like every image of an evaluation run shares the dataset's).  Returns (img_u8, dict incl. mask_cls)."""
    key = ("ref_with", size, id(heads), seed)
    if key in _MEMO:
        return _MEMO[key]
    img, r = head_reference(bb, head, ext, size, seed, use_cache)
    r = dict(r)
    r["mask_cls"] = classify_reference(heads, r)
    _MEMO[key] = (img, r)
    return img, r


def reference(bb, head, ext, size, num_classes, num_strings, seed=0, use_cache=True):
    """Oracle pass over one size x size image (head_reference), then the spread vocabulary of `num_classes` classes / `num_strings` prompt
    strings and CategoryODISE.forward after the head (odise.py:285-323).  Returns (img_u8, heads, dict of torch tensors incl. mask_cls)."""
    key = ("ref", size, num_classes, num_strings, seed)
    if key in _MEMO:
        return _MEMO[key]
    img, r = head_reference(bb, head, ext, size, seed, use_cache)
    r = dict(r)
    cat, clp, sizes, overlap = synthetic_vocabulary(num_classes, num_strings, 768)
    heads = om.OpenVocabHeads(ext.clip, [int(s) for s in sizes], projection_dim=256, overlap=torch.from_numpy(overlap.astype(bool)))
    # at most ~2 present classes per (non-null) query: all of COCO-133 / ADE-150, a subset of the 847
    spread_vocabulary(heads, r["mask_embed"][0], r["clip_embed"][0], anchored=None if num_classes <= 200 else 188)
    r["mask_cls"] = classify_reference(heads, r)
    _MEMO[key] = (img, heads, r)
    return img, heads, r


def ideal_on_device_features(ext, head, heads, feats_dev, img, tag=None):
    """The fp32 ORACLE head + classification (CPU) on the DEVICE's backbone features of one picture: what an ideal, infinitely precise head
    would make of the features the device computed (tools/parity_attribution.py cell B, profiles/r05_parity_attribution.txt).  A device
    result that differs from the pure oracle by more than rounding noise on some query must differ from THIS by rounding noise only - then
    the re-decision is the reference's own decision chain reacting to the ~3e-3 feature error, not an error of the device's head.
    feats_dev: dict s2..s5 of fp32 NCHW arrays [1, 512, h, w] (HipODISE.backbone of that picture); img: uint8 CHW tensor.
    -> dict with pred_masks [1,Q,h,w], mask_embed, clip_embed, logit_scale, mask_cls [1,Q,K+1] (torch, fp32).  Memoised per (features, vocabulary)."""
    h = hashlib.sha256()
    for k in ("s2", "s3", "s4", "s5"):
        a = np.ascontiguousarray(feats_dev[k], np.float32)
        h.update(a[..., ::7].tobytes())
    key = ("ideal", h.hexdigest())
    if key not in _MEMO:
        img01 = img.float()[None] / 255.0
        with torch.no_grad():
            out = head({k: torch.from_numpy(np.ascontiguousarray(feats_dev[k], np.float32)) for k in ("s2", "s3", "s4", "s5")})
            ce = om.mask_clip_embed(ext.clip, img01, out["pred_masks"])
        _MEMO[key] = {"pred_masks": out["pred_masks"], "mask_embed": out["mask_embed"], "clip_embed": ce, "logit_scale": float(out["logit_scale"])}
    r = dict(_MEMO[key])
    r["mask_cls"] = classify_reference(heads, r)
    return r


def reference_instability(ext, head, heads, feats_dev, feats_ref, img, trials=4, seed=1234):
    """How far the REFERENCE's own class probabilities move, per query, when its backbone features are perturbed by errors of exactly the device's
    size: the fp32 oracle head + classifier on `feats_ref + s * (feats_dev - feats_ref)` for `trials` random sign patterns s (element-wise +-1;
    s = +1 everywhere is ideal_on_device_features), against its output on the unperturbed features.  A query whose distribution moves under such
    a perturbation sits on one of the decoder's hard decisions (attention masks `sigmoid < 0.5` at 9 layers, MaskCLIP's visibility bits): any error
    of that size - the backbone's or the head's own rounding - can send it either way, in the reference as on the device.
    feats_*: dict s2..s5 [1, 512, h, w] fp32 (numpy / torch).  -> [Q] max over the trials of max_k |p_perturbed - p_unperturbed|."""
    img01 = img.float()[None] / 255.0
    g = torch.Generator().manual_seed(seed)

    def run(feats):
        with torch.no_grad():
            out = head(feats)
            ce = om.mask_clip_embed(ext.clip, img01, out["pred_masks"])
            r = {"mask_embed": out["mask_embed"], "clip_embed": ce, "logit_scale": float(out["logit_scale"])}
            return torch.exp(classify_reference(heads, r)[0].double())

    ref = {k: torch.as_tensor(np.ascontiguousarray(feats_ref[k]), dtype=torch.float32) for k in ("s2", "s3", "s4", "s5")}
    dev = {k: torch.as_tensor(np.ascontiguousarray(feats_dev[k]), dtype=torch.float32) for k in ("s2", "s3", "s4", "s5")}
    p0 = run(ref)
    move = torch.zeros(p0.shape[0], dtype=torch.float64)
    for _ in range(trials):
        pert = {}
        for k in ref:
            sign = torch.randint(0, 2, ref[k].shape, generator=g, dtype=torch.int8).float() * 2 - 1
            pert[k] = ref[k] + sign * (dev[k] - ref[k])
        move = torch.maximum(move, (run(pert) - p0).abs().amax(-1))
    return move.numpy()
