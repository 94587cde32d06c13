"""Margin contract shared by the GPU parity tests (test infrastructure).

The device path computes in fp16 with fp32 accumulation, the references in fp32: continuous quantities differ by a measured error, and a
DECISION taken on them (arg-max label, `logit > 0`, panoptic arg-max over score x sigmoid) can only differ legitimately where the
reference's own margin is inside that error.  The helpers below turn the MEASURED per-query errors into the set of decisions that must be
identical: if a decision kernel upsamples, crops, multiplies or breaks ties differently from maskformer_model.py:280-342, it shows up as
a mismatch on a decided element, however small the numeric error is.  Everything here is numpy / torch on the host."""
import numpy as np
import torch
import torch.nn.functional as F


def per_query_errors(p_got, p_ref, logits_got, logits_ref):
    """-> (eprob [Q], elogit [Q]): max class-probability error and max mask-logit error of every query (logits at the head's resolution:
    a bilinear resampling is a convex combination, so the bound carries over to every output resolution)."""
    eprob = np.abs(np.asarray(p_got, np.float64) - np.asarray(p_ref, np.float64)).max(-1)
    elogit = np.abs(np.asarray(logits_got, np.float64) - np.asarray(logits_ref, np.float64)).reshape(p_ref.shape[0], -1).max(-1)
    return eprob, elogit


def decided_labels(p_ref, eprob):
    """Queries whose arg-max label cannot move: reference top-2 margin above twice the query's own measured probability error."""
    top2 = np.sort(np.asarray(p_ref, np.float64), axis=-1)[:, -2:]
    return (top2[:, 1] - top2[:, 0]) > 2.0 * eprob


def upsampled_reference_logits(pred_masks, padded_hw, image_size, out_size):
    """odise.py:326-331 + sem_seg_postprocess for ONE image: [Q,h4,w4] -> [Q,oh,ow] (fp32 torch)."""
    up = F.interpolate(torch.as_tensor(pred_masks)[None].float(), size=tuple(padded_hw), mode="bilinear", align_corners=False)[0]
    up = up[:, : image_size[0], : image_size[1]]
    if tuple(out_size) != tuple(up.shape[-2:]):
        up = F.interpolate(up[None], size=tuple(out_size), mode="bilinear", align_corners=False)[0]
    return up


def decided_panoptic_pixels(mask_cls_ref, logits_up_ref, num_classes, eprob, elogit, object_mask_threshold=0.0):
    """Pixels whose panoptic arg-max winner AND inside-mask flag (maskformer_model.py:286-320) are fixed by the reference's margins:
    with d_q = eprob_q + elogit_q / 4 bounding the error of score_q x sigmoid(logit_q) (|d sigmoid| <= |d logit| / 4), the winner w of a pixel
    is decided iff score_w sigma_w - d_w > max_{q != w} (score_q sigma_q + d_q), and its `sigmoid >= 0.5` flag iff |logit_w| > elogit_w.
    mask_cls_ref [Q,K+1] log-probabilities, logits_up_ref [Q,oh,ow].  -> bool [oh,ow] (all True when at most one query is kept)."""
    probs = torch.softmax(torch.as_tensor(mask_cls_ref).float(), -1)
    scores, labels = probs.max(-1)
    keep = (labels != num_classes) & (scores > object_mask_threshold)
    up = torch.as_tensor(logits_up_ref).float()
    if int(keep.sum()) == 0:
        return np.ones(tuple(up.shape[-2:]), bool)
    d = torch.as_tensor(eprob + 0.25 * elogit, dtype=torch.float32)[keep]
    lg = up[keep]
    pm = scores[keep].view(-1, 1, 1) * lg.sigmoid()
    w = pm.argmax(0)
    lower = torch.gather(pm - d.view(-1, 1, 1), 0, w[None])[0]
    ok = torch.ones_like(lower, dtype=torch.bool)
    if pm.shape[0] > 1:
        upper = pm + d.view(-1, 1, 1)
        top = upper.topk(2, dim=0)
        rival = torch.where(top.indices[0] == w, top.values[1], top.values[0])
        ok = lower > rival
    flag = torch.gather(lg.abs(), 0, w[None])[0] > torch.as_tensor(elogit, dtype=torch.float32)[keep][w]
    return (ok & flag).numpy()


def decided_semantic_pixels(sem_ref, max_err):
    """Pixels whose semantic arg-max cannot move: reference top-2 score margin above twice the measured score error."""
    top2 = np.partition(np.asarray(sem_ref), -2, axis=0)[-2:]
    return (top2[1] - top2[0]) > 2.0 * max_err


def instance_keys(mask_cls_ref, num_classes, thing_ids, topk, panoptic_on=True):
    """The reference's instance selection (maskformer_model.py:344-369) as {(query, class): rank-free index} plus the k-th score."""
    scores = torch.softmax(torch.as_tensor(mask_cls_ref).float(), -1)[:, :-1].flatten()
    k = min(topk, scores.numel())
    top = scores.topk(k, sorted=False).indices
    q, c = (top // num_classes).numpy(), (top % num_classes).numpy()
    keep = np.array([(int(x) in thing_ids) or not panoptic_on for x in c], bool)
    kth = float(np.sort(scores.numpy())[-k])
    return {(int(a), int(b)): i for i, (a, b) in enumerate(zip(q[keep], c[keep]))}, kth, scores.numpy()


def segments_decided(mask_cls_ref, logits_up_ref, num_classes, thing_ids, eprob, elogit, overlap_threshold, trials=10, seed=0):
    """Is the reference's `segments_info` FIXED by its own margins at the measured error?  The panoptic table is the end of a chain of
    thresholds (kept queries, per-pixel arg-max, `mask_area / original_area >= overlap_threshold` per query, stuff merging;
    maskformer_model.py:286-342) and a closed-form margin for the whole chain would have to be very conservative; instead the oracle's
    own inputs are perturbed `trials` times inside the MEASURED per-query error bounds - class probabilities by uniform noise in
    +-eprob_q (renormalised), mask logits by uniform noise in +-elogit_q per pixel - and the table must come out identical every time.
    Deterministic given the measured errors.  -> (decided, number of perturbed tables that differ)."""
    from oracle import odise_model as om
    g = torch.Generator().manual_seed(seed)
    lp = torch.as_tensor(mask_cls_ref).float().reshape(-1, num_classes + 1)
    up = torch.as_tensor(logits_up_ref).float()
    # amplitude 1.5 x the measured errors: random draws inside a bound find its worst case less often than a structured error does
    ep = 1.5 * torch.as_tensor(np.asarray(eprob), dtype=torch.float32).view(-1, 1)
    el = 1.5 * torch.as_tensor(np.asarray(elogit), dtype=torch.float32).view(-1, 1, 1)
    _, want = om.panoptic_inference(lp, up, num_classes, thing_ids, 0.0, overlap_threshold)
    differ = 0
    for _ in range(trials):
        p = (lp.exp() + ep * (2 * torch.rand(lp.shape, generator=g) - 1)).clamp_min(1e-9)
        p = p / p.sum(-1, keepdim=True)
        noisy = up + el * (2 * torch.rand(up.shape, generator=g) - 1)
        _, info = om.panoptic_inference(p.log(), noisy, num_classes, thing_ids, 0.0, overlap_threshold)
        differ += info != want
    return differ == 0, differ
