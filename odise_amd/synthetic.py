"""Synthetic weights of the real architecture for benchmarking without checkpoints (no network here): every tensor of the full-size
ODISE(label) state - SD-v1 UNet and VAE, CLIP ViT-L/14@336, projections, Mask2Former heads, category head; 1630 tensors, 1.28 G
parameters - is drawn N(mean, std) from `weight_spec.json` (name -> [shape, mean, std]; the statistics of a fan-in scaled
initialisation that keeps activations O(1) through the residual stacks, tools/make_weight_spec.py).  Names are the checkpoint keys the
library looks up, so the result loads exactly like `checkpoint.assemble_state(...)` would."""
from __future__ import annotations

import json
import os
import zlib
from typing import Dict, Iterable, Optional

import numpy as np

SPEC_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "weight_spec.json")


def load_spec() -> Dict[str, list]:
    with open(SPEC_PATH) as f:
        return json.load(f)


def synthetic_state(prefixes: Optional[Iterable[str]] = None, strip: str = "", seed: int = 0) -> Dict[str, np.ndarray]:
    """fp32 tensors for every spec entry whose name starts with one of `prefixes` (all when None); `strip` is removed from the
    front of the returned names.  Per-tensor seeds derive from the name, so any subset reproduces the same values."""
    spec = load_spec()
    pre = tuple(prefixes) if prefixes is not None else None
    out = {}
    for name, (shape, mean, std) in spec.items():
        if pre is not None and not name.startswith(pre):
            continue
        rng = np.random.default_rng((zlib.crc32(name.encode()) << 16) ^ seed)
        a = rng.standard_normal(shape, dtype=np.float32) if len(shape) else np.float32(rng.standard_normal())
        a = np.asarray(a, np.float32) * np.float32(std) + np.float32(mean)
        out[name[len(strip):] if strip and name.startswith(strip) else name] = a
    if not out:
        raise KeyError(f"no tensors with prefixes {pre} in {SPEC_PATH}")
    return out


def synthetic_state_shared(prefixes: Optional[Iterable[str]] = None, strip: str = "", seed: int = 0, directory: Optional[str] = None):
    """`synthetic_state` once per HOST: the ranks of a multi-GPU run (bench.py --gpus N: one process per GPU, each loading the same 1.28 G
    synthetic parameters) share ONE fp32 copy in /dev/shm instead of generating and holding 5.1 GB each - the first process to take the file lock
    draws the tensors (same per-tensor seeds: bit-identical to `synthetic_state`), the others map the file read-only.  8 ranks: 5.1 GB of shared
    pages + each rank's transient staging copy inside the library, instead of 8 x (5.1 + 5.1) GB.
    -> (state dict of read-only views, handle); `release_shared(handle, unlink=...)` when the weights are on the device: every rank drops its
    mapping, ONE rank per host (after a barrier) unlinks the file - a mapped file lives until its last mapping goes."""
    import fcntl
    import hashlib
    import tempfile
    spec = load_spec()
    pre = tuple(prefixes) if prefixes is not None else None
    names = [n for n in spec if pre is None or n.startswith(pre)]
    if not names:
        raise KeyError(f"no tensors with prefixes {pre} in {SPEC_PATH}")
    sizes = [int(np.prod(spec[n][0])) if len(spec[n][0]) else 1 for n in names]
    total = int(sum(sizes))
    key = hashlib.sha1(json.dumps([[n, spec[n]] for n in names] + [seed]).encode()).hexdigest()[:16]
    root = directory or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
    base = os.path.join(root, f"odise_synth_{os.getuid()}_{key}")
    data, ok, lock = base + ".f32", base + ".ok", base + ".lock"
    with open(lock, "w") as lf:
        fcntl.flock(lf, fcntl.LOCK_EX)
        try:
            if not (os.path.exists(ok) and os.path.exists(data) and os.path.getsize(data) == total * 4):
                mm = np.memmap(data, np.float32, mode="w+", shape=(total,))
                off = 0
                for n, cnt in zip(names, sizes):
                    shape, mean, std = spec[n]
                    rng = np.random.default_rng((zlib.crc32(n.encode()) << 16) ^ seed)
                    a = rng.standard_normal(shape, dtype=np.float32) if len(shape) else np.float32(rng.standard_normal())
                    mm[off:off + cnt] = (np.asarray(a, np.float32) * np.float32(std) + np.float32(mean)).reshape(-1)
                    off += cnt
                mm.flush()
                del mm
                with open(ok, "w") as f:
                    f.write(str(total))
        finally:
            fcntl.flock(lf, fcntl.LOCK_UN)
    mm = np.memmap(data, np.float32, mode="r", shape=(total,))
    out, off = {}, 0
    for n, cnt in zip(names, sizes):
        shape = spec[n][0]
        out[n[len(strip):] if strip and n.startswith(strip) else n] = mm[off:off + cnt].reshape(shape) if len(shape) else mm[off:off + 1].reshape(())
        off += cnt
    return out, {"files": (data, ok, lock), "map": mm}


def release_shared(handle, unlink: bool) -> None:
    """Drop this process's mapping of a `synthetic_state_shared` file; `unlink=True` (one process per host, after every rank has mapped the
    file) removes it from /dev/shm - the pages go when the last mapping does."""
    handle.pop("map", None)
    if unlink:
        for f in handle.get("files", ()):
            try:
                os.unlink(f)
            except OSError:
                pass


def synthetic_vocabulary(num_classes: int = 133, num_strings: int = 254, dim: int = 768, seed: int = 7):
    """Arguments of `set_vocabulary` for a random text bank: `num_strings` prompt embeddings over `num_classes` synonym groups
    (COCO panoptic: 133 classes / 254 prompt-engineered strings)."""
    rng = np.random.default_rng(seed)
    sizes = np.ones(num_classes, np.int32)
    for i in rng.integers(0, num_classes, size=num_strings - num_classes):
        sizes[i] += 1
    cat = rng.standard_normal((num_strings, dim), dtype=np.float32)
    clp = rng.standard_normal((num_strings, dim), dtype=np.float32)
    overlap = (rng.random(num_classes) < 0.6).astype(np.int32)
    return cat, clp, sizes, overlap


# ---------------------------------------------------------------------------------------------------------------------------------
# Non-degenerate decisions on synthetic weights (bench.py).  Random weights as drawn collapse the 100 queries onto one mask and one
# label - every attention sublayer of the masked decoder adds the same vector to all queries - so the panoptic / instance decision
# kernels would run on an empty table.  The three adjustments below are the ones the full-size parity tests use (tests/fullsize.py,
# which applies them to the fp32 oracle): (a) residual-branch gain 0.3 in the masked decoder and the learned temperature at its clamp,
# (b) `mask_features.bias` shifted so that ~15 % of the mask logits are positive, (c) text banks - INPUTS of the path - spread over the
# queries' own embeddings.  Here they are functions of the DEVICE's outputs on the first image (numpy only; no oracle on this side).
# ---------------------------------------------------------------------------------------------------------------------------------
BRANCH_GAIN = 0.3


def apply_branch_gain(state: Dict[str, np.ndarray], gain: float = BRANCH_GAIN) -> None:
    """In place: scale the residual branches of the masked decoder (attention out_proj and FFN linear2 weights) and set
    `post_mask_embed.logit_scale` to ln(100), its clamp (odise.py:1013)."""
    for name in state:
        if name.startswith("sem_seg_head.predictor.") and (name.endswith("out_proj.weight") or name.endswith("linear2.weight")):
            state[name] = state[name] * np.float32(gain)
    state["sem_seg_head.predictor.post_mask_embed.logit_scale"] = np.asarray(np.log(100.0), np.float32)


def mask_embeddings_from(pred_masks: np.ndarray, mask_features: np.ndarray) -> np.ndarray:
    """The mask embeddings of the final prediction head - the [Q, C] matrix ME with pred_masks = ME . mask_features, an internal of the
    decoder that no API returns (`mask_embed` of the outputs is the POOLED embedding the classifier uses) - by least squares from the
    head's own outputs: ME = PM MF^T (MF MF^T)^-1.  pred_masks [Q, h, w], mask_features [C, h, w]."""
    pm = np.asarray(pred_masks, np.float64).reshape(pred_masks.shape[0], -1)
    mf = np.asarray(mask_features, np.float64).reshape(mask_features.shape[0], -1)
    return np.linalg.solve(mf @ mf.T, mf @ pm.T).T


def mask_bias_shift(mask_embed: np.ndarray, pred_masks: np.ndarray, positive_fraction: float = 0.15) -> np.ndarray:
    """delta [C] for `sem_seg_head.pixel_decoder.mask_features.bias`: me_q . delta = -s for every query, s = the (1 - positive_fraction)
    quantile of the current mask logits.  mask_embed [Q, C] = mask embeddings of the final prediction head (mask_embeddings_from),
    pred_masks [Q, h, w]."""
    me = np.asarray(mask_embed, np.float64)
    s = np.quantile(np.asarray(pred_masks, np.float64).reshape(-1)[::97], 1.0 - positive_fraction)
    return (-(s * (np.linalg.pinv(me) @ np.ones((me.shape[0], 1))))[:, 0]).astype(np.float32)


def spread_vocabulary(mask_embed: np.ndarray, clip_embed: np.ndarray, group_sizes, text_proj_w: np.ndarray, text_proj_b: np.ndarray,
                      seed: int = 5, null_queries: int = 6, null_bias: float = 0.07, anchored: Optional[int] = None):
    """Text banks that make the open-vocabulary decisions non-degenerate: every (present) class gets an anchor query and its prompt
    strings point along that query's embedding minus the mean embedding (plus seeded noise), mapped back through the pseudo-inverse of
    `category_head.text_proj`; the null embedding points at a handful of queries.  -> (cat_text [K_tot, 768], clip_text [K_tot, 768],
    null_embed [1, 768]).  mask_embed [Q, 256], clip_embed [Q, 768]."""
    rng = np.random.default_rng(seed)
    Q = mask_embed.shape[0]
    sizes = [int(s) for s in group_sizes]
    Kc = len(sizes)

    def directions(e):
        e = np.asarray(e, np.float64)
        e = e / np.linalg.norm(e, axis=-1, keepdims=True)
        mu = e.mean(0, keepdims=True)
        d = e - mu
        return d / np.linalg.norm(d, axis=-1, keepdims=True), mu / np.linalg.norm(mu)

    def off_mean(t, mu):
        return t - (t @ mu.T) * mu

    (d1, mu1), (d2, mu2) = directions(mask_embed), directions(clip_embed)
    perm = rng.permutation(Q)
    usable = Q - null_queries
    present = np.ones(Kc, bool)
    if anchored is not None and anchored < Kc:
        present[:] = False
        present[rng.permutation(Kc)[:anchored]] = True
    anchor1 = np.zeros(Kc, np.int64)
    for j, k in enumerate(np.flatnonzero(present)):
        anchor1[k] = perm[(j * 3) % usable]
    anchor2 = np.where(rng.random(Kc) < 0.7, anchor1, perm[rng.integers(0, usable, Kc)])
    t1, t2 = [], []
    for k, n in enumerate(sizes):
        for _ in range(n):
            n1 = rng.standard_normal(d1.shape[1]) / d1.shape[1] ** 0.5
            n2 = rng.standard_normal(d2.shape[1]) / d2.shape[1] ** 0.5
            t1.append(d1[anchor1[k]] + 0.5 * n1 if present[k] else 1.1 * n1)
            t2.append(d2[anchor2[k]] + 0.5 * n2 if present[k] else 1.1 * n2)
    t1, t2 = off_mean(np.stack(t1), mu1), off_mean(np.stack(t2), mu2)
    nq = d1[perm[-null_queries:]].sum(0)
    tn = off_mean((nq / np.linalg.norm(nq))[None], mu1) + null_bias * mu1
    W, b = np.asarray(text_proj_w, np.float64), np.asarray(text_proj_b, np.float64)
    pinv = np.linalg.pinv(W)                                                # [768, 256]: text = pinv (target - b) solves text_proj(text) = target
    return ((t1 - b) @ pinv.T).astype(np.float32), t2.astype(np.float32), ((tn - b) @ pinv.T).astype(np.float32)
