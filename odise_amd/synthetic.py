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
