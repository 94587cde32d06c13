"""How much of a full-size parity gap is the fp32 model's own sensitivity?  Runs the CPU oracle of the mask generator (oracle/m2f.py) on
the cached full-size backbone features twice - fp32, and with every weight and input rounded to fp16 and back (still fp32 arithmetic) -
for several synthetic initialisations, and prints the change of the mask logits and the per-query mask IoU.  A device path that computes
in fp16 cannot be closer to the fp32 oracle than this.  (CPU only; needs tests/.oracle_cache/feats_1024_0_*.npz, written by
`ODISE_ORACLE_CACHE_WRITE=1 python -m pytest tests/test_gpu_fullsize.py` or tests/fullsize.reference.)"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.m2f import SemSegHead, init_synthetic_  # noqa: E402

path = sorted(glob.glob(os.path.join(ROOT, "tests", ".oracle_cache", "feats_1024_0_*.npz")))[-1]
z = np.load(path)
feats = {k: torch.from_numpy(z[k]) for k in ("s2", "s3", "s4", "s5")}
feats16 = {k: v.half().float() for k, v in feats.items()}
for tag, kw in {"defaults (queries collapse)": {}, "branch_gain 0.3 (tests/fullsize.py)": dict(branch_gain=0.3), "qk_gain 2, level_gain 0.1": dict(qk_gain=2.0, level_gain=0.1),
                "qk_gain 4, level_gain 0.1": dict(qk_gain=4.0, level_gain=0.1)}.items():
    ref = init_synthetic_(SemSegHead(num_classes=133), **kw)(feats)
    h16 = init_synthetic_(SemSegHead(num_classes=133), **kw)
    with torch.no_grad():
        for p in h16.parameters():
            p.copy_(p.half().float())
    got = h16(feats16)
    pm, pr = got["pred_masks"], ref["pred_masks"]
    gb, rb = pm[0] > 0, pr[0] > 0
    iou = (gb & rb).flatten(1).sum(1).float() / (gb | rb).flatten(1).sum(1).float().clamp(min=1)
    b = rb.flatten(1).float()
    inter = b @ b.t()
    area = b.sum(1)
    pair = inter / (area[:, None] + area[None] - inter + 1e-9)
    print(f"{tag:38s}: mask logits move by {float((pm - pr).abs().max() / pr.abs().max()):.2e} of their max; per-query IoU min {float(iou.min()):.4f} "
          f"median {float(iou.median()):.5f}; mask_embed {float((got['mask_embed'] - ref['mask_embed']).abs().max() / ref['mask_embed'].abs().max()):.2e}; "
          f"mean IoU between different queries' masks {float((pair.sum() - pair.trace()) / (100 * 99)):.2f}", flush=True)
