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

# ---- second question: the END-TO-END bound.  The device's backbone features differ from the oracle's by up to ~3e-3 of their maximum
# (tests/test_gpu_fullsize.py::test_backbone_full_size); how far does the fp32 head itself move when its input is perturbed by that much?
# Gaussian noise per stride, scaled so that its largest deviation equals `rel` x max|feature|, several seeds: the worst-pixel and the
# 99.9 % figures below are what an end-to-end comparison has to allow on top of the head's own fp16 error.
print("\nfp32 head (set-up of tests/fullsize.py) under perturbed backbone features:", flush=True)
head = init_synthetic_(SemSegHead(num_classes=133), branch_gain=0.3)
with torch.no_grad():
    base = head(feats)["pred_masks"]
    scale = float(base.abs().max())
    for rel in (1e-3, 3e-3):
        worst, p999, ious = [], [], []
        for seed in range(4):
            g = torch.Generator().manual_seed(seed)
            noisy = {}
            for k, v in feats.items():
                n = torch.randn(v.shape, generator=g)
                noisy[k] = v + n * (rel * float(v.abs().max()) / float(n.abs().max()))
            pm = head(noisy)["pred_masks"]
            err = (pm - base).abs() / scale
            worst.append(float(err.max()))
            p999.append(float(torch.quantile(err.flatten()[::7], 0.999)))
            gb, rb = pm[0] > 0, base[0] > 0
            ious.append(float(((gb & rb).flatten(1).sum(1).float() / (gb | rb).flatten(1).sum(1).float().clamp(min=1)).min()))
        print(f"  feature noise with max deviation {rel:.0e} x max|feature|: mask logits move by (worst pixel) " + " ".join(f"{w:.2e}" for w in worst) +
              "; 99.9 % of the pixels below " + " ".join(f"{p:.2e}" for p in p999) + "; per-query IoU min " + " ".join(f"{i:.4f}" for i in ious), flush=True)

# ---- third question: how much does ANY fp16-storage implementation of this head differ from the fp32 one, and how much does that figure
# vary between realisations?  Same fp32 arithmetic, but the output of every leaf module (linear, conv, norm, activation, attention) is
# rounded to fp16 - what a device path that keeps its activations in fp16 does - and the input carries a perturbation of 1e-6 of its
# maximum (five seeds): the decisions inside the head (the masked attention's `sigmoid(mask) < 0.5`, 9 layers x 100 queries x 16 384 keys)
# flip for logits near zero, so the error is a draw from a distribution, not a constant.  The end-to-end tolerance of
# tests/test_gpu_fullsize.py has to cover that distribution.
print("\nfp32 head with every leaf-module output rounded to fp16, inputs perturbed by 1e-6 of their maximum:", flush=True)


def round_out(_m, _inp, out):
    if torch.is_tensor(out) and out.dtype == torch.float32:
        return out.half().float()
    if isinstance(out, tuple):
        return tuple(o.half().float() if torch.is_tensor(o) and o.dtype == torch.float32 else o for o in out)
    return out


h16 = init_synthetic_(SemSegHead(num_classes=133), branch_gain=0.3)
hooks = [m.register_forward_hook(round_out) for m in h16.modules() if len(list(m.children())) == 0]
with torch.no_grad():
    for p in h16.parameters():
        p.copy_(p.half().float())
    for seed in range(5):
        g = torch.Generator().manual_seed(100 + seed)
        noisy = {}
        for k, v in feats16.items():
            n = torch.randn(v.shape, generator=g)
            noisy[k] = v + n * (1e-6 * float(v.abs().max()) / float(n.abs().max())) if seed else v
        pm = h16(noisy)["pred_masks"]
        err = (pm - base).abs() / scale
        gb, rb = pm[0] > 0, base[0] > 0
        iou = (gb & rb).flatten(1).sum(1).float() / (gb | rb).flatten(1).sum(1).float().clamp(min=1)
        print(f"  realisation {seed}: worst pixel {float(err.max()):.2e}; 99.9 % of the pixels below {float(torch.quantile(err.flatten()[::7], 0.999)):.2e}; "
              f"per-query IoU min {float(iou.min()):.4f} median {float(iou.median()):.5f}", flush=True)
for hk in hooks:
    hk.remove()
