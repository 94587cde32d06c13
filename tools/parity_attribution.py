"""Where do the re-decided queries of the full-size parity tests come from - the backbone's error or the head's precision?  

2 x 2 attribution at 1024 x 1024 (BASELINE configs[2] shapes, the weights / vocabulary of tests/fullsize.py), per picture:

                              | head = fp32 ORACLE (CPU)            | head = DEVICE (fp16 storage, fp32 accumulate)
    features = ORACLE (fp32)  | the reference itself                | A: the head's own error
    features = DEVICE         | B: an IDEAL fp32 head on the        | C: what the product computes
                              |    device's backbone features       |

B is the upper bound of what any higher-precision head on the device (an fp32 residual stream, fp32 mask_features / mask_embed) could
deliver: if B still re-decides queries, the 3e-3 backbone error alone moves the decoder's hard decisions and the head's precision is not
the lever; if B is clean and A is not, the head is.  Every cell is compared with the reference: per-query worst mask-logit error / max|logit|,
queries above TAU_MASK, raw per-query IoU of the binary masks, per-query class-probability error, queries above TAU_PROB, label agreement.
The classification of a cell runs where its head ran (device: odise_hip_classify; oracle: oracle/odise_model.py on the CPU).

    python tools/parity_attribution.py [pictures=2]       (GPU; ~1 min of host oracle time per picture)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fullsize import build_models, category_head_state, classify_reference, export_state, reference, reference_with  # noqa: E402
from odise_amd.pipeline import HipCategoryODISE  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402
from oracle import odise_model as om  # noqa: E402

TAU_MASK, TAU_PROB = 2.5e-2, 3e-2
K, K_TOT = 133, 254
THINGS = set(range(80))
torch.set_num_threads(min(32, torch.get_num_threads()))


def cell(tag, pm, logp, pm_ref, logp_ref):
    scale = np.abs(pm_ref).max()
    err = np.abs(pm - pm_ref) / scale
    qerr = err.reshape(err.shape[0], -1).max(1)
    gb, rb = pm > 0, pm_ref > 0
    iou = (gb & rb).sum((1, 2)) / np.maximum((gb | rb).sum((1, 2)), 1)
    p, pr = np.exp(np.asarray(logp, np.float64)), np.exp(np.asarray(logp_ref, np.float64))
    eprob = np.abs(p - pr).max(-1)
    same = int((p.argmax(-1) == pr.argmax(-1)).sum())
    print(f"  {tag:42s} mask: max {qerr.max():.2e} median-query {np.median(qerr):.2e} queries>{TAU_MASK}: {int((qerr >= TAU_MASK).sum()):2d}  IoU min {iou.min():.4f} "
          f"med {np.median(iou):.5f} <1-1e-3: {int((iou < 1 - 1e-3).sum()):3d} | prob: max {eprob.max():.2e} median-query {np.median(eprob):.2e} "
          f"queries>{TAU_PROB}: {int((eprob >= TAU_PROB).sum()):2d} labels same {same}/100", flush=True)
    return dict(qerr=qerr, eprob=eprob, iou=iou)


def main():
    n_pic = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ctx = Context(0)
    ext, bb, head = build_models(K)
    img0, heads, r0 = reference(bb, head, ext, 1024, K, K_TOT)
    hip = HipCategoryODISE(ctx, export_state(ext, bb, head, heads), overlap_threshold=0.8)
    hip.load_category_head(category_head_state(heads))
    hip.set_vocabulary(heads.text_embed.numpy(), heads.clip_text_embed.numpy(), heads.group_sizes, heads.category_overlapping_mask.numpy(), THINGS,
                       heads.alpha, heads.beta)
    tot = {}
    for seed in range(n_pic):
        img, r = (img0, r0) if seed == 0 else reference_with(bb, head, ext, 1024, heads, seed)
        img01 = img.float()[None] / 255.0
        pm_ref, logp_ref = r["pred_masks"][0].numpy(), r["mask_cls"][0].numpy()
        feats_ref = {k: r[k] for k in ("s2", "s3", "s4", "s5")}
        dev01 = ctx.to_device(img01.numpy())
        print(f"picture {seed}:", flush=True)
        # device features
        feats_dev = hip.backbone(img01.numpy())
        for k in ("s2", "s3", "s4", "s5"):
            ref = feats_ref[k].numpy()
            print(f"  features {k}: max-err/scale {np.abs(feats_dev[k] - ref).max() / np.abs(ref).max():.2e}")
        # C: device head on device features (the arena still holds the backbone's maps; classify reads the head's outputs)
        outC = hip.head(feats_dev)
        logpC = hip.classify_device(dev01).numpy()[0]
        res = {"C device feats + device head": cell("C: device features -> device head", outC["pred_masks"][0], logpC, pm_ref, logp_ref)}
        # A: device head on oracle features
        outA = hip.head({k: v.numpy() for k, v in feats_ref.items()})
        logpA = hip.classify_device(dev01).numpy()[0]
        res["A oracle feats + device head"] = cell("A: oracle features -> device head", outA["pred_masks"][0], logpA, pm_ref, logp_ref)
        # B: oracle head (fp32, CPU) on device features
        with torch.no_grad():
            outB = head({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in feats_dev.items()})
            ceB = om.mask_clip_embed(ext.clip, img01, outB["pred_masks"])
            rB = {"mask_embed": outB["mask_embed"], "clip_embed": ceB, "logit_scale": float(outB["logit_scale"])}
            logpB = classify_reference(heads, rB)[0].numpy()
        res["B device feats + fp32 oracle head"] = cell("B: device features -> fp32 ORACLE head", outB["pred_masks"][0].numpy(), logpB, pm_ref, logp_ref)
        # the head's own error at the operating point: C against B (same features)
        cell("C against B (same device features)", outC["pred_masks"][0], logpC, outB["pred_masks"][0].numpy(), logpB)
        for k, v in res.items():
            t = tot.setdefault(k, dict(mask=0, prob=0, iou=[], n=0))
            t["mask"] += int((v["qerr"] >= TAU_MASK).sum())
            t["prob"] += int((v["eprob"] >= TAU_PROB).sum())
            t["iou"].append(float(v["iou"].min()))
            t["n"] += 1
    print("summary over", n_pic, "pictures (queries above the bound, summed; worst raw IoU):")
    for k, t in tot.items():
        print(f"  {k:36s} mask-logit re-decided {t['mask']:3d}   class-probability re-decided {t['prob']:3d}   worst IoU {min(t['iou']):.4f}")


if __name__ == "__main__":
    main()
