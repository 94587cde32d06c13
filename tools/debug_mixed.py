"""DEBUG (temporary): locate the stage where a mixed-size batch of two images leaves the oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_model as T  # noqa: E402
from odise_amd.runtime import Context  # noqa: E402
from oracle import odise_model as om  # noqa: E402

ctx = Context(0)
from odise_amd.pipeline import HipCategoryODISE  # noqa: E402
from oracle.backbone import FeatureExtractorBackbone  # noqa: E402
from oracle.ldm_extractor import ImplicitCaptionerExtractor  # noqa: E402
from oracle.m2f import SemSegHead, init_synthetic_  # noqa: E402
ext = ImplicitCaptionerExtractor(**T.SMALL)
bb = FeatureExtractorBackbone(ext, [128, 128, 512, 384, 192, 128, 128, 128])
head = init_synthetic_(SemSegHead(small=True, num_classes=len(T.GROUPS)))
heads = om.OpenVocabHeads(ext.clip, T.GROUPS, projection_dim=64)
state = ext.export_state()
state.update({"backbone.feature_projections." + k: v for k, v in bb.feature_projections.state_dict().items()})
state.update({"sem_seg_head." + k: v for k, v in head.state_dict().items()})
state["category_head.text_proj.weight"] = heads.text_proj.weight.detach()
state["category_head.text_proj.bias"] = heads.text_proj.bias.detach()
state["category_head.null_embed"] = heads.null_embed.detach()
hip = HipCategoryODISE(ctx, state, overlap_threshold=0.0)
hip.set_vocabulary(heads.text_embed.numpy(), heads.clip_text_embed.numpy(), T.GROUPS, heads.category_overlapping_mask.numpy(), T.THINGS, heads.alpha, heads.beta)
a, b = T._image_u8(512, 704, seed=21), T._image_u8(576, 512, seed=22)
H, W = 576, 704
padded = torch.zeros(2, 3, H, W)
padded[0, :, :512, :704] = a.float() / 255
padded[1, :, :576, :512] = b.float() / 255


def rel(g, r):
    r = np.asarray(r, np.float64)
    return float(np.abs(np.asarray(g, np.float64) - r).max() / np.abs(r).max())


feats_ref = bb(padded)
got = hip.backbone(padded.numpy())
for k in ("s2", "s3", "s4", "s5"):
    print("backbone B=2 canvas", k, [rel(got[k][i], feats_ref[k][i].numpy()) for i in range(2)])
for i in range(2):
    g1 = hip.backbone(padded[i:i + 1].numpy())
    print("backbone single", i, [rel(g1[k][0], feats_ref[k][i].numpy()) for k in ("s2", "s3", "s4", "s5")])
out_ref = head(feats_ref)
h = hip.head({k: v.numpy() for k, v in feats_ref.items()})
print("head from ref feats: pred_masks", [rel(h["pred_masks"][i], out_ref["pred_masks"][i].numpy()) for i in range(2)], "mask_embed",
      [rel(h["mask_embed"][i], out_ref["mask_embed"][i].numpy()) for i in range(2)])
cls_ref = heads.classify(out_ref, padded)
cls = hip.classify_device(ctx.to_device(padded.numpy())).numpy()
print("class prob err", [float(np.abs(np.exp(cls[i]) - np.exp(cls_ref[i].numpy())).max()) for i in range(2)])
sizes = [(512, 704), (576, 512)]
ref = om.postprocess(cls_ref, out_ref["pred_masks"], (H, W), sizes, sizes, len(T.GROUPS), T.THINGS, 0.0)
res = hip.postprocess_batch(cls_ref.numpy(), (H, W), sizes, sizes)
for i in range(2):
    print("post from ref cls", i, "sem", rel(res[i]["sem_seg"], ref[i]["sem_seg"].numpy()), "pan agree", float((res[i]["panoptic_seg"][0] == ref[i]["panoptic_seg"][0].numpy()).mean()))
both = hip.forward([{"image": a}, {"image": b}])
for i in range(2):
    print("forward", i, "sem", rel(both[i]["sem_seg"], ref[i]["sem_seg"].numpy()), "pan agree", float((both[i]["panoptic_seg"][0] == ref[i]["panoptic_seg"][0].numpy()).mean()))
