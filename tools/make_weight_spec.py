"""Write odise_amd/weight_spec.json: name -> [shape, mean, std] of every tensor of the full-size synthetic ODISE(label) state
(SD-v1 UNet + VAE, CLIP ViT-L/14@336, ODISE heads), measured on the oracle's synthetic initialisation.  The product-side generator
(odise_amd/synthetic.py) draws N(mean, std) per tensor from this table, so bench.py needs neither checkpoints nor the oracle to build
weights of the real architecture.  Test infrastructure: run once (CPU, ~2 min), commit the JSON."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import odise_model as om  # noqa: E402
from oracle.backbone import FeatureExtractorBackbone  # noqa: E402
from oracle.ldm_extractor import ImplicitCaptionerExtractor  # noqa: E402
from oracle.m2f import SemSegHead, init_synthetic_  # noqa: E402

K = 133
ext = ImplicitCaptionerExtractor()
bb = FeatureExtractorBackbone(ext, [512, 512, 2560, 1920, 960, 640, 512, 512])
head = init_synthetic_(SemSegHead(num_classes=K))
heads = om.OpenVocabHeads(ext.clip, [1] * K, projection_dim=256)
state = ext.export_state()
state.update({"backbone.feature_projections." + k: v for k, v in bb.feature_projections.state_dict().items()})
state.update({"sem_seg_head." + k: v for k, v in head.state_dict().items()})
state["category_head.text_proj.weight"] = heads.text_proj.weight.detach()
state["category_head.text_proj.bias"] = heads.text_proj.bias.detach()
state["category_head.null_embed"] = heads.null_embed.detach()
spec = {}
for k, v in state.items():
    a = v.detach().float().numpy()
    spec[k] = [list(a.shape), float(a.mean()), float(a.std()) if a.size > 1 else 0.0]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "odise_amd", "weight_spec.json")
with open(out, "w") as f:
    json.dump(spec, f, separators=(",", ":"))
print(len(spec), "tensors,", sum(int(np.prod(s[0])) for s in spec.values()) / 1e6, "M parameters ->", out)
