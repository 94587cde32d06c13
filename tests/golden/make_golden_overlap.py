"""Writes tests/golden/overlap_ade150.json: PoolingCLIPHead's category_overlapping_mask for the ADE20K-150 vocabulary against the default
training vocabulary, computed with the REFERENCE's own `get_openseg_labels` (odise/data/build.py:17-51) and the set logic of
odise/modeling/meta_arch/odise.py:1479-1491.  Build container only."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
from odise.data.build import get_openseg_labels  # noqa: E402

train = get_openseg_labels("coco_panoptic", prompt_engineered=True)
test = get_openseg_labels("ade20k_150", prompt_engineered=True)
train_set = {l for label in train for l in label}
overlap = [int(not set(train_set).isdisjoint(set(t))) for t in test]
with open(os.path.join(HERE, "overlap_ade150.json"), "w") as f:
    json.dump({"train": "coco_panoptic_with_prompt_eng", "test": "ade20k_150_with_prompt_eng", "overlap": overlap}, f)
print(sum(overlap), "of", len(overlap), "ADE-150 categories overlap the COCO training vocabulary")
