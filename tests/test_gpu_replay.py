"""GPU: the reference's calling conventions around the model (tools/replay.py: demo.py:153-171 `predict`, evaluator.py:60-142 loop) on a
narrow model - plumbing (formats, shapes, dtypes, a 1486-class vocabulary through every head), not arithmetic."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
pytestmark = pytest.mark.gpu


def test_demo_predict_and_eval_loop_replay(ctx):
    import replay
    import test_gpu_model as T
    from odise_amd.pipeline import HipCategoryODISE
    from odise_amd.synthetic import synthetic_vocabulary
    from oracle import odise_model as om
    from oracle.backbone import FeatureExtractorBackbone
    from oracle.ldm_extractor import ImplicitCaptionerExtractor
    from oracle.m2f import SemSegHead, init_synthetic_
    ext = ImplicitCaptionerExtractor(**T.SMALL)
    bb = FeatureExtractorBackbone(ext, [128, 128, 512, 384, 192, 128, 128, 128])
    head = init_synthetic_(SemSegHead(small=True, num_classes=1486))
    heads = om.OpenVocabHeads(ext.clip, [1], projection_dim=64)
    state = ext.export_state()
    state.update({"backbone.feature_projections." + k: v for k, v in bb.feature_projections.state_dict().items()})
    state.update({"sem_seg_head." + k: v for k, v in head.state_dict().items()})
    state["category_head.text_proj.weight"], state["category_head.text_proj.bias"] = heads.text_proj.weight.detach(), heads.text_proj.bias.detach()
    state["category_head.null_embed"] = heads.null_embed.detach()
    hip = HipCategoryODISE(ctx, state, overlap_threshold=0.0)
    K, K_TOT = 1486, 2482                                                  # demo.py's default COCO + ADE + LVIS vocabulary shape
    cat, clp, sizes, overlap = synthetic_vocabulary(K, K_TOT, 64)
    hip.set_vocabulary(cat, clp, sizes, overlap, set(range(80 + 100 + 1203)), 0.35, 0.65)
    pic = replay.bench.image_u8(256, 0)[:, :192]                           # a 256 x 192 picture: resized to 1365 x 1024 by ResizeShortestEdge
    pred, net_hw = replay.demo_predict(hip, ctx, np.ascontiguousarray(pic))
    assert net_hw == (1365, 1024)
    segs, n = replay.check_demo_outputs(pred, K, 256, 192)
    print("demo replay on the narrow model: segments", segs, "instances", n)
    batches = [[{"image": ctx.to_device(replay.bench.image_u8(512, i)), "height": 512, "width": 512}] for i in range(8)]
    comp, tot, ips = replay.inference_on_dataset(hip, ctx, batches, len(batches))
    assert comp > 0 and tot >= comp * 0.99 and ips > 0
