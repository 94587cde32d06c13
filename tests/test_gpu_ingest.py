"""GPU: weight ingestion + vocabulary building end to end.  The three weight sources are synthesised in their
real key layouts (SD checkpoint incl. the HF-named cond-stage text encoder, OpenAI CLIP incl. its text tower, ODISE checkpoint),
assembled by odise_amd.checkpoint.assemble_state, the vocabulary is tokenised and embedded on the device, and the resulting model
must reproduce the CPU oracle that was given the same constants (uncond_inputs from the oracle text tower, text banks from the oracle
text tower).  Tolerances as in tests/test_gpu_model.py."""
import numpy as np
import pytest
import torch

from odise_amd import checkpoint as ck
from odise_amd.pipeline import HipCategoryODISE, HipOpenPanopticInference
from odise_amd.text import HipTextEncoder
from odise_amd.tokenizer import SimpleTokenizer
from oracle import odise_model as om
from oracle.backbone import FeatureExtractorBackbone
from oracle.clip_text import CLIPText, empty_prompt_tokens, encode_hidden, encode_text, init_synthetic_ as init_text_
from oracle.ldm_extractor import ImplicitCaptionerExtractor
from oracle.m2f import SemSegHead, init_synthetic_
from tests.test_gpu_model import SMALL, _image_u8, _oracle_forward

pytestmark = pytest.mark.gpu
torch.set_num_threads(min(16, torch.get_num_threads()))

LABELS = [["person", "child"], ["sky"], ["tree", "trees", "bush"], ["car"], ["road", "street"], ["building"], ["dog"], ["grass"]]
THINGS = {0, 3, 6}


def _hf_names(m: CLIPText, prefix="cond_stage_model.transformer.text_model."):
    sd, W, out = m.state_dict(), m.positional_embedding.shape[1], {}
    out[prefix + "embeddings.token_embedding.weight"] = sd["token_embedding.weight"]
    out[prefix + "embeddings.position_embedding.weight"] = sd["positional_embedding"]
    out[prefix + "final_layer_norm.weight"], out[prefix + "final_layer_norm.bias"] = sd["ln_final.weight"], sd["ln_final.bias"]
    for i in range(len(m.transformer.resblocks)):
        r, q = f"transformer.resblocks.{i}.", prefix + f"encoder.layers.{i}."
        w, b = sd[r + "attn.in_proj_weight"], sd[r + "attn.in_proj_bias"]
        for j, n in enumerate("qkv"):
            out[q + f"self_attn.{n}_proj.weight"], out[q + f"self_attn.{n}_proj.bias"] = w[j * W:(j + 1) * W], b[j * W:(j + 1) * W]
        for a, c in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"),
                     ("mlp.c_proj", "mlp.fc2")):
            out[q + c + ".weight"], out[q + c + ".bias"] = sd[r + a + ".weight"], sd[r + a + ".bias"]
    return out


def test_ingested_checkpoints_and_device_vocabulary_match_oracle(ctx):
    ext = ImplicitCaptionerExtractor(**SMALL)
    bb = FeatureExtractorBackbone(ext, [128, 128, 512, 384, 192, 128, 128, 128])
    head = init_synthetic_(SemSegHead(small=True, num_classes=len(LABELS)))
    sd_text = init_text_(CLIPText(width=768, layers=2, heads=12, output_dim=768), seed=1).eval()       # SD cond stage (HF names)
    clip_text = init_text_(CLIPText(width=128, layers=2, heads=2, output_dim=64), seed=2).eval()       # OpenAI CLIP text tower
    exported = ext.export_state()
    # ---- the three sources in their container layouts
    sd_state = {k: v for k, v in exported.items() if k.startswith(("model.diffusion_model.", "first_stage_model."))}
    sd_state.update(_hf_names(sd_text))
    clip_state = {k[len("clip."):]: v for k, v in exported.items() if k.startswith("clip.")}
    clip_state.update(clip_text.state_dict())
    odise_state = {k: v for k, v in exported.items() if k.startswith("backbone.feature_extractor.") and "ldm_extractor" not in k}
    odise_state.update({"backbone.feature_projections." + k: v for k, v in bb.feature_projections.state_dict().items()})
    odise_state.update({"sem_seg_head." + k: v for k, v in head.state_dict().items()})
    heads = om.OpenVocabHeads(ext.clip, [len(l) for l in LABELS], projection_dim=64)
    odise_state["category_head.text_proj.weight"] = heads.text_proj.weight.detach()
    odise_state["category_head.text_proj.bias"] = heads.text_proj.bias.detach()
    odise_state["category_head.null_embed"] = heads.null_embed.detach()
    state = ck.assemble_state(ctx, sd_state=sd_state, clip_state=clip_state, odise_state=odise_state)
    # ---- derived constants
    with torch.no_grad():
        unc_ref = encode_hidden(sd_text, empty_prompt_tokens())
    unc = state["backbone.feature_extractor.ldm_extractor.ldm.uncond_inputs"]
    err = np.abs(unc - unc_ref.numpy()).max() / np.abs(unc_ref.numpy()).max()
    print("uncond_inputs err", err)
    assert unc.shape == (1, 77, 768) and err < 3e-3
    np.testing.assert_array_equal(state["backbone.feature_extractor.ldm_extractor.shared_noise"], ext.shared_noise.numpy())
    for k in ("model.diffusion_model.time_embed.0.weight", "clip.visual.proj", "sem_seg_head.predictor.query_feat.weight"):
        assert k in state
    # ---- vocabulary on the device (byte-level tokens: the real merges file is not available offline)
    tok = SimpleTokenizer(merges=[(f"\\u0001{i}", f"\\u0002{i}") for i in range(49152 - 256 - 2)])
    enc = HipTextEncoder(ctx, clip_state, heads=2)
    cat, clp, sizes, overlap = ck.build_vocabulary(LABELS, tok, enc, train_labels=[["sky"], ["car", "truck"], ["dog"]])
    flat = lambda ls: [s for l in ls for s in l]
    with torch.no_grad():
        cat_ref = encode_text(clip_text, torch.from_numpy(tok(flat(LABELS))))
        clp_ref = encode_text(clip_text, torch.from_numpy(tok(flat(ck.prompt_labels(LABELS, "photo")))))
    for got, ref, name in ((cat, cat_ref, "category bank"), (clp, clp_ref, "clip bank")):
        e = np.abs(got - ref.numpy()).max() / np.abs(ref.numpy()).max()
        print(name, "err", e)
        assert got.shape == ref.shape and e < 3e-3
    assert sizes.tolist() == [len(l) for l in LABELS] and overlap.tolist() == [0, 1, 0, 1, 0, 0, 1, 0]
    # ---- the assembled model against the oracle that uses the same constants
    ext.uncond_inputs.copy_(unc_ref)
    heads.text_embed, heads.clip_text_embed = cat_ref.clone(), clp_ref.clone()
    heads.category_overlapping_mask = torch.as_tensor(overlap).long()
    hip = HipCategoryODISE(ctx, state, overlap_threshold=0.0)
    hip.set_vocabulary(cat, clp, sizes, overlap, THINGS, heads.alpha, heads.beta)
    img = _image_u8(512, 512, seed=21)
    _, _, ref = _oracle_forward(bb, head, heads, img, (512, 512), 0.0)
    # _oracle_forward post-processes with the module-level vocabulary of test_gpu_model: redo it for this vocabulary
    imgf = img.float()[None] / 255.0
    outputs = head(bb(imgf))
    mask_cls = heads.classify(outputs, imgf)
    ref = om.postprocess(mask_cls, outputs["pred_masks"], (512, 512), [(512, 512)], [(512, 512)], len(LABELS), THINGS, 0.0)[0]
    got = hip.forward([{"image": img}])[0]
    pan_ref, info_ref = ref["panoptic_seg"]
    pan, info = got["panoptic_seg"]
    agree = (pan == pan_ref.numpy()).mean()
    print("segments", info, "ref", info_ref, "panoptic agreement", agree)
    assert info == info_ref and agree > 0.995
    sem_ref = ref["sem_seg"].numpy()
    assert np.abs(got["sem_seg"] - sem_ref).max() / np.abs(sem_ref).max() < 2e-2
    # ---- OpenPanopticInference (pano_wrapper.py:20-70): another label set for one call, the model's own vocabulary afterwards
    train = [["sky"], ["car", "truck"], ["dog"]]
    hip.attach_text(tok, enc, train_labels=train)
    labels2, things2 = LABELS[:5], {0, 3}
    wrap = HipOpenPanopticInference(hip, labels2, metadata={"thing_ids": sorted(things2)}, instance_on=False)
    got2 = wrap([{"image": img}])[0]
    assert "instances" not in got2 and got2["sem_seg"].shape[0] == len(labels2)
    after = hip.forward([{"image": img}])[0]
    np.testing.assert_array_equal(after["panoptic_seg"][0], pan)
    np.testing.assert_array_equal(after["sem_seg"], got["sem_seg"])
    assert "instances" in after
    cat2, clp2, sizes2, ov2 = ck.build_vocabulary(labels2, tok, enc, train_labels=train)
    hip.set_vocabulary(cat2, clp2, sizes2, ov2, things2, heads.alpha, heads.beta)
    hip.instance_on = False
    exp2 = hip.forward([{"image": img}])[0]
    np.testing.assert_array_equal(got2["panoptic_seg"][0], exp2["panoptic_seg"][0])
    np.testing.assert_array_equal(got2["sem_seg"], exp2["sem_seg"])
    # ---- an image already resident in HBM as uint8 [H,W,3] (HipDatasetMapper's output) takes the same path without a host round trip
    hip.load_open_state_dict({"category_head.test_labels": LABELS, "thing_ids": THINGS, "instance_on": True})
    dimg = ctx.to_device(np.ascontiguousarray(img.numpy().transpose(1, 2, 0)))
    res = hip.forward([{"image": dimg, "height": 512, "width": 512}])[0]
    assert (res["panoptic_seg"][0] == pan).mean() > 0.999 and res["panoptic_seg"][1] == info
