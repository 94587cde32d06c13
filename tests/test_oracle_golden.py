"""The oracle against golden vectors produced by the REFERENCE's own modules.  The fixtures under tests/golden/ were
written in the build container by tests/golden/make_golden_m2f.py / make_golden_heads.py, which import
third_party/Mask2Former/mask2former/... and odise/modeling/meta_arch/odise.py from /root/reference (third-party imports stubbed by
tests/golden/ref_stubs.py) and run them on seeded inputs with the oracle's seeded weights; here only the .npz files are read."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle.m2f import SemSegHead, init_synthetic_

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _close(got, ref, tol, what):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)
    assert err < tol, f"{what}: {err:.3e}"


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "m2f_head_*.npz"))))
def test_mask_generator_matches_reference_modules(path):
    """MaskFormerHead = MSDeformAttnPixelDecoder + ODISEMultiScaleMaskedTransformerDecoder (+ PooledMaskEmbed / MaskPooling): every
    tensor the hot path consumes, fp32, within 5e-5 of max|ref| (summation order differs, nothing else may)."""
    z = np.load(path)
    head = init_synthetic_(SemSegHead(small=True, num_classes=int(z["num_classes"]), in_channels=int(z["in_channels"])), seed=int(z["seed"]))
    feats = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    with torch.no_grad():
        mask_features, _enc, multi_scale = head.pixel_decoder.forward_features(feats)
        out = head(feats)
    _close(mask_features, z["out_mask_features"], 2e-5, "mask_features")
    for i, m in enumerate(multi_scale):
        _close(m, z[f"out_multi_scale_{i}"], 2e-5, f"multi_scale[{i}]")
    for k in ("pred_logits", "pred_masks", "mask_embed", "mask_pooled_features"):
        _close(out[k], z["out_" + k], 5e-5, k)
    _close(out["logit_scale"], z["out_logit_scale"], 1e-6, "logit_scale")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "heads_*.npz"))))
def test_classification_and_postprocessing_match_reference_forward(path):
    """`CategoryODISE.forward` eval branch of the reference (category logits + ensemble, MaskCLIP with mask tokens, PoolingCLIPHead,
    null merge, upsampling, sem_seg_postprocess, semantic / panoptic / instance inference) replayed through oracle/odise_model.py."""
    from oracle import clip_vit, odise_model as om
    z = np.load(path)
    from golden_heads import build
    head, clip, heads, groups, things, caption = build(z)
    K = len(groups)
    feats = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("feat_")}
    sizes, out_sizes = [tuple(s) for s in z["sizes"].tolist()], [tuple(s) for s in z["out_sizes"].tolist()]
    B = len(sizes)
    H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)          # ImageList.from_tensors pads to the batch maximum (odise.py:242-244)
    images01 = torch.zeros(B, 3, H, W)
    for b in range(B):
        im = torch.from_numpy(z[f"image_{b}"]).float() / 255.0
        images01[b, :, :im.shape[-2], :im.shape[-1]] = im
    Hp, Wp = feats["s2"].shape[-2] * 4, feats["s2"].shape[-1] * 4
    with torch.no_grad():
        outputs = head(feats)
        mask_cls = om.caption_classify(heads, outputs, images01) if caption else heads.classify(outputs, images01)
        res = om.postprocess(mask_cls, outputs["pred_masks"], (Hp, Wp), sizes, out_sizes, K, things, float(z["overlap_threshold"]), int(z["topk"]))
    for b in range(B):
        ref_cls = z[f"mask_cls_{b}"]
        assert np.abs(mask_cls[b].numpy() - ref_cls).max() < 2e-4 * max(1.0, np.abs(ref_cls).max()), "mask_cls"
        sem = res[b]["sem_seg"].numpy()
        assert np.abs(sem - z[f"sem_seg_{b}"].astype(np.float32)).max() < 2e-3 * np.abs(sem).max()          # fixture stored as fp16
        assert (sem.argmax(0) == z[f"sem_argmax_{b}"]).mean() > 0.999
        pan, info = res[b]["panoptic_seg"]
        want_info = [{"id": int(i), "isthing": bool(t), "category_id": int(c)} for i, t, c in z[f"pan_info_{b}"]]
        assert info == want_info
        assert (pan.numpy() == z[f"pan_{b}"]).mean() > 0.999
        inst = res[b]["instances"]
        order, want_order = np.argsort(-inst["scores"].numpy(), kind="stable"), np.argsort(-z[f"inst_scores_{b}"], kind="stable")
        np.testing.assert_allclose(inst["scores"].numpy()[order], z[f"inst_scores_{b}"][want_order], rtol=2e-4, atol=1e-6)
        np.testing.assert_array_equal(inst["pred_classes"].numpy()[order], z[f"inst_classes_{b}"][want_order])
        area = inst["pred_masks"].flatten(1).sum(1).numpy()
        assert np.abs(area[order] - z[f"inst_area_{b}"][want_order]).max() <= 2


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "backbone_*.npz"))))
def test_sliding_window_backbone_matches_reference_class(path):
    """`FeatureExtractorBackbone.slide_forward / single_forward / forward_features` of the reference (window placement, bicubic resize of
    small windows, nearest restore, per-tap projection and sum, overlap averaging) over a stand-in tap extractor."""
    from oracle.backbone import FeatureExtractorBackbone, TapStandIn
    z = np.load(path)
    seed, crop = int(z["seed"]), int(z["crop"])
    dims = [16, 16, 32, 24, 16, 16, 16, 16]
    bb = FeatureExtractorBackbone(TapStandIn(dims, seed), dims, projection_dim=256, backbone_in_size=(crop, crop), seed=seed + 1)
    with torch.no_grad():
        out = bb(torch.from_numpy(z["image"]))
    for k in ("s2", "s3", "s4", "s5"):
        ref = z["out_" + k]
        _close(out[k], ref.astype(np.float32), 2e-3 if ref.dtype == np.float16 else 2e-5, k)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "extractor_*.npz"))))
def test_feature_extractor_driver_matches_reference_forward(path):
    """`LdmImplicitCaptionerExtractor.forward` -> `LdmExtractor.forward` of the reference (normalisation, deterministic latent, the
    reference's own GaussianDiffusion.q_sample on the shared noise, implicit-caption conditioning, tap selection and order) walking
    the oracle's narrow UNet / VAE / CLIP modules, against oracle/ldm_extractor.py on the same modules."""
    from oracle.backbone import FEATURE_STRIDES
    from oracle.ldm_extractor import ImplicitCaptionerExtractor
    z = np.load(path)
    ext = ImplicitCaptionerExtractor(unet_div=10, vae_div=4, clip_kw=dict(image_size=56, patch_size=14, width=128, layers=2, heads=2, output_dim=32),
                                     context_dim=64, seed=int(z["seed"]))
    assert z["feature_strides"].tolist() == FEATURE_STRIDES
    with torch.no_grad():
        feats = ext(torch.from_numpy(z["image"]))
    assert len(feats) == 8 and [f.shape[1] for f in feats] == z["feature_dims"].tolist()
    for i, f in enumerate(feats):
        ref = z[f"feat_{i}"]
        _close(f, ref.astype(np.float32), 2e-3 if ref.dtype == np.float16 else 5e-5, f"tap {i}")


def test_text_encoder_and_prompts_match_reference_code():
    """`ClipAdapter._encode_text` (clip.py:148-162) over the oracle's text tower, and `prompt_labels` (odise/data/build.py:54-71)."""
    import json
    from odise_amd.checkpoint import prompt_labels
    from oracle.clip_text import CLIPText, encode_hidden, encode_text, init_synthetic_ as init_text
    z = np.load(os.path.join(GOLD, "text_encode.npz"))
    model = init_text(CLIPText(vocab_size=49408, context_length=77, width=64, layers=2, heads=2, output_dim=32), seed=7).eval()
    tokens = torch.from_numpy(z["tokens"])
    with torch.no_grad():
        _close(encode_text(model, tokens), z["embed"], 2e-5, "text_embed")
        _close(encode_hidden(model, tokens), z["hidden"].astype(np.float32), 2e-3, "text_encodings")
    j = json.load(open(os.path.join(GOLD, "prompt_labels.json")))
    for prompt, want in j["prompted"].items():
        assert prompt_labels(j["labels"], None if prompt == "None" else prompt) == want


def test_checkpoint_uri_resolution_matches_reference_handlers(tmp_path, monkeypatch):
    """`odise://` / `sd://` (odise/utils/file_io.py:22-96): same names, same URLs, same `$ODISE_MODEL_ZOO/<basename>` rule for the SD v1
    checkpoints this path supports (the v2 entries of the reference's table belong to a different UNet)."""
    import json
    from odise_amd import checkpoint as ck
    j = json.load(open(os.path.join(GOLD, "file_io.json")))
    assert ck.ODISE_URLS == j["odise"]
    assert ck.SD_URLS == {k: v for k, v in j["sd"].items() if k.startswith("v1-")}
    monkeypatch.setenv("ODISE_MODEL_ZOO", str(tmp_path))
    for uri, rel in j["zoo_relative"].items():
        if uri.startswith("sd://v2"):
            with pytest.raises(KeyError):
                ck.resolve(uri)
            continue
        (tmp_path / rel).write_bytes(b"x")
        assert ck.resolve(uri) == str(tmp_path / rel)


def test_label_file_parser(tmp_path):
    """`get_openseg_labels` (odise/data/build.py:17-51) -> odise_amd.checkpoint.read_openseg_labels: the format on a hand-written file,
    and - where the reference checkout exists - every label file of the reference against the digests its own function produced."""
    import hashlib
    import json
    from odise_amd.checkpoint import read_openseg_labels
    f = tmp_path / "labels.txt"
    f.write_text("0:invalid_class_id\n1:person,child\n7:traffic light\n3:sky\n")
    assert read_openseg_labels(str(f)) == [["person", "child"], ["traffic light"], ["sky"]]
    summary = json.load(open(os.path.join(GOLD, "openseg_labels.json")))
    assert summary["coco_panoptic_with_prompt_eng"]["categories"] == 133 and summary["coco_panoptic_with_prompt_eng"]["strings"] == 254
    root = "/root/reference/odise/data/datasets/openseg_labels"
    if os.path.isdir(root):
        for name, want in summary.items():
            ls = read_openseg_labels(os.path.join(root, name + ".txt"))
            assert (len(ls), sum(len(l) for l in ls)) == (want["categories"], want["strings"])
            assert hashlib.sha256(json.dumps(ls).encode()).hexdigest() == want["sha256"], name
