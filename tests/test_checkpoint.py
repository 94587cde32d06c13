"""CPU: weight-ingestion plumbing (odise_amd/checkpoint.py <- odise/utils/file_io.py:22-96, odise_checkpointer.py:54-140, ldm.py:66-74,
116, 273-277) on synthetic files in the three real container formats, and the vocabulary helpers (data/build.py:54-71,
odise.py:1479-1491)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from odise_amd import checkpoint as ck


def test_resolve_uses_the_model_zoo_and_never_downloads(tmp_path, monkeypatch):
    monkeypatch.setenv("ODISE_MODEL_ZOO", str(tmp_path))
    with pytest.raises(FileNotFoundError) as e:
        ck.resolve("sd://v1-3")
    assert "sd-v1-3.ckpt" in str(e.value) and "huggingface.co" in str(e.value)
    (tmp_path / "sd-v1-3.ckpt").write_bytes(b"x")
    (tmp_path / "odise_label_coco_50e-b67d2efc.pth").write_bytes(b"x")
    assert ck.resolve("sd://v1-3") == str(tmp_path / "sd-v1-3.ckpt")
    assert ck.resolve("odise://Panoptic/odise_label_coco_50e").endswith("odise_label_coco_50e-b67d2efc.pth")
    with pytest.raises(KeyError):
        ck.resolve("odise://Panoptic/nope")
    assert ck.resolve(str(tmp_path / "sd-v1-3.ckpt")) == str(tmp_path / "sd-v1-3.ckpt")
    with pytest.raises(FileNotFoundError):
        ck.resolve(str(tmp_path / "missing.pth"))


def test_sd_and_odise_containers(tmp_path):
    sd = {"model.diffusion_model.time_embed.0.weight": torch.randn(8, 4), "first_stage_model.encoder.conv_in.weight": torch.randn(4, 3, 3, 3),
          "cond_stage_model.transformer.text_model.final_layer_norm.weight": torch.ones(4), "model_ema.decay": torch.tensor(0.9999),
          "betas": torch.zeros(10)}
    torch.save({"state_dict": sd, "global_step": 1}, tmp_path / "sd.ckpt")
    got = ck.load_sd_checkpoint(str(tmp_path / "sd.ckpt"))
    assert sorted(got) == sorted(k for k in sd if k.split(".")[0] in ("model", "first_stage_model", "cond_stage_model") and not k.startswith("model_ema"))
    np.testing.assert_array_equal(got["model.diffusion_model.time_embed.0.weight"], sd["model.diffusion_model.time_embed.0.weight"].numpy())
    torch.save({"state_dict": {"betas": torch.zeros(3)}}, tmp_path / "bad.ckpt")
    with pytest.raises(ValueError):
        ck.load_sd_checkpoint(str(tmp_path / "bad.ckpt"))
    od = {"backbone.feature_extractor.alpha_cond": torch.zeros(1, 77, 8), "criterion.empty_weight": torch.ones(3),
          "category_head.null_embed": torch.randn(1, 8).half()}
    torch.save({"model": od, "iteration": 7}, tmp_path / "odise.pth")
    got = ck.load_odise_checkpoint(str(tmp_path / "odise.pth"))
    assert sorted(got) == ["backbone.feature_extractor.alpha_cond", "category_head.null_embed"] and got["category_head.null_embed"].dtype == np.float32


class _Visual(nn.Module):
    def __init__(self):
        super().__init__()
        self.proj = nn.Parameter(torch.randn(6, 4))

    def forward(self, x):
        return x @ self.proj


class _Clip(nn.Module):
    def __init__(self):
        super().__init__()
        self.visual = _Visual()
        self.text_projection = nn.Parameter(torch.randn(4, 4))
        self.register_buffer("input_resolution", torch.tensor(336))

    def forward(self, x):
        return self.visual(x) @ self.text_projection


def test_openai_clip_torchscript_and_plain(tmp_path):
    m = _Clip().eval()
    torch.jit.script(m).save(str(tmp_path / "clip_jit.pt"))           # the format OpenAI publishes
    torch.save(m.state_dict(), tmp_path / "clip_plain.pt")
    for name in ("clip_jit.pt", "clip_plain.pt"):
        got = ck.load_openai_clip(str(tmp_path / name))
        assert sorted(got) == ["text_projection", "visual.proj"]
        np.testing.assert_array_equal(got["visual.proj"], m.visual.proj.detach().numpy())
    torch.save({"text_projection": torch.zeros(2, 2)}, tmp_path / "noclip.pt")
    with pytest.raises(ValueError):
        ck.load_openai_clip(str(tmp_path / "noclip.pt"))


def test_shared_noise_and_vocabulary_helpers():
    ref = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(42)).numpy()
    np.testing.assert_array_equal(ck.shared_noise(), ref)
    labels = [["person", "child"], ["sky"], ["traffic light"]]
    assert ck.prompt_labels(labels, None) == labels
    assert ck.prompt_labels(labels, "photo")[0] == ["a photo of a person.", "a photo of a child."]
    assert ck.prompt_labels(labels, "scene")[2] == ["a photo of a traffic light in the scene."] and ck.prompt_labels(labels, "a")[1] == ["a sky"]
    np.testing.assert_array_equal(ck.category_overlapping_mask([["person"], ["tree", "trees"]], labels + [["trees", "bush"]]), [1, 0, 0, 1])

    class _Enc:
        def build_text_embed(self, tok):
            return tok[:, :3].astype(np.float32)

    tok = lambda texts: np.array([[len(t), t.count("photo"), i] for i, t in enumerate(texts)], np.int64)
    cat, clp, sizes, ov = ck.build_vocabulary(labels, tok, _Enc(), train_labels=[["sky"]])
    assert sizes.tolist() == [2, 1, 1] and ov.tolist() == [0, 1, 0] and cat.shape == clp.shape == (4, 3)
    assert (cat[:, 1] == 0).all() and (clp[:, 1] == 1).all()            # category head: raw synonyms; clip head: "a photo of a {}."
