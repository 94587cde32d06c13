"""CPU: cross-check the CLIP ViT oracle (oracle/clip_vit.py) against the independent HF `transformers` implementation
(CLIPVisionModelWithProjection, hidden_act=quick_gelu) with the oracle's synthetic weights remapped.  This is the
secondary pin SURVEY.md §8c prescribes for the open_clip arithmetic that is absent from /root/reference."""
import numpy as np
import pytest
import torch

from oracle.clip_vit import CLIPVisual, clip_preprocess, embed_image, encode_tokens, init_synthetic_

transformers = pytest.importorskip("transformers")


def _to_hf(clip: CLIPVisual, cfg):
    from transformers import CLIPVisionModelWithProjection
    hf = CLIPVisionModelWithProjection(cfg).eval()
    v = clip.visual
    sd = {}
    p = "vision_model."
    sd[p + "embeddings.patch_embedding.weight"] = v.conv1.weight
    sd[p + "embeddings.class_embedding"] = v.class_embedding
    sd[p + "embeddings.position_embedding.weight"] = v.positional_embedding
    sd[p + "pre_layrnorm.weight"], sd[p + "pre_layrnorm.bias"] = v.ln_pre.weight, v.ln_pre.bias
    sd[p + "post_layernorm.weight"], sd[p + "post_layernorm.bias"] = v.ln_post.weight, v.ln_post.bias
    for i, r in enumerate(v.transformer.resblocks):
        q = f"{p}encoder.layers.{i}."
        w, b = r.attn.in_proj_weight, r.attn.in_proj_bias
        d = w.shape[1]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            sd[q + f"self_attn.{n}.weight"], sd[q + f"self_attn.{n}.bias"] = w[j * d:(j + 1) * d], b[j * d:(j + 1) * d]
        sd[q + "self_attn.out_proj.weight"], sd[q + "self_attn.out_proj.bias"] = r.attn.out_proj.weight, r.attn.out_proj.bias
        sd[q + "layer_norm1.weight"], sd[q + "layer_norm1.bias"] = r.ln_1.weight, r.ln_1.bias
        sd[q + "layer_norm2.weight"], sd[q + "layer_norm2.bias"] = r.ln_2.weight, r.ln_2.bias
        sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"] = r.mlp.c_fc.weight, r.mlp.c_fc.bias
        sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"] = r.mlp.c_proj.weight, r.mlp.c_proj.bias
    sd["visual_projection.weight"] = v.proj.t()
    missing, unexpected = hf.load_state_dict({k: t.detach().clone() for k, t in sd.items()}, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    return hf


def test_clip_oracle_matches_hf_transformers():
    from transformers import CLIPVisionConfig
    kw = dict(image_size=56, patch_size=14, width=64, layers=3, heads=4, output_dim=32)
    clip = init_synthetic_(CLIPVisual(**kw), seed=7).eval()
    cfg = CLIPVisionConfig(hidden_size=64, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4, image_size=56,
                           patch_size=14, hidden_act="quick_gelu", projection_dim=32, layer_norm_eps=1e-5)
    hf = _to_hf(clip, cfg)
    x = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ours = encode_tokens(clip, x)[:, 0]
        theirs = hf(pixel_values=x).image_embeds
    np.testing.assert_allclose(ours.numpy(), theirs.numpy(), rtol=1e-4, atol=1e-5)


def test_preprocess_is_bicubic_resize_center_crop_normalize():
    img = torch.rand(1, 3, 80, 120, generator=torch.Generator().manual_seed(2))
    out = clip_preprocess(img, 56)
    assert out.shape == (1, 3, 56, 56)
    # identity resize when the image already has the target size
    sq = torch.rand(1, 3, 56, 56, generator=torch.Generator().manual_seed(3))
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)
    np.testing.assert_allclose(clip_preprocess(sq, 56).numpy(), ((sq - mean) / std).numpy(), rtol=1e-6)


def test_embed_image_shape():
    clip = init_synthetic_(CLIPVisual(image_size=56, patch_size=14, width=64, layers=2, heads=4, output_dim=32)).eval()
    e = embed_image(clip, torch.rand(3, 3, 96, 96))
    assert e.shape == (3, 32) and e.dtype == torch.float32
