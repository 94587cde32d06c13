"""Oracle: CLIP text tower as ODISE drives it (TEST INFRASTRUCTURE — see oracle/__init__.py).

PARITY UNPINNED against open_clip (open-clip-torch==2.0.2, reference setup.py:81, absent here); cross-checked against the
independent HF `transformers.CLIPTextModelWithProjection` in tests/test_oracle_clip_text.py.  Restates the OpenAI CLIP text encoder
(open_clip/model.py `CLIP.encode_text`) with parameter names identical to the OpenAI checkpoint keys (`token_embedding.weight`,
`positional_embedding`, `transformer.resblocks.*`, `ln_final.*`, `text_projection`) and follows ODISE's call sites:
  ClipAdapter._encode_text            odise/modeling/meta_arch/clip.py:148-162   (causal mask, ln_final, EOT pooling @ text_projection)
  build_clip_text_embed               clip.py:29-73                              (one embedding per prompt string, batches of 256)
  get_openseg_labels / prompt ensembling      odise/data/build.py:54-71, odise.py:1273-1288 (mean over a category's strings is taken by the caller)
  FrozenCLIPEmbedder (SD-v1 cond stage, `ldm.embed_text([""])`, ldm.py:116)   = the same tower, `last_hidden_state` after the final
  LayerNorm, HF weight names (`hf_to_openai` converts them).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .clip_vit import Transformer

SOT, EOT = 49406, 49407


class CLIPText(nn.Module):
    def __init__(self, vocab_size=49408, context_length=77, width=768, layers=12, heads=12, output_dim=768):
        super().__init__()
        self.context_length = context_length
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width))
        self.transformer = Transformer(width, layers, heads)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, output_dim))

    def attn_mask(self, L=None):
        L = L or self.context_length
        return torch.full((L, L), float("-inf")).triu_(1)  # causal (open_clip CLIP.build_attention_mask)


def init_synthetic_(model: CLIPText, seed: int = 99) -> CLIPText:
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name == "token_embedding.weight":
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
            elif name == "positional_embedding":
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            elif name == "text_projection":
                p.copy_(torch.randn(p.shape, generator=g) * p.shape[0] ** -0.5)
            elif p.ndim == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05 + (1.0 if (("ln_" in name) and name.endswith("weight")) else 0.0))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * p.shape[-1] ** -0.5)
    return model


def encode_hidden(m: CLIPText, tokens: torch.Tensor) -> torch.Tensor:
    """tokens [N, L] int64 -> ln_final(transformer(...)) [N, L, width]   (clip.py:149-155)"""
    L = tokens.shape[1]
    x = m.token_embedding(tokens) + m.positional_embedding[:L]
    x = m.transformer(x.permute(1, 0, 2), attn_mask=m.attn_mask(L)).permute(1, 0, 2)
    return m.ln_final(x)


def encode_text(m: CLIPText, tokens: torch.Tensor) -> torch.Tensor:
    """EOT pooling (the EOT id is the largest token id of every row) and projection (clip.py:158-160): [N, output_dim]"""
    x = encode_hidden(m, tokens)
    return x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)] @ m.text_projection


def empty_prompt_tokens(context_length: int = 77, pad_with_eot: bool = True) -> torch.Tensor:
    """Token ids of "" : <SOT><EOT> then padding - HF's CLIPTokenizer (SD cond stage) pads with EOT, open_clip.tokenize with 0."""
    t = torch.full((1, context_length), EOT if pad_with_eot else 0, dtype=torch.long)
    t[0, 0], t[0, 1] = SOT, EOT
    return t


def hf_to_openai(hf_state: dict, prefix: str = "text_model.") -> dict:
    """HF CLIPTextModel keys (`cond_stage_model.transformer.text_model.*` in an SD checkpoint, prefix stripped by the caller) ->
    OpenAI names.  q/k/v projections are stacked into `attn.in_proj_*`."""
    out = {}
    g = lambda k: hf_state[prefix + k]
    out["token_embedding.weight"] = g("embeddings.token_embedding.weight")
    out["positional_embedding"] = g("embeddings.position_embedding.weight")
    out["ln_final.weight"], out["ln_final.bias"] = g("final_layer_norm.weight"), g("final_layer_norm.bias")
    i = 0
    while prefix + f"encoder.layers.{i}.layer_norm1.weight" in hf_state:
        q, r = f"encoder.layers.{i}.", f"transformer.resblocks.{i}."
        out[r + "attn.in_proj_weight"] = torch.cat([g(q + f"self_attn.{n}_proj.weight") for n in "qkv"], 0)
        out[r + "attn.in_proj_bias"] = torch.cat([g(q + f"self_attn.{n}_proj.bias") for n in "qkv"], 0)
        out[r + "attn.out_proj.weight"], out[r + "attn.out_proj.bias"] = g(q + "self_attn.out_proj.weight"), g(q + "self_attn.out_proj.bias")
        out[r + "ln_1.weight"], out[r + "ln_1.bias"] = g(q + "layer_norm1.weight"), g(q + "layer_norm1.bias")
        out[r + "ln_2.weight"], out[r + "ln_2.bias"] = g(q + "layer_norm2.weight"), g(q + "layer_norm2.bias")
        out[r + "mlp.c_fc.weight"], out[r + "mlp.c_fc.bias"] = g(q + "mlp.fc1.weight"), g(q + "mlp.fc1.bias")
        out[r + "mlp.c_proj.weight"], out[r + "mlp.c_proj.bias"] = g(q + "mlp.fc2.weight"), g(q + "mlp.fc2.bias")
        i += 1
    return out
