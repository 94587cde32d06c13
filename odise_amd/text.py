"""CLIP text tower on the device, composed from the library's C-ABI operators (GEMM, LayerNorm, masked attention).

Mirrors `ClipAdapter._encode_text` / `build_clip_text_embed` (odise/modeling/meta_arch/clip.py:29-73, 148-162) for the OpenAI ViT-L/14
text encoder (12 layers, width 768, 12 heads, 77 tokens, causal mask, EOT pooling, `text_projection`) and `FrozenCLIPEmbedder`
(`ldm.embed_text([""])`, ldm.py:116: the same tower with HF key names, `last_hidden_state` after the final LayerNorm), so that
vocabularies and the constant `uncond_inputs` are built without any PyTorch model on the steady path.

Weights are addressed by the OpenAI checkpoint keys (`token_embedding.weight`, `positional_embedding`,
`transformer.resblocks.N.{ln_1,attn.in_proj_*,attn.out_proj,ln_2,mlp.c_fc,mlp.c_proj}`, `ln_final`, `text_projection`).
Sequences are padded from 77 to 80 rows so every matrix keeps 16-byte rows; the padding keys are masked for every query.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

from ._lib import ACT_QUICKGELU
from .runtime import Context, DeviceArray

SOT, EOT = 49406, 49407


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def hf_text_to_openai(state: Dict[str, "np.ndarray"], prefix: str = "cond_stage_model.transformer.text_model.") -> Dict[str, np.ndarray]:
    """HF `CLIPTextModel` keys (as stored in an SD-v1 checkpoint) -> OpenAI names; q/k/v projections are stacked."""
    g = lambda k: _np(state[prefix + k])
    out = {"token_embedding.weight": g("embeddings.token_embedding.weight"), "positional_embedding": g("embeddings.position_embedding.weight"),
           "ln_final.weight": g("final_layer_norm.weight"), "ln_final.bias": g("final_layer_norm.bias")}
    i = 0
    while prefix + f"encoder.layers.{i}.layer_norm1.weight" in state:
        q, r = f"encoder.layers.{i}.", f"transformer.resblocks.{i}."
        out[r + "attn.in_proj_weight"] = np.concatenate([g(q + f"self_attn.{n}_proj.weight") for n in "qkv"], 0)
        out[r + "attn.in_proj_bias"] = np.concatenate([g(q + f"self_attn.{n}_proj.bias") for n in "qkv"], 0)
        for a, b in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"),
                     ("mlp.c_proj", "mlp.fc2")):
            out[r + a + ".weight"], out[r + a + ".bias"] = g(q + b + ".weight"), g(q + b + ".bias")
        i += 1
    return out


class HipTextEncoder:
    def __init__(self, ctx: Context, state: Dict[str, "np.ndarray"], heads: Optional[int] = None, prefix: str = ""):
        self.ctx = ctx
        g = lambda k: _np(state[prefix + k]).astype(np.float32)
        self.tok = g("token_embedding.weight")                     # host: rows are gathered per call
        self.pos = g("positional_embedding")
        self.L, self.W = self.pos.shape
        self.heads = heads or max(1, self.W // 64)
        assert self.W % 8 == 0 and (self.W // self.heads) % 8 == 0, "width / head dim must be multiples of 8"
        self.Lp = (self.L + 7) // 8 * 8
        f16 = lambda a: ctx.to_device(np.ascontiguousarray(a, np.float16))
        f32 = lambda a: ctx.to_device(np.ascontiguousarray(a, np.float32))
        self.layers = []
        i = 0
        while prefix + f"transformer.resblocks.{i}.ln_1.weight" in state:
            r = f"transformer.resblocks.{i}."
            w, b, W = g(r + "attn.in_proj_weight"), g(r + "attn.in_proj_bias"), self.W
            self.layers.append(dict(
                ln1=(f32(g(r + "ln_1.weight")), f32(g(r + "ln_1.bias"))), ln2=(f32(g(r + "ln_2.weight")), f32(g(r + "ln_2.bias"))),
                wq=f16(w[:W]), wk=f16(w[W:2 * W]), wv=f16(w[2 * W:]), bq=f32(b[:W]), bk=f32(b[W:2 * W]), bv=f32(b[2 * W:]),
                wo=f16(g(r + "attn.out_proj.weight")), bo=f32(g(r + "attn.out_proj.bias")),
                w1=f16(g(r + "mlp.c_fc.weight")), b1=f32(g(r + "mlp.c_fc.bias")), w2=f16(g(r + "mlp.c_proj.weight")), b2=f32(g(r + "mlp.c_proj.bias"))))
            i += 1
        assert self.layers, "no transformer.resblocks.* keys found"
        self.lnf = (f32(g("ln_final.weight")), f32(g("ln_final.bias")))
        self.proj = None
        if prefix + "text_projection" in state:
            p = g("text_projection")                               # [width, out]
            self.out_dim = p.shape[1]
            self.proj = f16(p.T)                                   # GEMM weight layout [out, width]

    # ---- forward ------------------------------------------------------------------------------------------------------
    def hidden(self, tokens: np.ndarray) -> np.ndarray:
        """tokens [N, L'] (L' <= context length) -> ln_final(transformer(tok_emb + pos)) [N, L', width] fp32   (clip.py:149-155)"""
        tokens = np.asarray(tokens, np.int64)
        N, L = tokens.shape
        assert L <= self.L
        ctx, W, Lp, H = self.ctx, self.W, self.Lp, self.heads
        x0 = np.zeros((N, Lp, W), np.float16)
        x0[:, :L] = (self.tok[tokens] + self.pos[:L]).astype(np.float16)
        x = ctx.to_device(x0)
        m = np.ones((Lp, Lp), np.uint8)                            # 1 = not visible: future tokens (causal) and the padding rows
        m[np.tril_indices(Lp)] = 0
        m[:, L:] = 1
        m[L:, 0] = 0                                               # padding queries keep one visible key (their rows are discarded)
        mask = ctx.to_device(np.ascontiguousarray(np.broadcast_to(m, (N, Lp, Lp))))
        scale = (W // H) ** -0.5
        flat = lambda a: a.view((N * Lp, W))
        for ly in self.layers:
            h = ctx.layer_norm(x, *ly["ln1"])
            q = ctx.gemm(flat(h), ly["wq"], bias_n=ly["bq"]).view((N, Lp, W))
            k = ctx.gemm(flat(h), ly["wk"], bias_n=ly["bk"]).view((N, Lp, W))
            vt = ctx.gemm(ly["wv"], h, bias_m=ly["bv"])            # V^T [N, W, Lp]: the swapped product hands attention its operand
            a = ctx.attention(q, k, vt, H, scale, mask=mask, Lk=Lp)
            x = ctx.gemm(flat(a), ly["wo"], bias_n=ly["bo"], residual=flat(x)).view((N, Lp, W))
            h = ctx.layer_norm(x, *ly["ln2"])
            u = ctx.gemm(flat(h), ly["w1"], bias_n=ly["b1"], act=ACT_QUICKGELU)
            x = ctx.gemm(u, ly["w2"], bias_n=ly["b2"], residual=flat(x)).view((N, Lp, W))
        y = ctx.layer_norm(x, *self.lnf).numpy().astype(np.float32)
        return y[:, :L]

    def encode(self, tokens: np.ndarray) -> np.ndarray:
        """EOT pooling (the EOT id is the largest id of every row) + text_projection (clip.py:158-160): [N, out] fp32, not normalised."""
        assert self.proj is not None, "this tower has no text_projection (SD cond-stage weights)"
        tokens = np.asarray(tokens, np.int64)
        hid = self.hidden(tokens)
        pooled = hid[np.arange(len(tokens)), tokens.argmax(-1)]
        n8 = (len(tokens) + 7) // 8 * 8
        p16 = np.zeros((n8, self.W), np.float16)
        p16[:len(tokens)] = pooled
        out = self.ctx.gemm(self.ctx.to_device(p16), self.proj, out_dtype=np.float32).numpy()
        return out[:len(tokens)]

    def build_text_embed(self, token_rows: np.ndarray, batch: int = 256) -> np.ndarray:
        """build_clip_text_embed (clip.py:29-73): one embedding per prompt string, in batches of 256."""
        outs = [self.encode(token_rows[i:i + batch]) for i in range(0, len(token_rows), batch)]
        return np.concatenate(outs, 0)


def empty_prompt_tokens(context_length: int = 77, pad_with_eot: bool = True) -> np.ndarray:
    """Token ids of "": <SOT><EOT> + padding (HF CLIPTokenizer, the SD cond stage, pads with EOT; open_clip.tokenize with 0)."""
    t = np.full((1, context_length), EOT if pad_with_eot else 0, np.int64)
    t[0, 0], t[0, 1] = SOT, EOT
    return t
