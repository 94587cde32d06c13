"""GPU parity of the device CLIP text tower (odise_amd/text.py, composed from the C-ABI GEMM / LayerNorm / masked-attention operators)
against the CPU oracle (oracle/clip_text.py <- clip.py:29-73, 148-162; HF-cross-checked in tests/test_oracle_clip_text.py).
Tolerance: 3e-3 of max|ref| on the final hidden states and on the projected embeddings (fp16 activations, fp32 accumulate)."""
import numpy as np
import pytest
import torch

from odise_amd.text import HipTextEncoder, empty_prompt_tokens, hf_text_to_openai
from oracle.clip_text import CLIPText, EOT, SOT, encode_hidden, encode_text, init_synthetic_

pytestmark = pytest.mark.gpu


def _tokens(n_rows=5, L=77, seed=3):
    g = torch.Generator().manual_seed(seed)
    t = torch.zeros(n_rows, L, dtype=torch.long)
    for i, n in enumerate([0, 2, 9, 40, L - 2][:n_rows]):
        t[i, 0] = SOT
        t[i, 1:1 + n] = torch.randint(1000, 40000, (n,), generator=g)
        t[i, 1 + n] = EOT
    return t


@pytest.mark.parametrize("width,layers,heads", [(128, 2, 2), (256, 3, 4)])
def test_text_tower_matches_oracle(ctx, width, layers, heads):
    m = init_synthetic_(CLIPText(width=width, layers=layers, heads=heads, output_dim=96), seed=width).eval()
    tok = _tokens()
    with torch.no_grad():
        hid_ref = encode_hidden(m, tok).numpy()
        emb_ref = encode_text(m, tok).numpy()
    enc = HipTextEncoder(ctx, {k: v for k, v in m.state_dict().items()}, heads=heads)
    hid = enc.hidden(tok.numpy())
    emb = enc.build_text_embed(tok.numpy(), batch=2)
    e1 = np.abs(hid - hid_ref).max() / np.abs(hid_ref).max()
    e2 = np.abs(emb - emb_ref).max() / np.abs(emb_ref).max()
    print("hidden err", e1, "embed err", e2)
    assert hid.shape == hid_ref.shape and e1 < 3e-3
    assert emb.shape == emb_ref.shape and e2 < 3e-3


def test_uncond_inputs_from_hf_named_weights(ctx):
    """`ldm.embed_text([""])` (ldm.py:116): HF-named cond-stage weights, EOT-padded empty prompt, last_hidden_state."""
    m = init_synthetic_(CLIPText(width=128, layers=2, heads=2, output_dim=128), seed=5).eval()
    sd = m.state_dict()
    hf = {}
    p = "cond_stage_model.transformer.text_model."
    hf[p + "embeddings.token_embedding.weight"] = sd["token_embedding.weight"]
    hf[p + "embeddings.position_embedding.weight"] = sd["positional_embedding"]
    hf[p + "final_layer_norm.weight"], hf[p + "final_layer_norm.bias"] = sd["ln_final.weight"], sd["ln_final.bias"]
    for i in range(2):
        r, q = f"transformer.resblocks.{i}.", p + f"encoder.layers.{i}."
        w, b = sd[r + "attn.in_proj_weight"], sd[r + "attn.in_proj_bias"]
        for j, n in enumerate("qkv"):
            hf[q + f"self_attn.{n}_proj.weight"], hf[q + f"self_attn.{n}_proj.bias"] = w[j * 128:(j + 1) * 128], b[j * 128:(j + 1) * 128]
        for a, c in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"),
                     ("mlp.c_proj", "mlp.fc2")):
            hf[q + c + ".weight"], hf[q + c + ".bias"] = sd[r + a + ".weight"], sd[r + a + ".bias"]
    enc = HipTextEncoder(ctx, hf_text_to_openai(hf), heads=2)
    tok = empty_prompt_tokens()
    with torch.no_grad():
        ref = encode_hidden(m, torch.from_numpy(tok)).numpy()
    got = enc.hidden(tok)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print("uncond_inputs err", err)
    assert got.shape == (1, 77, 128) and err < 3e-3


def test_reference_generated_text_fixture_on_device(ctx):
    """tests/golden/text_encode.npz was written by the REFERENCE's `ClipAdapter._encode_text` (clip.py:148-162) walking the oracle's seeded text
    tower (tests/golden/make_golden_text.py); the device tower with the same weights must reproduce it."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text_encode.npz"))
    m = init_synthetic_(CLIPText(vocab_size=49408, context_length=77, width=64, layers=2, heads=2, output_dim=32), seed=7).eval()
    enc = HipTextEncoder(ctx, {k: v for k, v in m.state_dict().items()}, heads=2)
    emb, hid = enc.build_text_embed(z["tokens"]), enc.hidden(z["tokens"])
    e1 = np.abs(hid - z["hidden"].astype(np.float32)).max() / np.abs(z["hidden"]).max()
    e2 = np.abs(emb - z["embed"]).max() / np.abs(z["embed"]).max()
    print("reference text fixture on the device: hidden err", e1, "embed err", e2)
    assert e1 < 4e-3 and e2 < 4e-3
