"""CPU: cross-check the CLIP text-tower oracle (oracle/clip_text.py) against the independent HF `transformers` implementation
(CLIPTextModelWithProjection, hidden_act=quick_gelu) - the secondary pin SURVEY.md 8c prescribes for open_clip arithmetic that is
absent from /root/reference - and check the HF -> OpenAI key conversion used for the SD cond-stage weights."""
import numpy as np
import pytest
import torch

from oracle.clip_text import CLIPText, EOT, SOT, empty_prompt_tokens, encode_hidden, encode_text, hf_to_openai, init_synthetic_

transformers = pytest.importorskip("transformers")

KW = dict(vocab_size=49408, context_length=77, width=64, layers=3, heads=4, output_dim=32)


def _hf_model():
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    cfg = CLIPTextConfig(vocab_size=49408, hidden_size=64, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4,
                         max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=32, layer_norm_eps=1e-5,
                         pad_token_id=1, bos_token_id=SOT, eos_token_id=EOT)
    torch.manual_seed(0)
    return CLIPTextModelWithProjection(cfg).eval()


def _tokens():
    g = torch.Generator().manual_seed(5)
    t = torch.zeros(4, 77, dtype=torch.long)
    for i, n in enumerate((0, 3, 10, 75)):
        t[i, 0] = SOT
        t[i, 1:1 + n] = torch.randint(1000, 40000, (n,), generator=g)
        t[i, 1 + n] = EOT
    return t


def test_text_oracle_matches_hf_transformers():
    hf = _hf_model()
    m = CLIPText(**KW).eval()
    sd = hf_to_openai({k: v for k, v in hf.state_dict().items()}, prefix="text_model.")
    sd["text_projection"] = hf.text_projection.weight.t()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    tok = _tokens()
    with torch.no_grad():
        out = hf(input_ids=tok)
        hid = encode_hidden(m, tok)
        emb = encode_text(m, tok)
    np.testing.assert_allclose(hid.numpy(), out.last_hidden_state.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(emb.numpy(), out.text_embeds.numpy(), rtol=1e-4, atol=2e-5)


def test_empty_prompt_tokens_and_causality():
    t = empty_prompt_tokens()
    assert t.shape == (1, 77) and t[0, 0] == SOT and (t[0, 1:] == EOT).all()
    t0 = empty_prompt_tokens(pad_with_eot=False)
    assert t0[0, 1] == EOT and (t0[0, 2:] == 0).all()
    # causal: hidden states up to the first EOT do not depend on the padding convention
    m = init_synthetic_(CLIPText(**KW)).eval()
    with torch.no_grad():
        a, b = encode_hidden(m, t), encode_hidden(m, t0)
    np.testing.assert_allclose(a[:, :2].numpy(), b[:, :2].numpy(), rtol=1e-5, atol=1e-6)
    assert not np.allclose(a[:, 2:].numpy(), b[:, 2:].numpy())
