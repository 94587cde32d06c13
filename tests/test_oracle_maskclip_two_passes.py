"""The algebra behind the library's two-pass MaskCLIP (csrc/engine.h ClipKV, extractor.cpp maskclip_image_pass / maskclip_mask_pass), on the
CPU with the oracle's restatement of the reference (oracle/odise_model.py mask_clip_embed = clip.py:252-338).

The reference runs ONE tower over [Q mask tokens | class token | patches] with an attention mask whose first Q columns are all True
(clip.py:314-315: "mask+cls+image token to mask token attention is masked out"): no query ever reads a mask token.  So
  * the class + patch tokens evolve exactly as in the plain image tower, whatever the masks are, and
  * a mask token of block l reads the keys / values of that block's class + patch tokens only (its own visibility row, clip.py:318), never
    another mask token or itself.
Restated here as two passes in fp64 and compared with the one-pass oracle in fp64: equal to rounding.  That is what lets the library compute the
pictures' image tokens ahead of the mask head (inside the crops' CLIP tower) and leave only B x Q token rows behind it."""
import torch
import torch.nn.functional as F

from oracle import clip_vit
from oracle import odise_model as om


def _two_passes(clip, image, mask):
    v = clip.visual
    size = (v.image_size, v.image_size)
    image = F.interpolate(image, size=size, mode="bilinear", align_corners=False)
    mask = F.interpolate(mask, size=size, mode="bilinear", align_corners=False)
    image = clip_vit.clip_preprocess(image, v.image_size)
    B, Q = mask.shape[:2]
    hidden = (F.max_pool2d(mask.sigmoid(), kernel_size=v.patch_size, stride=v.patch_size) < 0.5).reshape(B, Q, -1)   # [B, Q, patches]
    heads = v.conv1.out_channels // 64
    # ---- pass 1: class + patch tokens through the plain tower; per block the keys / values every mask token will read
    x = v.conv1(image)
    x = x.reshape(B, x.shape[1], -1).permute(0, 2, 1)
    x = torch.cat([v.class_embedding + torch.zeros(B, 1, x.shape[-1], dtype=x.dtype), x], dim=1) + v.positional_embedding
    x = v.ln_pre(x)                                     # [B, T, D]
    T, D = x.shape[1:]
    dh = D // heads
    start = x[:, 0:1].expand(B, Q, D)                   # every mask token starts as the class token after ln_pre (clip.py:268-270)
    kv = []
    for blk in v.transformer.resblocks:
        n = blk.ln_1(x)
        qkv = F.linear(n, blk.attn.in_proj_weight, blk.attn.in_proj_bias)
        q, k, val = qkv.split(D, dim=-1)
        kv.append((k, val))
        qh, kh, vh = (t.reshape(B, T, heads, dh).transpose(1, 2) for t in (q, k, val))
        a = torch.softmax(qh @ kh.transpose(-1, -2) / dh ** 0.5, dim=-1) @ vh
        x = x + blk.attn.out_proj(a.transpose(1, 2).reshape(B, T, D))
        x = x + blk.mlp(blk.ln_2(x))
    # ---- pass 2: the mask tokens, B x Q rows, over pass 1's keys / values (column 0 = the class token: always visible)
    vis = torch.cat([torch.zeros(B, Q, 1, dtype=torch.bool), hidden], dim=-1)   # True = hidden
    m = start
    for blk, (k, val) in zip(v.transformer.resblocks, kv):
        n = blk.ln_1(m)
        q = F.linear(n, blk.attn.in_proj_weight[:D], blk.attn.in_proj_bias[:D])
        qh = q.reshape(B, Q, heads, dh).transpose(1, 2)
        kh, vh = (t.reshape(B, T, heads, dh).transpose(1, 2) for t in (k, val))
        s = qh @ kh.transpose(-1, -2) / dh ** 0.5
        s = s.masked_fill(vis[:, None], float("-inf"))
        a = torch.softmax(s, dim=-1) @ vh
        m = m + blk.attn.out_proj(a.transpose(1, 2).reshape(B, Q, D))
        m = m + blk.mlp(blk.ln_2(m))
    return torch.einsum("nld,dc->nlc", v.ln_post(m), v.proj)


def test_two_passes_equal_the_reference_layout():
    torch.manual_seed(0)
    clip = clip_vit.init_synthetic_(clip_vit.CLIPVisual(image_size=56, patch_size=14, width=128, layers=3, heads=2, output_dim=32)).double().eval()
    g = torch.Generator().manual_seed(3)
    image = torch.rand(2, 3, 80, 72, generator=g, dtype=torch.float64)
    coarse = torch.randn(2, 6, 5, 4, generator=g, dtype=torch.float64) * 3.0
    mask = F.interpolate(coarse, size=(20, 18), mode="bicubic", align_corners=False)
    mask[0, 0] = -5.0            # a mask that hides every patch: its token sees the class token only
    mask[1, 1] = 5.0             # ... and one that hides none
    with torch.no_grad():
        ref = om.mask_clip_embed(clip, image, mask)
        two = _two_passes(clip, image, mask)
    assert ref.shape == two.shape == (2, 6, 32)
    err = (ref - two).abs().max().item() / ref.abs().max().item()
    assert err < 1e-12, err
    # the visibility rows mattered (another mask gives another embedding), so the agreement above is not vacuous
    with torch.no_grad():
        other = om.mask_clip_embed(clip, image, -mask)
    assert (other - ref).abs().max().item() > 1e-3 * ref.abs().max().item()
