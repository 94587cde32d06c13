"""Oracle: CLIP ViT image tower as ODISE drives it (TEST INFRASTRUCTURE — see oracle/__init__.py).

PARITY UNPINNED against open_clip (open-clip-torch==2.0.2, reference setup.py:81, absent here); cross-checked against the
independent HF `transformers.CLIPVisionModel` implementation in tests/test_oracle_clip.py.  Restates the OpenAI CLIP
VisionTransformer (open_clip/model.py `VisualTransformer`, `ResidualAttentionBlock`, `QuickGELU`) with parameter names
identical to the OpenAI checkpoint keys `visual.*`, and follows ODISE's call sites:
  ClipAdapter._encode_image / embed_image      odise/modeling/meta_arch/clip.py:177-231
  clip_preprocess = Resize(bicubic) + CenterCrop + Normalize     clip.py:94  (torchvision 0.14.1 tensor path, antialias off)
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class _MLP(nn.Module):
    def __init__(self, width, hidden):
        super().__init__()
        self.c_fc = nn.Linear(width, hidden)
        self.gelu = QuickGELU()
        self.c_proj = nn.Linear(hidden, width)

    def forward(self, x):
        return self.c_proj(self.gelu(self.c_fc(x)))


class ResidualAttentionBlock(nn.Module):
    def __init__(self, width, heads, mlp_ratio=4.0):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = _MLP(width, int(width * mlp_ratio))

    def attention(self, x, attn_mask=None):  # x: [L, N, D]
        return self.attn(x, x, x, need_weights=False, attn_mask=attn_mask)[0]

    def forward(self, x, attn_mask=None):
        x = x + self.attention(self.ln_1(x), attn_mask=attn_mask)
        x = x + self.mlp(self.ln_2(x))
        return x


class Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward(self, x, attn_mask=None):
        for r in self.resblocks:
            x = r(x, attn_mask=attn_mask)
        return x


class VisualTransformer(nn.Module):
    """ViT-L/14@336: image 336, patch 14, width 1024, 24 layers, 16 heads, output 768.  `small=True` builds a narrow
    stand-in (width 128, 2 layers, 2 heads... same graph) for fast CPU tests."""

    def __init__(self, image_size=336, patch_size=14, width=1024, layers=24, heads=16, output_dim=768):
        super().__init__()
        self.image_size, self.patch_size = image_size, patch_size
        self.conv1 = nn.Conv2d(3, width, patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((image_size // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


class CLIPVisual(nn.Module):
    """Holds the tower as `.visual` so that state_dict keys read `visual.*` like the OpenAI archive."""

    def __init__(self, **kw):
        super().__init__()
        self.visual = VisualTransformer(**kw)


def init_synthetic_(model: nn.Module, seed: int = 4321) -> nn.Module:
    g = torch.Generator().manual_seed(seed)
    for name, p in sorted(model.named_parameters()):
        with torch.no_grad():
            if name.endswith("in_proj_weight") or (p.ndim >= 2 and "positional" not in name and not name.endswith(".proj")):
                gain = 0.5 if any(s in name for s in ("out_proj", "c_proj")) else 1.0
                p.copy_(torch.randn(p.shape, generator=g) * (gain / math.sqrt(p[0].numel())))
            elif name.endswith(".proj"):
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / math.sqrt(p.shape[0])))
            elif "positional" in name or "class_embedding" in name:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            elif "ln_" in name:
                p.copy_((1.0 if name.endswith("weight") else 0.0) + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return model


def clip_preprocess(image: torch.Tensor, size: int) -> torch.Tensor:
    """clip.py:94: T.Resize(size, BICUBIC) + CenterCrop(size) + Normalize on a float tensor [B,3,H,W] in [0,1].
    torchvision 0.14.1 resizes tensors with F.interpolate(mode='bicubic', align_corners=False) (no antialias) so that the
    SHORT side becomes `size`, then crops the centre."""
    _, _, h, w = image.shape
    if h <= w:
        nh, nw = size, int(size * w / h)
    else:
        nh, nw = int(size * h / w), size
    x = F.interpolate(image, size=(nh, nw), mode="bicubic", align_corners=False) if (nh, nw) != (h, w) else image
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    x = x[:, :, top:top + size, left:left + size]
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return (x - mean) / std


@torch.no_grad()
def encode_tokens(clip: CLIPVisual, image: torch.Tensor, attn_mask=None, extra_tokens=None) -> torch.Tensor:
    """ClipAdapter._encode_image up to ln_post+proj (clip.py:177-206): returns [B, 1+grid^2, output_dim]."""
    v = clip.visual
    x = v.conv1(image)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    x = torch.cat([v.class_embedding + torch.zeros(x.shape[0], 1, x.shape[-1]), x], dim=1)
    x = x + v.positional_embedding
    x = v.ln_pre(x)
    x = x.permute(1, 0, 2)
    x = v.transformer(x)
    x = x.permute(1, 0, 2)
    x = v.ln_post(x)
    return x @ v.proj


@torch.no_grad()
def embed_image(clip: CLIPVisual, image: torch.Tensor) -> torch.Tensor:
    """ClipAdapter.embed_image(...).image_embed with normalize=False (clip.py:225-231; ldm.py:705): [B, output_dim]."""
    x = encode_tokens(clip, clip_preprocess(image, clip.visual.image_size))
    return x[:, 0, :].float()
