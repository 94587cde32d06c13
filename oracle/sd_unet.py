"""Oracle: Stable-Diffusion v1 UNet and ODISE's single-step tap extraction (TEST INFRASTRUCTURE — see oracle/__init__.py).

PARITY UNPINNED: the UNet arithmetic lives in `ldm` (stable-diffusion-sdkit==2.1.3, reference setup.py:83), which is
neither under /root/reference nor installable here.  This file restates the published architecture
(ldm/modules/diffusionmodules/openaimodel.py `UNetModel`, `ResBlock`, `Downsample`, `Upsample`, `timestep_embedding`;
ldm/modules/attention.py `SpatialTransformer`, `BasicTransformerBlock`, `CrossAttention`, `FeedForward/GEGLU`; config
v1-inference.yaml: in 4, model_channels 320, channel_mult [1,2,4,4], num_res_blocks 2, attention_resolutions [4,2,1],
num_heads 8, context_dim 768, legacy False) in plain fp32 torch, with module/parameter names identical to the
checkpoint keys `model.diffusion_model.*`, and anchors on the reference's call site
LdmExtractor.unet_forward (odise/modeling/meta_arch/ldm.py:469-491), which `unet_forward` below follows line by line.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """ldm openaimodel/util.timestep_embedding (repeat_only=False): cat[cos(t f), sin(t f)]."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


def normalization(ch):
    return GroupNorm32(32, ch)


class ResBlock(nn.Module):
    def __init__(self, channels, emb_channels, out_channels):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(), nn.Conv2d(channels, out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, out_channels))
        self.out_layers = nn.Sequential(normalization(out_channels), nn.SiLU(), nn.Dropout(p=0.0),
                                        nn.Conv2d(out_channels, out_channels, 3, padding=1))
        self.skip_connection = nn.Identity() if out_channels == channels else nn.Conv2d(channels, out_channels, 1)

    def forward(self, x, emb, context=None):
        h = self.in_layers(x)
        emb_out = self.emb_layers(emb).type(h.dtype)
        h = h + emb_out[..., None, None]
        h = self.out_layers(h)
        return self.skip_connection(x) + h


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = dim_head * heads
        context_dim = context_dim if context_dim is not None else query_dim
        self.scale, self.heads = dim_head ** -0.5, heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))

    def forward(self, x, context=None):
        h = self.heads
        q = self.to_q(x)
        context = context if context is not None else x
        k, v = self.to_k(context), self.to_v(context)
        b, n, _ = q.shape
        q, k, v = (t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3) for t in (q, k, v))
        sim = torch.einsum("bhid,bhjd->bhij", q, k) * self.scale
        attn = sim.softmax(dim=-1)
        out = torch.einsum("bhij,bhjd->bhid", attn, v)
        out = out.permute(0, 2, 1, 3).reshape(b, n, -1)
        return self.to_out(out)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim))

    def forward(self, x):
        return self.net(x)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, context_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, n_heads, d_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)

    def forward(self, x, context=None):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context=context) + x
        x = self.ff(self.norm3(x)) + x
        return x


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, context_dim):
        super().__init__()
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, n_heads, d_head, context_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, x, emb=None, context=None):
        b, c, h, w = x.shape
        x_in = x
        x = self.proj_in(self.norm(x))
        x = x.reshape(b, c, h * w).permute(0, 2, 1)
        for blk in self.transformer_blocks:
            x = blk(x, context=context)
        x = x.permute(0, 2, 1).reshape(b, c, h, w)
        return self.proj_out(x) + x_in


class Downsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.op = nn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x, emb=None, context=None):
        return self.op(x)


class Upsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x, emb=None, context=None):
        return self.conv(F.interpolate(x, scale_factor=2, mode="nearest"))


class ConvIn(nn.Conv2d):
    def forward(self, x, emb=None, context=None):
        return super().forward(x)


class TimestepEmbedSequential(nn.Sequential):
    def forward(self, x, emb, context=None):
        for layer in self:
            x = layer(x, emb, context)
        return x


class UNetModel(nn.Module):
    """SD v1 UNet; `width_div` shrinks every channel count for fast CPU tests (1 = the real 859.5 M-parameter model)."""

    def __init__(self, in_channels=4, model_channels=320, out_channels=4, num_res_blocks=2, attention_resolutions=(4, 2, 1),
                 channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768, width_div=1):
        super().__init__()
        model_channels //= width_div
        self.model_channels = model_channels
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(ConvIn(in_channels, model_channels, 3, padding=1))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(SpatialTransformer(ch, num_heads, ch // num_heads, context_dim))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(ResBlock(ch, ted, ch), SpatialTransformer(ch, num_heads, ch // num_heads, context_dim),
                                                    ResBlock(ch, ted, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(SpatialTransformer(ch, num_heads, ch // num_heads, context_dim))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))


def init_synthetic_(model: nn.Module, seed: int = 1234) -> nn.Module:
    """Deterministic synthetic weights keyed like the real checkpoint: fan-in scaled so that
    activations stay O(1) through the residual stack; layers that are zero-initialised in the public code (out_layers.3,
    proj_out, `out`) get non-zero values so they are exercised."""
    g = torch.Generator().manual_seed(seed)
    for name, p in sorted(model.named_parameters()):
        with torch.no_grad():
            if p.ndim >= 2:
                fan_in = p[0].numel()
                gain = 0.5 if any(s in name for s in ("out_layers.3", "proj_out", "to_out.0", "ff.net.2")) else 1.0
                p.copy_(torch.randn(p.shape, generator=g) * (gain / math.sqrt(fan_in)))
            elif "norm" in name or ".in_layers.0." in name or ".out_layers.0." in name or name.startswith("out.0"):
                if name.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return model


@torch.no_grad()
def unet_forward(unet: UNetModel, x, timesteps, context, cond_emb=None, unet_block_indices: Tuple[int, ...] = (2, 5, 8, 11),
                 run_dead_code: bool = False) -> Tuple[Optional[torch.Tensor], List[torch.Tensor]]:
    """LdmExtractor.unet_forward (odise/modeling/meta_arch/ldm.py:469-491).  Returns (unet.out(h) or None, taps).
    `run_dead_code=True` also executes output_blocks[11] and `out`, whose result the reference discards (ldm.py:600)."""
    ret_features = []
    hs = []
    t_emb = timestep_embedding(timesteps, unet.model_channels)      # ldm.py:473
    emb = unet.time_embed(t_emb)                                     # :474
    if cond_emb is not None:
        emb = emb + cond_emb                                         # :475-476
    h = x
    for module in unet.input_blocks:                                 # :480-482
        h = module(h, emb, context)
        hs.append(h)
    h = unet.middle_block(h, emb, context)                           # :483
    last = len(unet.output_blocks) - 1
    for idx, module in enumerate(unet.output_blocks):                # :484-489
        h = torch.cat([h, hs.pop()], dim=1)
        if idx in unet_block_indices:
            ret_features.append(h.contiguous())
        if idx == last and not run_dead_code:
            return None, ret_features
        h = module(h, emb, context)
    return unet.out(h), ret_features                                 # :491


def config2_inputs(batch: int = 1, latent: int = 64, width_div: int = 1):
    """SURVEY.md §8d config 2 inputs: x_t N(0,1) seed 1, context = uncond-like + 0.1 N(0,1) seed 2, cond_emb N(0,0.02) seed 3."""
    x = torch.randn(batch, 4, latent, latent, generator=torch.Generator().manual_seed(1))
    g2 = torch.Generator().manual_seed(2)
    base = torch.randn(1, 77, 768, generator=g2)
    context = base + 0.1 * torch.randn(batch, 77, 768, generator=g2)
    cond_emb = 0.02 * torch.randn(batch, 1280 // width_div, generator=torch.Generator().manual_seed(3))
    return x, context, cond_emb
