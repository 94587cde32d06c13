"""Oracle: Stable-Diffusion AutoencoderKL encoder/decoder and ODISE's tap extraction (TEST INFRASTRUCTURE — see oracle/__init__.py).

PARITY UNPINNED: the arithmetic lives in `ldm` (stable-diffusion-sdkit==2.1.3, reference setup.py:83), absent from
/root/reference.  Restated from the published architecture (ldm/modules/diffusionmodules/model.py `Encoder`, `Decoder`,
`ResnetBlock`, `AttnBlock`, `Downsample`, `Upsample`, `Normalize`; v1-inference.yaml first_stage_config ddconfig:
ch 128, ch_mult [1,2,4,4], num_res_blocks 2, z_channels 4, double_z, attn_resolutions [], embed_dim 4,
scale_factor 0.18215) with parameter names identical to the checkpoint keys `first_stage_model.*`, anchored on
LdmExtractor.encoder_forward / encode_to_latent / decoder_forward / decode_to_image
(odise/modeling/meta_arch/ldm.py:424-467, 493-541), which the functions below follow line by line.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

SCALE_FACTOR = 0.18215


def Normalize(ch):
    return nn.GroupNorm(num_groups=32, num_channels=ch, eps=1e-6, affine=True)


def nonlinearity(x):
    return x * torch.sigmoid(x)


class ResnetBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def forward(self, x, temb=None):
        h = self.conv1(nonlinearity(self.norm1(x)))
        h = self.conv2(nonlinearity(self.norm2(h)))
        if self.in_channels != self.out_channels:
            x = self.nin_shortcut(x)
        return x + h


class AttnBlock(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.norm = Normalize(ch)
        self.q = nn.Conv2d(ch, ch, 1)
        self.k = nn.Conv2d(ch, ch, 1)
        self.v = nn.Conv2d(ch, ch, 1)
        self.proj_out = nn.Conv2d(ch, ch, 1)

    def forward(self, x):
        h_ = self.norm(x)
        q, k, v = self.q(h_), self.k(h_), self.v(h_)
        b, c, h, w = q.shape
        q = q.reshape(b, c, h * w).permute(0, 2, 1)
        k = k.reshape(b, c, h * w)
        w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
        w_ = F.softmax(w_, dim=2)
        v = v.reshape(b, c, h * w)
        h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
        return x + self.proj_out(h_)


class Downsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class Upsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, 1, 1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Level(nn.Module):
    pass


class _Mid(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.block_1 = ResnetBlock(ch, ch)
        self.attn_1 = AttnBlock(ch)
        self.block_2 = ResnetBlock(ch, ch)


class Encoder(nn.Module):
    def __init__(self, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4):
        super().__init__()
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block = nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            down = _Level()
            down.block = block
            down.attn = nn.ModuleList()
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in)
            self.down.append(down)
        self.mid = _Mid(block_in)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels, 3, 1, 1)


class Decoder(nn.Module):
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4):
        super().__init__()
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        block_in = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = _Mid(block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            up = _Level()
            up.block = block
            up.attn = nn.ModuleList()
            if i_level != 0:
                up.upsample = Upsample(block_in)
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)


class AutoencoderKL(nn.Module):
    """`width_div` shrinks `ch` for fast CPU tests (1 = the real model)."""

    def __init__(self, width_div=1, embed_dim=4, z_channels=4):
        super().__init__()
        ch = 128 // width_div
        self.encoder = Encoder(ch=ch, z_channels=z_channels)
        self.decoder = Decoder(ch=ch, z_channels=z_channels)
        self.quant_conv = nn.Conv2d(2 * z_channels, 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, z_channels, 1)


def init_synthetic_(model: nn.Module, seed: int = 1234) -> nn.Module:
    g = torch.Generator().manual_seed(seed)
    for name, p in sorted(model.named_parameters()):
        with torch.no_grad():
            if p.ndim >= 2:
                gain = 0.5 if any(s in name for s in ("conv2", "proj_out")) else 1.0
                p.copy_(torch.randn(p.shape, generator=g) * (gain / math.sqrt(p[0].numel())))
            elif "norm" in name:
                p.copy_((1.0 if name.endswith("weight") else 0.0) + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return model


@torch.no_grad()
def encoder_forward(vae: AutoencoderKL, x, encoder_block_indices: Tuple[int, ...] = (5, 7)):
    """LdmExtractor.encoder_forward (ldm.py:424-457): taps are the INPUTS of the flat-indexed down blocks."""
    enc = vae.encoder
    ret_features = []
    hs = [enc.conv_in(x)]
    flat = 0
    for i_level in range(enc.num_resolutions):
        for i_block in range(enc.num_res_blocks):
            if flat in encoder_block_indices:            # :437-438
                ret_features.append(hs[-1].contiguous())
            hs.append(enc.down[i_level].block[i_block](hs[-1]))
            flat += 1
        if i_level != enc.num_resolutions - 1:
            hs.append(enc.down[i_level].downsample(hs[-1]))
    h = hs[-1]
    h = enc.mid.block_1(h)
    h = enc.mid.attn_1(h)
    h = enc.mid.block_2(h)
    h = enc.norm_out(h)
    h = h * torch.sigmoid(h)
    h = enc.conv_out(h)
    return h, ret_features


@torch.no_grad()
def encode_to_latent(vae: AutoencoderKL, image):
    """LdmExtractor.encode_to_latent (ldm.py:459-467): deterministic posterior mean * scale_factor."""
    h, feats = encoder_forward(vae, image)
    moments = vae.quant_conv(h)
    mean, _logvar = torch.chunk(moments, 2, dim=1)       # DiagonalGaussianDistribution.mean
    return SCALE_FACTOR * mean, feats


@torch.no_grad()
def decoder_forward(vae: AutoencoderKL, z, decoder_block_indices: Tuple[int, ...] = (2, 5), run_dead_code: bool = False):
    """LdmExtractor.decoder_forward (ldm.py:493-533).  With run_dead_code=False the walk stops once the last tap has been
    collected (everything after it only feeds the discarded reconstruction, ldm.py:606)."""
    dec = vae.decoder
    ret_features: List[torch.Tensor] = []
    h = dec.conv_in(z)
    h = dec.mid.block_1(h)
    h = dec.mid.attn_1(h)
    h = dec.mid.block_2(h)
    flat = 0
    last = max(decoder_block_indices)
    for i_level in reversed(range(dec.num_resolutions)):
        for i_block in range(dec.num_res_blocks + 1):
            if flat in decoder_block_indices:            # :515-516
                ret_features.append(h.contiguous())
                if flat == last and not run_dead_code:
                    return None, ret_features
            h = dec.up[i_level].block[i_block](h)
            flat += 1
        if i_level != 0:
            h = dec.up[i_level].upsample(h)
    h = dec.norm_out(h)
    h = h * torch.sigmoid(h)
    h = dec.conv_out(h)
    return h, ret_features


@torch.no_grad()
def decode_to_image(vae: AutoencoderKL, z, run_dead_code: bool = False):
    """LdmExtractor.decode_to_image (ldm.py:535-541)."""
    z = 1.0 / SCALE_FACTOR * z
    z = vae.post_quant_conv(z)
    return decoder_forward(vae, z, run_dead_code=run_dead_code)
