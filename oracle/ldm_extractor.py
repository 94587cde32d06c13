"""Oracle: LdmImplicitCaptionerExtractor / LdmExtractor forward (TEST INFRASTRUCTURE — see oracle/__init__.py).

Follows odise/modeling/meta_arch/ldm.py line by line:
  LdmImplicitCaptionerExtractor.forward   ldm.py:697-718   (CLIP image embed -> cond_inputs [B,77,768], cond_emb [B,1,1280])
  PositionalLinear                        ldm.py:624-635
  LdmExtractor.forward                    ldm.py:543-621   (normalise, encode, q_sample(t=0, shared noise seed 42), unet taps,
                                                            decoder taps; 8 features in the order enc5, enc7, u2, u5, u8, u11, dec2, dec5)
  GaussianDiffusion.q_sample / ldm_linear odise/modeling/diffusion/gaussian_diffusion.py:104-137, 275-292, 1038-1051
The SD / CLIP sub-networks are the restatements in oracle/sd_unet.py, oracle/sd_vae.py, oracle/clip_vit.py (parity unpinned
for ldm, HF-cross-checked for CLIP).  `uncond_inputs` (= frozen text encoder of "", ldm.py:116) is a constant [1,77,768]
buffer; with synthetic weights it is a seeded random tensor.
PINNED (driver logic): tests/test_oracle_golden.py replays golden vectors written by the reference's own forward walking these
modules (tests/golden/make_golden_extractor.py); the UNet / VAE / CLIP arithmetic itself lives in absent pip packages (unpinned).
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import clip_vit, sd_unet, sd_vae


def ldm_linear_alphas_cumprod(num_timesteps: int = 1000) -> np.ndarray:
    """gaussian_diffusion.py:125-135 ("ldm_linear") + GaussianDiffusion.__init__ cumprod (float64)."""
    scale = 1000 / num_timesteps
    betas = np.linspace((scale * 0.00085) ** 0.5, (scale * 0.012) ** 0.5, num_timesteps, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas, axis=0)


def q_sample_coeffs(t: int = 0):
    """(sqrt(alpha_bar_t), sqrt(1-alpha_bar_t)) as float32 scalars, the way _extract_into_tensor casts them (:1038-1051)."""
    ac = ldm_linear_alphas_cumprod()
    return float(np.float32(np.sqrt(ac[t]))), float(np.float32(np.sqrt(1.0 - ac[t])))


def shared_noise(latent_dim: int = 4, latent_hw=(64, 64)) -> torch.Tensor:
    """LdmExtractor.__init__ (ldm.py:271-277): CPU generator seeded with 42."""
    rng = torch.Generator().manual_seed(42)
    return torch.randn(1, latent_dim, *latent_hw, generator=rng)


class PositionalLinear(nn.Module):
    def __init__(self, in_features, out_features, seq_len=77):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features)
        self.positional_embedding = nn.Parameter(torch.zeros(1, seq_len, out_features))

    def forward(self, x):
        return self.linear(x).unsqueeze(1) + self.positional_embedding


class ImplicitCaptionerExtractor(nn.Module):
    """width_div: 1 = real shapes (UNet 859.5M, VAE 83.7M, ViT-L/14@336); otherwise narrow stand-ins with the same graph."""

    def __init__(self, unet_div=1, vae_div=1, clip_kw: Optional[dict] = None, context_dim=768, seed=1234):
        super().__init__()
        self.unet = sd_unet.init_synthetic_(sd_unet.UNetModel(width_div=unet_div, context_dim=context_dim), seed)
        self.vae = sd_vae.init_synthetic_(sd_vae.AutoencoderKL(width_div=vae_div), seed + 1)
        clip_kw = clip_kw or {}
        self.clip = clip_vit.init_synthetic_(clip_vit.CLIPVisual(**clip_kw), seed + 2)
        dim_latent = self.clip.visual.proj.shape[1]
        ted = self.unet.time_embed[-1].out_features
        g = torch.Generator().manual_seed(seed + 3)
        self.register_buffer("uncond_inputs", torch.randn(1, 77, context_dim, generator=g))
        self.clip_project = PositionalLinear(dim_latent, context_dim, 77)
        self.alpha_cond = nn.Parameter(torch.zeros(1, 77, context_dim))
        self.time_embed_project = PositionalLinear(dim_latent, ted, 1)
        self.alpha_cond_time_embed = nn.Parameter(torch.zeros(ted))
        with torch.no_grad():  # trained values are non-zero; give every trainable tensor a seeded non-trivial value
            for p, std in ((self.clip_project.linear.weight, 1 / math.sqrt(dim_latent)), (self.clip_project.linear.bias, 0.02),
                           (self.clip_project.positional_embedding, 0.02), (self.alpha_cond, 0.5),
                           (self.time_embed_project.linear.weight, 1 / math.sqrt(dim_latent)), (self.time_embed_project.linear.bias, 0.02),
                           (self.time_embed_project.positional_embedding, 0.02), (self.alpha_cond_time_embed, 0.5)):
                p.copy_(torch.randn(p.shape, generator=g) * std)
        self.register_buffer("shared_noise", shared_noise())
        self.eval()

    # ---- LdmImplicitCaptionerExtractor.forward (ldm.py:697-718) -------------------------------------------------------
    @torch.no_grad()
    def conditioning(self, image: torch.Tensor):
        prefix = clip_vit.embed_image(self.clip, image)                                           # :705
        prefix_embed = self.clip_project(prefix)                                                  # :706
        cond_inputs = self.uncond_inputs + torch.tanh(self.alpha_cond) * prefix_embed             # :707-709
        cond_emb = torch.tanh(self.alpha_cond_time_embed) * self.time_embed_project(prefix)       # :711-714  [B,1,ted]
        return cond_inputs, cond_emb

    @torch.no_grad()
    def forward(self, image: torch.Tensor, run_dead_code: bool = False) -> List[torch.Tensor]:
        """image [B,3,H,W] in [0,1] (H,W multiples of 64; the reference always feeds 512x512 crops)."""
        cond_inputs, cond_emb = self.conditioning(image)
        # ---- LdmExtractor.forward (ldm.py:543-621) ----
        batch = image.shape[0]
        normalized = (image - 0.5) / 0.5                                                          # :556
        latent, enc_feats = sd_vae.encode_to_latent(self.vae, normalized)                         # :566
        t = torch.zeros(batch, dtype=torch.long)                                                  # :583 (steps=(0,))
        noise = self.shared_noise
        if noise.shape[2:] != latent.shape[2:]:                                                   # :586-591
            noise = torch.nn.functional.interpolate(noise, size=latent.shape[2:], mode="bicubic", align_corners=False)
        a, b = q_sample_coeffs(0)
        x_t = a * latent + b * noise.expand_as(latent)                                            # :598
        _, unet_feats = sd_unet.unet_forward(self.unet, x_t, t, cond_inputs, cond_emb[:, 0], run_dead_code=run_dead_code)  # :599
        _, dec_feats = sd_vae.decode_to_image(self.vae, latent, run_dead_code=run_dead_code)      # :606
        return [*enc_feats, *unet_feats, *dec_feats]                                              # :608

    def export_state(self):
        """Flat state dict keyed like the three real weight sources: SD ckpt (`model.diffusion_model.*`,
        `first_stage_model.*`), OpenAI CLIP (`visual.*`) and the ODISE checkpoint (`backbone.feature_extractor.*`)."""
        sd = {}
        for k, v in self.unet.state_dict().items():
            sd["model.diffusion_model." + k] = v
        for k, v in self.vae.state_dict().items():
            sd["first_stage_model." + k] = v
        for k, v in self.clip.state_dict().items():
            sd["clip." + k] = v
        fe = "backbone.feature_extractor."
        for k in ("clip_project.linear.weight", "clip_project.linear.bias", "clip_project.positional_embedding",
                  "time_embed_project.linear.weight", "time_embed_project.linear.bias", "time_embed_project.positional_embedding"):
            mod, _, leaf = k.rpartition(".")
            obj = self
            for part in mod.split("."):
                obj = getattr(obj, part)
            sd[fe + k] = getattr(obj, leaf)
        sd[fe + "alpha_cond"] = self.alpha_cond
        sd[fe + "alpha_cond_time_embed"] = self.alpha_cond_time_embed
        sd[fe + "ldm_extractor.ldm.uncond_inputs"] = self.uncond_inputs
        sd[fe + "ldm_extractor.shared_noise"] = self.shared_noise
        return {k: v.detach() for k, v in sd.items()}
