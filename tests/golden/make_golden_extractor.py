"""Golden fixture for the diffusion feature extractor's DRIVER code, produced by the REFERENCE's own
`LdmImplicitCaptionerExtractor.forward` -> `LdmExtractor.forward` (odise/modeling/meta_arch/ldm.py:424-621, 697-718),
`ClipAdapter.embed_image` (clip.py:177-231) and the reference's own `GaussianDiffusion.q_sample` with the "ldm_linear" schedule
(odise/modeling/diffusion/gaussian_diffusion.py) in the build container.

The networks those methods walk (SD UNet, VAE: the `ldm` package; CLIP: open_clip) are absent, so the oracle's narrow stand-ins - which
keep ldm's / open_clip's module tree and attribute names - are plugged in as `ldm.encoder / ldm.unet / ldm.decoder / clip.visual`.
What is pinned is everything the reference repository itself contributes: pixel normalisation, deterministic latent (posterior mean x
scale_factor), shared-noise q_sample at t = 0, implicit-caption conditioning (`uncond + tanh(alpha) * clip_project(prefix)`, the
time-embedding term), which blocks are tapped and in what order, feature strides / dims.  ldm helpers (`timestep_embedding`,
`DiagonalGaussianDistribution.mean`) and torchvision's tensor Resize / CenterCrop / Normalize are restated (tests/golden/ref_stubs.py,
oracle.clip_vit.clip_preprocess).

    python tests/golden/make_golden_extractor.py
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_stubs  # noqa: E402

ref_stubs.install()
import odise.modeling.meta_arch.ldm as rl  # noqa: E402
from odise.modeling.diffusion import create_gaussian_diffusion  # noqa: E402
from odise.modeling.meta_arch.clip import ClipAdapter  # noqa: E402

ref_stubs.seal()
from oracle import clip_vit  # noqa: E402
from oracle.ldm_extractor import ImplicitCaptionerExtractor  # noqa: E402


class FakeLatentDiffusion(nn.Module):
    """The attributes of ldm.py's `LatentDiffusion` wrapper that LdmExtractor touches, over the oracle's modules."""

    def __init__(self, ext: ImplicitCaptionerExtractor):
        super().__init__()
        self.vae, self.unet_model = ext.vae, ext.unet
        self.ldm = types.SimpleNamespace(scale_factor=0.18215, first_stage_model=ext.vae, model=ext.unet)   # .model: set_requires_grad walks it
        self.diffusion = create_gaussian_diffusion(steps=1000, learn_sigma=False, noise_schedule="ldm_linear")   # ldm.py:106-111
        self.register_buffer("uncond_inputs", ext.uncond_inputs.clone())
        self.register_buffer("pixel_mean", torch.tensor((0.5, 0.5, 0.5)).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor((0.5, 0.5, 0.5)).view(-1, 1, 1), False)
        self.image_size, self.latent_image_size, self.latent_dim = (512, 512), (64, 64), 4

    encoder = property(lambda self: self.vae.encoder)
    decoder = property(lambda self: self.vae.decoder)
    unet = property(lambda self: self.unet_model)
    device = property(lambda self: self.pixel_mean.device)

    def embed_text(self, text):
        # ldm.py:567 evaluates this as the DEFAULT of `batched_inputs.get("cond_inputs", ...)` on every call and discards it
        return torch.full((len(text), 77, self.uncond_inputs.shape[-1]), float("nan"))


def case(name, seed, B, H, W):
    ext = ImplicitCaptionerExtractor(unet_div=10, vae_div=4, clip_kw=dict(image_size=56, patch_size=14, width=128, layers=2, heads=2, output_dim=32),
                                     context_dim=64, seed=seed)
    # attributes ldm's modules carry and the reference reads
    for m in ext.vae.modules():
        if m.__class__.__name__ == "ResnetBlock":
            m.in_channels = m.norm1.num_channels
    for blk in ext.unet.output_blocks:
        blk[0].channels = blk[0].in_layers[0].num_channels
    ext.vae.decoder.give_pre_end, ext.vae.decoder.tanh_out = False, False

    ldm = FakeLatentDiffusion(ext)
    lx = rl.LdmExtractor(ldm=ldm)                                          # the real constructor: shared_noise (seed 42), reset_dim_stride
    assert torch.equal(lx.shared_noise, ext.shared_noise), "shared noise differs from the oracle's"
    ic = rl.LdmImplicitCaptionerExtractor.__new__(rl.LdmImplicitCaptionerExtractor)   # its constructor downloads CLIP
    nn.Module.__init__(ic)
    ic.ldm_extractor = lx
    clip = ClipAdapter.__new__(ClipAdapter)
    nn.Module.__init__(clip)
    clip.clip = ext.clip
    clip.clip_preprocess = lambda image: clip_vit.clip_preprocess(image, ext.clip.visual.image_size)
    clip.name, clip.normalize = "synthetic", False
    ic.clip = clip
    ic.clip_project = rl.PositionalLinear(32, 64, 77)
    ic.clip_project.load_state_dict(ext.clip_project.state_dict())
    ic.alpha_cond = nn.Parameter(ext.alpha_cond.detach().clone())
    ic.learnable_time_embed = True
    ted = ext.unet.time_embed[-1].out_features
    ic.time_embed_project = rl.PositionalLinear(32, ted, 1)
    ic.time_embed_project.load_state_dict(ext.time_embed_project.state_dict())
    ic.alpha_cond_time_embed = nn.Parameter(ext.alpha_cond_time_embed.detach().clone())
    ic.eval()

    img = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(seed + 9))
    with torch.no_grad():
        feats = ic(dict(img=img))
    print(name, "dims", ic.feature_dims, "strides", ic.feature_strides, "groups", ic.grouped_indices, [tuple(f.shape) for f in feats])
    np.savez_compressed(os.path.join(HERE, f"extractor_{name}.npz"), image=img.numpy(), seed=np.int64(seed), feature_dims=np.array(ic.feature_dims),
                        feature_strides=np.array(ic.feature_strides), **{f"feat_{i}": (f.numpy().astype(np.float16) if i in (0, 7) else f.numpy()) for i, f in enumerate(feats)})   # stride-4 taps as fp16


if __name__ == "__main__":
    case("a", seed=11, B=1, H=64, W=64)        # shared noise is resized (bicubic) to the 8x8 latent
    case("b", seed=12, B=2, H=64, W=128)
