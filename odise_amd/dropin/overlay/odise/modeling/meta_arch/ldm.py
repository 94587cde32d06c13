"""`odise.modeling.meta_arch.ldm` on libodise_hip.so: `LdmImplicitCaptionerExtractor` with the reference's constructor arguments,
properties and parameter names (/root/reference odise/modeling/meta_arch/ldm.py:638-722, 236-388, 624-635).  The frozen LDM / CLIP
networks are not PyTorch modules here: their weights go from the checkpoint files straight into the library (odise_amd.dropin.frozen_state)."""
from collections import OrderedDict
from typing import Tuple

import numpy as np
import torch
from torch import nn

from odise_amd import dropin

# SD v1 architecture constants the reference reads off the instantiated LDM (ldm.py:284-346 `reset_dim_stride`):
# encoder ResnetBlock input channels (ch 128, ch_mult 1-2-4-4, 2 blocks per level), UNet output-block input channels
# (model_channels 320, mult 1-2-4-4: concat inputs), decoder block input channels (3 blocks per level, top level first)
_ENC_IN = [128, 128, 128, 256, 256, 512, 512, 512]
_UNET_OUT_IN = [2560, 2560, 2560, 2560, 2560, 1920, 1920, 1280, 960, 960, 640, 640]
_DEC_IN = [512, 512, 512, 512, 512, 512, 512, 256, 256, 256, 128, 128]
_CONTEXT = (77, 768)        # ldm.embed_text([""]).shape[1:]
_TIME_EMBED = 1280          # unet.time_embed[-1].out_features
_CLIP_DIM = {"ViT-L-14-336": 768, "ViT-L-14": 768, "ViT-B-16": 512, "ViT-B-32": 512}


class PositionalLinear(nn.Module):
    """ldm.py:624-635 (parameters only; the arithmetic runs in the library's cond_inputs kernel)."""

    def __init__(self, in_features, out_features, seq_len=77, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias)
        self.positional_embedding = nn.Parameter(torch.zeros(1, seq_len, out_features))
        nn.init.trunc_normal_(self.positional_embedding, std=0.02)


class LdmImplicitCaptionerExtractor(nn.Module):
    def __init__(self, learnable_time_embed=True, num_timesteps=1, clip_model_name="ViT-L-14", encoder_block_indices: Tuple[int, ...] = (5, 7),
                 unet_block_indices: Tuple[int, ...] = (2, 5, 8, 11), decoder_block_indices: Tuple[int, ...] = (2, 5), steps: Tuple[int, ...] = (0,),
                 share_noise: bool = True, enable_resize: bool = False, ldm=None, init_checkpoint: str = "sd://v1-3"):
        super().__init__()
        if (tuple(encoder_block_indices), tuple(unet_block_indices), tuple(decoder_block_indices), tuple(steps)) != ((5, 7), (2, 5, 8, 11), (2, 5), (0,)):
            raise NotImplementedError("libodise_hip builds the taps of the released ODISE models: encoder (5, 7), unet (2, 5, 8, 11), decoder (2, 5), steps (0,)")
        if not share_noise or enable_resize or ldm is not None or not learnable_time_embed or num_timesteps != 1:
            raise NotImplementedError("libodise_hip implements the released configuration (shared noise, learnable time embedding, one timestep)")
        self.encoder_block_indices, self.unet_block_indices = tuple(encoder_block_indices), tuple(unet_block_indices)
        self.decoder_block_indices, self.steps = tuple(decoder_block_indices), tuple(steps)
        self.clip_model_name, self.init_checkpoint = clip_model_name, init_checkpoint
        dim = _CLIP_DIM[clip_model_name]
        self.text_embed_shape = _CONTEXT
        self.clip_project = PositionalLinear(dim, _CONTEXT[1], _CONTEXT[0])
        self.alpha_cond = nn.Parameter(torch.zeros(1, *_CONTEXT))
        self.learnable_time_embed = learnable_time_embed
        self.time_embed_project = PositionalLinear(dim, _TIME_EMBED, num_timesteps)
        self.alpha_cond_time_embed = nn.Parameter(torch.zeros(_TIME_EMBED))
        self._hip = None

    # ---- the properties FeatureExtractorBackbone reads (ldm.py:348-388, 672-692) ---------------------------------------------------
    @property
    def feature_size(self):
        return 512

    @property
    def feature_dims(self):
        return ([_ENC_IN[i] for i in self.encoder_block_indices] + [_UNET_OUT_IN[i] for i in self.unet_block_indices] * len(self.steps)
                + [_DEC_IN[i] for i in self.decoder_block_indices])

    @property
    def feature_strides(self):
        enc = [2 ** ((i + 2) // 2 - 1) for i in self.encoder_block_indices]
        unet = [64 // (2 ** ((i + 3) // 3 - 1)) for i in self.unet_block_indices]
        dec = [8 // (2 ** ((i + 3) // 3 - 1)) for i in self.decoder_block_indices]
        return enc + unet * len(self.steps) + dec

    @property
    def num_groups(self) -> int:
        return len(self.encoder_block_indices) + len(self.unet_block_indices) + len(self.decoder_block_indices)

    @property
    def grouped_indices(self):
        ret = [[i] for i in range(len(self.encoder_block_indices))]
        off = len(self.encoder_block_indices)
        nu = len(self.unet_block_indices)
        ret += [[i + t * nu + off for t in range(len(self.steps))] for i in range(nu)]
        off += len(self.steps) * nu
        return ret + [[i + off] for i in range(len(self.decoder_block_indices))]

    def extra_repr(self):
        return f"learnable_time_embed={self.learnable_time_embed}"

    def ignored_state_dict(self, destination=None, prefix=""):
        """The frozen LDM / CLIP weights never are part of this module's state (helper.py:35-46 in the reference): nothing to list."""
        return destination if destination is not None else OrderedDict()

    # ---- state for the library ---------------------------------------------------------------------------------------------------------
    def library_state(self, prefix="backbone.feature_extractor."):
        state = dict(dropin.frozen_state(self.init_checkpoint, self.clip_model_name))
        state.update({prefix + k: v.detach().cpu().numpy() for k, v in self.state_dict().items()})
        return state

    def forward(self, batched_inputs):
        """{"img": [B,3,H,W] in [0,1]} -> the 8 feature maps (fp32 NCHW), ldm.py:697-718 / 608."""
        from odise_amd.extractor import HipFeatureExtractor
        img = batched_inputs["img"]
        if self._hip is None:
            self._hip = HipFeatureExtractor(dropin.get_context(), self.library_state())
        taps = self._hip.features(img.detach().cpu().numpy().astype(np.float32))
        return [torch.from_numpy(t).to(img.device) for t in taps]
