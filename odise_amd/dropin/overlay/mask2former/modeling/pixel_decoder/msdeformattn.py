"""`mask2former.modeling.pixel_decoder.msdeformattn.MSDeformAttnPixelDecoder` on libodise_hip.so (reference:
third_party/Mask2Former/mask2former/modeling/pixel_decoder/msdeformattn.py:162-358): constructor arguments and parameter names of the
reference; `forward_features` runs `odise_hip_pixel_decoder_forward` with this module's own weights."""
import ctypes as C
from typing import Callable, Dict, List, Optional, Union

import numpy as np
import torch
from torch import nn

from odise_amd import dropin


class _ConvNorm(nn.Module):
    """detectron2 Conv2d(bias=False, norm=GN): `weight`, `norm.{weight,bias}`."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.xavier_uniform_(self.weight)
        self.norm = nn.GroupNorm(32, cout)


class _MSDeformAttnParams(nn.Module):
    """ops/modules/ms_deform_attn.py:27-81 (parameters only)."""

    def __init__(self, d_model, n_levels, n_heads, n_points):
        super().__init__()
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)


class _EncoderLayerParams(nn.Module):
    def __init__(self, d_model, d_ffn, n_levels, n_heads, n_points):
        super().__init__()
        self.self_attn = _MSDeformAttnParams(d_model, n_levels, n_heads, n_points)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1, self.linear2 = nn.Linear(d_model, d_ffn), nn.Linear(d_ffn, d_model)
        self.norm2 = nn.LayerNorm(d_model)


class _Encoder(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)


class _EncoderOnly(nn.Module):
    """MSDeformAttnTransformerEncoderOnly (msdeformattn.py:27-89): `level_embed`, `encoder.layers.{i}`."""

    def __init__(self, d_model, nhead, num_layers, d_ffn, n_levels=3, n_points=4):
        super().__init__()
        self.encoder = _Encoder(_EncoderLayerParams(d_model, d_ffn, n_levels, nhead, n_points) for _ in range(num_layers))
        self.level_embed = nn.Parameter(torch.empty(n_levels, d_model).normal_())


class MSDeformAttnPixelDecoder(nn.Module):
    def __init__(self, input_shape: Dict[str, object], *, transformer_dropout: float, transformer_nheads: int, transformer_dim_feedforward: int,
                 transformer_enc_layers: int, conv_dim: int, mask_dim: int, norm: Optional[Union[str, Callable]] = None,
                 transformer_in_features: List[str], common_stride: int):
        super().__init__()
        shapes = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, _ in shapes]
        tr = sorted(((k, v) for k, v in input_shape.items() if k in transformer_in_features), key=lambda x: x[1].stride)
        self.transformer_in_features = [k for k, _ in tr]
        if norm != "GN" or common_stride != 4 or self.in_features != ["s2", "s3", "s4", "s5"] and self.in_features != ["res2", "res3", "res4", "res5"]:
            raise NotImplementedError("libodise_hip implements the released pixel decoder: GN, common_stride 4, four input maps, three transformer levels")
        if len(self.transformer_in_features) != 3 or conv_dim != mask_dim:
            raise NotImplementedError("libodise_hip implements three transformer levels with conv_dim == mask_dim")
        self.transformer_num_feature_levels, self.conv_dim, self.mask_dim, self.common_stride = 3, conv_dim, mask_dim, common_stride
        self.maskformer_num_feature_levels, self.num_fpn_levels = 3, 1
        # input_proj: highest stride first (msdeformattn.py:212-226), Conv2d(1x1, bias) + GroupNorm
        self.input_proj = nn.ModuleList(nn.Sequential(nn.Conv2d(v.channels, conv_dim, 1), nn.GroupNorm(32, conv_dim)) for _, v in tr[::-1])
        self.transformer = _EncoderOnly(conv_dim, transformer_nheads, transformer_enc_layers, transformer_dim_feedforward)
        self.mask_features = nn.Conv2d(conv_dim, mask_dim, 1)
        in_ch = shapes[0][1].channels
        self.adapter_1, self.layer_1 = _ConvNorm(in_ch, conv_dim, 1), _ConvNorm(conv_dim, conv_dim, 3)
        self._state_prefix = "sem_seg_head.pixel_decoder."

    def forward_features(self, features):
        """{"s2".."s5": [B,C,H/stride,W/stride]} -> (mask_features, multi_scale[0], multi_scale[:3])  (msdeformattn.py:314-358)."""
        owner = getattr(self, "_head", None)
        if owner is None:
            raise RuntimeError("MSDeformAttnPixelDecoder runs inside a MaskFormerHead (which owns the library's head build)")
        return owner()._pixel_decoder(features)
