"""`odise.modeling.backbone.feature_extractor.FeatureExtractorBackbone` on libodise_hip.so (reference:
odise/modeling/backbone/feature_extractor.py:29-256): same constructor, `output_shape()` / `size_divisibility`, parameter names of the
per-tap BottleneckBlock projections (`feature_projections.{i}.0.{conv1,conv2,conv3[,shortcut]}.{weight,norm.weight,norm.bias}`)."""
import math
from collections import OrderedDict, defaultdict, namedtuple
from typing import List, Tuple, Union

import numpy as np
import torch
from torch import nn

from odise_amd import dropin

try:  # detectron2 present (a real deployment): its ShapeSpec, so instantiate_odise hands the genuine type on
    from detectron2.layers import ShapeSpec
    if not isinstance(ShapeSpec, type):
        raise ImportError
except Exception:  # noqa: BLE001
    ShapeSpec = namedtuple("ShapeSpec", ["channels", "height", "width", "stride"], defaults=(None, None, None, None))


class _ConvNorm(nn.Module):
    """detectron2.layers.Conv2d with norm="GN": parameters `weight`, `norm.weight`, `norm.bias`."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        self.norm = nn.GroupNorm(32, cout)


class _Bottleneck(nn.Module):
    """detectron2 BottleneckBlock(in, out, bottleneck_channels=out // 4, norm="GN") - parameters only."""

    def __init__(self, cin, cout, mid):
        super().__init__()
        if cin != cout:
            self.shortcut = _ConvNorm(cin, cout, 1)
        self.conv1, self.conv2, self.conv3 = _ConvNorm(cin, mid, 1), _ConvNorm(mid, mid, 3), _ConvNorm(mid, cout, 1)


class FeatureExtractorBackbone(nn.Module):
    def __init__(self, feature_extractor, out_features: List[str], backbone_in_size: Union[int, Tuple[int]] = (512, 512), min_stride: int = 4,
                 max_stride: int = 32, projection_dim: int = 512, num_res_blocks: int = 1, use_checkpoint: bool = False, slide_training: bool = False):
        super().__init__()
        if isinstance(backbone_in_size, int) or tuple(backbone_in_size) != (512, 512) or num_res_blocks != 1 or (min_stride, max_stride) != (4, 32):
            raise NotImplementedError("libodise_hip implements the released configuration: 512x512 slide windows, one BottleneckBlock per tap, strides 4..32")
        if not slide_training:
            # feature_extractor.py:197-203: without slide_training the inference window is the image's uncapped short side, resized down to
            # 512x512 (one huge window for a 1024x1024 picture) - different features from the window = min(512, short side) rule the library
            # implements; both released configurations set slide_training=True (configs/common/models/odise_with_label.py:28)
            raise NotImplementedError("libodise_hip implements slide_training=True (window = min(512, short side)), the released configurations' setting")
        self.feature_extractor = feature_extractor
        self.use_checkpoint = use_checkpoint
        self.feature_projections = nn.ModuleList(nn.Sequential(_Bottleneck(d, projection_dim, projection_dim // 4)) for d in feature_extractor.feature_dims)
        self.backbone_in_size, self._slide_inference, self._slide_training = (512, 512), True, slide_training
        self.min_stride, self.max_stride = min_stride, max_stride
        idx_to_stride, stride_to_indices = {}, defaultdict(list)
        for indices in feature_extractor.grouped_indices:                    # feature_extractor.py:88-97
            for idx in indices:
                stride = min(max(feature_extractor.feature_strides[idx], min_stride), max_stride)
                idx_to_stride[idx] = stride
                stride_to_indices[stride].append(idx)
        self._sorted_grouped_indices = [stride_to_indices[s] for s in sorted(stride_to_indices)]
        self._out_feature_channels, self._out_feature_strides = {}, {}
        for indices in self._sorted_grouped_indices:
            stride = idx_to_stride[indices[0]]
            name = f"s{int(math.log2(stride))}"
            if name not in out_features:
                continue
            assert name not in self._out_feature_strides, f"Duplicate feature name {name}"
            self._out_feature_strides[name], self._out_feature_channels[name] = stride, projection_dim
        self._out_features = list(self._out_feature_strides)
        if self._out_features != ["s2", "s3", "s4", "s5"]:
            raise NotImplementedError("libodise_hip produces the four maps s2..s5 of the released models")
        self._hip = None

    @property
    def size_divisibility(self) -> int:
        return 64

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name]) for name in self._out_features}

    def ignored_state_dict(self, destination=None, prefix=""):
        return destination if destination is not None else OrderedDict()

    def library_state(self, prefix="backbone."):
        state = self.feature_extractor.library_state(prefix + "feature_extractor.")
        state.update({prefix + k: v.detach().cpu().numpy() for k, v in self.state_dict().items() if k.startswith("feature_projections.")})
        return state

    def forward(self, img):
        """[B,3,H,W] in [0,1] (H, W multiples of 64) -> {"s2".."s5": [B,512,H/stride,W/stride]} (feature_extractor.py:252-256)."""
        from odise_amd.pipeline import HipODISE
        if self._hip is None:
            self._hip = HipODISE(dropin.get_context(), self.library_state(), with_head=False)
        out = self._hip.backbone(img.detach().cpu().numpy().astype(np.float32))
        return {k: torch.from_numpy(v).to(img.device) for k, v in out.items()}
