"""Oracle: FeatureExtractorBackbone (TEST INFRASTRUCTURE — see oracle/__init__.py).

Follows odise/modeling/backbone/feature_extractor.py line by line: `single_forward` 139-155, `forward_features` 157-179,
`slide_forward` 181-250.  PINNED for the window placement / resize / projection-sum / overlap averaging: tests/test_oracle_golden.py
replays golden vectors written by the reference's own class (tests/golden/make_golden_backbone.py).  detectron2 v0.6 is absent from
/root/reference (its pieces are restated here AND in tests/golden/ref_stubs.py, so they are not independently pinned): `BottleneckBlock`
(detectron2/modeling/backbone/resnet.py) and `ImageList.from_tensors` are restated from SURVEY.md Appendix A.4:
conv1 1x1 (in->128) + GN32 + ReLU -> conv2 3x3 (128->128) + GN32 + ReLU -> conv3 1x1 (128->512) + GN32;
shortcut 1x1 (in->512) + GN32 iff in != 512; ReLU(out + shortcut); convs bias-free; keys `convN.weight`, `convN.norm.*`.
"""
from __future__ import annotations

import math
from collections import defaultdict
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ldm_extractor import ImplicitCaptionerExtractor

FEATURE_STRIDES = [4, 8, 64, 32, 16, 8, 8, 4]  # LdmExtractor.reset_dim_stride (ldm.py:284-346) for taps enc5, enc7, u2, u5, u8, u11, dec2, dec5


class _ConvNorm(nn.Conv2d):
    """detectron2.layers.Conv2d with a GroupNorm stored as `.norm` and an optional activation."""

    def __init__(self, cin, cout, k, padding=0, relu=False):
        super().__init__(cin, cout, k, padding=padding, bias=False)
        self.norm = nn.GroupNorm(32, cout)
        self._relu = relu

    def forward(self, x):
        x = self.norm(super().forward(x))
        return F.relu(x) if self._relu else x


class BottleneckBlock(nn.Module):
    def __init__(self, in_channels, out_channels, bottleneck_channels):
        super().__init__()
        self.shortcut = _ConvNorm(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.conv1 = _ConvNorm(in_channels, bottleneck_channels, 1, relu=True)
        self.conv2 = _ConvNorm(bottleneck_channels, bottleneck_channels, 3, padding=1, relu=True)
        self.conv3 = _ConvNorm(bottleneck_channels, out_channels, 1)

    def forward(self, x):
        out = self.conv3(self.conv2(self.conv1(x)))
        shortcut = self.shortcut(x) if self.shortcut is not None else x
        return F.relu(out + shortcut)


class FeatureExtractorBackbone(nn.Module):
    def __init__(self, extractor: ImplicitCaptionerExtractor, feature_dims: List[int], projection_dim=512, min_stride=4, max_stride=32,
                 backbone_in_size=(512, 512), seed=99):
        super().__init__()
        self.feature_extractor = extractor
        self.feature_projections = nn.ModuleList(
            nn.Sequential(BottleneckBlock(d, projection_dim, projection_dim // 4)) for d in feature_dims)
        self.backbone_in_size = tuple(backbone_in_size)
        idx_to_stride, stride_to_indices = {}, defaultdict(list)
        for idx, stride in enumerate(FEATURE_STRIDES):                       # feature_extractor.py:88-94
            stride = min(max(stride, min_stride), max_stride)
            idx_to_stride[idx] = stride
            stride_to_indices[stride].append(idx)
        self._sorted_grouped_indices = [stride_to_indices[s] for s in sorted(stride_to_indices)]
        self._out_feature_strides = {f"s{int(math.log2(idx_to_stride[ind[0]]))}": idx_to_stride[ind[0]] for ind in self._sorted_grouped_indices}
        self._out_features = list(self._out_feature_strides)
        self.projection_dim = projection_dim
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, p in sorted(self.feature_projections.named_parameters()):
                if p.ndim >= 2:
                    p.copy_(torch.randn(p.shape, generator=g) / math.sqrt(p[0].numel()))
                else:
                    p.copy_((1.0 if name.endswith("weight") else 0.0) + 0.1 * torch.randn(p.shape, generator=g))
        self.eval()

    @torch.no_grad()
    def single_forward(self, img):                                            # :139-155
        input_image_size = img.shape[-2:]
        if tuple(input_image_size) != self.backbone_in_size:                  # T.Resize(size, BICUBIC) (:75-77); identity at 512
            img = F.interpolate(img, size=self.backbone_in_size, mode="bicubic", align_corners=False)
        features = self.feature_extractor(img)                                # ImageList pad to /64 is a no-op at 512
        return self.forward_features(features, input_image_size)

    @torch.no_grad()
    def forward_features(self, features, input_image_size):                   # :157-179
        out = {}
        for name, indices in zip(self._out_features, self._sorted_grouped_indices):
            acc = None
            stride = self._out_feature_strides[name]
            for idx in indices:
                restored = F.interpolate(features[idx], size=(input_image_size[-2] // stride, input_image_size[-1] // stride))
                proj = self.feature_projections[idx](restored)
                acc = proj if acc is None else acc + proj
            out[name] = acc
        return out

    @torch.no_grad()
    def forward(self, img) -> Dict[str, torch.Tensor]:                        # slide_forward :181-250 (slide_training=True)
        b, _, h_img, w_img = img.shape
        outs = {k: torch.zeros(b, self.projection_dim, h_img // s, w_img // s) for k, s in self._out_feature_strides.items()}
        counts = {k: torch.zeros_like(v) for k, v in outs.items()}
        short_side = min(min(self.backbone_in_size), min(img.shape[-2:]))
        h_crop = w_crop = h_stride = w_stride = short_side
        h_grids = max(h_img - h_crop + h_stride - 1, 0) // h_stride + 1
        w_grids = max(w_img - w_crop + w_stride - 1, 0) // w_stride + 1
        for h_idx in range(h_grids):
            for w_idx in range(w_grids):
                y1, x1 = h_idx * h_stride, w_idx * w_stride
                y2, x2 = min(y1 + h_crop, h_img), min(x1 + w_crop, w_img)
                y1, x1 = max(y2 - h_crop, 0), max(x2 - w_crop, 0)
                crop = self.single_forward(img[:, :, y1:y2, x1:x2])
                for k, s in self._out_feature_strides.items():
                    outs[k][:, :, y1 // s:y2 // s, x1 // s:x2 // s] += crop[k]
                    counts[k][..., y1 // s:y2 // s, x1 // s:x2 // s] += 1
        for k in outs:
            outs[k] /= counts[k]
        return outs


class TapStandIn(nn.Module):
    """A cheap feature extractor with LdmExtractor's interface (8 maps at FEATURE_STRIDES): average pooling + a fixed 1x1 mixing of the
    RGB planes.  tests/golden/make_golden_backbone.py feeds it to the REFERENCE's FeatureExtractorBackbone and to the class above, which
    pins the sliding-window logic of this file to the reference's."""

    feature_strides = FEATURE_STRIDES
    grouped_indices = [[i] for i in range(8)]

    def __init__(self, dims: List[int], seed: int):
        super().__init__()
        self.feature_dims = list(dims)
        g = torch.Generator().manual_seed(seed)
        self.mix = [torch.randn(d, 3, generator=g) for d in dims]

    def forward(self, x):
        img = x["img"] if isinstance(x, dict) else x
        return [torch.einsum("oc,bchw->bohw", m, F.avg_pool2d(img, s)) for m, s in zip(self.mix, FEATURE_STRIDES)]


def crop_boxes(h_img: int, w_img: int, crop: int = 512):
    """The (y1, x1, y2, x2) windows of slide_forward (feature_extractor.py:197-222), e.g. 4 boxes at 1024^2, 9 at 1280^2."""
    short = min(crop, h_img, w_img)
    hg = max(h_img - short + short - 1, 0) // short + 1
    wg = max(w_img - short + short - 1, 0) // short + 1
    boxes = []
    for hi in range(hg):
        for wi in range(wg):
            y2, x2 = min(hi * short + short, h_img), min(wi * short + short, w_img)
            boxes.append((max(y2 - short, 0), max(x2 - short, 0), y2, x2))
    return boxes
