"""Import the REFERENCE's own Python modules in the build container (test-fixture generation only; never on the GPU box).

ODISE's inference code (odise/modeling/meta_arch/odise.py) and the vendored Mask2Former modules
(third_party/Mask2Former/mask2former/...) are pure PyTorch, but they import detectron2, fvcore, open_clip, ldm ... which are not
installed here.  This module makes those imports succeed WITHOUT running any third-party arithmetic that is not written out below:

  * the package `__init__`s of `odise` and `mask2former` (which pull in datasets, evaluators, trainers) are bypassed by registering
    empty package shells whose `__path__` points at the real directories, so sub-modules load from the reference's files as they are;
  * the handful of detectron2 / fvcore helpers the inference path really executes are restated here (detectron2 v0.6 semantics):
    `configurable` (explicit-kwargs construction only), `Conv2d` (conv -> norm -> activation), `get_norm("GN")`, `ShapeSpec`,
    `ImageList.from_tensors`, `sem_seg_postprocess`, `retry_if_cuda_oom`, `Boxes` / `Instances` / `BitMasks.get_bounding_boxes`,
    the registries' `register()` decorator, `c2_xavier_fill` / `c2_msra_fill`, `Backbone`, resnet `BottleneckBlock` / `ResNet.make_stage`,
    torchvision's tensor `Resize`;
  * every other name of those packages resolves to an inert placeholder that may be used as a decorator / base class / constant while
    the reference modules are being imported and raises as soon as anything CALLS it afterwards (`seal()`), so a fixture can never
    silently depend on a stub.

Usage: `import ref_stubs; ref_stubs.install(); from odise.modeling.meta_arch.odise import ...; ref_stubs.seal()`.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types

import torch
import torch.nn.functional as F
from torch import nn

REFERENCE = "/root/reference"
M2F = os.path.join(REFERENCE, "third_party", "Mask2Former")
_SEALED = False
STUB_ROOTS = ("detectron2", "torchvision", "fvcore", "diffdist", "open_clip", "ldm", "panopticapi", "pycocotools", "timm", "kornia", "omegaconf", "nltk", "wandb",
              "cv2", "scipy.optimize", "pytorch_lightning", "taming", "clip", "transformers_stub", "iopath", "termcolor", "tabulate_stub",
              "MultiScaleDeformableAttention", "mmcv", "shapely", "cityscapesscripts", "lvis", "h5py", "colorama", "git")


class _Placeholder:
    """Usable as decorator, base class, attribute bag and constant at import time; calling it after seal() is an error."""

    def __init__(self, name="placeholder"):
        object.__setattr__(self, "_name", name)

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Placeholder(f"{self._name}.{item}")

    def __call__(self, *args, **kwargs):
        if _SEALED:
            raise NotImplementedError(f"reference code called the stub {self._name}: restate it in tests/golden/ref_stubs.py")
        if len(args) == 1 and not kwargs and (isinstance(args[0], type) or callable(args[0])):
            return args[0]                                                # used as a decorator
        return _Placeholder(f"{self._name}()")

    def __mro_entries__(self, bases):                                      # used as a base class
        return (object,)

    def __iter__(self):
        return iter(())

    def __getitem__(self, item):
        return _Placeholder(f"{self._name}[]")

    def __or__(self, other):
        return self

    __ror__ = __or__


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Placeholder(f"{self.__name__}.{item}")


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if any(fullname == r or fullname.startswith(r + ".") for r in STUB_ROOTS):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        _populate(module)


# ---- the helpers the inference path executes, restated (detectron2 v0.6 / fvcore) ----------------------------------------------------
def configurable(init_func=None, *, from_config=None):
    """detectron2.config.configurable: with explicit keyword arguments the wrapped __init__ / function runs unchanged."""
    if init_func is not None:
        return init_func
    return lambda f: f


class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class Conv2d(nn.Conv2d):
    """detectron2.layers.Conv2d: nn.Conv2d with optional `norm` and `activation` applied after the convolution."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    if norm == "GN":
        return nn.GroupNorm(32, out_channels)
    raise NotImplementedError(f"get_norm({norm!r})")


class _Registry:
    def register(self, obj=None):
        return obj if obj is not None else (lambda o: o)

    def get(self, name):
        raise NotImplementedError("registry lookup")


def c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def c2_msra_fill(module):
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def retry_if_cuda_oom(func):
    return func


def sem_seg_postprocess(result, img_size, output_height, output_width):
    """detectron2.modeling.postprocessing.sem_seg_postprocess."""
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


class ImageList:
    """detectron2.structures.ImageList (from_tensors with size_divisibility, zero padding at the bottom / right)."""

    def __init__(self, tensor, image_sizes):
        self.tensor, self.image_sizes = tensor, image_sizes

    def __len__(self):
        return len(self.image_sizes)

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        sizes = [(t.shape[-2], t.shape[-1]) for t in tensors]
        mh, mw = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 1:
            mh = (mh + size_divisibility - 1) // size_divisibility * size_divisibility
            mw = (mw + size_divisibility - 1) // size_divisibility * size_divisibility
        out = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (mh, mw), pad_value)
        for i, t in enumerate(tensors):
            out[i, ..., : t.shape[-2], : t.shape[-1]].copy_(t)
        return ImageList(out.contiguous(), sizes)


class Boxes:
    def __init__(self, tensor):
        self.tensor = tensor


class Instances:
    def __init__(self, image_size, **kwargs):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self._fields[name] = val

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(name)
        return self._fields[name]


class BitMasks:
    def __init__(self, tensor):
        self.tensor = tensor.to(torch.bool)

    def get_bounding_boxes(self):
        boxes = torch.zeros(self.tensor.shape[0], 4, dtype=torch.float32)
        x_any, y_any = torch.any(self.tensor, dim=1), torch.any(self.tensor, dim=2)
        for idx in range(self.tensor.shape[0]):
            x, y = torch.where(x_any[idx, :])[0], torch.where(y_any[idx, :])[0]
            if len(x) > 0 and len(y) > 0:
                boxes[idx, :] = torch.as_tensor([x[0], y[0], x[-1] + 1, y[-1] + 1], dtype=torch.float32)
        return Boxes(boxes)


class Backbone(nn.Module):
    """detectron2.modeling.backbone.Backbone: an nn.Module with shape bookkeeping (nothing numeric)."""

    @property
    def size_divisibility(self):
        return 0

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name]) for name in self._out_features}


class BottleneckBlock(nn.Module):
    """detectron2.modeling.backbone.resnet.BottleneckBlock (v0.6): 1x1 -> 3x3 -> 1x1 convolutions, each bias-free with its norm, ReLU after
    the first two, projection shortcut iff in != out, ReLU(out + shortcut)."""

    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="BN", stride_in_1x1=False, dilation=1):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False, norm=get_norm(norm, out_channels))
        else:
            self.shortcut = None
        stride_1x1, stride_3x3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=stride_1x1, bias=False, norm=get_norm(norm, bottleneck_channels))
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=stride_3x3, padding=1 * dilation, bias=False,
                            groups=num_groups, dilation=dilation, norm=get_norm(norm, bottleneck_channels))
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False, norm=get_norm(norm, out_channels))
        for layer in [self.conv1, self.conv2, self.conv3, self.shortcut]:
            if layer is not None:
                c2_msra_fill(layer)

    def forward(self, x):
        out = F.relu_(self.conv1(x))
        out = F.relu_(self.conv2(out))
        out = self.conv3(out)
        shortcut = self.shortcut(x) if self.shortcut is not None else x
        out += shortcut
        return F.relu_(out)


class ResNet:
    @staticmethod
    def make_stage(block_class, num_blocks, *, in_channels, out_channels, **kwargs):
        blocks = []
        for i in range(num_blocks):
            curr = {}
            for k, v in kwargs.items():
                if k.endswith("_per_block"):
                    curr[k[: -len("_per_block")]] = v[i]
                else:
                    curr[k] = v
            blocks.append(block_class(in_channels=in_channels, out_channels=out_channels, **curr))
            in_channels = out_channels
        return blocks


class InterpolationMode:
    BICUBIC = "bicubic"
    BILINEAR = "bilinear"


class Resize(nn.Module):
    """torchvision.transforms.Resize on a float tensor (0.14: F.interpolate, align_corners=False, no antialias); `size` = (h, w)."""

    def __init__(self, size, interpolation="bilinear", max_size=None, antialias=None):
        super().__init__()
        if not isinstance(size, (tuple, list)) or len(size) != 2 or max_size is not None:
            raise NotImplementedError("Resize: only an explicit (h, w)")
        self.size, self.interpolation = tuple(size), interpolation

    def forward(self, img):
        if tuple(img.shape[-2:]) == self.size:
            return img
        return F.interpolate(img, size=self.size, mode=self.interpolation, align_corners=False)


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """ldm.modules.diffusionmodules.util.timestep_embedding (sinusoidal, cos first)."""
    import math
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(device=timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    embedding = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        embedding = torch.cat([embedding, torch.zeros_like(embedding[:, :1])], dim=-1)
    return embedding


class DiagonalGaussianDistribution:
    """ldm.modules.distributions.distributions.DiagonalGaussianDistribution: only `.mean` is used (deterministic encoding)."""

    def __init__(self, parameters):
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)


_REAL = {
    "detectron2.config": dict(configurable=configurable),
    "detectron2.layers": dict(Conv2d=Conv2d, ShapeSpec=ShapeSpec, get_norm=get_norm),
    "detectron2.modeling": dict(SEM_SEG_HEADS_REGISTRY=_Registry(), META_ARCH_REGISTRY=_Registry(), BACKBONE_REGISTRY=_Registry()),
    "detectron2.modeling.backbone": dict(Backbone=Backbone),
    "detectron2.modeling.backbone.resnet": dict(BottleneckBlock=BottleneckBlock, ResNet=ResNet),
    "torchvision.transforms": dict(Resize=Resize, InterpolationMode=InterpolationMode),
    "ldm.modules.diffusionmodules.openaimodel": dict(timestep_embedding=timestep_embedding),
    "ldm.modules.distributions.distributions": dict(DiagonalGaussianDistribution=DiagonalGaussianDistribution),
    "timm.models.layers": dict(trunc_normal_=nn.init.trunc_normal_),
    "detectron2.modeling.postprocessing": dict(sem_seg_postprocess=sem_seg_postprocess),
    "detectron2.structures": dict(ImageList=ImageList, Boxes=Boxes, Instances=Instances, BitMasks=BitMasks),
    "detectron2.utils.memory": dict(retry_if_cuda_oom=retry_if_cuda_oom),
    "detectron2.utils.registry": dict(Registry=lambda name: _Registry()),
    "fvcore.nn.weight_init": dict(c2_xavier_fill=c2_xavier_fill, c2_msra_fill=c2_msra_fill),
    "fvcore.nn": dict(),
}


def _populate(module):
    for k, v in _REAL.get(module.__name__, {}).items():
        setattr(module, k, v)


def _shell(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def install():
    """Register the stub finder and the package shells.  Idempotent."""
    if any(isinstance(f, _StubFinder) for f in sys.meta_path):
        return
    sys.meta_path.insert(0, _StubFinder())
    for name, path in (("mask2former", os.path.join(M2F, "mask2former")),
                       ("mask2former.modeling", os.path.join(M2F, "mask2former", "modeling")),
                       ("mask2former.modeling.transformer_decoder", os.path.join(M2F, "mask2former", "modeling", "transformer_decoder")),
                       ("mask2former.modeling.pixel_decoder", os.path.join(M2F, "mask2former", "modeling", "pixel_decoder")),
                       ("mask2former.modeling.pixel_decoder.ops", os.path.join(M2F, "mask2former", "modeling", "pixel_decoder", "ops")),
                       ("mask2former.modeling.meta_arch", os.path.join(M2F, "mask2former", "modeling", "meta_arch")),
                       ("mask2former.modeling.backbone", os.path.join(M2F, "mask2former", "modeling", "backbone")),
                       ("odise", os.path.join(REFERENCE, "odise")),
                       ("odise.modeling", os.path.join(REFERENCE, "odise", "modeling")),
                       ("odise.modeling.meta_arch", os.path.join(REFERENCE, "odise", "modeling", "meta_arch")),
                       ("odise.modeling.backbone", os.path.join(REFERENCE, "odise", "modeling", "backbone")),
                       ("odise.modeling.wrapper", os.path.join(REFERENCE, "odise", "modeling", "wrapper")),
                       ("odise.data", os.path.join(REFERENCE, "odise", "data")),
                       ("odise.utils", os.path.join(REFERENCE, "odise", "utils")),
                       ("odise.checkpoint", os.path.join(REFERENCE, "odise", "checkpoint"))):
        if name not in sys.modules:
            _shell(name, path)


def seal():
    """From now on any call into a placeholder raises."""
    global _SEALED
    _SEALED = True
