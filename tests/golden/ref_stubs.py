"""Import the REFERENCE's own Python modules in the build container (test-fixture generation only; never on the GPU box).

ODISE's inference code (odise/modeling/meta_arch/odise.py) and the vendored Mask2Former modules
(third_party/Mask2Former/mask2former/...) are pure PyTorch, but they import detectron2, fvcore, open_clip, ldm ... which are not
installed here.  This module makes those imports succeed WITHOUT running any third-party arithmetic that is not written out below:

  * the package `__init__`s of `odise` and `mask2former` (which pull in datasets, evaluators, trainers) are bypassed by registering
    empty package shells whose `__path__` points at the real directories, so sub-modules load from the reference's files as they are;
  * the handful of detectron2 / fvcore helpers the inference path really executes are restated here (detectron2 v0.6 semantics):
    `configurable` (explicit-kwargs construction only), `Conv2d` (conv -> norm -> activation), `get_norm("GN")`, `ShapeSpec`,
    `ImageList.from_tensors`, `sem_seg_postprocess`, `retry_if_cuda_oom`, `Boxes` / `Instances` / `BitMasks.get_bounding_boxes`,
    the registries' `register()` decorator, `c2_xavier_fill` / `c2_msra_fill`;
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


_REAL = {
    "detectron2.config": dict(configurable=configurable),
    "detectron2.layers": dict(Conv2d=Conv2d, ShapeSpec=ShapeSpec, get_norm=get_norm),
    "detectron2.modeling": dict(SEM_SEG_HEADS_REGISTRY=_Registry(), META_ARCH_REGISTRY=_Registry(), BACKBONE_REGISTRY=_Registry()),
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
