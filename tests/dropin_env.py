"""Test environment of the drop-in overlay (odise_amd/dropin): the reference checkout importable WITHOUT detectron2 / omegaconf.

`install()` puts the overlay directory in front of the reference's roots on sys.path, keeps tests/golden/ref_stubs.py's placeholders for
the third-party packages that are absent here, and registers a restatement of the two detectron2 pieces the reference's LazyConfig model
files and `instantiate_odise` need (detectron2.config.LazyCall / instantiate over omegaconf-style nodes with relative `${..key}`
interpolation; MetadataCatalog.get).  CPU test infrastructure only - never imported by the product or on the GPU box."""
import importlib
import importlib.util
import os
import re
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_stubs  # noqa: E402

from odise_amd.dropin import OVERLAY_DIR  # noqa: E402

OVERLAID = ("odise", "odise.modeling", "odise.modeling.meta_arch", "odise.modeling.backbone", "mask2former", "mask2former.modeling",
            "mask2former.modeling.meta_arch", "mask2former.modeling.pixel_decoder")


class Node:
    """One LazyCall node: keyword arguments as attributes, `_target_` the callable, `_parent` for relative interpolation."""

    def __init__(self, target, kwargs):
        object.__setattr__(self, "_target_", target)
        object.__setattr__(self, "_kw", {})
        object.__setattr__(self, "_parent", None)
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __setattr__(self, key, value):
        if isinstance(value, Node):
            object.__setattr__(value, "_parent", self)
        self._kw[key] = value

    def __getattr__(self, key):
        try:
            return self._kw[key]
        except KeyError:
            raise AttributeError(key) from None


def LazyCall(target):
    return lambda **kwargs: Node(target, kwargs)


_INTERP = re.compile(r"^\$\{(\.+)([\w\.]+)\}$")


def _resolve(node, value):
    """omegaconf relative interpolation: `${.x}` = sibling in the same node, every further dot one level up."""
    while isinstance(value, str):
        m = _INTERP.match(value)
        if not m:
            break
        cur = node
        for _ in range(len(m.group(1)) - 1):
            cur = cur._parent
        for part in m.group(2).split("."):
            node_of_value = cur
            cur = cur._kw[part] if isinstance(cur, Node) else getattr(cur, part)
        node, value = node_of_value, cur
    return value


def instantiate(cfg):
    if isinstance(cfg, Node):
        kwargs = {k: instantiate(_resolve(cfg, v)) for k, v in cfg._kw.items()}
        return cfg._target_(**kwargs)
    if isinstance(cfg, (list, tuple)) and any(isinstance(v, Node) for v in cfg):
        return type(cfg)(instantiate(v) for v in cfg)
    return cfg


class _Metadata(types.SimpleNamespace):
    def get(self, key, default=None):
        return getattr(self, key, default)


class _MetadataCatalog:
    def get(self, name):
        if "coco" in name:   # COCO panoptic: contiguous ids 0..79 are things, 80..132 stuff
            return _Metadata(name=name, thing_dataset_id_to_contiguous_id={i: i for i in range(80)}, stuff_dataset_id_to_contiguous_id={i: i for i in range(80, 133)})
        return _Metadata(name=name, thing_dataset_id_to_contiguous_id={})


def install():
    ref_stubs.install()
    for root in (ref_stubs.REFERENCE, ref_stubs.M2F):
        if root not in sys.path:
            sys.path.append(root)
    if OVERLAY_DIR not in sys.path:
        sys.path.insert(0, OVERLAY_DIR)
    for name in sorted((m for m in sys.modules if m.split(".")[0] in ("odise", "mask2former", "MultiScaleDeformableAttention") and
                        (m in OVERLAID or m.startswith(("odise.modeling.meta_arch.", "odise.modeling.backbone.", "mask2former.modeling.meta_arch.",
                                                        "mask2former.modeling.pixel_decoder.msdeformattn")) or m == "MultiScaleDeformableAttention")), reverse=True):
        del sys.modules[name]
    for name in OVERLAID:
        importlib.import_module(name)
    # ref_stubs' finder answers for `MultiScaleDeformableAttention` (the reference's compiled op is absent here): load the overlay's module

    spec = importlib.util.spec_from_file_location("MultiScaleDeformableAttention", os.path.join(OVERLAY_DIR, "MultiScaleDeformableAttention.py"))
    msda = importlib.util.module_from_spec(spec)
    sys.modules["MultiScaleDeformableAttention"] = msda
    spec.loader.exec_module(msda)
    cfgmod = importlib.import_module("detectron2.config")
    cfgmod.LazyCall, cfgmod.instantiate = LazyCall, instantiate
    importlib.import_module("detectron2.data").MetadataCatalog = _MetadataCatalog()
    return sys.modules["odise"]
