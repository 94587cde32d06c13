"""Host wrapper of the ODISE mask generator on the device: FeatureExtractorBackbone (feature_extractor.py:139-250) and
MaskFormerHead = MSDeformAttn pixel decoder + ODISE masked transformer decoder (msdeformattn.py:314-358, odise.py:642-776).

`HipODISE(ctx, state)` takes one flat state dict keyed like the real checkpoints (see odise_amd/extractor.py plus the ODISE
checkpoint keys `backbone.feature_projections.*`, `sem_seg_head.*`).  The stage methods return the same dict keys / shapes /
dtypes (fp32) as the reference modules so that each stage can be compared in isolation.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from ._lib import check
from .extractor import _SKIP_PREFIXES
from .runtime import Context, DeviceArray

_PREFIXES = ("model.diffusion_model.", "first_stage_model.", "clip.visual.", "backbone.feature_extractor.",
             "backbone.feature_projections.", "sem_seg_head.")


def load_state(ctx: Context, state: Dict[str, "np.ndarray"]) -> int:
    n = 0
    for key, val in state.items():
        if key.startswith(_SKIP_PREFIXES) or not key.startswith(_PREFIXES):
            continue
        if hasattr(val, "detach"):
            val = val.detach().cpu().numpy()
        arr = np.ascontiguousarray(val, dtype=np.float32)
        if arr.ndim > 4:
            continue
        shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
        check(ctx.lib.odise_hip_load_weight(ctx.h, key.encode(), arr.ctypes.data_as(C.POINTER(C.c_float)), shape, arr.ndim),
              f"load_weight({key})")
        n += 1
    return n


class HipODISE:
    def __init__(self, ctx: Context, state: Dict[str, "np.ndarray"], with_extractor: bool = True, with_head: bool = True):
        self.ctx = ctx
        lib = ctx.lib
        self.num_tensors = load_state(ctx, state)
        self.has_backbone = with_extractor
        if with_extractor:
            check(lib.odise_hip_extractor_build(ctx.h), "extractor_build")
            check(lib.odise_hip_backbone_build(ctx.h), "backbone_build")
        if with_head:
            check(lib.odise_hip_head_build(ctx.h), "head_build")
        check(lib.odise_hip_clear_host_weights(ctx.h), "clear_host_weights")
        q, c = C.c_int(), C.c_int()
        check(lib.odise_hip_maskgen_info(ctx.h, C.byref(q), C.byref(c), None), "maskgen_info")
        self.num_queries, self.hidden_dim = q.value, c.value

    # ---- FeatureExtractorBackbone.forward -----------------------------------------------------------------------------------
    def backbone_device(self, image: DeviceArray, want_outputs: bool = True):
        B, _, H, W = image.shape
        outs = [self.ctx.empty((B, 512, H // s, W // s), np.float32) for s in (4, 8, 16, 32)] if want_outputs else None
        arr = (C.c_void_p * 4)(*[o.ptr for o in outs]) if outs else None
        check(self.ctx.lib.odise_hip_backbone_forward(self.ctx.h, C.c_void_p(image.ptr), B, H, W, arr), "backbone_forward")
        return outs

    def backbone(self, image) -> Dict[str, np.ndarray]:
        outs = self.backbone_device(self.ctx.to_device(np.asarray(image, np.float32)))
        return {k: o.numpy() for k, o in zip(("s2", "s3", "s4", "s5"), outs)}

    # ---- MaskFormerHead.layers ------------------------------------------------------------------------------------------------
    def head_device(self, feats: Optional[list], B: int, H4: int, W4: int, cin: int = 512):
        Q, Cd = self.num_queries, self.hidden_dim
        pm = self.ctx.empty((B, Q, H4, W4), np.float32)
        me = self.ctx.empty((B, Q, Cd), np.float32)
        mp = self.ctx.empty((B, Q, Cd), np.float32)
        ls = C.c_float()
        arr = (C.c_void_p * 4)(*[f.ptr for f in feats]) if feats is not None else None
        check(self.ctx.lib.odise_hip_head_forward(self.ctx.h, arr, B, cin, H4, W4, C.c_void_p(pm.ptr), C.c_void_p(me.ptr), C.c_void_p(mp.ptr),
                                                   C.byref(ls)), "head_forward")
        return pm, me, mp, float(ls.value)

    def head(self, features: Optional[Dict[str, np.ndarray]] = None, image_hw=None) -> Dict[str, np.ndarray]:
        """features: dict s2..s5 fp32 NCHW (host) or None to reuse the maps of the last `backbone` call (pass image_hw then)."""
        if features is not None:
            f = [self.ctx.to_device(np.asarray(features[k], np.float32)) for k in ("s2", "s3", "s4", "s5")]
            B, cin, H4, W4 = f[0].shape
        else:
            f, cin = None, 512
            B, H, W = image_hw
            H4, W4 = H // 4, W // 4
        pm, me, mp, ls = self.head_device(f, B, H4, W4, cin)
        return {"pred_masks": pm.numpy(), "mask_embed": me.numpy(), "mask_pooled_features": mp.numpy(), "logit_scale": ls}
