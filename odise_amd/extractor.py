"""Host wrapper of LdmImplicitCaptionerExtractor.forward (odise/modeling/meta_arch/ldm.py:697-718 -> 543-621) on the device.

`HipFeatureExtractor(ctx, state)` takes one flat state dict with the keys of the three real weight sources
(SD v1 ckpt: `first_stage_model.*`, `model.diffusion_model.*`; OpenAI CLIP archive prefixed `clip.`: `clip.visual.*`;
ODISE ckpt: `backbone.feature_extractor.*`) and returns, per call, the same list of 8 feature maps (fp32 NCHW) the
reference's `feature_extractor(dict(img=...))` returns: enc5, enc7, u2, u5, u8, u11, dec2, dec5 (ldm.py:608).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import numpy as np

from ._lib import check
from .runtime import Context, DeviceArray

_SKIP_PREFIXES = ("model.diffusion_model.out.", "model.diffusion_model.output_blocks.11.", "first_stage_model.decoder.up.0.",
                  "first_stage_model.decoder.up.1.", "first_stage_model.decoder.up.2.block.2.", "first_stage_model.decoder.up.2.upsample.",
                  "first_stage_model.decoder.norm_out.", "first_stage_model.decoder.conv_out.", "first_stage_model.loss.",
                  "cond_stage_model.", "model_ema.")
TAP_NAMES = ("enc5", "enc7", "u2", "u5", "u8", "u11", "dec2", "dec5")


class HipFeatureExtractor:
    def __init__(self, ctx: Context, state: Dict[str, "np.ndarray"]):
        self.ctx = ctx
        lib = ctx.lib
        n = 0
        for key, val in state.items():
            if key.startswith(_SKIP_PREFIXES):
                continue  # dead in ODISE's extractor (never executed, ldm.py:491/515-516/600/606): not uploaded
            if not key.startswith(("model.diffusion_model.", "first_stage_model.", "clip.visual.", "backbone.feature_extractor.")):
                continue
            if hasattr(val, "detach"):
                val = val.detach().cpu().numpy()
            arr = np.ascontiguousarray(val, dtype=np.float32)
            if arr.ndim > 4:
                continue
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            check(lib.odise_hip_load_weight(ctx.h, key.encode(), arr.ctypes.data_as(C.POINTER(C.c_float)), shape, arr.ndim),
                  f"load_weight({key})")
            n += 1
        check(lib.odise_hip_extractor_build(ctx.h), "extractor_build")
        check(lib.odise_hip_clear_host_weights(ctx.h), "clear_host_weights")
        self.num_tensors = n
        ctx.model_owner = self

    def run_nhwc(self, image: DeviceArray):
        """Hot-path call: image [B,3,H,W] f32 on the device; returns ([ptr]*8, [(n,c,h,w)]*8) of fp16 NHWC taps in the arena."""
        B, _, H, W = image.shape
        ptrs = (C.c_void_p * 8)()
        shapes = (C.c_int * 32)()
        check(self.ctx.lib.odise_hip_extractor_forward_nhwc(self.ctx.h, C.c_void_p(image.ptr), B, H, W, ptrs, shapes), "extractor_forward")
        return [ptrs[i] for i in range(8)], [tuple(shapes[4 * i:4 * i + 4]) for i in range(8)]

    def features_device(self, image: DeviceArray) -> List[DeviceArray]:
        ptrs, shapes = self.run_nhwc(image)
        outs = []
        for p, (n, c, h, w) in zip(ptrs, shapes):
            o = self.ctx.empty((n, c, h, w), np.float32)
            check(self.ctx.lib.odise_hip_nhwc_f16_to_nchw_f32(self.ctx.h, C.c_void_p(p), C.c_void_p(o.ptr), n, c, h, w), "nhwc_to_nchw")
            outs.append(o)
        return outs

    def features(self, image) -> List[np.ndarray]:
        img = self.ctx.to_device(np.asarray(image, np.float32))
        return [o.numpy() for o in self.features_device(img)]

    def last_macs(self) -> float:
        m = C.c_double()
        check(self.ctx.lib.odise_hip_extractor_last_macs(self.ctx.h, C.byref(m)), "extractor_last_macs")
        return float(m.value)
