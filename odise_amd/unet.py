"""Host wrapper of the SD-v1 UNet single-step tap extraction (LdmExtractor.unet_forward, ldm.py:469-491).

Weights come in as a state dict keyed like `model.diffusion_model.*` of an SD v1 checkpoint (prefix optional); they are
handed to libodise_hip.so once (`odise_hip_load_weight` + `odise_hip_unet_build`), which packs them to fp16 device
layouts.  `features()` returns the four taps (concat inputs of output blocks 2, 5, 8, 11) as fp32 NCHW arrays — the
same tensors, layout and dtype the reference's `unet_forward` appends to `ret_features`.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np

from ._lib import check
from .runtime import Context, DeviceArray

SD_PREFIX = "model.diffusion_model."
TAP_CHANNEL_MULT = (8, 6, 3, 2)   # u2=2560, u5=1920, u8=960, u11=640 channels at model_channels=320
TAP_DOWN = (8, 4, 2, 1)


class HipUNet:
    def __init__(self, ctx: Context, state_dict: Dict[str, "np.ndarray"], use_graph: bool = False):
        self.ctx = ctx
        lib = ctx.lib
        n = 0
        for key, val in state_dict.items():
            if key.startswith(SD_PREFIX):
                key = key[len(SD_PREFIX):]
            if key.startswith("out.") or key.startswith("output_blocks.11."):
                continue  # dead in ODISE's extractor (ldm.py:491, 600): never uploaded
            if hasattr(val, "detach"):
                val = val.detach().cpu().numpy()
            arr = np.ascontiguousarray(val, dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            check(lib.odise_hip_load_weight(ctx.h, key.encode(), arr.ctypes.data_as(C.POINTER(C.c_float)), shape, arr.ndim),
                  f"load_weight({key})")
            n += 1
            if key == "time_embed.0.weight":
                self.model_channels = arr.shape[1]
                self.time_embed_dim = arr.shape[0]
        check(lib.odise_hip_unet_build(ctx.h), "unet_build")
        check(lib.odise_hip_clear_host_weights(ctx.h), "clear_host_weights")
        self.num_tensors = n
        ctx.model_owner = self
        if use_graph:
            self.use_graph(True)

    def use_graph(self, enable: bool):
        check(self.ctx.lib.odise_hip_unet_use_graph(self.ctx.h, int(enable)), "unet_use_graph")

    def tap_shapes(self, B: int, h: int, w: int):
        mc = self.model_channels
        return [(B, m * mc, h // d, w // d) for m, d in zip(TAP_CHANNEL_MULT, TAP_DOWN)]

    def features_device(self, x_t: DeviceArray, context: DeviceArray, cond_emb: Optional[DeviceArray], t: int = 0,
                        outs: Optional[List[DeviceArray]] = None) -> List[DeviceArray]:
        """Device-resident call: x_t [B,4,h,w] f32, context [B,77,cdim] f32, cond_emb [B,ted] f32 -> 4 fp32 NCHW taps."""
        B, _, h, w = x_t.shape
        if outs is None:
            outs = [self.ctx.empty(s, np.float32) for s in self.tap_shapes(B, h, w)]
        ce = C.c_void_p(cond_emb.ptr) if cond_emb is not None else C.c_void_p(None)
        check(self.ctx.lib.odise_hip_unet_features(self.ctx.h, C.c_void_p(x_t.ptr), C.c_void_p(context.ptr), ce, B, h, w, int(t),
                                                    *[C.c_void_p(o.ptr) for o in outs]), "unet_features")
        return outs

    def run_nhwc(self, x_t: DeviceArray, context: DeviceArray, cond_emb: Optional[DeviceArray], t: int = 0):
        """Hot-path call used by bench.py: taps stay fp16 NHWC inside the library's arena (no layout conversion)."""
        B, _, h, w = x_t.shape
        taps = (C.c_void_p * 4)()
        ce = C.c_void_p(cond_emb.ptr) if cond_emb is not None else C.c_void_p(None)
        check(self.ctx.lib.odise_hip_unet_features_nhwc(self.ctx.h, C.c_void_p(x_t.ptr), C.c_void_p(context.ptr), ce, B, h, w, int(t),
                                                         taps), "unet_features_nhwc")
        return [taps[i] for i in range(4)]

    def features(self, x_t, context, cond_emb=None, t: int = 0) -> List[np.ndarray]:
        d = self.ctx.to_device
        outs = self.features_device(d(np.asarray(x_t, np.float32)), d(np.asarray(context, np.float32)),
                                    d(np.asarray(cond_emb, np.float32)) if cond_emb is not None else None, t)
        return [o.numpy() for o in outs]

    def last_macs(self) -> float:
        m = C.c_double()
        check(self.ctx.lib.odise_hip_unet_last_macs(self.ctx.h, C.byref(m)), "unet_last_macs")
        return float(m.value)
