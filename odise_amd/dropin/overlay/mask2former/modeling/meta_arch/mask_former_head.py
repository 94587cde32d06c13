"""`mask2former.modeling.meta_arch.mask_former_head.MaskFormerHead` on libodise_hip.so (reference:
third_party/Mask2Former/mask2former/modeling/meta_arch/mask_former_head.py:21-132).  The head owns the library-side build of the pixel
decoder + transformer predictor weights (`odise_hip_head_build`, keys `sem_seg_head.*`); `layers(features)` is one library call."""
import ctypes as C
import weakref
from typing import Dict

import numpy as np
import torch
from torch import nn

from odise_amd import dropin
from odise_amd._lib import check


class MaskFormerHead(nn.Module):
    def __init__(self, input_shape: Dict[str, object], *, num_classes: int, pixel_decoder: nn.Module, loss_weight: float = 1.0, ignore_value: int = -1,
                 transformer_predictor: nn.Module, transformer_in_feature: str):
        super().__init__()
        shapes = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, _ in shapes]
        self.ignore_value, self.common_stride, self.loss_weight = ignore_value, 4, loss_weight
        self.pixel_decoder, self.predictor = pixel_decoder, transformer_predictor
        self.transformer_in_feature, self.num_classes = transformer_in_feature, num_classes
        if transformer_in_feature != "multi_scale_pixel_decoder":
            raise NotImplementedError("libodise_hip implements transformer_in_feature='multi_scale_pixel_decoder'")
        object.__setattr__(pixel_decoder, "_head", weakref.ref(self))
        object.__setattr__(transformer_predictor, "_head", weakref.ref(self))
        self._built_version = None

    # ---- library side ----------------------------------------------------------------------------------------------------------------
    def library_state(self, prefix="sem_seg_head."):
        return {prefix + k: v.detach().cpu().numpy() for k, v in self.state_dict().items()}

    def _ensure_built(self):
        version = tuple(p._version for p in self.parameters())
        if self._built_version != version:
            from odise_amd.pipeline import load_state
            ctx = dropin.get_context()
            load_state(ctx, self.library_state())
            check(ctx.lib.odise_hip_head_build(ctx.h), "head_build")
            check(ctx.lib.odise_hip_clear_host_weights(ctx.h), "clear_host_weights")
            self._built_version = version
        return dropin.get_context()

    def _pixel_decoder(self, features):
        ctx = self._ensure_built()
        f = [features[k] for k in self.in_features]
        B, cin, H4, W4 = f[0].shape
        keep, ptrs = [], []
        for t in f:
            p, k = dropin.to_device(t)
            ptrs.append(p)
            keep.append(k)
        Cd = self.pixel_decoder.conv_dim
        mf_p, mf_get = dropin.new_output((B, Cd, H4, W4), f[0])
        outs = [dropin.new_output((B, Cd, H4 >> (3 - l), W4 >> (3 - l)), f[0]) for l in range(3)]
        check(ctx.lib.odise_hip_pixel_decoder_forward(ctx.h, (C.c_void_p * 4)(*ptrs), B, cin, H4, W4, mf_p, (C.c_void_p * 3)(*[o[0] for o in outs])),
              "pixel_decoder_forward")
        ms = [o[1]() for o in outs]
        return mf_get(), ms[0], ms

    def _predictor(self, x, mask_features):
        ctx = self._ensure_built()
        B, Cd, H4, W4 = mask_features.shape
        keep, ptrs, hw = [], [], []
        for t in x:
            p, k = dropin.to_device(t)
            ptrs.append(p)
            keep.append(k)
            hw += [int(t.shape[-2]), int(t.shape[-1])]
        mp, mk = dropin.to_device(mask_features)
        return self._collect(ctx, lambda pm, me, pool, ls: check(ctx.lib.odise_hip_predictor_forward(
            ctx.h, (C.c_void_p * 3)(*ptrs), (C.c_int * 6)(*hw), mp, B, H4, W4, pm, me, pool, ls), "predictor_forward"), B, H4, W4, mask_features)

    def _collect(self, ctx, call, B, H4, W4, like):
        Q, Cd = self.predictor.num_queries, self.predictor.hidden_dim
        pm = dropin.new_output((B, Q, H4, W4), like)
        me = dropin.new_output((B, Q, Cd), like)
        pool = dropin.new_output((B, Q, Cd), like)
        ls = C.c_float()
        call(pm[0], me[0], pool[0], C.byref(ls))
        pred_masks = pm[1]()
        logits = self.predictor.class_embed(me[1]()) if getattr(self.predictor, "class_embed", None) is not None else None
        out = {"pred_logits": logits, "pred_masks": pred_masks, "aux_outputs": [], "mask_embed": me[1](), "mask_pooled_features": pool[1](),
               "logit_scale": torch.tensor(float(ls.value), device=pred_masks.device)}
        return out

    def layers(self, features, mask=None):
        ctx = self._ensure_built()
        f = [features[k] for k in self.in_features]
        B, cin, H4, W4 = f[0].shape
        keep, ptrs = [], []
        for t in f:
            p, k = dropin.to_device(t)
            ptrs.append(p)
            keep.append(k)
        return self._collect(ctx, lambda pm, me, pool, ls: check(ctx.lib.odise_hip_head_forward(
            ctx.h, (C.c_void_p * 4)(*ptrs), B, cin, H4, W4, pm, me, pool, ls), "head_forward"), B, H4, W4, f[0])

    def forward(self, features, mask=None):
        return self.layers(features, mask)
