"""`MultiScaleDeformableAttention` (imported as `MSDA` by ops/functions/ms_deform_attn_func.py:22-29 of the reference): the forward of the
reference's only native op on libodise_hip.so, same argument order as ms_deform_attn.h:25-44.  Inference only: no backward."""
import ctypes as C

import numpy as np
import torch

from odise_amd import dropin
from odise_amd._lib import F16, F32, check


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """value [B,S,M,D], spatial_shapes [L,2] int64, level_start_index [L] int64, sampling_loc [B,Lq,M,L,P,2], attn_weight [B,Lq,M,L,P]
    -> [B, Lq, M*D] (dtype and device of `value`; fp16 values are sampled natively, everything else in fp32)."""
    ctx = dropin.get_context()
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_loc.shape
    half = value.dtype == torch.float16
    vp, vk = dropin.to_device(value, np.float16 if half else np.float32)
    lp, lk = dropin.to_device(sampling_loc, np.float32)
    wp, wk = dropin.to_device(attn_weight, np.float32)
    ss = np.ascontiguousarray(spatial_shapes.detach().cpu().numpy(), np.int64)
    ls = np.ascontiguousarray(level_start_index.detach().cpu().numpy(), np.int64)
    op, fetch = dropin.new_output((B, Lq, M * D), value, np.float16 if half else np.float32)
    check(ctx.lib.odise_hip_ms_deform_attn_forward(ctx.h, vp, ss.ctypes.data_as(C.c_void_p), ls.ctypes.data_as(C.c_void_p), lp, wp, B, S, M, D, Lq, L, P,
                                                   int(im2col_step), F16 if half else F32, op), "ms_deform_attn_forward")
    return fetch().to(value.dtype)


def ms_deform_attn_backward(*args, **kwargs):
    raise RuntimeError("libodise_hip is an inference library: MultiScaleDeformableAttention has no backward (training is out of scope)")
