"""Oracle: multi-scale deformable attention forward (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates, in two independent ways, what MSDA.ms_deform_attn_forward computes
(third_party/Mask2Former/mask2former/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:242-303,
bilinear helper :38-89) and what the reference's own CPU fallback computes
(.../ops/functions/ms_deform_attn_func.py:52-72):

  * `msda_forward_loops`  — scalar numpy/python loops following the kernel (small cases only);
  * `msda_forward_torch`  — vectorised torch gather formulation (production shapes, seconds on CPU).

Pinned against the reference's `ms_deform_attn_core_pytorch` by tests/golden/make_golden.py.
"""
from __future__ import annotations

import numpy as np
import torch


def msda_forward_loops(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    """value [B,S,M,D], spatial_shapes [L,2] (H,W), level_start_index [L], sampling_loc [B,Lq,M,L,P,2] (x,y in [0,1]),
    attn_weight [B,Lq,M,L,P] -> [B,Lq,M*D].  float64 accumulation of the float inputs, kernel op order."""
    value = np.asarray(value, dtype=np.float64)
    loc = np.asarray(sampling_loc, dtype=np.float64)
    w = np.asarray(attn_weight, dtype=np.float64)
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.zeros((B, Lq, M, D), dtype=np.float64)
    for b in range(B):
        for q in range(Lq):
            for m in range(M):
                for l in range(L):
                    H, W = int(spatial_shapes[l][0]), int(spatial_shapes[l][1])
                    start = int(level_start_index[l])
                    for p in range(P):
                        # ms_deform_im2col_cuda.cuh:281-289: h_im = loc_h*H - 0.5 ; inside test is open (-1,H)x(-1,W)
                        h_im = loc[b, q, m, l, p, 1] * H - 0.5
                        w_im = loc[b, q, m, l, p, 0] * W - 0.5
                        if not (h_im > -1 and w_im > -1 and h_im < H and w_im < W):
                            continue
                        h_low, w_low = int(np.floor(h_im)), int(np.floor(w_im))
                        lh, lw = h_im - h_low, w_im - w_low
                        hh, hw = 1 - lh, 1 - lw
                        val = np.zeros(D)
                        # :38-89 — the four corners, zero outside the map
                        for (yy, xx, ww) in ((h_low, w_low, hh * hw), (h_low, w_low + 1, hh * lw),
                                             (h_low + 1, w_low, lh * hw), (h_low + 1, w_low + 1, lh * lw)):
                            if 0 <= yy <= H - 1 and 0 <= xx <= W - 1:
                                val += ww * value[b, start + yy * W + xx, m]
                        out[b, q, m] += w[b, q, m, l, p] * val
    return out.reshape(B, Lq, M * D)


def msda_forward_torch(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    """Vectorised restatement (explicit corner gathers; does NOT call grid_sample, so it is independent of the
    reference's ms_deform_attn_core_pytorch).  Computes in the dtype of `value` (use .double() for a tight oracle)."""
    value = torch.as_tensor(value)
    loc = torch.as_tensor(sampling_loc).to(value.dtype)
    w = torch.as_tensor(attn_weight).to(value.dtype)
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = torch.zeros(B, Lq, M, D, dtype=value.dtype)
    bidx = torch.arange(B).view(B, 1, 1, 1).expand(B, Lq, M, P)
    midx = torch.arange(M).view(1, 1, M, 1).expand(B, Lq, M, P)
    for l in range(L):
        H, W = int(spatial_shapes[l][0]), int(spatial_shapes[l][1])
        start = int(level_start_index[l])
        h_im = loc[:, :, :, l, :, 1] * H - 0.5  # [B,Lq,M,P]
        w_im = loc[:, :, :, l, :, 0] * W - 0.5
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        h_low, w_low = torch.floor(h_im), torch.floor(w_im)
        lh, lw = h_im - h_low, w_im - w_low
        hh, hw = 1 - lh, 1 - lw
        h_low, w_low = h_low.long(), w_low.long()
        val = torch.zeros(B, Lq, M, P, D, dtype=value.dtype)
        for dy, dx, ww in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
            yy, xx = h_low + dy, w_low + dx
            ok = inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
            idx = (start + yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1))
            v = value[bidx, idx, midx]  # [B,Lq,M,P,D]
            val = val + (ww * ok.to(value.dtype)).unsqueeze(-1) * v
        out = out + (w[:, :, :, l, :].unsqueeze(-1) * val).sum(dim=3)
    return out.reshape(B, Lq, M * D)


def make_inputs(B, M, D, Lq, shapes, P, seed, loc_range=(0.0, 1.0), value_scale=0.01):
    """Seeded inputs in the style of the reference's ops/test.py:24-39 (rand value*0.01, rand loc, normalised weights)."""
    g = torch.Generator().manual_seed(seed)
    shapes = [(int(h), int(w)) for h, w in shapes]
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    starts = [0]
    for h, w in shapes[:-1]:
        starts.append(starts[-1] + h * w)
    value = torch.rand(B, S, M, D, generator=g) * value_scale
    lo, hi = loc_range
    loc = torch.rand(B, Lq, M, L, P, 2, generator=g) * (hi - lo) + lo
    w = torch.rand(B, Lq, M, L, P, generator=g) + 1e-5
    w = w / w.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value, torch.tensor(shapes, dtype=torch.long), torch.tensor(starts, dtype=torch.long), loc, w
