"""Oracle: input resize and evaluator reductions of the eval loop (TEST INFRASTRUCTURE - see oracle/__init__.py).

  * `resize_shortest_edge_shape` + `pil_resize_bilinear_u8`: detectron2 `T.ResizeShortestEdge(1024, max 2560)` -> `ResizeTransform`
    -> `PIL.Image.resize((w, h), BILINEAR)` on uint8 RGB (configs/common/data/pano_open_d2_eval.py:74-107).  detectron2 and Pillow are
    third-party; the resampler (Pillow `src/libImaging/Resample.c`: `precompute_coeffs`, `normalize_coeffs_8bpc`,
    `ImagingResampleHorizontal_8bpc` / `Vertical_8bpc`, horizontal pass first, uint8 intermediate) is restated bit-exactly and
    PINNED against the Pillow installed here (tests/test_oracle_eval_ops.py).
  * `semantic_confusion`: detectron2 `SemSegEvaluator.process` (odise/evaluation/d2_evaluator.py:63 subclass): argmax over classes,
    ignore label -> K, `bincount((K+1) * pred + gt)` reshaped [(K+1), (K+1)] (rows = prediction).
  * `pair_histogram`: the per-pixel part of panopticapi `pq_compute_single_core` (COCOPanopticEvaluator, d2_evaluator.py:49):
    `np.unique(gt * OFFSET + pred, return_counts=True)` as a dense [n_gt, n_pred] count matrix of segment indices.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resize_shortest_edge_shape(h: int, w: int, short: int = 1024, max_size: int = 2560):
    """detectron2 ResizeShortestEdge.get_output_shape."""
    scale = short * 1.0 / min(h, w)
    newh, neww = (short, scale * w) if h < w else (scale * h, short)
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


def precompute_coeffs(in_size: int, out_size: int):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the whole input."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = [0.0] * ksize
        tot = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            tot += w[x]                                # sequential double accumulation, as in the C source
        if tot != 0.0:
            for x in range(xmax):
                w[x] /= tot
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            kk[xx, x] = int(0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] >= 0 else int(-0.5 + w[x] * (1 << PRECISION_BITS))
    return bounds, kk


def _pass(img: np.ndarray, bounds, kk, axis: int) -> np.ndarray:
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((len(bounds),) + src.shape[1:], np.uint8)
    for i, (xmin, xmax) in enumerate(bounds):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc += src[xmin + x] * int(kk[i, x])
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img uint8 [H, W, C] -> uint8 [out_h, out_w, C], bit-identical to PIL.Image.resize((out_w, out_h), BILINEAR)."""
    h, w = img.shape[:2]
    out = img
    if out_w != w:
        out = _pass(out, *precompute_coeffs(w, out_w), axis=1)
    if out_h != h:
        out = _pass(out, *precompute_coeffs(h, out_h), axis=0)
    return out


def semantic_confusion(sem_seg: np.ndarray, gt: np.ndarray, ignore_label: int = 255) -> np.ndarray:
    """sem_seg [K, H, W] float, gt [H, W] int -> int64 [(K+1), (K+1)] (rows = predicted class)."""
    K = sem_seg.shape[0]
    pred = sem_seg.argmax(0).astype(np.int64)
    g = gt.astype(np.int64).copy()
    g[g == ignore_label] = K
    return np.bincount((K + 1) * pred.reshape(-1) + g.reshape(-1), minlength=(K + 1) ** 2).reshape(K + 1, K + 1)


def pair_histogram(a: np.ndarray, b: np.ndarray, na: int, nb: int) -> np.ndarray:
    """a, b int index maps of equal shape (values in [0, na) / [0, nb)) -> int64 [na, nb] co-occurrence counts."""
    return np.bincount(a.reshape(-1).astype(np.int64) * nb + b.reshape(-1).astype(np.int64), minlength=na * nb).reshape(na, nb)
