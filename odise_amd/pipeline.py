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
        ctx.model_owner = self

    def reload_head(self, state: Dict[str, "np.ndarray"]) -> None:
        """(Re)load `sem_seg_head.*` (pixel decoder + masked transformer decoder) and rebuild that stage; the feature extractor and the
        tap projections stay as they are.  What loading another ODISE head checkpoint over the same frozen towers amounts to."""
        load_state(self.ctx, {k: v for k, v in state.items() if k.startswith("sem_seg_head.")})
        check(self.ctx.lib.odise_hip_head_build(self.ctx.h), "head_build")
        check(self.ctx.lib.odise_hip_clear_host_weights(self.ctx.h), "clear_host_weights")

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

    def backbone_maps(self) -> Dict[str, np.ndarray]:
        """The s2..s5 maps of the LAST backbone pass (`backbone_device` or a whole `infer_device` call) as the device holds them, fp32 NCHW on
        the host.  Test / attribution hook: valid until the next call that resets the activation arena."""
        shp = (C.c_int * 16)()
        none = (C.c_void_p * 4)()
        check(self.ctx.lib.odise_hip_backbone_maps(self.ctx.h, none, shp), "backbone_maps")
        outs = [self.ctx.empty(tuple(shp[4 * i:4 * i + 4]), np.float32) for i in range(4)]
        arr = (C.c_void_p * 4)(*[o.ptr for o in outs])
        check(self.ctx.lib.odise_hip_backbone_maps(self.ctx.h, arr, None), "backbone_maps")
        res = {k: o.numpy() for k, o in zip(("s2", "s3", "s4", "s5"), outs)}
        for o in outs:
            o.free()
        return res

    # ---- MaskFormerHead.layers ------------------------------------------------------------------------------------------------
    def head_device(self, feats: Optional[list], B: int, H4: int, W4: int, cin: int = 512, want_outputs: bool = True):
        Q, Cd = self.num_queries, self.hidden_dim
        pm = self.ctx.empty((B, Q, H4, W4), np.float32) if want_outputs else None
        me = self.ctx.empty((B, Q, Cd), np.float32) if want_outputs else None
        mp = self.ctx.empty((B, Q, Cd), np.float32) if want_outputs else None
        ls = C.c_float()
        arr = (C.c_void_p * 4)(*[f.ptr for f in feats]) if feats is not None else None
        p = lambda a: C.c_void_p(a.ptr) if a is not None else None
        check(self.ctx.lib.odise_hip_head_forward(self.ctx.h, arr, B, cin, H4, W4, p(pm), p(me), p(mp), C.byref(ls)), "head_forward")
        return pm, me, mp, float(ls.value)

    def mask_features_device(self, feats: list, B: int, H4: int, W4: int, cin: int = 512) -> DeviceArray:
        """MSDeformAttnPixelDecoder.forward_features on its own (msdeformattn.py:314-358): `mask_features` [B,C,H4,W4] fp32 of the s2..s5
        maps `feats` (fp32 NCHW DeviceArrays).  Resets the activation arena: the maps of an earlier `backbone_device` call held inside the
        library are gone afterwards, pass `feats` to `head_device` explicitly."""
        mf = self.ctx.empty((B, self.hidden_dim, H4, W4), np.float32)
        arr = (C.c_void_p * 4)(*[f.ptr for f in feats])
        check(self.ctx.lib.odise_hip_pixel_decoder_forward(self.ctx.h, arr, B, cin, H4, W4, C.c_void_p(mf.ptr), None), "pixel_decoder_forward")
        return mf

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


# ---------------------------------------------------------------------------------------------------------------------
# Open-vocabulary classification + post-processing (CategoryODISE.forward eval branch, odise.py:282-372)
# ---------------------------------------------------------------------------------------------------------------------
class HipCategoryODISE(HipODISE):
    """The whole CategoryODISE eval forward on the device.  `forward(batched_inputs)` takes the reference's input format
    (list of {"image": uint8/float CHW, "height", "width"}; images of one call may differ in size, odise.py:238-244) and returns the
    reference's output format (list of dicts with "sem_seg" [K,h,w] fp32, "panoptic_seg" (int32 [h,w], segments_info), "instances"
    {pred_masks, scores, pred_classes}).  One library call per batch (`odise_hip_infer`); every decision of the three heads is taken on
    the device, the host reads back the segment / instance tables (a few hundred bytes per image) once at the end."""

    HEAD_KEYS = ("category_head.text_proj.weight", "category_head.text_proj.bias", "category_head.null_embed")

    def __init__(self, ctx: Context, state, semantic_on=True, panoptic_on=True, instance_on=True, object_mask_threshold=0.0,
                 overlap_threshold=0.8, test_topk_per_image=100, size_divisibility=64):
        super().__init__(ctx, {k: v for k, v in state.items()})
        self.load_category_head(state)
        self.semantic_on, self.panoptic_on, self.instance_on = semantic_on, panoptic_on, instance_on
        self.semantic_argmax = False          # True: "sem_seg_argmax" int32 [h,w] instead of "sem_seg" [K,h,w] (never materialised)
        self.object_mask_threshold, self.overlap_threshold = object_mask_threshold, overlap_threshold
        self.test_topk_per_image, self.size_divisibility = test_topk_per_image, size_divisibility
        assert size_divisibility == 64, "the feature extractor fixes size_divisibility at 64 (feature_extractor.py:126-128)"
        self.num_classes = 0
        self.thing_ids = set()
        self.metadata, self.test_labels = None, None
        self._alpha, self._beta = 0.3, 0.7
        self._banks, self._vocab_cache = None, {}
        self._pool = {}

    def load_category_head(self, state) -> None:
        """(Re)load the classification head's own weights - `category_head.text_proj.*`, `category_head.null_embed` (CaptionODISE:
        `word_head.text_proj.*`) - and rebuild the stage; the mask generator and the frozen towers are untouched.  The device text banks
        are derived from these weights, so the active vocabulary has to be set again afterwards."""
        ctx = self.ctx
        for key in self.HEAD_KEYS:
            val = state[key]
            if hasattr(val, "detach"):
                val = val.detach().cpu().numpy()
            arr = np.ascontiguousarray(val, dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            check(ctx.lib.odise_hip_load_weight(ctx.h, key.encode(), arr.ctypes.data_as(C.POINTER(C.c_float)), shape, arr.ndim), key)
        check(ctx.lib.odise_hip_classify_build(ctx.h), "classify_build")
        check(ctx.lib.odise_hip_clear_host_weights(ctx.h), "clear_host_weights")
        self.num_classes, self._banks, self.test_labels = 0, None, None

    def set_vocabulary(self, cat_text, clip_text, group_sizes, overlap, thing_ids, alpha=0.3, beta=0.7):
        """cat_text / clip_text: [K_tot, dim] CLIP text embeddings of the category_head / clip_head prompt sets."""
        cat = np.ascontiguousarray(cat_text, np.float32)
        clp = np.ascontiguousarray(clip_text, np.float32)
        gs = np.ascontiguousarray(group_sizes, np.int32)
        ov = np.ascontiguousarray(overlap, np.int32)
        check(self.ctx.lib.odise_hip_set_vocabulary(self.ctx.h, cat.ctypes.data_as(C.c_void_p), clp.ctypes.data_as(C.c_void_p), cat.shape[0],
                                                     cat.shape[1], gs.ctypes.data_as(C.c_void_p), ov.ctypes.data_as(C.c_void_p), len(gs),
                                                     C.c_float(alpha), C.c_float(beta)), "set_vocabulary")
        self.num_classes = len(gs)
        self.thing_ids = set(int(t) for t in thing_ids)
        self._alpha, self._beta = float(alpha), float(beta)
        self._banks, self.test_labels = (cat, clp, gs, ov), None           # banks handed over directly: no label strings known

    # ---- open-vocabulary state (OpenPanopticInference's protocol, odise/modeling/wrapper/pano_wrapper.py:20-70) ------------------
    def attach_text(self, tokenizer, text_encoder, train_labels=None, clip_text_encoder=None):
        """Give the model what `load_open_state_dict` needs to turn label lists into text banks on the device
        (odise_amd.tokenizer.SimpleTokenizer, odise_amd.text.HipTextEncoder).  `train_labels` decide the seen / unseen ensemble weights
        (PoolingCLIPHead, odise.py:1446-1447, 1479-1491); None = the reference's default, COCO panoptic with prompt engineering, read
        from the label files (see odise_amd.checkpoint.default_train_labels)."""
        self._tokenizer, self._text_encoder, self._clip_text_encoder = tokenizer, text_encoder, clip_text_encoder
        self._train_labels = train_labels
        self._vocab_cache = {}

    def set_labels(self, labels, thing_ids=None, metadata=None):
        """Vocabulary from label strings (needs `attach_text`): what building the reference model with `labels=` / `metadata=` does."""
        st = {"category_head.test_labels": labels}
        if metadata is not None:
            st["metadata"] = metadata
        if thing_ids is not None:
            st["thing_ids"] = thing_ids
        self.load_open_state_dict(st)

    def open_state_dict(self) -> dict:
        """The inference-time switches the reference collects from its module tree (odise.py:1249-1262, 1440-1466,
        maskformer_model.py test-time attributes): a flat dict with the reference's key suffixes."""
        return {"category_head.test_labels": self.test_labels, "clip_head.test_labels": self.test_labels, "category_head.text_banks": self._banks,
                "metadata": self.metadata, "thing_ids": set(self.thing_ids), "sem_seg_head.num_classes": self.num_classes, "semantic_on": self.semantic_on,
                "instance_on": self.instance_on, "panoptic_on": self.panoptic_on, "test_topk_per_image": self.test_topk_per_image}

    def load_open_state_dict(self, state: dict) -> None:
        labels, banks = None, None
        for k, v in state.items():
            if k.endswith("test_labels"):
                labels = v
            elif k.endswith("text_banks"):
                banks = v
            elif k.endswith("thing_ids"):
                self.thing_ids = set(v)
            elif k.endswith("metadata"):
                self.metadata = v
            elif k.endswith(("semantic_on", "instance_on", "panoptic_on", "test_topk_per_image")):
                setattr(self, k.rsplit(".", 1)[-1], v)
        if state.get("metadata") is not None:
            md = self.metadata
            ids = md["thing_ids"] if isinstance(md, dict) else (getattr(md, "thing_ids", None) or
                                                                  list(getattr(md, "thing_dataset_id_to_contiguous_id", {}).values()))
            self.thing_ids = set(int(i) for i in ids)
        if labels is None and banks is not None:
            if banks is not self._banks:                                   # restore banks that were set without label strings
                self.set_vocabulary(*banks, self.thing_ids, self._alpha, self._beta)
        elif labels is not None and [list(l) for l in labels] != self.test_labels:
            from .checkpoint import build_vocabulary
            if getattr(self, "_tokenizer", None) is None or getattr(self, "_text_encoder", None) is None:
                raise RuntimeError("label strings need a tokenizer and a text encoder: call attach_text() first (or hand over banks with set_vocabulary)")
            key = tuple(tuple(l) for l in labels)
            if key not in self._vocab_cache:                               # the reference caches text embeddings per label tuple (odise.py:1281-1288)
                self._vocab_cache[key] = build_vocabulary(labels, self._tokenizer, self._text_encoder, train_labels=self._train_labels,
                                                          clip_text_encoder=self._clip_text_encoder)
            cat, clp, sizes, overlap = self._vocab_cache[key]
            self.set_vocabulary(cat, clp, sizes, overlap, self.thing_ids, self._alpha, self._beta)
            self.test_labels = [list(l) for l in labels]

    def classify_device(self, image01: DeviceArray, want_clip_embed=False):
        B, _, H, W = image01.shape
        out = self.ctx.empty((B, self.num_queries, self.num_classes + 1), np.float32)
        ce = self.ctx.empty((B, self.num_queries, 768), np.float32) if want_clip_embed else None
        check(self.ctx.lib.odise_hip_classify(self.ctx.h, C.c_void_p(image01.ptr), B, H, W, C.c_void_p(out.ptr),
                                               C.c_void_p(ce.ptr) if ce is not None else None), "classify")
        return (out, ce) if want_clip_embed else out

    def _buf(self, tag: str, shape, dtype) -> DeviceArray:
        """Pooled device buffer: post-processing outputs are hundreds of MB per image; allocating them per call costs more than
        the kernels that fill them (hipMalloc / hipFree synchronise the device)."""
        shape = tuple(int(x) for x in shape)
        key = (tag, np.dtype(dtype).str)
        need = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        cur = self._pool.get(key)
        if cur is None or cur.nbytes < need:
            cur = self.ctx.empty((max(need, 16),), np.uint8)
            self._pool[key] = cur
        return cur.view(shape, dtype)

    # ---- outputs of one call ------------------------------------------------------------------------------------------------------
    def _post_desc(self, n: int, out_sizes, pan_out=None, alloc=None) -> tuple:
        """Pooled output buffers of `n` images + a PostDesc pointing at them (B / pad / img_hw / mask_cls are left to the caller).
        `alloc(tag, shape, dtype) -> DeviceArray` (optional) provides the LARGE per-image outputs instead of the pool - the drop-in model hands
        out views of torch tensors on its own device, so the library writes the reference's result tensors in place (no copy at the edge)."""
        from ._lib import MAX_SEGMENTS, PostDesc
        K, topk = self.num_classes, int(self.test_topk_per_image)
        d = PostDesc()
        keep = []                                                          # ctypes arrays must outlive the call
        ohw = (C.c_int * (2 * n))(*[int(v) for s in out_sizes for v in s])
        thing = (C.c_uint8 * max(K, 1))(*[1 if k in self.thing_ids else 0 for k in range(K)])
        keep += [ohw, thing]
        d.out_hw, d.isthing = C.cast(ohw, C.c_void_p), C.cast(thing, C.c_void_p)
        d.semantic_on, d.panoptic_on, d.instance_on = int(bool(self.semantic_on)), int(bool(self.panoptic_on)), int(bool(self.instance_on))
        d.object_mask_threshold, d.overlap_threshold, d.topk = float(self.object_mask_threshold), float(self.overlap_threshold), topk
        bufs = {"sem": [None] * n, "amax": [None] * n, "pan": [None] * n, "masks": [None] * n, "pan_ext": [False] * n}
        big = alloc if alloc is not None else self._buf
        for i, (oh, ow) in enumerate(out_sizes):
            if self.semantic_on and not self.semantic_argmax:
                bufs["sem"][i] = big(f"sem{i}", (K, oh, ow), np.float32)
            if self.semantic_on and self.semantic_argmax:
                bufs["amax"][i] = big(f"amax{i}", (oh, ow), np.int32)
            if self.panoptic_on:
                ext = pan_out[i] if pan_out is not None else None
                if ext is not None:                                        # caller-owned record (this rank's slice of the gather buffer)
                    bufs["pan"][i], bufs["pan_ext"][i] = ext, True
                else:
                    bufs["pan"][i] = big(f"pan{i}", (oh * ow + 1 + 3 * MAX_SEGMENTS,), np.int32)
            if self.instance_on:
                bufs["masks"][i] = big(f"masks{i}", (topk, oh, ow), np.float32)

        def parr(lst):
            a = (C.c_void_p * n)(*[(b.ptr if isinstance(b, DeviceArray) else b) for b in lst])
            keep.append(a)
            return C.cast(a, C.c_void_p)

        if self.semantic_on and not self.semantic_argmax:
            d.sem_seg = parr(bufs["sem"])
        if self.semantic_on and self.semantic_argmax:
            d.sem_argmax = parr(bufs["amax"])
        if self.panoptic_on:
            d.panoptic = parr(bufs["pan"])
        if self.instance_on:
            d.inst_masks = parr(bufs["masks"])
            bufs["itable"] = self._buf("inst_table", (n, 1 + 2 * topk), np.int32)
            bufs["iscores"] = self._buf("inst_scores", (n, topk), np.float32)
            d.inst_table, d.inst_scores = bufs["itable"].ptr, bufs["iscores"].ptr
        return d, bufs, keep

    def _collect(self, bufs, out_sizes, to_host: bool) -> list:
        """Read the small tables back (one synchronisation) and assemble the reference's result dicts."""
        from ._lib import MAX_SEGMENTS
        n, topk = len(out_sizes), int(self.test_topk_per_image)
        results = [dict() for _ in range(n)]
        itable = bufs["itable"].numpy() if self.instance_on else None
        iscores = bufs["iscores"].numpy() if self.instance_on else None
        for i, (oh, ow) in enumerate(out_sizes):
            r = results[i]
            if self.semantic_on and not self.semantic_argmax:
                r["sem_seg"] = bufs["sem"][i].numpy() if to_host else bufs["sem"][i]
            if self.semantic_on and self.semantic_argmax:
                r["sem_seg_argmax"] = bufs["amax"][i].numpy() if to_host else bufs["amax"][i]
            if self.panoptic_on:
                rec = bufs["pan"][i]
                if bufs["pan_ext"][i]:
                    r["panoptic_seg"] = (None, None)                       # the record lives in the caller's buffer (decode with distributed.unpack_record)
                else:
                    tail = rec.view((1 + 3 * MAX_SEGMENTS,), np.int32, oh * ow * 4).numpy()
                    info = [{"id": int(a), "isthing": bool(b), "category_id": int(c)} for a, b, c in tail[1:1 + 3 * int(tail[0])].reshape(-1, 3)]
                    seg = rec.view((oh, ow), np.int32)
                    r["panoptic_seg"] = (seg.numpy() if to_host else seg, info)
            if self.instance_on:
                cnt = int(itable[i, 0])
                masks = bufs["masks"][i].view((cnt, oh, ow))
                r["instances"] = {"pred_masks": masks.numpy() if to_host else masks, "scores": iscores[i, :cnt].copy(),
                                  "pred_classes": itable[i, 1 + topk:1 + topk + cnt].astype(np.int64), "query_index": itable[i, 1:1 + cnt].copy()}
        return results

    def postprocess_batch(self, mask_cls, pad_hw, img_hw, out_sizes, to_host: bool = True, pan_out=None) -> list:
        """Post-processing (odise.py:336-370) of the images of the last head_forward from their class log-probabilities `mask_cls`
        ([B,Q,K+1], host array or DeviceArray).  img_hw: one (h, w) for all images or a list; out_sizes: list of (h, w)."""
        dcls = mask_cls if isinstance(mask_cls, DeviceArray) else self.ctx.to_device(np.ascontiguousarray(mask_cls, np.float32))
        n = dcls.shape[0]
        sizes = [tuple(out_sizes[b]) for b in range(n)] if not isinstance(out_sizes, dict) else [tuple(out_sizes[b]) for b in sorted(out_sizes)]
        ihw = [tuple(img_hw)] * n if isinstance(img_hw[0], (int, np.integer)) else [tuple(x) for x in img_hw]
        d, bufs, keep = self._post_desc(n, sizes, pan_out)
        iarr = (C.c_int * (2 * n))(*[int(v) for s in ihw for v in s])
        d.B, d.pad_h, d.pad_w, d.img_hw, d.mask_cls = n, int(pad_hw[0]), int(pad_hw[1]), C.cast(iarr, C.c_void_p), dcls.ptr
        check(self.ctx.lib.odise_hip_postprocess_batch(self.ctx.h, C.byref(d)), "postprocess_batch")
        return self._collect(bufs, sizes, to_host)

    def prefetch_device(self, images, layout: int, img_hw) -> None:
        """Register the NEXT batch (`odise_hip_infer_prefetch`): the following `infer_device` call enqueues this batch's input side and VAE
        encoder behind its own VAE lane, and the `infer_device` of exactly these pictures starts from the stored latent.  `images=None` cancels.
        The image buffers must stay alive and unchanged until their own `infer_device` call has returned."""
        if images is None:
            check(self.ctx.lib.odise_hip_infer_prefetch(self.ctx.h, None), "infer_prefetch")
            return
        n = len(images)
        from ._lib import InferDesc
        d = InferDesc()
        ptrs = (C.c_void_p * n)(*[(im.ptr if isinstance(im, DeviceArray) else im) for im in images])
        iarr = (C.c_int * (2 * n))(*[int(v) for s in img_hw for v in s])
        d.B, d.images, d.image_layout, d.img_hw = n, C.cast(ptrs, C.c_void_p), layout, C.cast(iarr, C.c_void_p)
        check(self.ctx.lib.odise_hip_infer_prefetch(self.ctx.h, C.byref(d)), "infer_prefetch")

    def infer_device(self, images, layout: int, img_hw, out_sizes, to_host: bool = False, pan_out=None, mask_cls_out=None, alloc=None) -> list:
        """One `odise_hip_infer` call: `images` = device pointers (DeviceArray or int) of uint8 HWC (layout 0) / uint8 CHW (1) / fp32 CHW
        0..255 (2) pictures with sizes img_hw [(h, w)]."""
        n = len(images)
        from ._lib import InferDesc
        d = InferDesc()
        post, bufs, keep = self._post_desc(n, out_sizes, pan_out, alloc)
        ptrs = (C.c_void_p * n)(*[(im.ptr if isinstance(im, DeviceArray) else im) for im in images])
        iarr = (C.c_int * (2 * n))(*[int(v) for s in img_hw for v in s])
        d.B, d.images, d.image_layout, d.img_hw = n, C.cast(ptrs, C.c_void_p), layout, C.cast(iarr, C.c_void_p)
        d.mask_cls_out = mask_cls_out.ptr if mask_cls_out is not None else None
        d.post = post
        check(self.ctx.lib.odise_hip_infer(self.ctx.h, C.byref(d)), "infer")
        return self._collect(bufs, out_sizes, to_host)

    def __call__(self, *args, **kwargs):                                   # nn.Module-style call: the reference's wrappers do `self.model(batched_inputs)`
        return self.forward(*args, **kwargs)

    def forward(self, batched_inputs, to_host: bool = True, alloc=None) -> list:
        """CategoryODISE.forward, eval branch (odise.py:236-246, 282-372).  "image" is a CHW uint8 / float array or CPU tensor (values
        0..255, the reference's format), or a DeviceArray uint8 [H,W,3] already in HBM (odise_amd.ingest.HipDatasetMapper).
        `to_host=False` leaves the large outputs (sem_seg, panoptic map, instance masks) on the device, like the reference does;
        `alloc` (see _post_desc) lets the caller own them."""
        first = batched_inputs[0]["image"]
        if isinstance(first, DeviceArray):
            ims = [x["image"] for x in batched_inputs]
            assert all(i.dtype == np.uint8 and len(i.shape) == 3 and i.shape[2] == 3 for i in ims), "device images: uint8 [H,W,3]"
            layout, hw = 0, [tuple(i.shape[:2]) for i in ims]
        else:
            host = []
            for x in batched_inputs:
                im = x["image"]
                if hasattr(im, "detach"):
                    im = im.detach().cpu().numpy()
                host.append(np.asarray(im))
            u8 = all(h.dtype == np.uint8 for h in host)
            layout = 1 if u8 else 2
            hw = [tuple(h.shape[-2:]) for h in host]
            ims = [self.ctx.to_device(np.ascontiguousarray(h, np.uint8 if u8 else np.float32)) for h in host]
        sizes = [(int(x.get("height", s[0])), int(x.get("width", s[1]))) for x, s in zip(batched_inputs, hw)]
        return self.infer_device(ims, layout, hw, sizes, to_host=to_host, alloc=alloc)


class HipCaptionODISE(HipCategoryODISE):
    """CaptionODISE's eval forward (odise.py:545-619): identical to the label model up to the classification stage, where the
    no-object probability comes from the decoder's learned 2-way `class_embed` (object / no-object) instead of the null text
    embedding, and the word bank is projected by `word_head.text_proj` (WordEmbed.forward eval, odise.py:1206-1216; prompt "photo",
    configs/common/models/mask_generator_with_caption.py:57-63).  The library switches on the presence of `word_head.*` /
    `sem_seg_head.predictor.class_embed.*` in the state; `set_vocabulary(cat_text=<word bank>, ...)` is unchanged."""

    HEAD_KEYS = ("word_head.text_proj.weight", "word_head.text_proj.bias")


class HipOpenPanopticInference:
    """OpenPanopticInference (odise/modeling/wrapper/pano_wrapper.py:13-70): run the wrapped model with another vocabulary / metadata /
    task switches and restore its own afterwards.  `model` must have been given a tokenizer and a text encoder (`attach_text`)."""

    def __init__(self, model: HipCategoryODISE, labels, metadata=None, semantic_on=True, instance_on=True, panoptic_on=True,
                 test_topk_per_image=100):
        self.model, self.labels, self.metadata = model, labels, metadata
        self.open_state_dict = {}
        for k in model.open_state_dict():
            if k.endswith("test_labels"):
                self.open_state_dict[k] = labels
            elif k.endswith("metadata"):
                self.open_state_dict[k] = metadata
            elif k.endswith("num_classes"):
                self.open_state_dict[k] = len(labels)
            elif k.endswith("semantic_on"):
                self.open_state_dict[k] = semantic_on
            elif k.endswith("instance_on"):
                self.open_state_dict[k] = instance_on
            elif k.endswith("panoptic_on"):
                self.open_state_dict[k] = panoptic_on
            elif k.endswith("test_topk_per_image"):
                self.open_state_dict[k] = test_topk_per_image

    @property
    def num_classes(self):
        return len(self.labels)

    def forward(self, batched_inputs):
        saved = self.model.open_state_dict()
        self.model.load_open_state_dict(self.open_state_dict)
        try:
            return self.model.forward(batched_inputs)
        finally:
            self.model.load_open_state_dict(saved)

    __call__ = forward
