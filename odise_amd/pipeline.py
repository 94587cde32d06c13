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
def _softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


class HipCategoryODISE(HipODISE):
    """The whole CategoryODISE eval forward on the device.  `forward(batched_inputs)` takes the reference's input format
    (list of {"image": uint8/float CHW, "height", "width"}) and returns the reference's output format (list of dicts with
    "sem_seg" [K,h,w] fp32, "panoptic_seg" (int32 [h,w], segments_info), "instances" {pred_masks, scores, pred_classes})."""

    HEAD_KEYS = ("category_head.text_proj.weight", "category_head.text_proj.bias", "category_head.null_embed")

    def __init__(self, ctx: Context, state, semantic_on=True, panoptic_on=True, instance_on=True, object_mask_threshold=0.0,
                 overlap_threshold=0.8, test_topk_per_image=100, size_divisibility=64):
        super().__init__(ctx, {k: v for k, v in state.items()})
        # category_head weights are loaded through the same store: re-register them (the store was cleared after build)
        n = 0
        for key in self.HEAD_KEYS:
            val = state[key]
            if hasattr(val, "detach"):
                val = val.detach().cpu().numpy()
            arr = np.ascontiguousarray(val, dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            check(ctx.lib.odise_hip_load_weight(ctx.h, key.encode(), arr.ctypes.data_as(C.POINTER(C.c_float)), shape, arr.ndim), key)
            n += 1
        check(ctx.lib.odise_hip_classify_build(ctx.h), "classify_build")
        check(ctx.lib.odise_hip_clear_host_weights(ctx.h), "clear_host_weights")
        self.semantic_on, self.panoptic_on, self.instance_on = semantic_on, panoptic_on, instance_on
        self.object_mask_threshold, self.overlap_threshold = object_mask_threshold, overlap_threshold
        self.test_topk_per_image, self.size_divisibility = test_topk_per_image, size_divisibility
        self.num_classes = 0
        self.thing_ids = set()
        self.metadata, self.test_labels = None, None
        self._alpha, self._beta = 0.3, 0.7
        self._banks, self._vocab_cache = None, {}
        self._pool = {}

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
        (odise_amd.tokenizer.SimpleTokenizer, odise_amd.text.HipTextEncoder; `train_labels` decide the seen/unseen ensemble weights)."""
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

    def postprocess_image(self, b: int, mask_cls: np.ndarray, pad_hw, img_hw, out_hw, to_host: bool = True, pan_out=None) -> dict:
        """Post-processing of image b alone (see postprocess_batch)."""
        return self.postprocess_batch({b: mask_cls}, pad_hw, img_hw, {b: out_hw}, to_host=to_host,
                                      pan_out={b: pan_out} if pan_out is not None else None)[0]

    def postprocess_batch(self, mask_cls, pad_hw, img_hw, out_sizes, to_host: bool = True, pan_out=None) -> list:
        """Post-processing (odise.py:336-370) of the images in `mask_cls` ({b: [Q,K+1]} or an array [B,Q,K+1], host) from the
        device-resident mask logits.  The work is staged ACROSS images - every host decision (kept queries, panoptic segment
        ids, instance top-k) is taken for the whole batch between two rounds of kernel launches - so a step has two device
        synchronisations instead of three per image.  to_host=False keeps the large results (sem_seg, panoptic map, instance
        masks) on the device, like the reference, whose outputs are device tensors; they live in pooled buffers that the next
        call reuses."""
        ctx, lib = self.ctx, self.ctx.lib
        Q, K = self.num_queries, self.num_classes
        items = list(mask_cls.items()) if isinstance(mask_cls, dict) else list(enumerate(mask_cls))
        sizes = out_sizes if isinstance(out_sizes, dict) else dict(enumerate(out_sizes))
        n = len(items)
        qpad = (Q + 7) // 8 * 8
        p = lambda a: C.c_void_p(a.ptr) if a is not None else None
        # ---- stage A (host): class probabilities, kept queries; one upload for the batch
        probs = [_softmax(np.asarray(mc, np.float32)) for _, mc in items]
        scores = [pr.max(-1) for pr in probs]
        labels = [pr.argmax(-1) for pr in probs]
        keep = [(lb != K) & (sc > self.object_mask_threshold) for lb, sc in zip(labels, scores)]   # maskformer_model.py:290
        up = np.zeros((n, Q + K * Q), np.float32)
        for i in range(n):
            up[i, :Q] = np.where(keep[i], scores[i], -1.0)
            if self.semantic_on:
                up[i, Q:] = np.ascontiguousarray(probs[i][:, :-1].T).reshape(-1)
        dup = self._buf("post_in", up.shape, np.float32).copy_from(up)
        dcnt = self._buf("post_counts", (n, 3 * Q + 2 * qpad), np.float32)   # [3,Q] int32 counters + [2,qpad] fp32 instance stats
        row = (3 * Q + 2 * qpad) * 4
        sems, idss = [], []
        # ---- stage B (device): per-pixel pass of every image
        for i, (b, _) in enumerate(items):
            oh, ow = sizes[b]
            kscore = dup.view((Q,), np.float32, i * up.shape[1] * 4)
            semT = dup.view((K, Q), np.float32, (i * up.shape[1] + Q) * 4) if self.semantic_on else None
            sem = self._buf(f"sem{i}", (K, oh, ow), np.float32) if self.semantic_on else None
            ids = self._buf(f"ids{i}", (oh * ow,), np.int32)
            counts = dcnt.view((3, Q), np.int32, i * row)
            inst = dcnt.view((2, qpad), np.float32, i * row + 3 * Q * 4) if self.instance_on else None
            check(lib.odise_hip_postprocess_pixels(ctx.h, b, p(kscore), p(semT), K, pad_hw[0], pad_hw[1], img_hw[0], img_hw[1], oh, ow, p(sem),
                                                   p(ids), p(counts), p(inst)), "postprocess_pixels")
            sems.append(sem)
            idss.append(ids)
        # ---- stage C: one read-back (synchronises), host decisions for the batch
        raw = dcnt.numpy()
        results = [dict() for _ in range(n)]
        maps = np.zeros((n, Q + self.test_topk_per_image), np.int32)   # [seg_map | instance query indices]
        inst_sel = []
        for i, (b, _) in enumerate(items):
            if self.panoptic_on:
                cnt = raw[i, :3 * Q].view(np.int32).reshape(3, Q)
                segments_info, stuff_memory, current = [], {}, 0
                for q in np.nonzero(keep[i])[0]:                                            # kept queries in order (maskformer_model.py:312-340)
                    pred_class = int(labels[i][q])
                    isthing = pred_class in self.thing_ids
                    mask_area, original_area, inter = int(cnt[0, q]), int(cnt[1, q]), int(cnt[2, q])
                    if mask_area > 0 and original_area > 0 and inter > 0:
                        if mask_area / original_area < self.overlap_threshold:
                            continue
                        if not isthing:
                            if pred_class in stuff_memory:
                                maps[i, q] = stuff_memory[pred_class]
                                continue
                            stuff_memory[pred_class] = current + 1
                        current += 1
                        maps[i, q] = current
                        segments_info.append({"id": current, "isthing": bool(isthing), "category_id": pred_class})
                results[i]["panoptic_seg"] = (None, segments_info)
            if self.instance_on:
                sc = probs[i][:, :-1].reshape(-1)                                           # maskformer_model.py:349-357
                topk = min(self.test_topk_per_image, sc.size)
                top = np.argpartition(-sc, topk - 1)[:topk]
                top = top[np.argsort(-sc[top], kind="stable")]
                cls, qidx, s = top % K, top // K, sc[top]
                if self.panoptic_on:
                    thing = np.array([int(c) in self.thing_ids for c in cls], bool)
                    cls, qidx, s = cls[thing], qidx[thing], s[thing]
                st = raw[i, 3 * Q:].reshape(2, qpad)
                mask_scores = st[0, qidx] / (st[1, qidx] + 1e-6)
                maps[i, Q:Q + len(qidx)] = qidx
                inst_sel.append((cls, qidx, s, mask_scores))
        dmaps = self._buf("post_maps", maps.shape, np.int32).copy_from(maps)
        # ---- stage D (device): panoptic map and instance masks of every image
        for i, (b, _) in enumerate(items):
            oh, ow = sizes[b]
            if self.panoptic_on:
                ext = pan_out[b] if pan_out is not None and pan_out[b] is not None else None
                seg = self._buf(f"seg{i}", (oh, ow), np.int32) if ext is None else None
                dst = p(seg) if seg is not None else C.c_void_p(int(ext))   # optionally write into a caller-owned buffer (gather slice)
                dmap = dmaps.view((Q,), np.int32, i * maps.shape[1] * 4)
                check(lib.odise_hip_panoptic_write(ctx.h, p(idss[i]), p(dmap), dst, oh * ow), "panoptic_write")
                results[i]["panoptic_seg"] = ((seg.numpy() if to_host else seg) if seg is not None else None, results[i]["panoptic_seg"][1])
            if self.semantic_on:
                results[i]["sem_seg"] = sems[i].numpy() if to_host else sems[i]
            if self.instance_on:
                cls, qidx, s, mask_scores = inst_sel[i]
                masks = self._buf(f"masks{i}", (max(len(qidx), 1), oh, ow), np.float32).view((len(qidx), oh, ow))
                if len(qidx):
                    didx = dmaps.view((len(qidx),), np.int32, (i * maps.shape[1] + Q) * 4)
                    check(lib.odise_hip_instance_masks(ctx.h, b, p(didx), len(qidx), pad_hw[0], pad_hw[1], img_hw[0], img_hw[1], oh, ow, p(masks)),
                          "instance_masks")
                results[i]["instances"] = {"pred_masks": masks.numpy() if to_host else masks, "scores": (s * mask_scores).astype(np.float32),
                                           "pred_classes": cls.astype(np.int64), "query_index": qidx}
        return results

    def forward_device(self, padded: DeviceArray, img01: DeviceArray, out_sizes, to_host: bool = False, pan_out=None) -> list:
        """Hot path with inputs already resident in HBM: padded [B,3,Hp,Wp] and img01 [B,3,H,W] fp32 in [0,1]."""
        B, _, Hp, Wp = padded.shape
        H, W = img01.shape[-2:]
        self.backbone_device(padded, want_outputs=False)
        self.head_device(None, B, Hp // 4, Wp // 4, want_outputs=False)
        mask_cls = self.classify_device(img01).numpy()
        return self.postprocess_batch(mask_cls, (Hp, Wp), (H, W), out_sizes, to_host=to_host,
                                      pan_out=dict(enumerate(pan_out)) if pan_out is not None else None)

    def __call__(self, *args, **kwargs):                                   # nn.Module-style call: the reference's wrappers do `self.model(batched_inputs)`
        return self.forward(*args, **kwargs)

    def forward(self, batched_inputs, to_host: bool = True) -> list:
        """CategoryODISE.forward, eval branch (odise.py:236-246, 282-372) for a batch of equally sized images.  "image" is a CHW uint8 /
        float array on the host (values 0..255), or a DeviceArray uint8 [H,W,3] already in HBM (odise_amd.ingest.HipDatasetMapper).
        `to_host=False` leaves the large outputs (sem_seg, panoptic map, instance masks) on the device, like the reference does."""
        if isinstance(batched_inputs[0]["image"], DeviceArray):
            return self._forward_resident(batched_inputs, to_host)
        imgs = []
        for x in batched_inputs:
            im = x["image"]
            if hasattr(im, "detach"):
                im = im.detach().cpu().numpy()
            imgs.append(np.asarray(im, np.float32) / 255.0)                                 # (x - 0) / 255  (pixel_mean 0, pixel_std 255)
        H, W = imgs[0].shape[-2:]
        assert all(i.shape[-2:] == (H, W) for i in imgs), "batched images must share one size (use one call per size)"
        d = self.size_divisibility
        Hp, Wp = (H + d - 1) // d * d, (W + d - 1) // d * d
        B = len(imgs)
        img01 = np.stack(imgs)
        padded = np.zeros((B, 3, Hp, Wp), np.float32)                                        # ImageList.from_tensors(images, 64): zero pad
        padded[:, :, :H, :W] = img01
        dpad = self.ctx.to_device(padded)
        self.backbone_device(dpad, want_outputs=False)
        self.head_device(None, B, Hp // 4, Wp // 4)
        mask_cls = self.classify_device(self.ctx.to_device(img01)).numpy()
        sizes = [(int(x.get("height", H)), int(x.get("width", W))) for x in batched_inputs]
        return self.postprocess_batch(mask_cls, (Hp, Wp), (H, W), sizes, to_host=to_host)

    def _forward_resident(self, batched_inputs, to_host: bool = True) -> list:
        ims = [x["image"] for x in batched_inputs]
        H, W = ims[0].shape[:2]
        assert all(i.dtype == np.uint8 and i.shape == (H, W, 3) for i in ims), "device images: uint8 [H,W,3] of one size"
        d = self.size_divisibility
        Hp, Wp, B = (H + d - 1) // d * d, (W + d - 1) // d * d, len(ims)
        padded = self._buf("in_padded", (B, 3, Hp, Wp), np.float32)
        img01 = padded if (Hp, Wp) == (H, W) else self._buf("in_img01", (B, 3, H, W), np.float32)
        for b, im in enumerate(ims):                                      # (x - 0) / 255 and ImageList.from_tensors(images, 64), on the device
            self.ctx.u8_hwc_to_f32_chw_padded(im, Hp, Wp, 1.0 / 255.0, out=padded.view((3, Hp, Wp), offset_bytes=b * 3 * Hp * Wp * 4))
            if img01 is not padded:
                self.ctx.u8_hwc_to_f32_chw_padded(im, H, W, 1.0 / 255.0, out=img01.view((3, H, W), offset_bytes=b * 3 * H * W * 4))
        sizes = [(int(x.get("height", H)), int(x.get("width", W))) for x in batched_inputs]
        return self.forward_device(padded, img01, sizes, to_host=to_host)


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
