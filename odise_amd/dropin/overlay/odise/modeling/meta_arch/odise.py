"""`odise.modeling.meta_arch.odise` on libodise_hip.so: the classes the reference's LazyConfig files and wrappers name
(configs/common/models/mask_generator_with_label.py:15-22, mask_generator_with_caption.py; pano_wrapper.py:36-52), with the reference's
constructor arguments, attribute tree, state-dict keys and call signatures (/root/reference odise/modeling/meta_arch/odise.py).

Every module is callable on its own; `CategoryODISE.forward` / `CaptionODISE.forward` bypass the inner calls and run
the whole eval branch as ONE library call (`odise_hip_infer`).  Inference only: `self.training` paths raise."""
import ctypes as C
import operator
from collections import OrderedDict
from typing import Any, Mapping

import numpy as np
import torch
from torch import nn

from odise_amd import dropin
from odise_amd._lib import check


def to_tuple(lst):
    return tuple(to_tuple(i) if isinstance(i, list) else i for i in lst)


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


class MLP(nn.Module):
    """mask2former_transformer_decoder.py:206-216 (parameters `layers.{i}`)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        h = [hidden_dim] * (num_layers - 1)
        self.num_layers = num_layers
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))


class PseudoClassEmbed(nn.Module):
    """odise.py:910-920: every query is foreground for all classes (the label model has no learned class head)."""

    def __init__(self, num_classes):
        super().__init__()
        self.num_classes = num_classes

    def forward(self, x):
        fg = torch.ones((*x.shape[:-1], self.num_classes), dtype=x.dtype, device=x.device)
        return torch.cat([fg, torch.zeros((*x.shape[:-1], 1), dtype=x.dtype, device=x.device)], dim=-1)


class MaskPooling(nn.Module):
    """odise.py:923-963 on `odise_hip_mask_pooling`."""

    def __init__(self, hard_pooling=True, mask_threshold=0.5):
        super().__init__()
        if not hard_pooling or mask_threshold != 0.5:
            raise NotImplementedError("libodise_hip implements hard pooling at threshold 0.5 (the released models)")
        self.hard_pooling, self.mask_threshold = hard_pooling, mask_threshold

    def extra_repr(self) -> str:
        return f"hard_pooling={self.hard_pooling}\nmask_threshold={self.mask_threshold}\n"

    def forward(self, x, mask):
        assert x.shape[-2:] == mask.shape[-2:]
        ctx = dropin.get_context()
        B, Cc, H, W = x.shape
        Q = mask.shape[1]
        xp, xk = dropin.to_device(x)
        mp, mk = dropin.to_device(mask)
        op, fetch = dropin.new_output((B, Q, Cc), x)
        check(ctx.lib.odise_hip_mask_pooling(ctx.h, xp, mp, op, B, Cc, Q, H * W), "mask_pooling")
        return {"mask_pooled_features": fetch()}


class PooledMaskEmbed(nn.Module):
    """odise.py:966-1015.  Stand-alone it is composed from the library's operators (mask pooling, LayerNorm, GEMM); inside the fused
    model the same arithmetic is part of `odise_hip_head_forward`."""

    def __init__(self, hidden_dim, mask_dim, projection_dim, temperature=0.07):
        super().__init__()
        self.pool_proj = nn.Sequential(nn.LayerNorm(hidden_dim), nn.Linear(hidden_dim, hidden_dim))
        self.mask_embed = nn.Sequential(nn.LayerNorm(mask_dim), MLP(mask_dim, hidden_dim, projection_dim, 3))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / temperature))
        self.mask_pooling = MaskPooling()

    def _device_weights(self, ctx):
        """The module's few weights on the device, uploaded once per parameter state, not per call.  The key holds `_version` (in-place
        updates: `load_state_dict`, optimiser steps) AND the storage identity / dtype / device of every parameter: `param.data = ...`
        (`Module.to(dtype)`, manual weight swaps, some checkpoint loaders) replaces the storage without bumping the version."""
        version = tuple((p._version, p.data_ptr(), p.dtype, str(p.device)) for p in self.parameters())
        if getattr(self, "_dev", None) is None or self._dev[0] != version or self._dev[1] is not ctx:
            w = {}
            for name, mod in (("pool_ln", self.pool_proj[0]), ("pool_lin", self.pool_proj[1]), ("embed_ln", self.mask_embed[0]),
                              *[(f"mlp{i}", lin) for i, lin in enumerate(self.mask_embed[1].layers)]):
                if isinstance(mod, nn.LayerNorm):
                    w[name] = (ctx.to_device(_np(mod.weight).astype(np.float32)), ctx.to_device(_np(mod.bias).astype(np.float32)), mod.eps)
                else:
                    w[name] = (ctx.to_device(_np(mod.weight).astype(np.float16)), ctx.to_device(_np(mod.bias).astype(np.float32)) if mod.bias is not None else None)
            self._dev = (version, ctx, w)
        return self._dev[2]

    # the cache holds ctypes-backed device buffers: never part of a pickle or a copy.deepcopy of the model (both go through __getstate__)
    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_dev", None)
        return state

    @staticmethod
    def _linear(ctx, x, wb, act=0, residual=None):
        return ctx.gemm(x, wb[0], bias_n=wb[1], act=act, residual=residual)

    @staticmethod
    def _layer_norm(ctx, x, gbe):
        return ctx.layer_norm(x, gbe[0], gbe[1], eps=gbe[2])

    def forward(self, decoder_output, input_mask_embed, mask_features, pred_logits, pred_masks):
        from odise_amd._lib import ACT_RELU
        ctx = dropin.get_context()
        pooled = self.mask_pooling(mask_features, pred_masks)["mask_pooled_features"]          # [B,Q,C] fp32
        B, Q, Cd = pooled.shape
        x = ctx.to_device(_np(pooled).reshape(B * Q, Cd).astype(np.float16))
        dec = ctx.to_device(_np(decoder_output).reshape(B * Q, Cd).astype(np.float16))
        w = self._device_weights(ctx)
        x = self._linear(ctx, self._layer_norm(ctx, x, w["pool_ln"]), w["pool_lin"], residual=dec)   # pool_proj(x) += decoder_output (:1000)
        h = self._layer_norm(ctx, x, w["embed_ln"])
        mlp = self.mask_embed[1]
        for i in range(mlp.num_layers):
            h = self._linear(ctx, h, w[f"mlp{i}"], act=ACT_RELU if i < mlp.num_layers - 1 else 0)
        dev = decoder_output.device
        to_t = lambda a: torch.from_numpy(a.numpy().astype(np.float32).reshape(B, Q, -1)).to(dev)
        return {"mask_embed": to_t(h), "mask_pooled_features": to_t(x), "logit_scale": torch.clamp(self.logit_scale.detach().exp(), max=100)}


class _Attn(nn.Module):
    """SelfAttentionLayer / CrossAttentionLayer (mask2former_transformer_decoder.py:17-143): `self_attn` | `multihead_attn`, `norm`."""

    def __init__(self, d_model, nhead, name):
        super().__init__()
        setattr(self, name, nn.MultiheadAttention(d_model, nhead, dropout=0.0))
        self.norm = nn.LayerNorm(d_model)


class _FFN(nn.Module):
    def __init__(self, d_model, dim_feedforward):
        super().__init__()
        self.linear1, self.linear2 = nn.Linear(d_model, dim_feedforward), nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)


class ODISEMultiScaleMaskedTransformerDecoder(nn.Module):
    """odise.py:622-776 over MultiScaleMaskedTransformerDecoder (mask2former_transformer_decoder.py:218-334): parameters of the
    reference; `forward(x, mask_features)` = `odise_hip_predictor_forward` through the owning MaskFormerHead's library build."""

    def __init__(self, *, class_embed=None, mask_embed=None, post_mask_embed=None, in_channels, mask_classification=True, num_classes: int,
                 hidden_dim: int, num_queries: int, nheads: int, dim_feedforward: int, dec_layers: int, pre_norm: bool, mask_dim: int,
                 enforce_input_project: bool):
        super().__init__()
        assert mask_classification
        if pre_norm or enforce_input_project or in_channels != hidden_dim or mask_embed is not None or post_mask_embed is None:
            raise NotImplementedError("libodise_hip implements the released decoder: post-norm, identity input projections, PooledMaskEmbed")
        self.mask_classification, self.num_heads, self.num_layers, self.num_feature_levels = True, nheads, dec_layers, 3
        self.num_queries, self.hidden_dim, self.num_classes = num_queries, hidden_dim, num_classes
        self.transformer_self_attention_layers = nn.ModuleList(_Attn(hidden_dim, nheads, "self_attn") for _ in range(dec_layers))
        self.transformer_cross_attention_layers = nn.ModuleList(_Attn(hidden_dim, nheads, "multihead_attn") for _ in range(dec_layers))
        self.transformer_ffn_layers = nn.ModuleList(_FFN(hidden_dim, dim_feedforward) for _ in range(dec_layers))
        self.decoder_norm = nn.LayerNorm(hidden_dim)
        self.query_feat, self.query_embed = nn.Embedding(num_queries, hidden_dim), nn.Embedding(num_queries, hidden_dim)
        self.level_embed = nn.Embedding(3, hidden_dim)
        self.input_proj = nn.ModuleList(nn.Sequential() for _ in range(3))
        # CategoryODISE: the parameter-free PseudoClassEmbed; CaptionODISE keeps Mask2Former's Linear(hidden_dim, num_classes + 1)
        self.class_embed = class_embed if class_embed is not None else nn.Linear(hidden_dim, num_classes + 1)
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)
        self.post_mask_embed = post_mask_embed

    def forward(self, x, mask_features, mask=None, *, inputs_dict=None):
        assert len(x) == self.num_feature_levels
        owner = getattr(self, "_head", None)
        if owner is None:
            raise RuntimeError("ODISEMultiScaleMaskedTransformerDecoder runs inside a MaskFormerHead (which owns the library's head build)")
        return owner()._predictor(x, mask_features)


# ---- text side ----------------------------------------------------------------------------------------------------------------------------
def _default_train_labels():
    from odise_amd.checkpoint import default_train_labels
    return default_train_labels()


class _TextBank(nn.Module):
    """Shared by CategoryEmbed / WordEmbed / PoolingCLIPHead: CLIP text embeddings of prompt strings, cached per label tuple
    (odise.py:1281-1288, 1092-1102) and produced on the device (odise_amd.text.HipTextEncoder)."""

    def _init_bank(self, clip_model_name, prompt):
        self.clip_model_name, self.prompt = clip_model_name, prompt
        self.test_labels = None
        self._test_text_embed_dict = dict()

    def extra_repr(self) -> str:
        return f"clip_model_name={self.clip_model_name},\n"

    def _open_state_dict(self):
        return {"test_labels": self.test_labels}

    def open_state_dict(self, destination=None, prefix=""):
        if destination is None:
            destination = OrderedDict()
        for k, v in self._open_state_dict().items():
            destination[prefix + k] = v
        return destination

    def build_text_embed(self, labels, verbose=False):
        """labels: nested list of prompt strings -> [n_strings, dim] fp32 (clip.py:29-73 `build_clip_text_embed`: one embedding per string)."""
        tools = dropin.text_tools()
        if tools is None:
            raise RuntimeError("label strings need a tokenizer and the CLIP text tower: odise_amd.dropin.set_text_tools(tokenizer, HipTextEncoder)")
        tok, enc = tools
        flat = [labels] if isinstance(labels, str) else [s for l in labels for s in ([l] if isinstance(l, str) else l)]
        return torch.from_numpy(enc.build_text_embed(tok(flat)).astype(np.float32))

    def get_and_cache_test_text_embed(self, labels):
        key = to_tuple(labels)
        if key not in self._test_text_embed_dict:
            if len(self._test_text_embed_dict) > 3:                            # the reference keeps a handful of vocabularies
                self._test_text_embed_dict.pop(next(iter(self._test_text_embed_dict)))
            self._test_text_embed_dict[key] = self.build_text_embed(labels)
        return self._test_text_embed_dict[key]


def _prompt_labels(labels, prompt):
    from odise_amd.checkpoint import prompt_labels
    return prompt_labels(labels, prompt)


class CategoryEmbed(_TextBank):
    """odise.py:1219-1307: `text_proj`, `null_embed`; eval forward -> {"text_embed", "null_embed", "labels"}."""

    def __init__(self, labels, projection_dim, clip_model_name="ViT-L-14", prompt=None):
        super().__init__()
        self.labels = labels
        self._init_bank(clip_model_name, prompt)
        dim = 768 if "L-14" in clip_model_name else 512
        self.text_proj = nn.Identity() if projection_dim < 0 else nn.Linear(dim, projection_dim)
        self.null_embed = nn.Parameter(torch.zeros(1, dim))    # = build_text_embed("") in the reference; every released checkpoint stores it

    def forward(self, outputs, targets=None):
        if self.training:
            raise RuntimeError("libodise_hip is an inference library (CategoryEmbed training branch, odise.py:1291-1296)")
        assert targets is None and self.test_labels is not None
        text_embed = self.get_and_cache_test_text_embed(_prompt_labels(self.test_labels, self.prompt))
        with torch.no_grad():
            return {"text_embed": self.text_proj(text_embed.to(self.null_embed.device)), "null_embed": self.text_proj(self.null_embed),
                    "labels": self.test_labels}


class WordEmbed(_TextBank):
    """odise.py:1018-1216 (eval side): `text_proj` over the word bank of the caption model."""

    def __init__(self, projection_dim, clip_model_name="ViT-L-14", word_dropout=0.0, word_tags="noun_phrase", num_words=8, prompt="photo"):
        super().__init__()
        self._init_bank(clip_model_name, prompt)
        dim = 768 if "L-14" in clip_model_name else 512
        self.text_proj = nn.Identity() if projection_dim < 0 else nn.Linear(dim, projection_dim)
        self.word_dropout, self.word_tags, self.num_words = word_dropout, word_tags, num_words

    def forward(self, outputs, targets=None):
        if self.training:
            raise RuntimeError("libodise_hip is an inference library (WordEmbed training branch)")
        assert targets is None and self.test_labels is not None
        text_embed = self.get_and_cache_test_text_embed(_prompt_labels(self.test_labels, self.prompt))
        with torch.no_grad():
            return {"text_embed": self.text_proj(text_embed.to(next(self.parameters()).device)), "labels": self.test_labels}


class PoolingCLIPHead(_TextBank):
    """odise.py:1422-1542: MaskCLIP embeddings of the predicted masks (library: `odise_hip_maskclip_embed`) against the "a photo of a {}."
    bank, max over synonyms, geometric ensemble with the in-vocabulary logits weighted by train / test label overlap."""

    def __init__(self, clip_model_name="ViT-L-14-336", alpha=0.35, beta=0.65, prompt="photo", train_labels=None, normalize_logits=True, bg_labels=None):
        super().__init__()
        self._init_bank(clip_model_name, prompt)
        if not normalize_logits or bg_labels is not None:
            raise NotImplementedError("libodise_hip implements normalize_logits=True without background labels (the released models)")
        self.alpha, self.beta = alpha, beta
        self._train_labels = train_labels
        self.bg_labels, self.normalize_logits = bg_labels, normalize_logits

    @property
    def train_labels(self):
        if self._train_labels is None:                                         # the reference's default: COCO panoptic, prompt engineered
            self._train_labels = _default_train_labels()
        return self._train_labels

    @property
    def with_bg(self):
        return False

    def category_overlapping_mask(self):
        from odise_amd.checkpoint import category_overlapping_mask
        return category_overlapping_mask(self.train_labels, self.test_labels)

    def forward(self, outputs, targets=None):
        assert not self.training, "PoolingCLIPHead only supports inference"
        assert targets is None and self.test_labels is not None
        from odise_amd.checkpoint import ensemble_max
        pred_open_logits = outputs.pop("pred_open_logits")
        labels = _prompt_labels(self.test_labels, self.prompt)
        text_embed = self.get_and_cache_test_text_embed(labels)
        images, masks = outputs["images"], outputs["pred_masks"]
        ctx = dropin.get_context()
        B, _, H, W = images.shape
        _, Q, h, w = masks.shape
        ip, ik = dropin.to_device(images)
        mp, mk = dropin.to_device(masks)
        op, fetch = dropin.new_output((B, Q, text_embed.shape[-1]), images)
        check(ctx.lib.odise_hip_maskclip_embed(ctx.h, ip, B, H, W, mp, Q, h, w, op), "maskclip_embed")
        me = torch.nn.functional.normalize(fetch().float().cpu(), dim=-1)
        te = torch.nn.functional.normalize(text_embed.float(), dim=-1)
        clip_logits = ensemble_max(torch.einsum("bqc,nc->bqn", me, te) * 100.0, [len(l) for l in self.test_labels])   # clamp(exp(ln 100), max=100)
        ovl = torch.from_numpy(self.category_overlapping_mask()).to(pred_open_logits.dtype)
        p, q = pred_open_logits.float().cpu().softmax(-1), clip_logits.softmax(-1)
        base = (p ** (1 - self.alpha) * q ** self.alpha).log() * ovl
        novel = (p ** (1 - self.beta) * q ** self.beta).log() * (1 - ovl)
        ret = {"pred_open_logits": (base + novel).to(pred_open_logits.device)}
        if "labels" in outputs:
            ret["labels"] = labels
        return ret


# ---- meta architectures -------------------------------------------------------------------------------------------------------------------
class ODISE(nn.Module):
    """MaskFormer's test-time attributes (maskformer_model.py:25-102) + ODISE's open-vocabulary protocol (odise.py:121-166)."""

    def __init__(self, *, backbone, sem_seg_head, criterion=None, num_queries: int, object_mask_threshold: float, overlap_threshold: float, metadata,
                 size_divisibility: int, sem_seg_postprocess_before_inference: bool, pixel_mean, pixel_std, semantic_on: bool, panoptic_on: bool,
                 instance_on: bool, test_topk_per_image: int):
        super().__init__()
        self.backbone, self.sem_seg_head, self.criterion = backbone, sem_seg_head, None   # losses are training-only
        self.num_queries, self.object_mask_threshold, self.overlap_threshold = num_queries, object_mask_threshold, overlap_threshold
        self.metadata = metadata
        self.size_divisibility = size_divisibility if size_divisibility >= 0 else backbone.size_divisibility
        self.sem_seg_postprocess_before_inference = sem_seg_postprocess_before_inference
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)
        if [float(v) for v in pixel_mean] != [0.0] * 3 or [float(v) for v in pixel_std] != [255.0] * 3 or not sem_seg_postprocess_before_inference:
            raise NotImplementedError("libodise_hip implements the released normalisation (mean 0, std 255) and sem_seg_postprocess_before_inference=True")
        self.semantic_on, self.instance_on, self.panoptic_on, self.test_topk_per_image = semantic_on, instance_on, panoptic_on, test_topk_per_image
        self._hip, self._built_version = None, None

    @property
    def device(self):
        return self.pixel_mean.device

    def ignored_state_dict(self, destination=None, prefix=""):
        return destination if destination is not None else OrderedDict()

    def _open_state_dict(self):
        return {"sem_seg_head.num_classes": self.sem_seg_head.num_classes, "metadata": self.metadata, "test_topk_per_image": self.test_topk_per_image,
                "semantic_on": self.semantic_on, "panoptic_on": self.panoptic_on, "instance_on": self.instance_on}

    def open_state_dict(self, destination=None, prefix=""):
        if destination is None:
            destination = OrderedDict()
        for k, v in self._open_state_dict().items():
            destination[prefix + k] = v
        for name, module in self.named_modules(remove_duplicate=True):
            if module is not self and module is not None and hasattr(module, "open_state_dict"):
                module.open_state_dict(destination, prefix + name + ".")
        return destination

    def load_open_state_dict(self, state_dict: Mapping[str, Any]):
        for k, v in state_dict.items():
            if len(k.rsplit(".", 1)) == 2:
                prefix, suffix = k.rsplit(".", 1)
                operator.attrgetter(prefix)(self).__setattr__(suffix, v)
            else:
                self.__setattr__(k, v)
            assert operator.attrgetter(k)(self) == v, f"{k} is not loaded correctly"

    # ---- the fused model on the library -----------------------------------------------------------------------------------------------
    _HIP_CLASS = "HipCategoryODISE"

    def _library_state(self):
        state = self.backbone.library_state("backbone.")
        state.update(self.sem_seg_head.library_state("sem_seg_head."))
        for name in ("category_head", "word_head"):
            mod = getattr(self, name, None)
            if mod is not None:
                state.update({f"{name}.{k}": _np(v) for k, v in mod.state_dict().items()})
        return state

    def _thing_ids(self):
        md = self.metadata
        ids = md.get("thing_ids") if isinstance(md, dict) else None
        if ids is None:
            ids = getattr(md, "thing_dataset_id_to_contiguous_id", None)
            ids = list(ids.values()) if ids is not None else getattr(md, "thing_ids", [])
        return set(int(i) for i in ids)

    def _engine(self):
        from odise_amd import pipeline
        version = tuple(p._version for p in self.parameters())
        if self._hip is None or self._built_version != version:
            cls = getattr(pipeline, self._HIP_CLASS)
            self._hip = cls(dropin.get_context(), self._library_state(), semantic_on=self.semantic_on, panoptic_on=self.panoptic_on,
                            instance_on=self.instance_on, object_mask_threshold=self.object_mask_threshold, overlap_threshold=self.overlap_threshold,
                            test_topk_per_image=self.test_topk_per_image, size_divisibility=self.size_divisibility)
            self._built_version, self._vocab_key = version, None
            self.sem_seg_head._built_version = tuple(p._version for p in self.sem_seg_head.parameters())   # the head's weights are in the library now
        hip = self._hip
        hip.semantic_on, hip.panoptic_on, hip.instance_on = self.semantic_on, self.panoptic_on, self.instance_on
        hip.object_mask_threshold, hip.overlap_threshold, hip.test_topk_per_image = self.object_mask_threshold, self.overlap_threshold, self.test_topk_per_image
        return hip

    def _set_vocabulary(self, hip, text_head):
        labels = text_head.test_labels
        assert labels is not None and self.clip_head.test_labels is not None, "test_labels are set through OpenPanopticInference / load_open_state_dict"
        key = (to_tuple(labels), self.clip_head.alpha, self.clip_head.beta, tuple(sorted(self._thing_ids())))
        if self._vocab_key != key:
            cat = text_head.get_and_cache_test_text_embed(_prompt_labels(labels, text_head.prompt))
            clp = self.clip_head.get_and_cache_test_text_embed(_prompt_labels(labels, self.clip_head.prompt))
            hip.set_vocabulary(cat.numpy(), clp.numpy(), [len(l) for l in labels], self.clip_head.category_overlapping_mask(), self._thing_ids(),
                               self.clip_head.alpha, self.clip_head.beta)
            self._vocab_key = key

    def _forward_eval(self, batched_inputs, text_head):
        if self.training:
            raise RuntimeError("libodise_hip is an inference library: call model.eval() (the training branch of odise.py:246-281 is out of scope)")
        hip = self._engine()
        self._set_vocabulary(hip, text_head)
        if self.device.type == "cuda":
            return self._forward_eval_device(hip, batched_inputs)
        # a model that lives on the CPU returns CPU tensors (as the reference would): the results cross PCIe once
        results = hip.forward([{**x, "image": x["image"]} for x in batched_inputs], to_host=True)
        out = []
        for r in results:
            o = {}
            if "sem_seg" in r:
                o["sem_seg"] = torch.from_numpy(r["sem_seg"]).to(self.device)
            if "panoptic_seg" in r:
                o["panoptic_seg"] = (torch.from_numpy(r["panoptic_seg"][0]).to(self.device), r["panoptic_seg"][1])
            if "instances" in r:
                o["instances"] = _instances(r["instances"], self.device)
            out.append(o)
        return out

    def _forward_eval_device(self, hip, batched_inputs):
        """The model lives on a ROCm device (`model.to("cuda")`, what tools/train_net.py does): `sem_seg`, the panoptic map and the instance
        masks are allocated as torch tensors on that device and the library writes them in place (odise.py:336-372 returns device tensors,
        odise/evaluation/evaluator.py:87-126 consumes them); pictures that are already device tensors are read where they are.  Nothing but
        the few hundred bytes of segment / instance tables crosses PCIe."""
        outs = dropin.TorchOutputs(self.device)
        inputs, keep = [], []
        for x in batched_inputs:
            im = x["image"]
            if torch.is_tensor(im) and im.is_cuda:                      # CHW uint8 / float, 0..255 (DatasetMapper's format, moved by the caller)
                im = im.contiguous() if im.dtype == torch.uint8 else im.float().contiguous()
                keep.append(im)
            inputs.append({**x, "image": im})
        if keep:
            assert len(keep) == len(inputs) and len({t.dtype for t in keep}) == 1, "a batch mixes host and device pictures (or dtypes)"
            torch.cuda.current_stream(self.device).synchronize()
            sizes = [(int(x.get("height", t.shape[-2])), int(x.get("width", t.shape[-1]))) for x, t in zip(inputs, keep)]
            results = hip.infer_device([t.data_ptr() for t in keep], 1 if keep[0].dtype == torch.uint8 else 2, [tuple(t.shape[-2:]) for t in keep], sizes,
                                       to_host=False, alloc=_ready_alloc(outs))
        else:
            results = hip.forward(inputs, to_host=False, alloc=_ready_alloc(outs))
        hip.ctx.sync()                                                  # the tensors are complete when the caller's stream touches them
        out = []
        for i, r in enumerate(results):
            o = {}
            if "sem_seg" in r:
                o["sem_seg"] = outs.tensors[f"sem{i}"]
            if "sem_seg_argmax" in r:
                o["sem_seg_argmax"] = outs.tensors[f"amax{i}"]
            if "panoptic_seg" in r:
                seg = r["panoptic_seg"][0]
                o["panoptic_seg"] = (outs.tensors[f"pan{i}"][: seg.shape[0] * seg.shape[1]].view(seg.shape[0], seg.shape[1]), r["panoptic_seg"][1])
            if "instances" in r:
                inst = r["instances"]
                n = len(inst["scores"])
                o["instances"] = _instances({"pred_masks": outs.tensors[f"masks{i}"][:n], "scores": inst["scores"], "pred_classes": inst["pred_classes"]},
                                            self.device)
            out.append(o)
        return out


def _ready_alloc(outs):
    """TorchOutputs as an allocator that drains torch's stream after each allocation (a recycled block may still be in use there)."""
    def alloc(tag, shape, dtype):
        v = outs(tag, shape, dtype)
        outs.ready()
        return v
    return alloc


def _instances(inst, device):
    """detectron2 `Instances` (maskformer_model.py:369-379: pred_masks, pred_boxes = zeros, scores, pred_classes) when detectron2 is there,
    a namespace with the same fields otherwise."""
    masks = inst["pred_masks"] if torch.is_tensor(inst["pred_masks"]) else torch.from_numpy(inst["pred_masks"]).to(device)
    fields = {"pred_masks": masks, "pred_boxes": torch.zeros(masks.shape[0], 4, device=device), "scores": torch.from_numpy(inst["scores"]).to(device),
              "pred_classes": torch.from_numpy(inst["pred_classes"]).to(device)}
    try:
        from detectron2.structures import Boxes, Instances
        if not isinstance(Instances, type):
            raise ImportError
        r = Instances(tuple(masks.shape[-2:]))
        r.pred_masks, r.pred_boxes, r.scores, r.pred_classes = masks, Boxes(fields["pred_boxes"]), fields["scores"], fields["pred_classes"]
        return r
    except Exception:  # noqa: BLE001
        from types import SimpleNamespace
        return SimpleNamespace(image_size=tuple(masks.shape[-2:]), **fields)


class CategoryODISE(ODISE):
    """odise.py:169-372."""

    def __init__(self, *, category_head=None, clip_head=None, **kwargs):
        super().__init__(**kwargs)
        if category_head is None or clip_head is None:
            raise NotImplementedError("libodise_hip implements the released label model: category_head + clip_head")
        self.category_head, self.clip_head = category_head, clip_head

    def forward(self, batched_inputs):
        return self._forward_eval(batched_inputs, self.category_head)


class CaptionODISE(ODISE):
    """odise.py:375-619 (eval branch 545-619): learned (object, no-object) class head + word bank."""

    _HIP_CLASS = "HipCaptionODISE"

    def __init__(self, *, word_head=None, clip_head=None, grounding_criterion=None, **kwargs):
        super().__init__(**kwargs)
        if word_head is None or clip_head is None:
            raise NotImplementedError("libodise_hip implements the released caption model: word_head + clip_head")
        self.word_head, self.clip_head, self.grounding_criterion = word_head, clip_head, None

    def forward(self, batched_inputs):
        return self._forward_eval(batched_inputs, self.word_head)
