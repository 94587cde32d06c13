"""Weight ingestion and vocabulary building: turn the reference's three weight sources into the flat state
dict that `HipCategoryODISE` uploads through `odise_hip_load_weight`, with the reference's own key names.

  * `sd://v1-3` / a path  -> `sd-v1-3.ckpt`, a torch pickle {"state_dict": {...}} with `model.diffusion_model.*` (UNet),
    `first_stage_model.*` (VAE) and `cond_stage_model.transformer.text_model.*` (HF CLIP text encoder)  (ldm.py:66-74, 121-122)
  * OpenAI `ViT-L-14-336px.pt` (TorchScript archive or plain state dict): `visual.*`, `token_embedding.weight`, `positional_embedding`,
    `transformer.resblocks.*`, `ln_final.*`, `text_projection`, `logit_scale`  (open_clip `pretrained="openai"`, clip.py:77-97)
  * `odise://Panoptic/odise_label_coco_50e` / a path -> {"model": {...}} with the trainable tensors of SURVEY.md Appendix B
    (odise/checkpoint/odise_checkpointer.py:54-140; frozen sub-networks are absent from it: helper.py:44-46, clip.py:120-122)

`odise://` and `sd://` resolve like odise/utils/file_io.py:22-96: `$ODISE_MODEL_ZOO/<basename of the release URL>`.  Nothing is
downloaded (no network): a missing file raises FileNotFoundError naming the URL to fetch.

The two constants the reference derives at construction time are derived here too, on the device:
  * `ldm_extractor.ldm.uncond_inputs` = the SD text encoder applied to "" (ldm.py:116)  -> `HipTextEncoder.hidden`
  * `ldm_extractor.shared_noise` = `torch.randn(1, 4, 64, 64, generator=manual_seed(42))` (ldm.py:273-277)
torch is used only to unpickle the files and for that seeded generator.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

ODISE_URLS = {
    "Panoptic/odise_caption_coco_50e": "https://github.com/NVlabs/ODISE/releases/download/v1.0.0/odise_caption_coco_50e-853cc971.pth",
    "Panoptic/odise_label_coco_50e": "https://github.com/NVlabs/ODISE/releases/download/v1.0.0/odise_label_coco_50e-b67d2efc.pth",
}
SD_URLS = {
    "v1-3": "https://huggingface.co/CompVis/stable-diffusion-v-1-3-original/resolve/main/sd-v1-3.ckpt",
    "v1-4": "https://huggingface.co/CompVis/stable-diffusion-v-1-4-original/resolve/main/sd-v1-4.ckpt",
    "v1-5": "https://huggingface.co/runwayml/stable-diffusion-v1-5/resolve/main/v1-5-pruned-emaonly.ckpt",
}
OPENAI_CLIP_URLS = {
    "ViT-L-14-336": "https://openaipublic.azureedge.net/clip/models/3035c92b350959924f9f00213499208652fc7ea050643e8b385c2dac08641f02/ViT-L-14-336px.pt",
    "ViT-L-14": "https://openaipublic.azureedge.net/clip/models/b8cca3fd41ae0c99ba7e8951adf17d267cdb84cd88be6f7c2e0eca1737a03836/ViT-L-14.pt",
}


def resolve(uri: str) -> str:
    """Local path of a checkpoint URI (`odise://…`, `sd://…`, `clip://…` or a plain path).  file_io.py:22-96 semantics, offline."""
    for prefix, table in (("odise://", ODISE_URLS), ("sd://", SD_URLS), ("clip://", OPENAI_CLIP_URLS)):
        if uri.startswith(prefix):
            name = uri[len(prefix):]
            if name not in table:
                raise KeyError(f"{name} is not a valid {prefix} model: {sorted(table)}")
            url = table[name]
            zoo = os.environ.get("ODISE_MODEL_ZOO", "")
            local = os.path.join(zoo, os.path.basename(url)) if zoo else ""
            if local and os.path.exists(local):
                return local
            raise FileNotFoundError(f"{uri}: {os.path.basename(url)} not found under ODISE_MODEL_ZOO={zoo!r}; fetch {url} (no network here)")
    if not os.path.exists(uri):
        raise FileNotFoundError(uri)
    return uri


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().float().numpy()
    return np.asarray(v)


def _torch_load(path: str):
    import torch
    try:
        return torch.load(path, map_location="cpu", weights_only=False)
    except TypeError:  # older torch
        return torch.load(path, map_location="cpu")


def load_sd_checkpoint(uri: str) -> Dict[str, np.ndarray]:
    """UNet + VAE + cond-stage tensors of an SD-v1 checkpoint, keys unchanged; EMA copies and loss buffers are dropped."""
    ck = _torch_load(resolve(uri))
    sd = ck["state_dict"] if "state_dict" in ck else ck
    keep = ("model.diffusion_model.", "first_stage_model.", "cond_stage_model.")
    out = {k: _np(v) for k, v in sd.items() if k.startswith(keep)}
    if not any(k.startswith("model.diffusion_model.") for k in out):
        raise ValueError(f"{uri}: no model.diffusion_model.* tensors (not an SD-v1 checkpoint?)")
    return out


def load_openai_clip(uri: str) -> Dict[str, np.ndarray]:
    """OpenAI CLIP weights with their native names (TorchScript archive or plain state dict)."""
    import torch
    path = resolve(uri)
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except RuntimeError:
        sd = _torch_load(path)
        sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd.state_dict()
    out = {k: _np(v) for k, v in sd.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    if "visual.proj" not in out:
        raise ValueError(f"{uri}: no visual.proj (not an OpenAI CLIP ViT checkpoint?)")
    return out


def load_odise_checkpoint(uri: str) -> Dict[str, np.ndarray]:
    """Trainable tensors of an ODISE checkpoint ({"model": ...}); training-only `criterion.*` buffers are dropped."""
    ck = _torch_load(resolve(uri))
    sd = ck["model"] if "model" in ck else ck
    return {k: _np(v) for k, v in sd.items() if not k.startswith("criterion.")}


def shared_noise(latent_dim: int = 4, hw: Tuple[int, int] = (64, 64)) -> np.ndarray:
    """ldm.py:273-277: the fixed noise every image shares."""
    import torch
    return torch.randn(1, latent_dim, *hw, generator=torch.Generator().manual_seed(42)).numpy()


def assemble_state(ctx, sd_uri: str = "sd://v1-3", clip_uri: str = "clip://ViT-L-14-336", odise_uri: str = "odise://Panoptic/odise_label_coco_50e",
                   sd_state=None, clip_state=None, odise_state=None) -> Dict[str, np.ndarray]:
    """The flat state dict `HipCategoryODISE(ctx, state)` expects, from the three sources (already loaded dicts may be passed).
    Key layout (= oracle/ldm_extractor.py:export_state, the names the C side looks up):
      model.diffusion_model.*, first_stage_model.*            <- SD checkpoint
      clip.visual.* (+ the text tower as clip.*)              <- OpenAI CLIP
      backbone.*, sem_seg_head.*, category_head.*             <- ODISE checkpoint
      backbone.feature_extractor.ldm_extractor.{ldm.uncond_inputs, shared_noise}   <- derived (see module docstring)"""
    from .text import HipTextEncoder, empty_prompt_tokens, hf_text_to_openai
    sd = sd_state if sd_state is not None else load_sd_checkpoint(sd_uri)
    clip = clip_state if clip_state is not None else load_openai_clip(clip_uri)
    od = odise_state if odise_state is not None else load_odise_checkpoint(odise_uri)
    state: Dict[str, np.ndarray] = {}
    state.update({k: v for k, v in sd.items() if not k.startswith("cond_stage_model.")})
    state.update({"clip." + k: v for k, v in clip.items()})
    state.update(od)
    fe = "backbone.feature_extractor.ldm_extractor."
    enc = HipTextEncoder(ctx, hf_text_to_openai(sd))
    state[fe + "ldm.uncond_inputs"] = enc.hidden(empty_prompt_tokens(pad_with_eot=True)).astype(np.float32)
    state[fe + "shared_noise"] = shared_noise()
    required = ("model.diffusion_model.time_embed.0.weight", "first_stage_model.encoder.conv_in.weight", "clip.visual.proj",
                "backbone.feature_extractor.clip_project.linear.weight", "backbone.feature_extractor.alpha_cond",
                "sem_seg_head.predictor.query_feat.weight", "category_head.text_proj.weight", "category_head.null_embed")
    missing = [k for k in required if k not in state]
    if missing:
        raise KeyError(f"assembled state lacks {missing}")
    return state


# ---- vocabulary (the open-vocabulary label sets of an evaluation / a demo run) --------------------------------------------------------
def prompt_labels(labels: Sequence[Sequence[str]], prompt: Optional[str]) -> List[List[str]]:
    """odise/data/build.py:54-71."""
    if prompt is None:
        return [list(l) for l in labels]
    assert prompt in ("a", "photo", "scene")
    fmt = {"a": "a {}", "photo": "a photo of a {}.", "scene": "a photo of a {} in the scene."}[prompt]
    return [[fmt.format(s) for s in l] for l in labels]


def read_openseg_labels(path: str, invalid_name: str = "invalid_class_id") -> List[List[str]]:
    """The label files the reference ships under odise/data/datasets/openseg_labels/ (`<id>:<name>[,<synonym>...]` per line; the
    `*_with_prompt_eng.txt` variants carry the synonyms) -> nested label list, as `get_openseg_labels` returns it
    (odise/data/build.py:17-51: lines whose name is `invalid_class_id` are skipped, ids are not used for ordering)."""
    out = []
    with open(path, "r") as f:
        for line in f.read().splitlines():
            _id, name = line.split(":")
            if name == invalid_name:
                continue
            int(_id)                                                   # malformed ids are errors there too
            out.append(name.split(","))
    return out


def default_train_labels() -> List[List[str]]:
    """The training vocabulary PoolingCLIPHead falls back to (odise.py:1446-1447: `get_openseg_labels("coco_panoptic",
    prompt_engineered=True)`): read from the label file the reference ships, found in `$ODISE_OPENSEG_LABELS` or in an `odise` package
    directory on sys.path (odise/data/datasets/openseg_labels/coco_panoptic_with_prompt_eng.txt).  There is no silent substitute: the
    seen / unseen split decides the ensemble weights (alpha for seen, beta for unseen categories)."""
    import sys
    name = "coco_panoptic_with_prompt_eng.txt"
    cands = [os.path.join(os.environ["ODISE_OPENSEG_LABELS"], name)] if os.environ.get("ODISE_OPENSEG_LABELS") else []
    cands += [os.path.join(p or ".", "odise", "data", "datasets", "openseg_labels", name) for p in sys.path]
    for c in cands:
        if os.path.exists(c):
            return read_openseg_labels(c)
    raise FileNotFoundError(f"{name} not found: pass train_labels explicitly, or point ODISE_OPENSEG_LABELS at the reference's "
                            "odise/data/datasets/openseg_labels directory (or put the reference checkout on sys.path)")


def ensemble_max(logits, group_sizes):
    """helper.py:79-109 `ensemble_logits_with_labels(..., ensemble_method="max")` on a torch tensor [..., K_tot] -> [..., K]."""
    import torch
    out, start = [], 0
    for n in group_sizes:
        out.append(logits[..., start:start + n].max(dim=-1).values)
        start += n
    assert start == logits.shape[-1]
    return torch.stack(out, dim=-1)


def category_overlapping_mask(train_labels: Sequence[Sequence[str]], test_labels: Sequence[Sequence[str]]) -> np.ndarray:
    """PoolingCLIPHead.forward, odise.py:1479-1491: 1 where a test category shares a name with any training category."""
    train = {s for l in train_labels for s in l}
    return np.array([int(not train.isdisjoint(set(l))) for l in test_labels], np.int32)


def build_vocabulary(test_labels: Sequence[Sequence[str]], tokenizer, text_encoder, train_labels: Optional[Sequence[Sequence[str]]] = None,
                     category_prompt: Optional[str] = None, clip_prompt: Optional[str] = "photo", clip_text_encoder=None):
    """Arguments of `HipCategoryODISE.set_vocabulary` for a label set: (cat_text [K_tot, dim], clip_text [K_tot, dim], group_sizes [K],
    overlap [K]).  One un-normalised CLIP text embedding per prompt string, strings grouped per category
    (CategoryEmbed.forward odise.py:1296-1307 with prompt=None - the label lists are already prompt-engineered synonyms -,
    PoolingCLIPHead.forward odise.py:1468-1495 with prompt="photo", build_clip_text_embed clip.py:29-73; both ViT-L-14-336)."""
    sizes = np.array([len(l) for l in test_labels], np.int32)
    flat = lambda ls: [s for l in ls for s in l]
    cat = text_encoder.build_text_embed(tokenizer(flat(prompt_labels(test_labels, category_prompt))))
    enc2 = clip_text_encoder or text_encoder
    if enc2 is text_encoder and clip_prompt == category_prompt:
        clp = cat
    else:
        clp = enc2.build_text_embed(tokenizer(flat(prompt_labels(test_labels, clip_prompt))))
    overlap = category_overlapping_mask(train_labels if train_labels is not None else default_train_labels(), test_labels)
    return cat, clp, sizes, overlap
