"""Drop-in boundary on the reference's side: importable modules at the REFERENCE's dotted paths whose classes keep the
reference's names, keyword arguments, attribute tree, state-dict keys and call signatures, and run on libodise_hip.so.

    PYTHONPATH=$(python -m odise_amd.dropin):$PYTHONPATH python demo/demo.py ...        # or tools/train_net.py --eval-only

`odise_amd/dropin/overlay/` holds package shells `odise/...`, `mask2former/...` and the module `MultiScaleDeformableAttention` (the
reference's compiled op, `MSDA`).  Each shell chains the reference's own directory of the same package behind its own (`chain_reference`),
so everything this repository does not replace - configs, data, checkpoint, evaluation, the wrappers - is still imported from the
reference checkout, unchanged, while the dotted paths the LazyConfig files name
(configs/common/models/mask_generator_with_label.py:15-26, odise_with_label.py:12-13) resolve here.

The classes are `torch.nn.Module`s that only HOLD state: parameters with the reference's names and shapes (so `ODISECheckpointer.load`
/ `load_state_dict` work and `state_dict()` has the reference's keys), the constructor arguments and the open-vocabulary attributes the
wrappers swap (`test_labels`, `metadata`, `num_classes`, ...).  `forward` hands device pointers to the C ABI; on the fused path
(`CategoryODISE.forward` -> `odise_hip_infer`) nothing is computed by PyTorch.  Two classes, when called STAND-ALONE, finish small host-side
arithmetic in torch on CPU tensors: `PoolingCLIPHead.forward` (cosine logits of 100 x K embeddings and the geometric ensemble, after the
library's MaskCLIP tower) and `PooledMaskEmbed.forward` (its few weights are uploaded once per parameter version); the fused model does the same arithmetic in
`odise_hip_classify` / `odise_hip_head_forward`.  Frozen Stable-Diffusion / CLIP weights never enter a state dict (as in the reference, helper.py:35-46): they come from the
checkpoint files named by `init_checkpoint` / `clip_model_name` through odise_amd.checkpoint, or from `set_frozen_state` (tests,
synthetic benchmarks).  One process drives one GPU through one library context (`get_context`).
"""
from __future__ import annotations

import os
import sys
from typing import Dict, Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OVERLAY_DIR = os.path.join(HERE, "overlay")

_ctx = None
_frozen: Optional[Dict[str, np.ndarray]] = None
_text = None            # (tokenizer, text_encoder) for label strings -> CLIP text embeddings


def chain_reference(pkg_name: str, pkg_path) -> list:
    """__path__ of an overlay package: this repository's directory first, then every other directory that provides the same package
    (the reference checkout on sys.path, or its parent package's __path__), so sub-modules not replaced here load from the reference."""
    own = [os.path.abspath(p) for p in pkg_path]
    parts = pkg_name.split(".")
    if len(parts) == 1:
        roots = [p or os.getcwd() for p in sys.path]
    else:
        roots = list(getattr(sys.modules.get(".".join(parts[:-1])), "__path__", []))
    out = list(own)
    for r in roots:
        cand = os.path.abspath(os.path.join(r, parts[-1] if len(parts) > 1 else parts[0]))
        if os.path.isdir(cand) and cand not in out and not cand.startswith(OVERLAY_DIR):
            out.append(cand)
    return out


def get_context():
    """The process's library context (device = LOCAL_RANK, one process per GPU)."""
    global _ctx
    if _ctx is None:
        from ..runtime import Context
        _ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _ctx


def set_context(ctx) -> None:
    global _ctx
    _ctx = ctx


def set_frozen_state(state: Optional[Dict[str, "np.ndarray"]]) -> None:
    """Frozen weights keyed like the real files: `model.diffusion_model.*`, `first_stage_model.*` (SD checkpoint), `clip.visual.*` (OpenAI
    CLIP archive), plus the two derived buffers `backbone.feature_extractor.ldm_extractor.{ldm.uncond_inputs, shared_noise}`."""
    global _frozen
    _frozen = state


def set_text_tools(tokenizer, text_encoder) -> None:
    """What turns label strings into CLIP text embeddings (odise_amd.tokenizer.SimpleTokenizer, odise_amd.text.HipTextEncoder)."""
    global _text
    _text = (tokenizer, text_encoder)


def text_tools():
    return _text


def frozen_state(init_checkpoint: str = "sd://v1-3", clip_model_name: str = "ViT-L-14-336") -> Dict[str, "np.ndarray"]:
    """The frozen weights of LdmImplicitCaptionerExtractor / MaskCLIP: `set_frozen_state`'s, else read from the checkpoint files
    (ldm.py:66-74 `init_checkpoint`, clip.py:31-35 `open_clip ... pretrained="openai"`) via odise_amd.checkpoint."""
    if _frozen is not None:
        return _frozen
    from .. import checkpoint as ck
    ctx = get_context()
    sd = ck.load_sd_checkpoint(init_checkpoint)
    clip = ck.load_openai_clip("clip://" + clip_model_name if "://" not in clip_model_name else clip_model_name)
    state = {k: v for k, v in sd.items() if not k.startswith("cond_stage_model.")}
    state.update({"clip." + k: v for k, v in clip.items()})
    from ..text import HipTextEncoder, empty_prompt_tokens, hf_text_to_openai
    fe = "backbone.feature_extractor.ldm_extractor."
    state[fe + "ldm.uncond_inputs"] = HipTextEncoder(ctx, hf_text_to_openai(sd)).hidden(empty_prompt_tokens(pad_with_eot=True)).astype(np.float32)
    state[fe + "shared_noise"] = ck.shared_noise()
    set_frozen_state(state)
    return state


# ---- torch <-> device buffers ----------------------------------------------------------------------------------------------------------
def to_device(t, dtype=np.float32):
    """A torch tensor (CPU or ROCm) as a contiguous device buffer of `dtype`: (pointer, keep-alive object)."""
    import torch
    ctx = get_context()
    if torch.is_tensor(t):
        if t.is_cuda:
            tt = t.detach().to(getattr(torch, np.dtype(dtype).name)).contiguous()
            torch.cuda.current_stream(t.device).synchronize()
            return tt.data_ptr(), tt
        t = t.detach().cpu().numpy()
    d = ctx.to_device(np.ascontiguousarray(t, dtype))
    return d.ptr, d


def new_output(shape, like=None, dtype=np.float32):
    """An output buffer the library writes: (pointer, fetch() -> torch tensor on `like`'s device)."""
    import torch
    ctx = get_context()
    if like is not None and torch.is_tensor(like) and like.is_cuda:
        out = torch.empty(tuple(int(s) for s in shape), dtype=getattr(torch, np.dtype(dtype).name), device=like.device)

        def fetch():
            ctx.sync()
            return out
        return out.data_ptr(), fetch
    d = ctx.empty(shape, dtype)
    return d.ptr, lambda: torch.from_numpy(d.numpy())


def tensor_view(t):
    """A torch ROCm tensor as a non-owning runtime.DeviceArray over the same memory (keeps the tensor alive)."""
    from ..runtime import DeviceArray
    assert t.is_cuda and t.is_contiguous()
    v = DeviceArray.__new__(DeviceArray)
    v.ctx = get_context()
    v.shape = tuple(int(s) for s in t.shape)
    v.dtype = np.dtype(str(t.dtype).replace("torch.", ""))
    v.nbytes = int(t.numel()) * v.dtype.itemsize
    v.ptr = t.data_ptr()
    v._owned = False
    v._base = t
    return v


class TorchOutputs:
    """`alloc` callable for HipCategoryODISE.forward (pipeline._post_desc): the large per-image outputs are torch tensors on `device`, written
    in place by the library - what the reference's `model(batched_inputs)` returns (odise.py:336-372) without a copy at the edge.  Allocation
    happens on torch's current stream; that stream is drained once before the library (its own stream) writes the buffers."""

    def __init__(self, device):
        import torch
        self.device, self.tensors = device, {}
        self._torch = torch

    def __call__(self, tag, shape, dtype):
        torch = self._torch
        t = torch.empty(tuple(int(s) for s in shape), dtype=getattr(torch, np.dtype(dtype).name), device=self.device)
        self.tensors[tag] = t
        return tensor_view(t)

    def ready(self):
        self._torch.cuda.current_stream(self.device).synchronize()


if __name__ == "__main__":
    print(OVERLAY_DIR)
