"""Golden fixtures for the text side, produced by the REFERENCE's own code (build container only):
  * `ClipAdapter._encode_text` (odise/modeling/meta_arch/clip.py:148-162) walking the oracle's narrow CLIP text tower (open_clip is
    absent; the tower keeps its attribute names: token_embedding, positional_embedding, transformer(x, attn_mask=), ln_final,
    text_projection, attn_mask) -> text_encode.npz: pins oracle.clip_text.encode_hidden / encode_text (causal mask, EOT pooling);
  * `prompt_labels` (odise/data/build.py:54-71) -> prompt_labels.json: pins odise_amd.checkpoint.prompt_labels (product host code);
  * `ODISEHandler` / `StableDiffusionHandler` (odise/utils/file_io.py:22-96) -> file_io.json: pins odise_amd.checkpoint.resolve.

    python tests/golden/make_golden_text.py
"""
import json
import os
import sys

import numpy as np
import torch
from torch import nn

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_stubs  # noqa: E402

ref_stubs.install()
from odise.data.build import get_openseg_labels, prompt_labels  # noqa: E402
from odise.modeling.meta_arch.clip import ClipAdapter  # noqa: E402
from odise.utils.file_io import ODISEHandler, StableDiffusionHandler  # noqa: E402

ref_stubs.seal()
from oracle.clip_text import EOT, SOT, CLIPText, init_synthetic_  # noqa: E402


def main():
    model = init_synthetic_(CLIPText(vocab_size=49408, context_length=77, width=64, layers=2, heads=2, output_dim=32), seed=7).eval()
    view = nn.Module()                                                     # the oracle tower under open_clip's attribute names
    view.token_embedding, view.transformer, view.ln_final = model.token_embedding, model.transformer, model.ln_final
    view.positional_embedding, view.text_projection = model.positional_embedding, model.text_projection
    view.register_buffer("attn_mask", model.attn_mask(), persistent=False)    # open_clip keeps the causal mask as a buffer
    adapter = ClipAdapter.__new__(ClipAdapter)
    nn.Module.__init__(adapter)
    adapter.clip = view
    g = torch.Generator().manual_seed(3)
    tokens = torch.zeros(6, 77, dtype=torch.long)
    for i, n in enumerate((1, 3, 7, 20, 40, 75)):                           # <SOT> n word tokens <EOT>, zero padded (open_clip.tokenize)
        tokens[i, 0] = SOT
        tokens[i, 1:1 + n] = torch.randint(1000, 40000, (n,), generator=g)
        tokens[i, 1 + n] = EOT
    with torch.no_grad():
        embed, hidden = adapter._encode_text(tokens)
    np.savez_compressed(os.path.join(HERE, "text_encode.npz"), tokens=tokens.numpy(), embed=embed.numpy(), hidden=hidden.numpy().astype(np.float16))
    labels = [["person", "child"], ["sky"], ["tree", "trees", "bush"], ["traffic light"]]
    out = {str(p): prompt_labels(labels, p) for p in (None, "a", "photo", "scene")}
    json.dump({"labels": labels, "prompted": out}, open(os.path.join(HERE, "prompt_labels.json"), "w"), indent=1)
    # odise:// and sd:// resolution (odise/utils/file_io.py:22-96): the name -> URL tables and the model-zoo rule of _get_local_path
    import tempfile
    import odise.utils.file_io as fio
    with tempfile.TemporaryDirectory() as zoo:
        os.environ["ODISE_MODEL_ZOO"] = zoo
        fio.PathManager = type("PM", (), {"get_local_path": staticmethod(lambda path, **kw: path)})   # the handlers' last step: identity here
        local = {}
        for h in (ODISEHandler(), StableDiffusionHandler()):
            for name, url in h.URLS.items():
                open(os.path.join(zoo, os.path.basename(url)), "w").close()
                local[h.PREFIX + name] = os.path.relpath(h._get_local_path(h.PREFIX + name), zoo)
    json.dump({"odise": ODISEHandler.URLS, "sd": StableDiffusionHandler.URLS, "zoo_relative": local}, open(os.path.join(HERE, "file_io.json"), "w"), indent=1)
    # label files (odise/data/build.py:17-51): only counts and a digest are recorded - the files themselves stay in the reference
    import hashlib
    summary = {}
    for ds in ("coco_panoptic", "ade20k_150", "ade20k_847", "lvis_1203"):
        for pe in (False, True):
            ls = get_openseg_labels(ds, prompt_engineered=pe)
            summary[f"{ds}{'_with_prompt_eng' if pe else ''}"] = dict(categories=len(ls), strings=sum(len(l) for l in ls),
                                                                     sha256=hashlib.sha256(json.dumps(ls).encode()).hexdigest())
    json.dump(summary, open(os.path.join(HERE, "openseg_labels.json"), "w"), indent=1)
    print({k: v[0] for k, v in out.items()}, embed.shape, len(local), summary["coco_panoptic_with_prompt_eng"])


if __name__ == "__main__":
    main()
