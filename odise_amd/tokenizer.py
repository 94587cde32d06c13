"""CLIP byte-level BPE tokenizer (host side of the text-bank builder).

The reference tokenises with `open_clip.tokenize` (open-clip-torch==2.0.2, reference setup.py:81; call sites
odise/modeling/meta_arch/clip.py:66, 166), a dependency that is not vendored under /root/reference.  This module restates its published
algorithm (OpenAI CLIP `simple_tokenizer.py`): text is whitespace-cleaned and lower-cased, split by the CLIP regular expression, every
piece is mapped byte-wise to printable unicode symbols, merged greedily by the ranked merges of `bpe_simple_vocab_16e6.txt.gz`
(lines 1 .. 49152-256-2), and wrapped as <start_of_text> ids <end_of_text>, zero padded to the context length; over-long prompts are
truncated and closed with <end_of_text>.  The merges file is NOT shipped here (no network): pass its path or set ODISE_CLIP_BPE /
ODISE_MODEL_ZOO.  `ftfy` (unicode repair) is not installed; plain-ASCII prompts - every label file of the reference - are unaffected.
"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import numpy as np
import regex as re

SOT_TEXT, EOT_TEXT = "<start_of_text>", "<end_of_text>"


@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    """The reversible byte -> printable unicode table of GPT-2 / CLIP."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


def _pairs(word: Tuple[str, ...]):
    return set(zip(word[:-1], word[1:]))


def default_bpe_path() -> str:
    cands = [os.environ.get("ODISE_CLIP_BPE", "")]
    zoo = os.environ.get("ODISE_MODEL_ZOO", "")
    if zoo:
        cands.append(os.path.join(zoo, "bpe_simple_vocab_16e6.txt.gz"))
    for c in cands:
        if c and os.path.exists(c):
            return c
    raise FileNotFoundError("CLIP merges file bpe_simple_vocab_16e6.txt.gz not found: set ODISE_CLIP_BPE or put it under ODISE_MODEL_ZOO")


class SimpleTokenizer:
    def __init__(self, bpe_path: str = None, merges: Sequence[Tuple[str, str]] = None):
        """`merges`: explicit ranked merge list (tests); otherwise read from the CLIP merges file."""
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        if merges is None:
            with gzip.open(bpe_path or default_bpe_path()) as f:
                lines = f.read().decode("utf-8").split("\n")
            merges = [tuple(m.split()) for m in lines[1:49152 - 256 - 2 + 1]]
        merges = [tuple(m) for m in merges]
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(m) for m in merges]
        vocab += [SOT_TEXT, EOT_TEXT]
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.decoder = {v: k for k, v in self.encoder.items()}
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {SOT_TEXT: SOT_TEXT, EOT_TEXT: EOT_TEXT}
        self.pat = re.compile(r"""<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", re.IGNORECASE)
        self.sot, self.eot = self.encoder[SOT_TEXT], self.encoder[EOT_TEXT]

    def bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = _pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                except ValueError:
                    new.extend(word[i:])
                    break
                new.extend(word[i:j])
                i = j
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
            if len(word) == 1:
                break
            pairs = _pairs(word)
        out = " ".join(word)
        self.cache[token] = out
        return out

    @staticmethod
    def clean(text: str) -> str:
        text = html.unescape(html.unescape(text)).strip()      # basic_clean without ftfy
        return re.sub(r"\s+", " ", text).strip().lower()       # whitespace_clean + lower

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for tok in re.findall(self.pat, self.clean(text)):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(tok).split(" "))
        return ids

    def decode(self, ids: Iterable[int]) -> str:
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def __call__(self, texts: Union[str, Sequence[str]], context_length: int = 77) -> np.ndarray:
        """open_clip.tokenize: int64 [N, context_length], zero padded; truncated prompts end with <end_of_text>."""
        if isinstance(texts, str):
            texts = [texts]
        out = np.zeros((len(texts), context_length), np.int64)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, :len(ids)] = ids
        return out
