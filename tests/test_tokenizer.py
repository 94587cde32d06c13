"""CPU: mechanics of the CLIP byte-level BPE tokenizer restatement (odise_amd/tokenizer.py).  The real merges file is not available
offline, so the tests drive the algorithm with explicit merge lists: id layout (256 bytes, 256 end-of-word bytes, merges, SOT, EOT =
49406 / 49407 at the real merge count), greedy ranked merging, cleaning, round trip, padding and truncation (open_clip.tokenize)."""
import numpy as np

from odise_amd.tokenizer import SimpleTokenizer, bytes_to_unicode


def test_byte_table_is_a_bijection():
    t = bytes_to_unicode()
    assert len(t) == 256 and len(set(t.values())) == 256 and t[ord("a")] == "a" and t[ord(" ")] != " "


def test_special_ids_at_real_merge_count():
    merges = [(f"x{i}", f"y{i}") for i in range(49152 - 256 - 2)]
    tok = SimpleTokenizer(merges=merges)
    assert (tok.sot, tok.eot) == (49406, 49407) and len(tok.encoder) == 49408
    ids = tok([""])
    assert ids.shape == (1, 77) and ids[0, 0] == 49406 and ids[0, 1] == 49407 and (ids[0, 2:] == 0).all()


def test_ranked_merges_cleaning_roundtrip_truncation():
    merges = [("l", "l"), ("h", "e"), ("he", "ll"), ("hell", "o</w>"), ("c", "a"), ("ca", "t</w>"), ("a", "</w>")]  # "a</w>" is not a pair of symbols: never applies
    tok = SimpleTokenizer(merges=merges[:6])
    e = tok.encoder
    assert tok.encode("Hello") == [e["hello</w>"]]                       # lower-cased, fully merged by rank order
    assert tok.encode("  hello   CAT ") == [e["hello</w>"], e["cat</w>"]]  # whitespace cleaned
    assert tok.encode("hell") == [e["he"], e["l"], e["l</w>"]]            # ("l","l</w>") is not a merge: end-of-word symbol differs
    assert tok.encode("a photo") [0] == e["a</w>"]
    text = "a photo of a cat, isn't it? 42"
    # decode puts a space after every word piece (CLIP's convention): compare modulo spaces
    assert tok.decode(tok.encode(text)).replace(" ", "") == text.replace(" ", "")
    assert tok.decode(tok.encode("café &amp; bar")).strip() == "café & bar"   # utf-8 bytes, html unescape
    ids = tok(["cat", "cat " * 100], context_length=16)
    assert ids.dtype == np.int64 and ids[0].tolist()[:3] == [tok.sot, e["cat</w>"], tok.eot] and (ids[0, 3:] == 0).all()
    assert ids[1, 0] == tok.sot and ids[1, -1] == tok.eot and (ids[1, 1:-1] == e["cat</w>"]).all()
