"""Host logic of the open-vocabulary state protocol (OpenPanopticInference, odise/modeling/wrapper/pano_wrapper.py:20-70): the wrapper
swaps labels / metadata / task switches for one call and the model's own come back afterwards.  No device: the model's device calls
are stubbed, the text encoder is a deterministic fake."""
import os

import numpy as np
import pytest

from odise_amd.pipeline import HipCategoryODISE, HipOpenPanopticInference


class _Tok:
    def __call__(self, texts, context_length=77):
        out = np.zeros((len(texts), context_length), np.int64)
        for i, t in enumerate(texts):
            b = list(t.encode())[:context_length]
            out[i, :len(b)] = b
        return out


class _Enc:
    calls = 0

    def build_text_embed(self, rows):
        _Enc.calls += 1
        return np.stack([np.full(4, float(r.sum() % 97), np.float32) for r in rows])


class _Model(HipCategoryODISE):
    def __init__(self):                                                   # no context, no weights: only the state protocol
        self.semantic_on, self.panoptic_on, self.instance_on, self.test_topk_per_image = True, True, True, 100
        self.num_classes, self.thing_ids, self.metadata, self.test_labels = 0, set(), None, None
        self._alpha, self._beta, self._banks, self._vocab_cache = 0.3, 0.7, None, {}
        self.uploads = []

    def set_vocabulary(self, cat, clp, sizes, overlap, thing_ids, alpha=0.3, beta=0.7):
        self.uploads.append((np.asarray(cat).copy(), list(sizes), list(overlap), set(thing_ids)))
        self.num_classes, self.thing_ids = len(sizes), set(thing_ids)
        self._banks, self.test_labels = (cat, clp, sizes, overlap), None

    def forward(self, batched_inputs):
        return [dict(labels=self.test_labels, K=self.num_classes, things=set(self.thing_ids), inst=self.instance_on,
                     topk=self.test_topk_per_image) for _ in batched_inputs]


A = [["person", "child"], ["sky"], ["car"]]
B = [["cat"], ["sofa", "couch"]]


def test_wrapper_swaps_and_restores_label_vocabulary():
    m = _Model()
    m.attach_text(_Tok(), _Enc(), train_labels=[["sky"], ["cat"]])
    m.set_labels(A, thing_ids={0, 2})
    assert m.num_classes == 3 and m.test_labels == A and m.uploads[-1][2] == [0, 1, 0] and m.uploads[-1][1] == [2, 1, 1]
    w = HipOpenPanopticInference(m, B, metadata={"thing_ids": [0]}, instance_on=False, test_topk_per_image=7)
    assert w.num_classes == 2 and w.open_state_dict["sem_seg_head.num_classes"] == 2
    assert {k.rsplit(".", 1)[-1] for k in w.open_state_dict} == {"test_labels", "metadata", "num_classes", "semantic_on", "instance_on",
                                                                  "panoptic_on", "test_topk_per_image"}
    out = w([{"image": None}])[0]
    assert out == dict(labels=B, K=2, things={0}, inst=False, topk=7)
    own = m.forward([0])[0]
    assert own == dict(labels=A, K=3, things={0, 2}, inst=True, topk=100) and m.metadata is None
    n_up, n_enc = len(m.uploads), _Enc.calls
    w([{"image": None}])                                                  # second call: banks come from the per-label-set cache
    assert _Enc.calls == n_enc and len(m.uploads) == n_up + 2
    assert m.uploads[-1][2] == [0, 1, 0] and m.uploads[-2][2] == [1, 0]    # seen/unseen flags follow the training labels


def test_wrapper_restores_banks_set_without_labels():
    m = _Model()
    m.attach_text(_Tok(), _Enc())
    cat = np.arange(12, dtype=np.float32).reshape(3, 4)
    m.set_vocabulary(cat, cat, [1, 2], [0, 0], {1})
    w = HipOpenPanopticInference(m, B)
    assert w([{"image": None}])[0]["K"] == 2
    assert m.num_classes == 2 and m.thing_ids == {1} and m.test_labels is None
    np.testing.assert_array_equal(m.uploads[-1][0], cat)


def test_labels_without_a_text_encoder_are_refused():
    m = _Model()
    with pytest.raises(RuntimeError, match="attach_text"):
        m.set_labels(A, thing_ids={0})
    assert m.uploads == [] and m.test_labels is None


def test_wrapper_restores_after_an_exception():
    m = _Model()
    m.attach_text(_Tok(), _Enc())
    m.set_labels(A, thing_ids={0})
    w = HipOpenPanopticInference(m, B, panoptic_on=False)
    m.forward = lambda x: (_ for _ in ()).throw(RuntimeError("boom"))
    try:
        w([{"image": None}])
    except RuntimeError:
        pass
    assert m.test_labels == A and m.panoptic_on is True


REF_WRAPPER = "/root/reference/odise/modeling/wrapper/pano_wrapper.py"


@pytest.mark.skipif(not os.path.exists(REF_WRAPPER), reason="the reference checkout exists only in the build container")
def test_the_reference_wrapper_itself_drives_this_model():
    """Drop-in at the caller's side: the REFERENCE's own `OpenPanopticInference` (loaded from its file, it needs nothing but torch) wraps
    this model class through `open_state_dict / load_open_state_dict / __call__` and behaves exactly like `HipOpenPanopticInference`."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_pano_wrapper", REF_WRAPPER)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    m = _Model()
    m.attach_text(_Tok(), _Enc(), train_labels=[["sky"], ["cat"]])
    m.set_labels(A, thing_ids={0, 2})
    ref = mod.OpenPanopticInference(m, B, metadata={"thing_ids": [0]}, instance_on=False, test_topk_per_image=7).eval()
    mine = HipOpenPanopticInference(m, B, metadata={"thing_ids": [0]}, instance_on=False, test_topk_per_image=7)
    assert dict(ref.open_state_dict) == mine.open_state_dict
    assert ref([{"image": None}])[0] == dict(labels=B, K=2, things={0}, inst=False, topk=7)
    assert m.forward([0])[0] == dict(labels=A, K=3, things={0, 2}, inst=True, topk=100)
    assert ref([{"image": None}]) == mine([{"image": None}])
