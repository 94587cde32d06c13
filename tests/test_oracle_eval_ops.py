"""CPU: pin the input-resize oracle against the Pillow installed here (bit-exact) and check the evaluator reductions against their
numpy definitions (oracle/eval_ops.py)."""
import numpy as np
import pytest

from oracle.eval_ops import pair_histogram, pil_resize_bilinear_u8, resize_shortest_edge_shape, semantic_confusion

PIL = pytest.importorskip("PIL")


@pytest.mark.parametrize("h,w,oh,ow", [(37, 53, 74, 106), (120, 90, 64, 48), (64, 64, 64, 31), (50, 70, 123, 70), (33, 47, 8, 5), (16, 16, 16, 16),
                                       (211, 173, 96, 128)])
def test_resize_matches_pillow_bit_exactly(h, w, oh, ow):
    from PIL import Image
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    got = pil_resize_bilinear_u8(img, oh, ow)
    np.testing.assert_array_equal(got, ref)


def test_resize_shortest_edge_shape():
    assert resize_shortest_edge_shape(480, 640) == (1024, 1365)
    assert resize_shortest_edge_shape(640, 480) == (1365, 1024)
    assert resize_shortest_edge_shape(500, 2000) == (640, 2560)       # max_size caps the long edge
    assert resize_shortest_edge_shape(1024, 1024) == (1024, 1024)


def test_reductions():
    rng = np.random.default_rng(0)
    K = 5
    sem = rng.standard_normal((K, 12, 9)).astype(np.float32)
    gt = rng.integers(0, K, (12, 9))
    gt[0, :3] = 255
    conf = semantic_confusion(sem, gt)
    assert conf.shape == (K + 1, K + 1) and conf.sum() == 12 * 9 and conf[:, K].sum() == 3 and conf[K].sum() == 0
    pred = sem.argmax(0)
    assert conf[2, 3] == ((pred == 2) & (gt == 3)).sum()
    a, b = rng.integers(0, 4, (12, 9)), rng.integers(0, 7, (12, 9))
    h = pair_histogram(a, b, 4, 7)
    assert h.sum() == 108 and h[1, 2] == ((a == 1) & (b == 2)).sum()
