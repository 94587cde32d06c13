"""GPU: the eval-loop helpers are bit-exact integer / byte work: resize against Pillow itself and against the
pinned oracle, confusion matrix and segment-pair histogram against numpy."""
import numpy as np
import pytest

from oracle.eval_ops import pair_histogram, pil_resize_bilinear_u8, resize_shortest_edge_shape, semantic_confusion

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w,oh,ow", [(37, 53, 74, 106), (120, 90, 64, 48), (64, 64, 64, 31), (50, 70, 123, 70), (33, 47, 8, 5), (16, 16, 16, 16),
                                       (480, 640, 1024, 1365), (1500, 2000, 768, 1024)])
def test_resize_is_bit_identical_to_pillow(ctx, h, w, oh, ow):
    from PIL import Image
    rng = np.random.default_rng(h + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = ctx.resize_bilinear_u8(ctx.to_device(img), oh, ow).numpy()
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    np.testing.assert_array_equal(got, ref)
    if h * w < 20000:
        np.testing.assert_array_equal(got, pil_resize_bilinear_u8(img, oh, ow))


def test_shortest_edge_resize_to_chw_float(ctx):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)
    oh, ow = resize_shortest_edge_shape(300, 400, 512, 1280)
    assert (oh, ow) == (512, 683)
    d = ctx.resize_bilinear_u8(ctx.to_device(img), oh, ow)
    chw = ctx.u8_hwc_to_f32_chw(d, 1.0 / 255.0).numpy()
    ref = d.numpy().transpose(2, 0, 1).astype(np.float32) * np.float32(1.0 / 255.0)
    np.testing.assert_array_equal(chw, ref)


@pytest.mark.parametrize("K,h,w", [(5, 33, 47), (150, 256, 300), (133, 512, 512)])
def test_semantic_confusion_matches_numpy(ctx, K, h, w):
    rng = np.random.default_rng(K)
    sem = rng.standard_normal((K, h, w)).astype(np.float32)
    gt = rng.integers(0, K, (h, w)).astype(np.int32)
    gt[rng.random((h, w)) < 0.1] = 255
    ref = semantic_confusion(sem, gt)
    g = gt.copy()
    g[g == 255] = K
    conf = ctx.semantic_confusion(ctx.to_device(sem), ctx.to_device(g))
    conf = ctx.semantic_confusion(ctx.to_device(sem), ctx.to_device(g), conf)   # accumulates
    np.testing.assert_array_equal(conf.numpy(), 2 * ref)


@pytest.mark.parametrize("na,nb", [(7, 12), (101, 200)])
def test_pair_histogram_matches_numpy(ctx, na, nb):
    rng = np.random.default_rng(na)
    a = rng.integers(0, na, (400, 500)).astype(np.int32)
    b = rng.integers(0, nb, (400, 500)).astype(np.int32)
    got = ctx.pair_histogram(ctx.to_device(a), ctx.to_device(b), na, nb).numpy()
    np.testing.assert_array_equal(got, pair_histogram(a, b, na, nb))
