"""Pin the JPEG oracle (oracle/jpeg.py) bit-exactly against Pillow = libjpeg-turbo, the decoder behind the reference's
`read_image(file, "RGB")`: sizes that are not MCU multiples, the three luma samplings, grey images, optimised
Huffman tables, restart intervals, low qualities (large coefficients), 1- and 2-sample-wide chroma (plain replication rule) and EXIF
orientations."""
import io

import numpy as np
import pytest
from PIL import Image

from oracle import jpeg as oj


def _picture(h, w, seed, smooth=True):
    rng = np.random.default_rng(seed)
    if smooth:
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([128 + 100 * np.sin(xx / 7.0 + seed) * np.cos(yy / 5.0), 128 + 90 * np.cos(xx / 3.0 - yy / 11.0),
                         (xx * 5 + yy * 3 + seed * 17) % 256], -1)
        img = base + rng.normal(0, 12, (h, w, 3))
    else:
        img = rng.integers(0, 256, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def _jpeg(img, mode="RGB", **kw):
    buf = io.BytesIO()
    Image.fromarray(img if mode == "RGB" else img[..., 0], mode).save(buf, "JPEG", **kw)
    return buf.getvalue()


def _pil(data):
    from PIL import ImageOps
    im = ImageOps.exif_transpose(Image.open(io.BytesIO(data)))
    return np.asarray(im.convert("RGB"))


CASES = [(h, w, sub, q) for (h, w) in ((8, 8), (16, 16), (17, 23), (33, 70), (64, 48), (1, 1), (5, 3), (2, 40), (40, 2), (3, 4), (100, 131))
         for sub, q in ((0, 90), (1, 75), (2, 75), (2, 20))]


@pytest.mark.parametrize("h,w,sub,q", CASES)
def test_matches_pillow(h, w, sub, q):
    data = _jpeg(_picture(h, w, seed=h * 131 + w), quality=q, subsampling=sub)
    np.testing.assert_array_equal(oj.decode(data), _pil(data))


@pytest.mark.parametrize("kw", [dict(quality=3, subsampling=2), dict(quality=100, subsampling=0), dict(quality=60, subsampling=2, optimize=True),
                                dict(quality=85, subsampling=1, restart_marker_blocks=3), dict(quality=85, subsampling=2, restart_marker_rows=1)])
def test_tables_restarts_and_extremes(kw):
    for seed, smooth in ((1, True), (2, False)):
        data = _jpeg(_picture(75, 99, seed, smooth), **kw)
        np.testing.assert_array_equal(oj.decode(data), _pil(data))


def test_grey_and_orientation():
    img = _picture(37, 52, 5)
    data = _jpeg(img, mode="L", quality=80)
    np.testing.assert_array_equal(oj.decode(data), _pil(data))
    for orient in range(1, 9):
        ex = Image.Exif()
        ex[0x0112] = orient
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", quality=90, exif=ex.tobytes())
        data = buf.getvalue()
        assert oj.parse(data)["orientation"] == orient
        np.testing.assert_array_equal(oj.decode(data), _pil(data))


def test_unsupported_files_are_refused():
    img = _picture(32, 32, 9)
    with pytest.raises(oj.Unsupported):
        oj.decode(_jpeg(img, quality=80, progressive=True))
    buf = io.BytesIO()
    Image.fromarray(img).convert("CMYK").save(buf, "JPEG")
    with pytest.raises(oj.Unsupported):
        oj.decode(buf.getvalue())
    with pytest.raises(ValueError):
        oj.decode(b"not a jpeg")
