"""GPU: JPEG decoding through the C ABI (`odise_hip_jpeg_decode`: host Huffman decoding, IDCT / fancy upsampling / colour conversion /
EXIF transpose on the device) is bit-identical to Pillow = what the reference's `read_image(file, "RGB")` returns, and to the pinned
oracle; `HipDatasetMapper` (decode + ResizeShortestEdge) reproduces the DatasetMapper's image."""
import io

import numpy as np
import pytest
from PIL import Image

from odise_amd._lib import UnsupportedInput
from odise_amd.ingest import HipDatasetMapper, resize_shortest_edge_shape
from oracle import jpeg as oj
from tests.test_oracle_jpeg import CASES, _jpeg, _picture, _pil

pytestmark = pytest.mark.gpu


def test_case_matrix_bit_exact(ctx):
    for h, w, sub, q in CASES:
        data = _jpeg(_picture(h, w, seed=h * 131 + w), quality=q, subsampling=sub)
        got = ctx.jpeg_decode(data).numpy()
        np.testing.assert_array_equal(got, _pil(data), err_msg=f"{h}x{w} subsampling {sub} quality {q}")
        np.testing.assert_array_equal(got, oj.decode(data))


@pytest.mark.parametrize("kw", [dict(quality=3, subsampling=2), dict(quality=100, subsampling=0), dict(quality=60, subsampling=2, optimize=True),
                                dict(quality=85, subsampling=1, restart_marker_blocks=3), dict(quality=85, subsampling=2, restart_marker_rows=1)])
def test_tables_restarts_and_extremes(ctx, kw):
    for seed, smooth in ((1, True), (2, False)):
        data = _jpeg(_picture(75, 99, seed, smooth), **kw)
        np.testing.assert_array_equal(ctx.jpeg_decode(data).numpy(), _pil(data))


def test_grey_orientation_and_buffer_reuse(ctx):
    img = _picture(37, 52, 5)
    data = _jpeg(img, mode="L", quality=80)
    np.testing.assert_array_equal(ctx.jpeg_decode(data).numpy(), _pil(data))
    for orient in range(1, 9):
        ex = Image.Exif()
        ex[0x0112] = orient
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", quality=90, exif=ex.tobytes())
        data = buf.getvalue()
        np.testing.assert_array_equal(ctx.jpeg_decode(data).numpy(), _pil(data), err_msg=f"orientation {orient}")
        raw = ctx.jpeg_decode(data, apply_orientation=False).numpy()
        np.testing.assert_array_equal(raw, np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))
    big = _jpeg(_picture(768, 1024, 7), quality=85, subsampling=2)       # staging buffers grow, then are reused by a smaller image
    np.testing.assert_array_equal(ctx.jpeg_decode(big).numpy(), _pil(big))
    small = _jpeg(_picture(40, 24, 8), quality=70, subsampling=1)
    np.testing.assert_array_equal(ctx.jpeg_decode(small).numpy(), _pil(small))
    np.testing.assert_array_equal(ctx.jpeg_decode(big).numpy(), _pil(big))


def test_progressive_files(ctx):
    """Progressive JPEG (SOF2): the host half assembles the coefficients over the scans, the device half is the same."""
    for (h, w, kw) in [(75, 99, dict(quality=60, subsampling=1, optimize=True)), (100, 131, dict(quality=20, subsampling=2)),
                       (64, 48, dict(quality=90, subsampling=0)), (90, 120, dict(quality=85, subsampling=2, restart_marker_blocks=7))]:
        data = _jpeg(_picture(h, w, seed=h + w), progressive=True, **kw)
        np.testing.assert_array_equal(ctx.jpeg_decode(data).numpy(), _pil(data), err_msg=f"{h}x{w} {kw}")
    grey = _jpeg(_picture(61, 83, 4), mode="L", quality=70, progressive=True)
    np.testing.assert_array_equal(ctx.jpeg_decode(grey).numpy(), _pil(grey))


def test_unsupported_and_malformed(ctx):
    img = _picture(32, 32, 9)
    buf = io.BytesIO()
    Image.fromarray(img).convert("CMYK").save(buf, "JPEG")
    with pytest.raises(UnsupportedInput):
        ctx.jpeg_decode(buf.getvalue())
    with pytest.raises(RuntimeError):
        ctx.jpeg_decode(b"\xff\xd8\xff\xd9")
    good = _jpeg(img, quality=80)
    start = oj.parse(good)["data_start"]
    cut = good[:start + 40]                                               # truncated scan: decodes (zero bits), like libjpeg with a warning
    np.testing.assert_array_equal(ctx.jpeg_decode(cut).numpy(), oj.decode(cut))


def test_dataset_mapper_matches_read_image_and_resize(ctx):
    data = _jpeg(_picture(480, 640, 11), quality=90, subsampling=2)
    mapper = HipDatasetMapper(ctx, short_edge_length=256, max_size=300)
    out = mapper({"jpeg": data, "image_id": 7})
    nh, nw = resize_shortest_edge_shape(480, 640, 256, 300)
    assert (nh, nw) == (225, 300) and out["height"] == 480 and out["width"] == 640 and out["image_id"] == 7
    ref = np.asarray(Image.fromarray(_pil(data)).resize((nw, nh), Image.BILINEAR))
    np.testing.assert_array_equal(out["image"].numpy(), ref)
    keep = HipDatasetMapper(ctx, short_edge_length=None)({"jpeg": data})
    np.testing.assert_array_equal(keep["image"].numpy(), _pil(data))
    padded = ctx.u8_hwc_to_f32_chw_padded(out["image"], 256, 320, 1.0 / 255.0).numpy()
    want = np.zeros((3, 256, 320), np.float32)
    want[:, :nh, :nw] = ref.transpose(2, 0, 1).astype(np.float32) * np.float32(1.0 / 255.0)
    np.testing.assert_array_equal(padded, want)


def test_loader_threads_give_the_same_images(ctx):
    from odise_amd.runtime import jpeg_decode_coefs, jpeg_entropy_decode
    dicts = [{"jpeg": _jpeg(_picture(h, w, seed=i), quality=80, subsampling=sub), "image_id": i}
             for i, (h, w, sub) in enumerate([(120, 160, 2), (97, 64, 0), (64, 201, 1), (300, 200, 2), (33, 47, 2), (128, 128, 0), (250, 99, 1)])]
    mapper = HipDatasetMapper(ctx, short_edge_length=96, max_size=200)
    one_by_one = [mapper(d) for d in dicts]
    threaded = list(mapper.map_many(dicts, workers=3))
    assert [d["image_id"] for d in threaded] == list(range(len(dicts)))
    for a, b in zip(one_by_one, threaded):
        assert (a["height"], a["width"]) == (b["height"], b["width"])
        np.testing.assert_array_equal(a["image"].numpy(), b["image"].numpy())
    info, coefs, qt = jpeg_entropy_decode(dicts[3]["jpeg"])
    np.testing.assert_array_equal(jpeg_decode_coefs(ctx, info, coefs, qt).numpy(), _pil(dicts[3]["jpeg"]))
    bad = dict(info, width=info["width"] + 64)
    with pytest.raises(RuntimeError):
        jpeg_decode_coefs(ctx, bad, coefs, qt)
