"""Input side of the eval loop on the device: file bytes -> resized uint8 image resident in HBM.

Mirrors detectron2's `DatasetMapper(is_train=False, augmentations=[T.ResizeShortestEdge(short_edge_length=1024, max_size=2560,
sample_style="choice")], image_format="RGB")` as configured in configs/common/data/pano_open_d2_eval.py:74-107:
`utils.read_image(file_name, "RGB")` (Pillow decode + EXIF transpose) -> `ResizeTransform` (PIL bilinear) -> CHW tensor, with
`height` / `width` of the ORIGINAL image kept for `sem_seg_postprocess`.  JPEG decoding (`odise_hip_jpeg_decode`) and the resize
(`odise_hip_resize_bilinear_u8`) are bit-identical to Pillow; the image stays on the device as uint8 [H,W,3] and
`HipCategoryODISE.forward` converts / pads it there.  Files the decoder refuses (arithmetic-coded, CMYK, PNG ...) raise
`UnsupportedInput`: there is no CPU decode path here - the caller may hand such images over as arrays, as before.
"""
from __future__ import annotations

from typing import Optional

from concurrent.futures import ThreadPoolExecutor
from typing import Iterable, Iterator

from ._lib import UnsupportedInput
from .runtime import Context, DeviceArray, jpeg_decode_coefs, jpeg_entropy_decode


def resize_shortest_edge_shape(h: int, w: int, short: int = 1024, max_size: int = 2560):
    """detectron2 `ResizeShortestEdge.get_output_shape` (transforms/augmentation_impl.py)."""
    scale = short * 1.0 / min(h, w)
    newh, neww = (short, scale * w) if h < w else (scale * h, short)
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


class HipDatasetMapper:
    def __init__(self, ctx: Context, short_edge_length: Optional[int] = 1024, max_size: int = 2560):
        self.ctx, self.short, self.max_size = ctx, short_edge_length, max_size

    def __call__(self, dataset_dict: dict) -> dict:
        """{"file_name": path} or {"jpeg": bytes} (+ any other keys, passed through) -> the same dict with "image" = DeviceArray uint8
        [h,w,3] (resized) and "height" / "width" = the decoded image's size."""
        d = dict(dataset_dict)
        data = d.pop("jpeg", None)
        if data is None:
            with open(d["file_name"], "rb") as f:
                data = f.read()
        if data[:2] != b"\xff\xd8":
            raise UnsupportedInput("HipDatasetMapper: not a JPEG stream")
        return self._finish(d, self.ctx.jpeg_decode(data))

    def _finish(self, d: dict, img: DeviceArray) -> dict:
        h, w = img.shape[:2]
        d.setdefault("height", h)
        d.setdefault("width", w)
        if self.short is not None:
            nh, nw = resize_shortest_edge_shape(h, w, self.short, self.max_size)
            if (nh, nw) != (h, w):
                img = self.ctx.resize_bilinear_u8(img, nh, nw)
        d["image"] = img
        return d

    # ---- loader threads: the serial half (file read + Huffman decoding, GIL released inside the library) runs `workers` images ahead
    @staticmethod
    def _host_half(dataset_dict: dict):
        d = dict(dataset_dict)
        data = d.pop("jpeg", None)
        if data is None:
            with open(d["file_name"], "rb") as f:
                data = f.read()
        if data[:2] != b"\xff\xd8":
            raise UnsupportedInput("HipDatasetMapper: not a JPEG stream")
        info, coefs, qt = jpeg_entropy_decode(data, flat=True)
        return d, info, coefs, qt

    def map_many(self, dataset_dicts: Iterable[dict], workers: int = 4) -> Iterator[dict]:
        """Same results as `map(self, dataset_dicts)`, in order; the device half of image i overlaps the host half of i+1 .. i+workers
        (the reference's DataLoader does this with `num_workers` processes, pano_open_d2_eval.py:88)."""
        with ThreadPoolExecutor(max_workers=workers) as pool:
            pending = []
            it = iter(dataset_dicts)
            for dd in it:
                pending.append(pool.submit(self._host_half, dd))
                if len(pending) > workers:
                    d, info, coefs, qt = pending.pop(0).result()
                    yield self._finish(d, jpeg_decode_coefs(self.ctx, info, coefs, qt))
            for fut in pending:
                d, info, coefs, qt = fut.result()
                yield self._finish(d, jpeg_decode_coefs(self.ctx, info, coefs, qt))
