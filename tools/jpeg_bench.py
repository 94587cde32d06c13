"""Throughput of the JPEG input path at COCO-like sizes: host Huffman decoding vs the device stages, against Pillow on the host cores."""
import io
import sys
import time

import numpy as np
from PIL import Image

sys.path.insert(0, ".")
from odise_amd.ingest import HipDatasetMapper  # noqa: E402
from odise_amd.runtime import default_context, jpeg_entropy_decode  # noqa: E402
from tests.test_oracle_jpeg import _jpeg, _picture  # noqa: E402

ctx = default_context()
for (h, w) in ((480, 640), (1024, 1024)):
    data = _jpeg(_picture(h, w, 3), quality=90, subsampling=2)
    n = 20
    ctx.jpeg_decode(data); ctx.sync()
    t = time.perf_counter()
    for _ in range(n):
        jpeg_entropy_decode(data)
    t_host = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for _ in range(n):
        ctx.jpeg_decode(data)
    ctx.sync()
    t_all = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for _ in range(n):
        np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    t_pil = (time.perf_counter() - t) / n
    mapper = HipDatasetMapper(ctx)
    mapper({"jpeg": data}); ctx.sync()
    t = time.perf_counter()
    for _ in range(n):
        mapper({"jpeg": data})
    ctx.sync()
    t_map = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for _ in range(n):
        im = Image.open(io.BytesIO(data)).convert("RGB")
        s = 1024 / min(h, w)
        np.asarray(im.resize((int(w * s + 0.5), int(h * s + 0.5)), Image.BILINEAR))
    t_pilmap = (time.perf_counter() - t) / n
    print(f"{h}x{w} {len(data)/1e3:.0f} kB: host entropy decode {t_host*1e3:.2f} ms, decode to HBM {t_all*1e3:.2f} ms (Pillow {t_pil*1e3:.2f} ms); "
          f"decode+resize(1024) {t_map*1e3:.2f} ms (Pillow {t_pilmap*1e3:.2f} ms)", flush=True)
