"""MaskCLIP (clip.py:252-323) on 4 pictures x 100 mask tokens alone on the chip, in the forms of ODISE_OPT_MASKCLIP_PASSES: one pass over
[577 image | 100 mask] token rows (2), two passes in place (1: image tokens, then mask tokens over their keys / values).  HIP-event timing on
the context's stream; under `rocprofv3 --kernel-trace --stats` with `--only 1` the per-kernel split of the two passes.

    python tools/maskclip_bench.py [--images 4] [--rounds 5] [--only MODE]          (GPU)"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", type=int, default=None)
    args = ap.parse_args()
    from odise_amd.runtime import Context, check
    ctx = Context(0)
    S, B, Q = 1024, args.images, 100
    u8 = [bench.image_u8(S, b) for b in range(B)]
    hip, _ = bench.calibrated_model(ctx, u8[0], S, 133, 254, set(range(80)), None)
    rng = np.random.default_rng(0)
    img = ctx.to_device(np.stack([np.ascontiguousarray(u.astype(np.float32).transpose(2, 0, 1) / 255.0) for u in u8]))
    # smooth random mask logits at 1/4 resolution: blobs, so that every mask token sees a different subset of the patches
    h = w = S // 4
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    masks = np.empty((B, Q, h, w), np.float32)
    for b in range(B):
        for q in range(Q):
            cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(12, 90)
            masks[b, q] = (r * r - (yy - cy) ** 2 - (xx - cx) ** 2) / (r * r) * 6.0
    dm = ctx.to_device(masks)
    out = ctx.empty((B, Q, 768), np.float32)

    def call():
        check(ctx.lib.odise_hip_maskclip_embed(ctx.h, C.c_void_p(img.ptr), B, S, S, C.c_void_p(dm.ptr), Q, h, w, C.c_void_p(out.ptr)), "maskclip_embed")

    modes = [args.only] if args.only is not None else [2, 1]
    res, outs = {}, {}
    for _ in range(args.rounds):
        for m in modes:
            ctx.set_option(ctx.OPT_MASKCLIP_PASSES, m)
            call()
            ctx.sync()
            ctx.timer_start()
            for _ in range(5):
                call()
            res.setdefault(m, []).append(ctx.timer_stop() / 5)
            outs[m] = out.numpy().copy()
    for m in modes:
        print(f"ODISE_OPT_MASKCLIP_PASSES {m}: {np.median(res[m]):.3f} ms per call ({B} pictures x {Q} mask tokens), min {min(res[m]):.3f}")
    if len(modes) == 2:
        a, b = outs[modes[0]], outs[modes[1]]
        print(f"embeddings, max |diff| / max |ref|: {np.abs(a - b).max() / np.abs(a).max():.3e}")
    ctx.close()


if __name__ == "__main__":
    main()
