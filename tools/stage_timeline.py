"""Stage boundaries of the benchmarked step on both clocks (csrc stage_mark, include/odise_hip_tools.h odise_hip_stage_timeline): when the DEVICE
reached each boundary (HIP event on the lane that enqueues the stage) and when the HOST had enqueued everything before it.  A host column that
runs ahead of the device column means the lanes are fed in time; where the device waits for the host the step is launch-bound.
usage: stage_timeline.py [--images 4] [--reps 3]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--pipeline", action="store_true", help="encoder prefetch: the calls alternate between two sets of pictures, each preparing the other")
    ap.add_argument("--prefetch-start", type=int, default=1)
    ap.add_argument("--prefetch-cus", type=int, default=0)
    ap.add_argument("--maskclip-passes", type=int, default=0, help="ODISE_OPT_MASKCLIP_PASSES")
    args = ap.parse_args()
    from odise_amd.runtime import Context
    ctx = Context(0)
    if args.lanes == 1:
        ctx.lib.odise_hip_set_lanes(ctx.h, 1)
    ctx.set_option(ctx.OPT_MASKCLIP_PASSES, args.maskclip_passes)
    S, B = 1024, args.images
    u8 = [bench.image_u8(S, b) for b in range(B)]
    hip, _ = bench.calibrated_model(ctx, u8[0], S, 133, 254, set(range(80)), None)
    d_img = [ctx.to_device(u) for u in u8]
    hw = [(S, S)] * B
    sets = [d_img]
    if args.pipeline:
        ctx.set_option(ctx.OPT_PREFETCH_CU_EIGHTHS, args.prefetch_cus)
        ctx.set_option(ctx.OPT_PREFETCH_START, args.prefetch_start)
        sets.append([ctx.to_device(bench.image_u8(S, B + b)) for b in range(B)])
    turn = [0]

    def call():
        cur = sets[turn[0] % len(sets)]
        turn[0] += 1
        if args.pipeline:
            hip.prefetch_device(sets[turn[0] % len(sets)], 0, hw)
        hip.infer_device(cur, 0, hw, hw, to_host=False)

    for _ in range(2):
        call()
    ctx.sync()
    rows = None
    for _ in range(args.reps):
        ctx.stage_timeline(True)
        call()
        ctx.sync()
        t = ctx.stage_timeline_read()
        rows = t if rows is None else [(a[0], a[1] + b[1], a[2] + b[2]) for a, b in zip(rows, t)]
    ctx.stage_timeline(False)
    print(f"# one odise_hip_infer over {B} x {S}x{S} pictures, {args.lanes} lane(s), mean of {args.reps} calls; ms since the first mark")
    print(f"{'device reached':>15} {'host enqueued':>14}  stage boundary")
    for name, g, h in sorted(rows, key=lambda r: r[1]):
        print(f"{g / args.reps:15.2f} {h / args.reps:14.2f}  {name}")


if __name__ == "__main__":
    main()
