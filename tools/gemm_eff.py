"""Where the MFMA kernels of the benchmarked step lose time: every GEMM / convolution launch of one step with its problem shape, tile
choice, duration and TFLOP/s.

  run  : the benchmarked step (bs = 4 x 1024x1024, full path) on ONE lane (durations do not overlap), with the tools build logging every
         launch (ODISE_GEMM_FLAGS=32 -> 'GEMMLOG ...' on stderr).  Meant to run under rocprofv3 --kernel-trace:
           ODISE_HIP_LIB=odise_amd/lib/libodise_hip_tools.so ODISE_GEMM_FLAGS=32 rocprofv3 --kernel-trace --output-format csv -d D -o t -- \
               python tools/gemm_eff.py run 2> D/gemm.log
  join : gemm_eff.py join D/t_kernel_trace.csv D/gemm.log [ceiling TFLOP/s = 1100]  ->  the last step's launches grouped by shape, sorted by
         the time they would save at the ceiling rate (the best rate the large GEMMs of this library reach on this chip)."""
import collections
import csv
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

FAMILIES = ("gemm_kernel", "gemm_pp_kernel", "gemm_pp2_kernel", "conv3_halo_kernel", "conv3_halo4_kernel")


def run(steps=3):
    import bench
    from odise_amd.pipeline import HipCategoryODISE
    from odise_amd.runtime import Context
    from odise_amd.synthetic import synthetic_state, synthetic_vocabulary

    ctx = Context(0)
    hip = HipCategoryODISE(ctx, synthetic_state(), overlap_threshold=0.8)
    cat, clp, sizes, overlap = synthetic_vocabulary(133, 254, 768)
    hip.set_vocabulary(cat, clp, sizes, overlap, set(range(80)), 0.3, 0.7)
    imgs = [ctx.to_device(bench.image_u8(1024, b)) for b in range(4)]
    hw = [(1024, 1024)] * 4
    assert ctx.lib.odise_hip_set_lanes(ctx.h, 1) == 0
    for s in range(steps):
        ctx.sync()
        sys.stderr.write(f"GEMMLOG step {s}\n")
        sys.stderr.flush()
        hip.infer_device(imgs, 0, hw, hw, to_host=False)
    ctx.sync()


def join(trace, log, ceiling=1100.0):
    launches = []
    for r in csv.DictReader(open(trace)):
        nm = r["Kernel_Name"]
        fam = next((f for f in FAMILIES if re.search(r"\b" + f + r"\b|" + f + "<|" + f + "I", nm)), None)
        if fam is None or "splitk" in nm:
            continue
        launches.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, fam,
                         (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))))
    launches.sort()
    entries, step_at = [], []
    for line in open(log, errors="replace"):
        if not line.startswith("GEMMLOG"):
            continue
        if line.startswith("GEMMLOG step"):
            step_at.append(len(entries))
            continue
        entries.append({k: int(v) for k, v in re.findall(r"(\w+)=(-?\d+)", line)})
    print(f"# {len(launches)} MFMA-kernel launches in the trace, {len(entries)} logged launches, steps start at {step_at}")
    assert len(launches) == len(entries), "trace and log disagree: not the same process?"
    first = step_at[-1]
    agg = collections.OrderedDict()
    for e, (_, us, fam, grid) in zip(entries[first:], launches[first:]):
        key = (e["conv"], e["M"], e["N"], e["K"], e["batch"], e["cin"] if e["conv"] else 0, e["kh"] if e["conv"] else 0, e["h"] if e["conv"] else 0,
               e["stride"] if e["conv"] else 0, e["ups"] if e["conv"] else 0, e["tile"], e["split"], fam, grid)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += us
    tot_us = sum(a[1] for a in agg.values())
    tot_fl = sum(2.0 * k[1] * k[2] * k[3] * k[4] * a[0] for k, a in agg.items())
    print(f"# last step: {sum(a[0] for a in agg.values())} launches, {tot_us / 1e3:.2f} ms, {tot_fl / 1e12:.2f} TFLOP -> {tot_fl / tot_us / 1e6:.0f} TFLOP/s on average; "
          f"'lost' = time above what the launch would take at {ceiling:.0f} TFLOP/s")
    print(f"{'kind':5s} {'M':>7s} {'N':>5s} {'K':>6s} {'b':>3s} {'conv (cin k h s u)':>20s} {'tile':>4s} {'sp':>3s} {'kernel':18s} {'grid':>14s} {'n':>3s} {'tot ms':>8s} {'avg us':>8s} {'TFLOP/s':>8s} {'lost ms':>8s}")
    rows = []
    for k, a in agg.items():
        fl = 2.0 * k[1] * k[2] * k[3] * k[4]
        lost = a[1] - a[0] * fl / (ceiling * 1e6)
        rows.append((lost, k, a, fl))
    cum = 0.0
    for lost, k, a, fl in sorted(rows, key=lambda r: -r[0]):
        cum += lost
        conv = f"{k[5]} {k[6]} {k[7]} {k[8]} {k[9]}" if k[0] else ""
        print(f"{'conv' if k[0] else 'gemm':5s} {k[1]:7d} {k[2]:5d} {k[3]:6d} {k[4]:3d} {conv:>20s} {k[10]:4d} {k[11]:3d} {k[12]:18s} {str(k[13]):>14s} {a[0]:3d} "
              f"{a[1] / 1e3:8.2f} {a[1] / a[0]:8.1f} {fl * a[0] / a[1] / 1e6:8.0f} {lost / 1e3:8.2f}")
    print(f"# total lost against the ceiling: {cum / 1e3:.2f} ms of {tot_us / 1e3:.2f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    else:
        join(sys.argv[2], sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 1100.0)
