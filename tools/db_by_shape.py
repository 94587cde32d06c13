"""Per-kernel and per-dispatch-shape summary of a rocprofv3 run stored as a rocpd sqlite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME --
<cmd>` writes DIR/NAME_results.db in ROCm 7): kernel-time totals, the per-(kernel, grid) table of the LAST step, and the busy time per
HIP stream.  usage: db_by_shape.py <results.db> [marker | <steps_in_trace>] [top=40]
"marker" (default) cuts the trace at the image_pad_kernel launches that open every model call and takes the last-but-one call (the last one is
followed by bench.py's own dominant-kernel timing launches)."""
import collections
import re
import sqlite3
import sys

path = sys.argv[1]
steps = sys.argv[2] if len(sys.argv) > 2 else "marker"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cur = sqlite3.connect(path).cursor()
rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, stream_id from kernels order by start").fetchall()


def short(n):
    n = n.replace("void odise::", "").replace("odise::", "").replace("(odise::GemmArgs)", "").replace("(odise::AttnArgs)", "")
    return re.sub(r"\(.*", "", n)[:46]


ours = [r for r in rows if "odise" in r[0]]
total = sum(r[2] - r[1] for r in rows) / 1e6
print(f"# {len(rows)} kernel dispatches, {total:.2f} ms of kernel time in the whole trace ({len(ours)} from libodise_hip)")
by_name = collections.Counter()
calls = collections.Counter()
for r in rows:
    by_name[short(r[0])] += (r[2] - r[1]) / 1e3
    calls[short(r[0])] += 1
print("# ---- kernel totals over the whole trace (us)")
print("Name,Calls,TotalDurationUs,AverageUs,Percentage")
for k, v in by_name.most_common(top):
    print(f"\"{k}\",{calls[k]},{v:.1f},{v / calls[k]:.2f},{100 * v / (total * 1e3):.2f}")
if steps == "marker":
    starts = [i for i, r in enumerate(ours) if "image_pad_kernel" in r[0] and (i == 0 or "image_pad_kernel" not in ours[i - 1][0])]
    assert len(starts) >= 2, "fewer than two model calls in the trace"
    last = ours[starts[-2]:starts[-1]]
    n = len(last)
else:
    steps = int(steps)
    n = len(ours) // steps
    last = ours[-n:] if steps > 0 else ours
span = (max(r[2] for r in last) - min(r[1] for r in last)) / 1e6
busy = sum(r[2] - r[1] for r in last) / 1e6
print(f"# ---- one model call ({steps}): {n} launches, {busy:.2f} ms of kernel time inside a {span:.2f} ms span (overlap across streams shortens the span)")
streams = collections.Counter()
for r in last:
    streams[r[9]] += (r[2] - r[1]) / 1e6
print("# kernel time per stream id (ms):", dict(streams))
agg = collections.OrderedDict()
for r in last:
    key = (short(r[0]), r[3] // max(r[6], 1), r[4] // max(r[7], 1), r[5] // max(r[8], 1))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += (r[2] - r[1]) / 1e3
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0]:46s} grid=({k[1]},{k[2]},{k[3]}) n={a[0]:3d} tot={a[1] / 1e3:7.2f}ms {100 * a[1] / (busy * 1e3):5.1f}% avg={a[1] / a[0]:8.1f}us")
