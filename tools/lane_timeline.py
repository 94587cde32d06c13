"""Timeline of ONE model call from a rocprofv3 kernel trace (rocpd sqlite: `rocprofv3 --kernel-trace -d DIR -o NAME -- python bench.py ...` ->
DIR/NAME_results.db): per HIP stream the first / last kernel, busy time, idle gaps above 0.25 ms, and where the big phases start.  Shows
whether the lanes of the feature extractor (VAE encoder -> decoder on the main stream, CLIP -> UNet on the second) actually overlap or wait
for the host to feed them.  usage: lane_timeline.py <results.db>"""
import collections
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, stream_id from kernels order by start").fetchall()
ours = [r for r in rows if "odise" in r[0]]
starts = [i for i, r in enumerate(ours) if "image_pad_kernel" in r[0] and (i == 0 or "image_pad_kernel" not in ours[i - 1][0])]
assert len(starts) >= 2, "fewer than two model calls in the trace"
last = ours[starts[-2]:starts[-1]]
t0 = last[0][1]


def short(n):
    return re.sub(r"\(.*", "", n.replace("void odise::", "").replace("odise::", ""))[:36]


def ms(t):
    return (t - t0) / 1e6


end = max(r[2] for r in last)
print(f"one model call: {len(last)} launches, span {ms(end):.2f} ms, kernel time {sum(r[2] - r[1] for r in last) / 1e6:.2f} ms")
by = collections.defaultdict(list)
for r in last:
    by[r[7]].append(r)
for sid, rs in sorted(by.items()):
    print(f"stream {sid}: {len(rs)} launches, first at {ms(rs[0][1]):.2f} ms, last ends {ms(max(r[2] for r in rs)):.2f} ms, busy {sum(r[2] - r[1] for r in rs) / 1e6:.2f} ms")
    prev = rs[0]
    for r in rs[1:]:
        gap = (r[1] - prev[2]) / 1e6
        if gap > 0.25:
            print(f"    idle {gap:6.2f} ms from t = {ms(prev[2]):6.2f} (after {short(prev[0])}, before {short(r[0])})")
        prev = r
    fam = collections.Counter()
    for r in rs:
        fam[short(r[0]).split("<")[0][:28]] += (r[2] - r[1]) / 1e6
    print("    by family (ms):", ", ".join(f"{k} {v:.2f}" for k, v in fam.most_common(7)))
