"""Per-kernel-family and per-dispatch-shape breakdown of the LAST step in a rocprofv3 kernel trace.
usage: trace_by_shape.py <kernel_trace.csv> [marker | <steps_in_trace>] [top=24]"""
import collections
import csv
import sys

path = sys.argv[1]
steps = sys.argv[2] if len(sys.argv) > 2 else "marker"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rows = [r for r in csv.DictReader(open(path)) if 'odise' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
if steps == "marker":
    # a model call starts with the image_pad_kernel launches of its batch (odise_hip_infer): the trace is cut there, and the LAST BUT ONE
    # segment is one whole timed step (the last one is followed by bench.py's own dominant-kernel timing launches)
    starts = [i for i, r in enumerate(rows) if 'image_pad_kernel' in r['Kernel_Name'] and (i == 0 or 'image_pad_kernel' not in rows[i - 1]['Kernel_Name'])]
    assert len(starts) >= 2, "fewer than two model calls in the trace"
    last = rows[starts[-2]:starts[-1]]
    n = len(last)
else:
    n = len(rows) // int(steps)
    last = rows[-n:]
agg = collections.OrderedDict()
fam = collections.Counter()
for r in last:
    nm = r['Kernel_Name'].replace('void odise::', '').replace('(odise::GemmArgs)', '').replace('(odise::AttnArgs)', '')
    if nm.startswith('_ZN5odise'):
        nm = nm[9:].lstrip('0123456789')
    nm = nm[:44]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    key = (nm, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += d
    fam[nm.split('<')[0].split('(')[0][:28]] += d
tot = sum(a[1] for a in agg.values())
print(f'step {tot/1e3:.2f} ms of kernel time, {n} launches')
for k, v in fam.most_common(14):
    print(f"  {k:30s} {v/1e3:7.2f} ms {100*v/tot:5.1f}%")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0]:44s} grid=({k[1]},{k[2]},{k[3]}) n={a[0]:3d} tot={a[1]/1e3:7.2f}ms {100*a[1]/tot:5.1f}% avg={a[1]/a[0]:8.1f}us")
