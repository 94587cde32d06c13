"""Join a GEMM launch log (ODISE_GEMM_FLAGS=32, stderr) with a rocprofv3 kernel trace: per-shape time and TFLOP/s of the LAST step.
usage: join_gemm_log.py <kernel_trace.csv> <gemmlog.txt> <steps_in_trace> [top]"""
import collections
import csv
import sys

trace, logf, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
rows = [r for r in csv.DictReader(open(trace)) if 'odise' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // steps
last = rows[-n:]
g = [r for r in last if 'gemm_kernel' in r['Kernel_Name'] or 'gemm_pp_kernel' in r['Kernel_Name']]
logs = [l for l in open(logf) if 'GEMMLOG' in l]
per = len(logs) // steps if len(logs) >= steps * len(g) else len(logs)
logs = logs[-len(g):]
BN = [128, 128, 64, 320, 256, 128, 128]
def parse(l):
    return {k: int(v) for k, v in (kv.split('=') for kv in l.split()[1:])}
bad = sum(1 for r, l in zip(g, logs) if int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) != -(-parse(l)['N'] // BN[parse(l)['tile']]))
print(f"{len(g)} gemm launches in the last step, {len(logs)} log lines used, {bad} grid mismatches")
agg = collections.OrderedDict()
for r, l in zip(g, logs):
    d = parse(l)
    key = tuple(d[k] for k in ('conv', 'M', 'N', 'K', 'batch', 'cin', 'h', 'stride', 'ups', 'tile', 'split'))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(a[1] for a in agg.values())
allk = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e6
print(f"gemm total {tot/1e3:.2f} ms of {allk:.2f} ms kernel time")
out = sorted(((a[1], k, a[0], 2.0 * k[1] * k[2] * k[3] * k[4] * a[0] / a[1] / 1e6) for k, a in agg.items()), reverse=True)
for t, k, cnt, tf in out[:top]:
    print(f"{t/1e3:6.2f}ms n={cnt:3d} {tf:7.1f} TF/s conv={k[0]} M={k[1]} N={k[2]} K={k[3]} b={k[4]} cin={k[5]} h={k[6]} s={k[7]} ups={k[8]} tile={k[9]} split={k[10]}")
