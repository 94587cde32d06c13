"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) totals of the LAST step of a bench run."""
import collections
import csv
import sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
name = sys.argv[4] if len(sys.argv) > 4 else "unet"
rows = [r for r in csv.DictReader(open(f"{d}/{name}_kernel_trace.csv")) if "odise" in r["Kernel_Name"]]
n = len(rows) // steps
agg = collections.OrderedDict()
for r in rows[-n:]:
    nm = r["Kernel_Name"].replace("void odise::", "").replace("(odise::GemmArgs)", "").replace("(odise::AttnArgs)", "")[:44]
    key = (nm, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
print(f"step kernel time {tot:.1f} us over {n} launches")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0]:44s} grid=({k[1]},{k[2]},{k[3]}) n={a[0]:3d} tot={a[1]:9.1f}us {100 * a[1] / tot:5.1f}% avg={a[1] / a[0]:8.1f}")
