"""Average rocprofv3 --pmc counter values per dispatch of the kernels matching a substring.  usage: pmc_avg.py <counter_collection.csv> <substr>"""
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
acc = collections.defaultdict(list)
for r in rows:
    acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    v = acc[k]
    print(f"   {k:28s} {sum(v)/len(v):.4e}  (n={len(v)})")
