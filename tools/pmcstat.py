"""Per-kernel mean of PMC counters from rocprofv3 counter_collection CSVs: python tools/pmcstat.py <csv>... [--match substr]"""
import csv, sys, collections
match = ""
files = []
a = sys.argv[1:]
while a:
    v = a.pop(0)
    if v == "--match": match = a.pop(0)
    else: files.append(v)
acc = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"]: continue
        acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    v = v[len(v) // 2:]          # skip warm-up launches
    print(f"{c:32s} {sum(v)/len(v):16.0f}  n={len(v)}  {k}")
