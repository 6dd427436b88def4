"""Median per-kernel durations from a rocprofv3 --kernel-trace CSV.
Usage: python tools/kstat.py <dir> [substr] [runs]
  default: one line per (kernel, grid);  runs=1: one line per maximal time-ordered run of matching launches that is not
  interrupted by any other kernel (so a benchmark that separates its cases by an unrelated kernel gets one line per case)."""
import csv, glob, sys, statistics as st
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""; runs = len(sys.argv) > 3 and sys.argv[3] not in ("", "0")
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"][:100], r.get("Grid_Size_X", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
if runs:
    cur = []
    def flush():
        if len(cur) >= 3:
            v = sorted(x[3] for x in cur)
            print(f"{st.median(v):9.1f} us (min {v[0]:.1f})  n={len(v):4d} grid={cur[0][2]} {cur[0][1]}")
        cur.clear()
    for r in rows:
        if sub in r[1] and (not cur or cur[0][1] == r[1]): cur.append(r)
        else:
            flush()
            if sub in r[1]: cur.append(r)
    flush()
else:
    g = {}
    for _, n, gx, t in rows:
        if sub in n: g.setdefault((n, gx), []).append(t)
    for (n, gx), v in g.items():
        v.sort()
        print(f"{st.median(v):9.1f} us (p25 {v[len(v)//4]:.1f} p75 {v[3*len(v)//4]:.1f})  n={len(v):4d} grid={gx} {n}")
