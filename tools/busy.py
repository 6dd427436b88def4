"""GPU occupancy of the steady-state steps from a rocprofv3 kernel trace of bench.py.
Steps are delimited by adamw_kernel launches; prints wall per step, union-busy time, per-queue busy time and the largest idle gaps."""
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
print("adamw launches:", len(ad))
lo, hi = ad[-4], ad[-2]          # two full steps well inside the timed region (the last step is the single-stream probe)
t0, t1 = rows[lo][1], rows[hi][1]
sel = [r for r in rows[lo + 1:hi + 1]]
wall = (t1 - t0) / 1e6
ev = sorted((max(s, t0), min(e, t1)) for s, e, _, _ in sel)
busy = 0; cur_s, cur_e = ev[0]; gaps = []
for s, e in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e - t0)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"2 steps: wall {wall:.2f} ms, union busy {busy/1e6:.2f} ms ({busy/1e6/wall*100:.1f}%), kernel-time sum {sum(e-s for s,e in ev)/1e6:.2f} ms, launches {len(sel)}")
q = collections.defaultdict(float)
for s, e, _, qq in sel: q[qq] += (e - s) / 1e6
for k, v in sorted(q.items(), key=lambda x: -x[1]): print(f"  queue {k}: {v:.2f} ms busy")
gaps.sort(reverse=True)
print("largest idle gaps (us @ offset ms):", [(round(g / 1e3, 1), round(o / 1e6, 2)) for g, o in gaps[:12]], "total idle in gaps", round(sum(g for g, _ in gaps) / 1e6, 2), "ms")
