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

# ---- main-queue view: idle gaps between consecutive kernels of the busiest queue, with their neighbours
mainq = max(q.items(), key=lambda x: x[1])[0]
mk = sorted((s, e, n) for s, e, n, qq in sel if qq == mainq)
gl = []
for (s0, e0, n0), (s1, e1, n1) in zip(mk, mk[1:]):
    if s1 > e0: gl.append((s1 - e0, n0[:48], n1[:48], (e0 - t0) / 1e6))
tot = sum(g[0] for g in gl) / 1e6
print(f"main queue {mainq}: {len(mk)} launches, busy {sum(e - s for s, e, _ in mk)/1e6:.2f} ms, idle between launches {tot:.2f} ms")
import collections as C
hist = C.Counter()
for g in gl:
    b = 2 if g[0] < 2e3 else 5 if g[0] < 5e3 else 10 if g[0] < 1e4 else 50 if g[0] < 5e4 else 1000
    hist[b] += g[0] / 1e6
print("idle by gap size (<2us,<5us,<10us,<50us,more) ms:", [round(hist[b], 2) for b in (2, 5, 10, 50, 1000)])
for g in sorted(gl, reverse=True)[:15]:
    print(f"  {g[0]/1e3:8.1f} us at {g[3]:7.2f} ms  after {g[1]}  before {g[2]}")
