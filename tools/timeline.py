"""Coarse timeline of one steady-state step from a rocprofv3 kernel trace of bench.py: per 2-ms bin, busy fraction of each
queue/stream, launches, and the dominant kernel."""
import csv, glob, sys, collections
d = sys.argv[1]; binms = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
rows = []; hdr = None
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rd = csv.DictReader(open(f)); hdr = rd.fieldnames
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
print("columns:", hdr)
rows.sort()
ad_all = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
ad = []                                    # last AdamW launch of every step (the reducer's pipelined optimizer launches one per bucket)
for i in ad_all:
    if ad and rows[i][1] - rows[ad[-1]][1] < 20e6:
        ad[-1] = i
    else:
        ad.append(i)
lo, hi = ad[-4], ad[-3]
t0, t1 = rows[lo][1], rows[hi][1]
sel = [r for r in rows if r[1] > t0 and r[0] < t1]
streams = sorted({r[3] for r in sel})
nb = int((t1 - t0) / 1e6 / binms) + 1
print(f"step wall {(t1-t0)/1e6:.2f} ms; streams {streams}")
busy = {s: [0.0] * nb for s in streams}; cnt = [0] * nb; names = [collections.Counter() for _ in range(nb)]
for s, e, n, q in sel:
    s = max(s, t0); e = min(e, t1)
    b0 = int((s - t0) / 1e6 / binms); b1 = int((e - t0) / 1e6 / binms)
    for b in range(b0, min(b1, nb - 1) + 1):
        bs = t0 + b * binms * 1e6; be = bs + binms * 1e6
        ov = max(0, min(e, be) - max(s, bs))
        busy[q][b] += ov / (binms * 1e6)
        names[b][n.replace("(anonymous namespace)::", "").replace("void ", "")[:28]] += ov
    cnt[b0] += 1
for b in range(nb):
    top = names[b].most_common(1)[0][0] if names[b] else ""
    print(f"{b*binms:6.1f} ms  " + "  ".join(f"{busy[s][b]*100:5.0f}%" for s in streams) + f"  launches {cnt[b]:4d}  {top}")
