"""Per-token kernel-time breakdown of the graph-replayed beam-search loop from a rocprofv3 kernel trace of tools/bench_generate.py."""
import csv, glob, collections, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:50]))
rows.sort()
bt = [i for i, r in enumerate(rows) if "beam_topk" in r[2]]
i0, i1 = bt[-40], bt[-10]
seg = rows[i0:i1]
wall = (rows[i1][0] - rows[i0][0]) / 1e3; busy = sum(e - s for s, e, _ in seg) / 1e3
print("30 tokens: wall %.1f us/token, kernel busy %.1f us/token, kernels/token %.1f" % (wall / 30, busy / 30, len(seg) / 30))
c = collections.Counter(); t = collections.Counter()
for s, e, n in seg: c[n] += 1; t[n] += (e - s) / 1e3
for n, v in t.most_common(12): print("%8.1f us/token  %5.1f launches/token  avg %6.1f us  %s" % (v / 30, c[n] / 30, v / c[n], n))
