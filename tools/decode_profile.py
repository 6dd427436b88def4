"""Per-token kernel-time breakdown of the graph-replayed beam-search loop from a rocprofv3 kernel trace of tools/bench_generate.py."""
import csv, glob, collections, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("vacgemm::", "")[:90]))
rows.sort()
bt = [i for i, r in enumerate(rows) if "beam_step" in r[2]]
i0, i1 = bt[-40], bt[-10]
seg = rows[i0:i1]
wall = (rows[i1][0] - rows[i0][0]) / 1e3; busy = sum(e - s for s, e, _ in seg) / 1e3
print("30 tokens: wall %.1f us/token, kernel busy %.1f us/token, kernels/token %.1f" % (wall / 30, busy / 30, len(seg) / 30))
c = collections.Counter(); t = collections.Counter()
for s, e, n in seg: c[n] += 1; t[n] += (e - s) / 1e3
for n, v in t.most_common(12): print("%8.1f us/token  %5.1f launches/token  avg %6.1f us  %s" % (v / 30, c[n] / 30, v / c[n], n))

# ---- what runs between two captions (ViT + encoder + cross K/V of the next one): kernels from the last beam_step of a caption to the
# first decoder step of the next
ds = [i for i, r in enumerate(rows) if "decoder_step" in r[2]]
cuts = [(a, b) for a, b in zip(ds, ds[1:]) if b - a > 60]
if cuts:
    a, b = cuts[-1]
    while a < b and "beam_step" not in rows[a][2]:
        a += 1
    seg = rows[a + 1:b]
    wall = (rows[b][0] - rows[a][1]) / 1e3; busy = sum(e - s for s, e, _ in seg) / 1e3
    print("\nbetween captions: wall %.1f us, kernel busy %.1f us, %d kernels" % (wall, busy, len(seg)))
    c = collections.Counter(); t = collections.Counter()
    for s, e, n in seg: c[n] += 1; t[n] += (e - s) / 1e3
    for n, v in t.most_common(28): print("%8.1f us  %4d launches  avg %6.1f us  %s" % (v, c[n], v / c[n], n))
