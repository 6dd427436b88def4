"""Per-step kernel-time summary from a rocprofv3 kernel trace of bench.py: python tools/stepstat.py <dir> <steps_total>"""
import csv, glob, sys, collections, re
d = sys.argv[1]; steps = float(sys.argv[2])
tot = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)[:70]
        tot[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; cnt[n] += 1
all_ms = sum(tot.values())
print(f"total kernel time {all_ms/steps:.2f} ms/step over {steps:.0f} steps")
for n, t in sorted(tot.items(), key=lambda x: -x[1])[:28]:
    print(f"{t/steps:8.3f} ms/step {cnt[n]/steps:8.1f} launches/step  avg {t/cnt[n]*1e3:8.1f} us  {n}")
