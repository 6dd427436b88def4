"""Per-step kernel-time summary from a rocprofv3 kernel trace of bench.py: python tools/stepstat.py <dir> <timed_steps K>

Only the plan-replayed timed steps are counted: the AdamW kernel runs once per step, so the window is
(end of the AdamW that precedes the last K steps, end of the AdamW of step K-1] — model initialisation (copyBuffer, ATen fills /
normal_), warm-up, plan recording and the final instrumented single-stream step of bench.py stay outside the table."""
import csv, glob, sys, collections, re
d = sys.argv[1]; K = int(float(sys.argv[2]))
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)[:70]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
adam_all = sorted(e for s, e, n in rows if n.startswith("adamw_kernel"))
adam = []                                       # one time stamp per step: the END of its last AdamW launch (the pipelined optimizer of the
for e in adam_all:                              # data-parallel reducer launches AdamW once per gradient bucket, a few ms apart)
    if adam and e - adam[-1] < 20e6:
        adam[-1] = e
    else:
        adam.append(e)
if len(adam) < K + 1:
    sys.exit(f"stepstat: {len(adam)} AdamW launches in the trace, need {K + 1} to delimit {K} timed steps")
steps = K - 1                                   # the last timed step of bench.py is the eager instrumented one
t0, t1 = adam[-K - 1], adam[-2]
tot = collections.defaultdict(float); cnt = collections.Counter()
for s, e, n in rows:
    if s >= t0 and e <= t1:
        tot[n] += (e - s) / 1e6; cnt[n] += 1
all_ms = sum(tot.values())
print(f"window: {steps} replayed steps, {(t1 - t0) / 1e6 / steps:.2f} ms wall per step; total kernel time {all_ms / steps:.2f} ms/step (sum over all streams)")
for n, t in sorted(tot.items(), key=lambda x: -x[1])[:32]:
    print(f"{t/steps:8.3f} ms/step {cnt[n]/steps:8.1f} launches/step  avg {t/cnt[n]*1e3:8.1f} us  {n}")
