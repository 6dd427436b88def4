import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60], r["Stream_Id"], r["Queue_Id"], r["Thread_Id"]))
rows.sort()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
lo = ad[-4]
t0 = rows[lo][1]
# first 12 kernels of stream 0 after adamw, and last kernel of each other stream before the first of them
main = [r for r in rows[lo + 1:] if r[3] == rows[lo][3]][:10]
print("adamw stream", rows[lo][3], "queue", rows[lo][4])
for r in main: print(f"  main  start {(r[0]-t0)/1e6:7.3f} ms dur {(r[1]-r[0])/1e3:7.1f} us  {r[2]}")
first = main[0][0]
for sid in sorted({r[3] for r in rows}):
    prev = [r for r in rows[lo:] if r[3] == sid and r[1] <= first]
    nxt = [r for r in rows[lo:] if r[3] == sid and r[0] > first]
    if prev: print(f"stream {sid} (queue {prev[-1][4]}): last kernel ending before main's first: end {(prev[-1][1]-t0)/1e6:7.3f} ms {prev[-1][2]}; kernels before {len(prev)}, after {len(nxt)}")
