"""Kernel sequence around a marker in a single-stream bench trace: python tools/seq_dump.py <dir> <marker substr> <before> <after>"""
import csv, glob, sys
d, marker, nb, na = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]))
rows.sort()
idx = [i for i, r in enumerate(rows) if marker in r[2]]
i = idx[len(idx) // 2]
prev_end = rows[i - nb - 1][1]
for r in rows[i - nb:i + na]:
    print(f"gap {(r[0]-prev_end)/1e3:6.1f} us  dur {(r[1]-r[0])/1e3:8.1f} us  {r[2]}")
    prev_end = r[1]
