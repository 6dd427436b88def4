#!/bin/bash
# GPU box: shader clock and MFMA-pipe utilisation of one GEMM variant: tools/pmc_clock.sh M N K hint  (hint may carry debug bits: bits*1000 + tile)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/pc1 /tmp/pc2
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d /tmp/pc1 -o p --output-format csv -- python3 $R/tools/bench_gemm_one.py $1 $2 $3 $4 > /dev/null 2>&1 || echo "pass 1 failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/pc2 -o p --output-format csv -- python3 $R/tools/bench_gemm_one.py $1 $2 $3 $4 > /dev/null 2>&1 || echo "pass 2 failed"
python3 - "$@" <<'PY'
import csv, glob, sys, statistics as st
def load(d):
    c, t = {}, []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_kernel" in r["Kernel_Name"]:
                c.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_kernel" in r["Kernel_Name"]:
                t.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: st.median(v) for k, v in c.items()}, st.median(t)
c1, t1 = load("/tmp/pc1"); c2, t2 = load("/tmp/pc2")
busy = c1["SQ_BUSY_CYCLES"] / 32.0            # 32 shader engines
mfma = c2["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0  # 1024 SIMDs
print(f"hint {sys.argv[4]}: {t1:7.1f} us  cycles/launch {busy:9.0f}  shader clock {busy / t1 / 1e3:5.2f} GHz  MFMA pipe busy {mfma / busy * 100:5.1f} % of cycles")
PY
