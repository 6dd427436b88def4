#!/bin/bash
# GPU box: trace the autotune replay and pick winners -> gpurun_out/gemm_tuned.json
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/at
rocprofv3 --kernel-trace -d /tmp/at -o at --output-format csv -- python $R/tools/autotune_gemm.py run /tmp/cases.json > $R/gpurun_out/autotune_run.log 2>&1 || { tail -20 $R/gpurun_out/autotune_run.log; exit 1; }
tail -2 $R/gpurun_out/autotune_run.log
python $R/tools/autotune_gemm.py pick /tmp/cases.json /tmp/at $R/gpurun_out/gemm_tuned.json > $R/gpurun_out/autotune_pick.txt 2>&1
tail -5 $R/gpurun_out/autotune_pick.txt
