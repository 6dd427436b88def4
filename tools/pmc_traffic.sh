#!/bin/bash
# GPU box: HBM traffic counters of the bench step -> gpurun_out/pmc_hbm_traffic.json (copy into profiles/)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/pf /tmp/pw
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-plan > $R/gpurun_out/pmc_f.log 2>&1 || { tail $R/gpurun_out/pmc_f.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-plan > $R/gpurun_out/pmc_w.log 2>&1 || { tail $R/gpurun_out/pmc_w.log; exit 1; }
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $R/gpurun_out/pmc_hbm_traffic.json
