#!/bin/bash
# GPU box: rocprofv3 kernel trace of the bench step with the reducer forced on over a ONE-rank RCCL communicator
tag=${1:-r4}
steps=${2:-8}
R=/root/repo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd
export VACNIC_BENCH_FORCE_DDP=${FORCE:-1}
rocprofv3 --kernel-trace --stats -d /tmp/pd -o pd --output-format csv -- python3 $R/bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-extras --no-roofline-step > $R/gpurun_out/${tag}_prof_ddp1.json 2> $R/gpurun_out/${tag}_prof_ddp1.err || { tail -5 $R/gpurun_out/${tag}_prof_ddp1.err; exit 1; }
python3 $R/tools/stepstat.py /tmp/pd $((steps + 1)) > $R/gpurun_out/${tag}_stepstat_ddp1.txt
python3 $R/tools/timeline.py /tmp/pd 1.0 > $R/gpurun_out/${tag}_timeline_ddp1.txt
head -30 $R/gpurun_out/${tag}_stepstat_ddp1.txt
