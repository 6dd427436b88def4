#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of the default bench step (multi-stream) and of the single-stream schedule;
# per-kernel ms/step tables + the coarse stream timeline go to gpurun_out/<tag>_*.txt (copy what should be judged to profiles/).
tag=${1:-r2}
steps=${2:-8}
R=/root/repo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps_ms /tmp/ps_ss
rocprofv3 --kernel-trace --stats -d /tmp/ps_ms -o ms --output-format csv -- python3 $R/bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/${tag}_prof_ms.json 2> $R/gpurun_out/${tag}_prof_ms.err || { tail -5 $R/gpurun_out/${tag}_prof_ms.err; exit 1; }
python3 $R/tools/stepstat.py /tmp/ps_ms $steps > $R/gpurun_out/${tag}_stepstat_multistream.txt
python3 $R/tools/timeline.py /tmp/ps_ms 2.0 > $R/gpurun_out/${tag}_timeline_streams.txt
cp $(find /tmp/ps_ms -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_kernel_stats_multistream.csv 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/ps_ss -o ss --output-format csv -- python3 $R/bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-extras --no-streams > $R/gpurun_out/${tag}_prof_ss.json 2> $R/gpurun_out/${tag}_prof_ss.err || { tail -5 $R/gpurun_out/${tag}_prof_ss.err; exit 1; }
python3 $R/tools/stepstat.py /tmp/ps_ss $steps > $R/gpurun_out/${tag}_stepstat_single_stream.txt
cp $(find /tmp/ps_ss -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_kernel_stats_single_stream.csv 2>/dev/null
head -45 $R/gpurun_out/${tag}_stepstat_single_stream.txt
