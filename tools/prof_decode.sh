#!/bin/bash
# GPU box: per-token kernel breakdown of the config-5 decode loop (rocprofv3 kernel trace of tools/bench_generate.py)
tag=${1:-r2}
R=/root/repo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd
rocprofv3 --kernel-trace -d /tmp/pd -o pd --output-format csv -- python3 $R/tools/bench_generate.py 6 1 > $R/gpurun_out/${tag}_gen.json 2> $R/gpurun_out/${tag}_gen.err || { tail -5 $R/gpurun_out/${tag}_gen.err; exit 1; }
tail -1 $R/gpurun_out/${tag}_gen.json
python3 $R/tools/decode_profile.py /tmp/pd > $R/gpurun_out/${tag}_decode_profile.txt
cat $R/gpurun_out/${tag}_decode_profile.txt
