#!/bin/bash
# kernel-trace a command and print per-kernel medians: tools/kt.sh <substr> <out.txt> -- python ...
sub=$1; out=$2; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt -o kt --output-format csv -- "$@" > /root/repo/gpurun_out/kt.log 2>&1 || { tail -20 /root/repo/gpurun_out/kt.log; exit 1; }
python /root/repo/tools/kstat.py /tmp/kt "$sub" $CHUNK > /root/repo/gpurun_out/$out && cat /root/repo/gpurun_out/$out
