#!/bin/bash
# GPU box: same-box A/B of the weight-gradient schedules inside the default bench step -> gpurun_out/<tag>.txt
tag=${1:-r3_step_ab_wgrad}
out=gpurun_out/$tag.txt
: > $out
run() {
  label=$1; shift
  extra=""; for kv in "$@"; do case $kv in VACNIC_BENCH_ARGS=*) extra=${kv#VACNIC_BENCH_ARGS=};; esac; done
  env "$@" python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras $extra > gpurun_out/_ab.json 2>/dev/null
  python - "$label" <<'PY' >> gpurun_out/_ab_line.txt
import json, sys
r = json.load(open("gpurun_out/_ab.json"))
print(f"{sys.argv[1]:46s} {r['ms_per_step']:6.2f} ms/step {r['value']:7.2f} samples/s   TT (single-stream instrumented step): {r['roofline']['all_gemm']['TT']}")
PY
  tail -1 gpurun_out/_ab_line.txt | tee -a $out
}
rm -f gpurun_out/_ab_line.txt
if [ "$2" = "split" ]; then
for rep in 1 2; do
  run "hybrid (default)" VACNIC_WGRAD_GROUP=1
  run "hybrid, split-K factors x2" VACNIC_WGRAD_SPLIT_SCALE=2
  run "hybrid, split-K factors x0.5" VACNIC_WGRAD_SPLIT_SCALE=0.5
  run "hybrid, no measured tile table" VACNIC_GEMM_TUNED=0
done
exit 0
fi
if [ "$2" = "prio" ]; then
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" | tee -a $out
for rep in 1 2; do
  run "default priorities" VACNIC_WGRAD_GROUP=1
  run "no branch stream (encoder branches on the compute stream)" VACNIC_NO_BRANCH_STREAM=1
  run "no weight-gradient stream" VACNIC_NO_WGRAD_STREAM=1
  run "two tower streams" VACNIC_TWO_TOWER_STREAMS=1
done
exit 0
fi
if [ "$2" = "units" ]; then
for rep in 1 2; do
  run "hybrid, flush at 16 blocks (default)" VACNIC_WGRAD_GROUP=1
  run "hybrid, flush at 56 blocks" VACNIC_WGRAD_GROUP_UNITS=56
  run "hybrid, flush at 168 blocks (= the decoder)" VACNIC_WGRAD_GROUP_UNITS=168
  run "hybrid, flush at the end of backward" VACNIC_WGRAD_GROUP_UNITS=100000
  run "split-K everything" VACNIC_WGRAD_GROUP=0
done
exit 0
fi
for rep in 1 2; do
  run "grouped everything, 4096-row phases" VACNIC_WGRAD_GROUP_MAX_M=1000000 VACNIC_WGRAD_PHASE_ROWS=4096
  run "grouped everything, 2048-row phases" VACNIC_WGRAD_GROUP_MAX_M=1000000 VACNIC_WGRAD_PHASE_ROWS=2048
  run "grouped everything, whole reduction per launch" VACNIC_WGRAD_GROUP_MAX_M=1000000
  run "split-K with fp32 atomics (round 2)" VACNIC_WGRAD_GROUP=0
  run "hybrid: grouped for M<=4096, split-K above (default)" VACNIC_WGRAD_GROUP=1
done
