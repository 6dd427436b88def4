#!/bin/bash
# same-box A/B of bench.py under environment switches: tools/ab_env.sh <tag> "<ENV=..>" "<ENV=..>" ...   ("-" = no switch)
# two interleaved rounds, ms/step of each run -> gpurun_out/<tag>.txt
tag=$1; shift
out=gpurun_out/${tag}.txt
: > $out
for round in 1 2; do
  for e in "$@"; do
    [ "$e" = "-" ] && e=""
    ms=$(env $e python3 bench.py --no-cpu-baseline --no-extras --steps 24 $BENCH_FLAGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "round $round  [${e:-default}]  ms/step, samples/s: $ms" | tee -a $out
  done
done
