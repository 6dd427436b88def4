#!/bin/bash
# same-box A/B of whole-step throughput across source trees: tools/ab_bench.sh <rounds> <dir1>[:ENV=VAL] <dir2> ...   ('.' = this tree)
# (device-to-device spread on the pool is ~10 %, larger than most deltas: only numbers from ONE box are comparable)
rounds=$1; shift
R=$(pwd)
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    d=${spec%%:*}; envs=""; [ "$spec" != "$d" ] && envs=${spec#*:}
    ( cd $R/$d && [ -n "$envs" ] && export $envs; timeout -k 10 200 python bench.py --steps 24 --warmup 3 --no-cpu-baseline $( [ -f $R/$d/tools/bench_gemm_r2.py ] || grep -q no-extras bench.py && echo --no-extras ) > /tmp/ab.json 2> /tmp/ab.err; python - <<PY
import json
try:
    d = json.load(open("/tmp/ab.json")); g = d["roofline"].get("all_gemm", {})
    print("$spec", d["value"], "samples/s", d["ms_per_step"], "ms/step  host", d.get("host_enqueue_ms_per_step"), {k: v["ms"] for k, v in g.items()})
except Exception as e:
    print("$spec FAILED", e, open("/tmp/ab.err").read()[-400:])
PY
    )
  done
done
