#!/bin/bash
# A/B of one environment switch on the bench. Usage: env_ab.sh VAR value_a value_b [bench args...]
cd $GRAFT_REPO_ROOT
var=$1; a=$2; b=$3; shift 3
for v in $a $b $a $b; do
  env $var=$v timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$var=$v FAILED"; tail -3 gpurun_out/ab_$v.err; exit 1; }
  python - "$var" "$v" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[2]}.json"))
print(sys.argv[1] + "=" + sys.argv[2], round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), {k: round(v, 1) for k, v in d["roofline"]["kernel_ms"].items()})
PY
done
