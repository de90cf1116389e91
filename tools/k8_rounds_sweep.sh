#!/bin/bash
# Sweep of SRX_K8_ROUNDS (a launch takes 8 rays per octet only if it has that many times 8 192 waves of them). Usage: k8_rounds_sweep.sh "<bench args>" r1 r2 ...
cd $GRAFT_REPO_ROOT
args=$1; shift
for r in "$@"; do
  env SRX_K8_ROUNDS=$r timeout -k 10 300 python bench.py --no-cpu-baseline $args > gpurun_out/k8r_$r.json 2> gpurun_out/k8r_$r.err || { echo "$r FAILED"; tail -3 gpurun_out/k8r_$r.err; exit 1; }
  python - "$r" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/k8r_{sys.argv[1]}.json"))
print("rounds", sys.argv[1], round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), {k: round(v, 1) for k, v in d["roofline"]["kernel_ms"].items()})
PY
done
