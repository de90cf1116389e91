#!/bin/bash
# Bench with the traversal stack wholly in LDS against forced LDS depths (spill path compiled in). Usage: spill_ab.sh "<bench args>" lib:depth ...
cd $GRAFT_REPO_ROOT
args=$1; shift
export SRX_SHADOW_STREAM=0
for spec in "$@"; do
  lib=${spec%%:*}; depth=${spec##*:}
  tag=${lib}_${depth}
  env SRX_LIB=$GRAFT_REPO_ROOT/raytracing_amd/$lib SRX_LDS_STACK=$depth timeout -k 10 300 python bench.py --no-cpu-baseline $args > gpurun_out/sp_$tag.json 2> gpurun_out/sp_$tag.err || { echo "$spec FAILED"; tail -3 gpurun_out/sp_$tag.err; exit 1; }
  python - "$spec" "gpurun_out/sp_$tag.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), {k: round(v, 1) for k, v in d["roofline"]["kernel_ms"].items()})
PY
done
