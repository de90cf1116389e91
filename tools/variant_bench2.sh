#!/bin/bash
# Like variant_bench.sh with extra bench arguments. Usage: variant_bench2.sh "<bench args>" lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
args=$1; shift
for l in "$@"; do
  SRX_LIB=$GRAFT_REPO_ROOT/raytracing_amd/$l timeout -k 10 300 python bench.py --no-cpu-baseline $args > gpurun_out/vb_$l.json 2> gpurun_out/vb_$l.err || { echo "$l FAILED"; tail -3 gpurun_out/vb_$l.err; exit 1; }
  python - "$l" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/vb_{sys.argv[1]}.json"))
print(sys.argv[1], round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), {k: round(v, 1) for k, v in d["roofline"]["kernel_ms"].items()})
PY
done
