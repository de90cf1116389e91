#!/bin/bash
# passes-per-call sweep. Usage: ppc_sweep.sh <share_of> <steps> B...
cd $GRAFT_REPO_ROOT
so=$1; steps=$2; shift 2
for b in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps $steps --warmup $b --share-of $so --passes-per-call $b > gpurun_out/ppc_${so}_$b.json 2> gpurun_out/ppc_${so}_$b.err || { echo "B=$b FAILED"; tail -3 gpurun_out/ppc_${so}_$b.err; exit 1; }
  python - "$so" "$b" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ppc_{sys.argv[1]}_{sys.argv[2]}.json"))
print("share 1/" + sys.argv[1], "B", sys.argv[2], "rate", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), {k: round(v, 1) for k, v in d["roofline"]["kernel_ms"].items()}, "launches", d["roofline"]["launches"])
PY
done
