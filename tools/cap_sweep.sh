#!/bin/bash
# ray_capacity sweep on the bench. Usage: cap_sweep.sh log2cap...
cd $GRAFT_REPO_ROOT
for c in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 --ray-capacity $((1 << c)) > gpurun_out/cap_$c.json 2> gpurun_out/cap_$c.err || { echo "cap 2^$c FAILED"; tail -3 gpurun_out/cap_$c.err; exit 1; }
  python - "$c" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/cap_{sys.argv[1]}.json"))
print("capacity 2^" + sys.argv[1], "rate", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), {k: round(v, 1) for k, v in d["roofline"]["kernel_ms"].items()}, "launches", d["roofline"]["launches"])
PY
done
