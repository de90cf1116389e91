#!/bin/bash
# One GPU renders rank 0's share of an N-rank run: per-GPU rate at 1/N of the frame (exchange excluded). Usage: share_sweep.sh N...
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --share-of $n > gpurun_out/share_$n.json 2> gpurun_out/share_$n.err || { echo "share $n FAILED"; tail -3 gpurun_out/share_$n.err; exit 1; }
  python - "$n" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/share_{sys.argv[1]}.json"))
print("share 1/" + sys.argv[1], "rate", round(d["value"], 1), "Mrays/s on this GPU ->", round(d["value"] * int(sys.argv[1]), 1), "projected; ms/step", round(d["ms_per_step"], 2), {k: round(v, 1) for k, v in d["roofline"]["kernel_ms"].items()}, "launches", d["roofline"]["launches"])
PY
done
