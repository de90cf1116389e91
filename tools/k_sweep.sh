#!/bin/bash
# Rays-per-octet sweep on one scene. Usage: k_sweep.sh <tris> <k...>   (k = 0 means the launcher's own choice)
cd $GRAFT_REPO_ROOT
tris=$1; shift
for k in "$@"; do
  if [ "$k" = "0" ]; then unset SRX_RAYS_PER_OCTET; else export SRX_RAYS_PER_OCTET=$k; fi
  timeout -k 10 400 python bench.py --tris $tris --no-cpu-baseline > gpurun_out/ks_${tris}_$k.json 2> gpurun_out/ks_${tris}_$k.err || { echo "k=$k FAILED"; tail -3 gpurun_out/ks_${tris}_$k.err; exit 1; }
  python - "$tris" "$k" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ks_{sys.argv[1]}_{sys.argv[2]}.json"))
print("tris", sys.argv[1], "k", sys.argv[2], "depth", d["config"].get("stack_depth"), round(d["value"], 1), {k: round(v, 1) for k, v in d["roofline"]["kernel_ms"].items()})
PY
done
