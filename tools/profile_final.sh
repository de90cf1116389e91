#!/bin/bash
# The round's committed bench lines, with the kernels as they are and the PMC-derived figures already under profiles/
# (bench.py looks them up per workload): the headline line as the driver runs it, then the other BASELINE.json workloads.
# Usage (on the GPU box): bash tools/profile_final.sh <tag>   -> gpurun_out/<tag>/bench*.json; then, here,
#   for f in gpurun_out/<tag>/bench*.json; do cp $f profiles/r03_$(basename $f); done
R=$GRAFT_REPO_ROOT; TAG=${1:-final}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
run() { name=$1; shift; echo "== $name: bench.py $*"; timeout -k 10 500 python3 bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -3 $OUT/$name.err; exit 1; }; }
run bench
run bench_64spp --steps 64 --warmup 1 --no-cpu-baseline --no-fast-tree
run bench_interior --scene interior --no-cpu-baseline --steps 3
run bench_diffuse_box --scene diffuse_box --tris 50000 --no-cpu-baseline --steps 5
run bench_diffuse_box_mcm32 --scene diffuse_box --tris 50000 --mcm 32 --no-cpu-baseline --steps 4
run bench_diffuse_box_8passes --scene diffuse_box --tris 50000 --no-cpu-baseline --steps 16 --warmup 8 --passes-per-call 8 --no-fast-tree
run bench_10Mtris --tris 10000000 --no-cpu-baseline --steps 3
run bench_instanced10M --scene instanced --tris 10000000 --no-cpu-baseline --steps 3
run bench_c4_instanced_share_of_8_fast --scene instanced --tris 10000000 --res 3840x2160 --share-of 8 --no-cpu-baseline --steps 4 --warmup 2 --fast-build
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
    a = json.load(open(f)); r = a["roofline"]
    ft = a.get("fast_tree") or {}
    print(f"{os.path.basename(f):44s} {a['value']:8.1f} Mrays/s  {a['ms_per_step']:8.2f} ms/step  frac {r.get('frac')}  fast tree {ft.get('value')}")
PY
