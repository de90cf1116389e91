#!/bin/bash
# Bench lines of the workloads that are not the headline, with the kernels as they are. Usage (on the GPU box):
# bash tools/profile_others.sh <tag>   -> gpurun_out/<tag>/bench_*.json (tools/profile_install.sh copies them to profiles/)
R=$GRAFT_REPO_ROOT; TAG=${1:-refresh}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
# the other workloads' bench lines (not the headline): interior (C3), 10 M triangles (C4 scale), C4 as one of 8 GPUs sees it, diffuse box (C5)
cd $R
timeout -k 10 300 python3 bench.py --scene interior --no-cpu-baseline --steps 3 > $OUT/bench_interior.json 2> $OUT/bench_interior.err || exit 1
timeout -k 10 300 python3 bench.py --scene diffuse_box --tris 50000 --no-cpu-baseline --steps 5 > $OUT/bench_diffuse_box.json 2> $OUT/bench_diffuse_box.err || exit 1
timeout -k 10 300 python3 bench.py --scene diffuse_box --tris 50000 --no-cpu-baseline --steps 16 --warmup 8 --passes-per-call 8 --no-fast-tree > $OUT/bench_diffuse_box_8passes.json 2> $OUT/bench_diffuse_box_8passes.err || exit 1
timeout -k 10 400 python3 bench.py --tris 10000000 --no-cpu-baseline --steps 3 > $OUT/bench_10Mtris.json 2> $OUT/bench_10Mtris.err || exit 1
timeout -k 10 400 python3 bench.py --tris 10000000 --res 3840x2160 --share-of 8 --no-cpu-baseline --steps 4 --warmup 2 --fast-build > $OUT/bench_c4_share_of_8_fast.json 2> $OUT/bench_c4_share_of_8_fast.err || exit 1
tail -c 400 $OUT/bench_10Mtris.json
