#!/bin/bash
# PMC passes + kernel stats + bench line for the non-headline BASELINE.json workloads, one
# tools/profile_refresh.sh run per workload. Usage (on the GPU box): bash tools/profile_workloads.sh <tag> <name> [<name> ...]
# names: atrium interior diffuse_box diffuse_box_mcm32 10Mtris instanced10M. Results: gpurun_out/<tag>_<name>/.
R=$GRAFT_REPO_ROOT; TAG=$1; shift
for n in "$@"; do
  case $n in
    atrium) A="";;
    interior) A="--scene interior";;
    diffuse_box) A="--scene diffuse_box --tris 50000";;
    diffuse_box_mcm32) A="--scene diffuse_box --tris 50000 --mcm 32";;
    10Mtris) A="--tris 10000000";;
    instanced10M) A="--scene instanced --tris 10000000";;
    *) echo "unknown workload $n"; exit 2;;
  esac
  echo "== $n: bench.py $A"
  BENCH_ARGS="$A" bash $R/tools/profile_refresh.sh ${TAG}_$n || { echo "profile of $n failed"; exit 1; }
done
