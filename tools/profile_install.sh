#!/bin/bash
# Copies what a tools/profile_refresh.sh run left under gpurun_out/<tag>/ into profiles/ under the round's names.
# Usage: bash tools/profile_install.sh <tag> <round prefix, e.g. r02>
set -e
T=gpurun_out/$1; P=profiles/$2
cp $T/pmc_raw.txt ${P}_pmc_raw.txt
cp $T/pmc_derived.json ${P}_pmc_derived.json
cp $(find $T/stats1 -name "*kernel_stats.csv" | head -1) ${P}_kernel_stats_single_stream.csv
cp $(find $T/stats2 -name "*kernel_stats.csv" | head -1) ${P}_kernel_stats.csv
cp $T/bench.json ${P}_bench.json
ls -la ${P}_*
for n in interior diffuse_box diffuse_box_8passes 10Mtris c4_share_of_8_fast; do [ -f $T/bench_$n.json ] && cp $T/bench_$n.json ${P}_bench_$n.json; done
ls ${P}_bench_*
