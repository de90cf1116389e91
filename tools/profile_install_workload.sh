#!/bin/bash
# Copies a tools/profile_workloads.sh result into profiles/. Usage: bash tools/profile_install_workload.sh <tag> <name> <round prefix, e.g. r03>
set -e
T=gpurun_out/$1_$2; P=profiles/$3
S=_$2; [ "$2" = atrium ] && S=""
cp $T/pmc_raw.txt ${P}_pmc_raw$S.txt
cp $T/pmc_derived.json ${P}_pmc_derived$S.json
cp $(find $T/stats1 -name "*kernel_stats.csv" | head -1) ${P}_kernel_stats_single_stream$S.csv
cp $(find $T/stats2 -name "*kernel_stats.csv" | head -1) ${P}_kernel_stats$S.csv
cp $T/bench.json ${P}_bench_pre$S.json
ls -la ${P}_*$S.*
