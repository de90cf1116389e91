#!/bin/bash
# Re-collects everything under profiles/ that depends on the kernels: PMC passes (one stream), kernel stats on one
# stream and on two, the default bench line. Usage (on the GPU box): bash tools/profile_refresh.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-refresh}
export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export SRX_SHADOW_STREAM=0
bash $R/tools/pmc_passes.sh $TAG/pmc && bash $R/tools/pmc_passes_mem.sh $TAG/pmcmem || exit 1
cd /tmp
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-tree $BENCH_ARGS > $OUT/stats1.log 2>&1 || exit 1
unset SRX_SHADOW_STREAM
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-tree $BENCH_ARGS > $OUT/stats2.log 2>&1 || exit 1
cd $R && timeout -k 10 300 python3 bench.py $BENCH_ARGS --no-fast-tree > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 tools/pmc_summary.py $OUT/pmc $OUT/pmcmem > $OUT/pmc_raw.txt
python3 tools/pmc_derive.py $OUT/pmc_raw.txt $(find $OUT/stats1 -name "*kernel_stats.csv" | head -1) 4 $OUT/bench.json > $OUT/pmc_derived.json
find $OUT -name "*kernel_stats.csv" | head
tail -c 300 $OUT/bench.json
