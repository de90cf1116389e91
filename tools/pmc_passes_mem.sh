#!/bin/bash
# Memory-pipeline PMC passes (texture addresser / L1 / data return) for one bench pass. Usage: pmc_passes_mem.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-pmcmem}
export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
run() { n=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fast-tree $BENCH_ARGS > $OUT/$n.log 2>&1
  echo "pass $n exit $?"
}
run m1 TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE && \
run m2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
