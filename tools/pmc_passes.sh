#!/bin/bash
# Collects rocprofv3 PMC passes for one bench pass (each pass is its own run). Usage: tools_pmc.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-pmc}
export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
run() { # name, counters...
  n=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fast-tree $BENCH_ARGS > $OUT/$n.log 2>&1
  echo "pass $n exit $?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU && \
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE && \
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
