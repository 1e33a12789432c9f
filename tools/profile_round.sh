#!/bin/bash
# Collects the judged artefacts of one round on the GPU box into gpurun_out/<tag>/ :
#   bench line, rocprofv3 kernel-trace stats, PMC passes (VALU / waits / HBM fetch / HBM write).
# Usage (through gpurun): bash tools/profile_round.sh r01
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
ARGS="--steps 4 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- python bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_valu -o p -- python bench.py $ARGS > $OUT/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_WR -f csv -d $OUT/pmc_wait -o p -- python bench.py $ARGS > $OUT/pmc_wait.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o p -- python bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o p -- python bench.py $ARGS > $OUT/pmc_write.log 2>&1
cat $OUT/bench.json
cat $OUT/trace/t_kernel_stats.csv | head -5
python tools/pmc_summary.py $OUT/pmc_valu/p_counter_collection.csv $OUT/pmc_wait/p_counter_collection.csv $OUT/pmc_fetch/p_counter_collection.csv $OUT/pmc_write/p_counter_collection.csv
