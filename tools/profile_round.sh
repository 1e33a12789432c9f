#!/bin/bash
# Collects the judged artefacts of one round on the GPU box into gpurun_out/<tag>/ :
#   bench line, rocprofv3 kernel-trace stats (every kernel: trace, plot, gather, exposure, tonemap), PMC passes
#   (VALU / waits / HBM fetch / HBM write) merged into <tag>_pmc.json for bench.py's roofline.executed.
# Usage (through gpurun): bash tools/profile_round.sh r02   -> copy gpurun_out/r02/* of interest into profiles/.
set -u
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --launches-per-step 3 --no-cpu-baseline --no-others --no-live-counters"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- python bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS SQ_INSTS_SALU GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_valu -o p -- python bench.py $ARGS > $OUT/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR -f csv -d $OUT/pmc_wait -o p -- python bench.py $ARGS > $OUT/pmc_wait.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o p -- python bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o p -- python bench.py $ARGS > $OUT/pmc_write.log 2>&1
grep '^{' $OUT/pmc_valu.log | tail -1 > $OUT/pmc_bench.json
python tools/pmc_summary.py --json $OUT/${TAG}_pmc.json --bench $OUT/pmc_bench.json $OUT/pmc_valu/p_counter_collection.csv $OUT/pmc_wait/p_counter_collection.csv $OUT/pmc_fetch/p_counter_collection.csv $OUT/pmc_write/p_counter_collection.csv > $OUT/${TAG}_pmc_summary.txt
cp $OUT/${TAG}_pmc.json profiles/ 2>/dev/null   # so that the bench line below quotes it (same build by construction)
# The other BASELINE configs ("report VALUUtilization on the glass scene"; config 5 from LDS and from global
# memory): the VALU and the wait pass of a short run each, merged into <tag>_<config>_<fetch>_pmc.json.
for CF in "glass-720p lds" "replicated-1080p lds" "replicated-1080p global" "demo-720p lds" "spill-1080p lds"; do
  set -- $CF
  OARGS="--steps 1 --warmup 1 --launches-per-step 3 --batches-per-launch 64 --no-cpu-baseline --no-others --no-live-counters --config $1 --fetch $2"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS SQ_INSTS_SALU GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_valu_$1_$2 -o p -- python bench.py $OARGS > $OUT/pmc_valu_$1_$2.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR -f csv -d $OUT/pmc_wait_$1_$2 -o p -- python bench.py $OARGS > $OUT/pmc_wait_$1_$2.log 2>&1
  grep '^{' $OUT/pmc_valu_$1_$2.log | tail -1 > $OUT/pmc_bench_$1_$2.json
  python tools/pmc_summary.py --json $OUT/${TAG}_$1_$2_pmc.json --bench $OUT/pmc_bench_$1_$2.json $OUT/pmc_valu_$1_$2/p_counter_collection.csv $OUT/pmc_wait_$1_$2/p_counter_collection.csv >> $OUT/${TAG}_pmc_summary.txt
  cp $OUT/${TAG}_$1_$2_pmc.json profiles/ 2>/dev/null
done
# the secondary kernels at 1080p: un-fused App run under the kernel trace (plot / gather / exposure / tonemap rows)
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_app -o t -- python -c "
import robigo_luculenta_amd as R
R.app_run(1920, 1080, 48, concurrency=4, fused=False, tonemap_interval_ms=50)" > $OUT/trace_app.log 2>&1
python tools/overlap_summary.py $OUT/trace_app/t_kernel_trace.csv > $OUT/app_overlap.txt 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
head -8 $OUT/trace/t_kernel_stats.csv
head -8 $OUT/trace_app/t_kernel_stats.csv
cat $OUT/${TAG}_pmc_summary.txt
