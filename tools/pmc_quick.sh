#!/bin/bash
# One counter pass (VALU instructions, active lanes, cycles) of a short bench run of one config; prints the per-segment figures.
# Usage (through gpurun): bash tools/pmc_quick.sh <tag> [config=demo-1080p] [fetch=lds]
set -u
TAG=${1:-q}; CFG=${2:-demo-1080p}; FETCH=${3:-lds}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --launches-per-step 2 --batches-per-launch 64 --no-cpu-baseline --no-others --no-live-counters --config $CFG --fetch $FETCH"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS SQ_INSTS_SALU GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_valu_$CFG-$FETCH -o p -- python bench.py $ARGS > $OUT/pmc_valu_$CFG-$FETCH.log 2>&1
grep '^{' $OUT/pmc_valu_$CFG-$FETCH.log | tail -1 > $OUT/pmc_bench_$CFG-$FETCH.json
python - <<PY
import csv, json, collections
b = json.load(open("$OUT/pmc_bench_$CFG-$FETCH.json"))
d = collections.defaultdict(lambda: collections.defaultdict(float)); ns = {}
for r in csv.DictReader(open("$OUT/pmc_valu_$CFG-$FETCH/p_counter_collection.csv")):
    if "rl_trace" in r["Kernel_Name"]:
        d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        ns[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
k = sorted(d, key=int)[-1]; c = d[k]
segs64 = b["roofline"]["rays_per_launch"] / 64.0
cyc = (c["GRBM_GUI_ACTIVE"] / 8.0) / (c["SQ_INSTS_VALU"] / 1024.0)
lanes = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
print("$CFG $FETCH: VALU insts / 64-ray segment %.0f, SALU %.0f, cycles / VALU inst / SIMD %.2f, active lanes %.3f, useful lane-slots vs 2 cycles %.3f, kernel %.2f ms"
      % (c["SQ_INSTS_VALU"] / segs64, c["SQ_INSTS_SALU"] / segs64, cyc, lanes, 2.0 / cyc * lanes, ns[k] / 1e6))
PY
