#!/bin/bash
# Two counter passes that say where a trace-kernel wave's time goes: issue per instruction class, waits, branches, instruction mix.
# Usage (through gpurun): bash tools/pmc_stalls.sh <tag> [config=demo-1080p]
set -u
TAG=${1:-q}; CFG=${2:-demo-1080p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --launches-per-step 2 --batches-per-launch 64 --no-cpu-baseline --no-others --no-live-counters --config $CFG"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -f csv -d $OUT/st1_$CFG -o p -- python bench.py $ARGS > $OUT/st1_$CFG.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CYCLES -f csv -d $OUT/st2_$CFG -o p -- python bench.py $ARGS > $OUT/st2_$CFG.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 -f csv -d $OUT/st3_$CFG -o p -- python bench.py $ARGS > $OUT/st3_$CFG.log 2>&1
grep '^{' $OUT/st1_$CFG.log | tail -1 > $OUT/st_bench_$CFG.json
python - <<PY
import csv, json, collections
b = json.load(open("$OUT/st_bench_$CFG.json"))
c = {}
for p in ("st1", "st2", "st3"):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open("$OUT/%s_$CFG/p_counter_collection.csv" % p)):
        if "rl_trace" in r["Kernel_Name"]:
            d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    c.update(d[sorted(d, key=int)[-1]])
segs64 = b["roofline"]["rays_per_launch"] / 64.0
wc = c["SQ_WAVE_CYCLES"]
print("$CFG per 64-ray segment: VALU %.0f SALU %.0f branch %.0f LDS %.0f trans_f32 %.0f  f64 add/mul/fma/trans %.0f/%.0f/%.0f/%.0f int32 %.0f int64 %.0f cvt %.0f fma_f32 %.0f"
      % tuple(c[k] / segs64 for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VALU_TRANS_F32",
                                       "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
                                       "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_FMA_F32")))
print("  share of wave time: issuing any %.3f (valu %.3f, scalar %.3f, lds %.3f, misc %.3f)  wait_inst_any %.3f (lds %.3f)  wait_any %.3f  salu inst cycles %.3f"
      % tuple(c[k] / wc for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC",
                                   "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_INST_CYCLES_SALU")))
PY
