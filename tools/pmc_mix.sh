#!/bin/bash
# The trace kernel's whole instruction mix per 64-ray segment -- vector, scalar, LDS, branch, scalar-memory, vector-memory --
# and how a wave's time splits into issuing / stalled / waiting, from two counter passes of a short bench run.
# Usage (through gpurun): [RL_LIBRARY=...] bash tools/pmc_mix.sh <tag> [config=demo-1080p] [fetch=lds]
set -u
TAG=${1:-mix}; CFG=${2:-demo-1080p}; FETCH=${3:-lds}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --launches-per-step 2 --batches-per-launch 64 --no-cpu-baseline --no-others --no-live-counters --config $CFG --fetch $FETCH"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS GRBM_GUI_ACTIVE -f csv -d $OUT/mix1_$CFG-$FETCH -o p -- python bench.py $ARGS > $OUT/mix1_$CFG-$FETCH.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC -f csv -d $OUT/mix2_$CFG-$FETCH -o p -- python bench.py $ARGS > $OUT/mix2_$CFG-$FETCH.log 2>&1
grep '^{' $OUT/mix1_$CFG-$FETCH.log | tail -1 > $OUT/mix_bench_$CFG-$FETCH.json
python - <<PY
import csv, json, collections
b = json.load(open("$OUT/mix_bench_$CFG-$FETCH.json"))
c = collections.defaultdict(float); ns = None
for f in ("$OUT/mix1_$CFG-$FETCH/p_counter_collection.csv", "$OUT/mix2_$CFG-$FETCH/p_counter_collection.csv"):
    d = collections.defaultdict(lambda: collections.defaultdict(float)); t = {}
    for r in csv.DictReader(open(f)):
        if "rl_trace" in r["Kernel_Name"]:
            d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            t[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = sorted(d, key=int)[-1]
    c.update(d[k]); ns = t[k]
seg = b["roofline"]["rays_per_launch"] / 64.0
print("$CFG $FETCH  %.1f Mrays/s  kernel %.2f ms" % (b["value"], ns / 1e6))
names = ["SQ_INSTS", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM"]
print("  per 64-ray segment: " + "  ".join("%s %.0f" % (n[3:], c[n] / seg) for n in names))
known = sum(c[n] for n in names[1:])
print("  others (s_waitcnt, s_nop, s_barrier, ...): %.0f" % ((c["SQ_INSTS"] - known) / seg))
w = c["SQ_WAVE_CYCLES"]
print("  wave time: waiting %.1f %%, issue-stalled %.1f %%, issuing %.1f %% (VALU %.1f, scalar %.1f, LDS %.1f, misc %.1f)" % tuple(
    100.0 * c[n] / w for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC")))
print("  wave cycles per segment %.0f (x4 quad-cycles), per instruction %.2f" % (4.0 * w / seg, 4.0 * w / c["SQ_INSTS"]))
PY
