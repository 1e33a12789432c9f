#!/bin/bash
# One counter pass on the trace kernel's instruction fetch: shader instruction cache requests / hits / misses and the
# time waves wait for instructions.  Usage (through gpurun): bash tools/pmc_icache.sh <tag> [config=demo-1080p] [extra bench args]
set -u
TAG=${1:-ic}; CFG=${2:-demo-1080p}; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --launches-per-step 2 --batches-per-launch 64 --no-cpu-baseline --no-others --no-live-counters --config $CFG $*"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -f csv -d $OUT/ic_$CFG -o p -- python bench.py $ARGS > $OUT/ic_$CFG.log 2>&1
python - <<PY
import csv, collections
d = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("$OUT/ic_$CFG/p_counter_collection.csv")):
    if "rl_trace" in r["Kernel_Name"]:
        d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
c = d[sorted(d, key=int)[-1]]
print("$CFG", {k: "%.4g" % v for k, v in sorted(c.items())})
req = c.get("SQC_ICACHE_REQ", 0.0)
if req:
    print("  icache hit rate %.4f  misses per request %.4f (duplicates %.4f)  requests per wave cycle %.4f" %
          (c["SQC_ICACHE_HITS"] / req, c["SQC_ICACHE_MISSES"] / req, c.get("SQC_ICACHE_MISSES_DUPLICATE", 0) / req, req / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
PY
