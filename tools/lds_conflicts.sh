#!/bin/bash
# Where the trace kernel's LDS bank conflicts come from (VERDICT r03 #1b).  Two measurements, one text file:
#   1. the LDS counters of the trace kernel on the built-in scene and on three ablations of it (no prisms; the 7 fixed objects
#      only; every material diffuse grey) -- differences attribute the conflict cycles to the prism machinery, the sphere
#      machinery and the bounce;
#   2. tools/lds_conflict_microbench.hip: LDS-array cycles and conflict cycles per wave-instruction of every access class the
#      kernel uses (cross-lane fetches, record gathers at the scene's strides, ring traffic, the min-merge), with the kernel's
#      shape (16 waves per CU).
# Usage (through gpurun): bash tools/lds_conflicts.sh <tag>  ->  gpurun_out/<tag>/lds_conflicts.txt
set -u
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CTR="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
{
echo "LDS counters of rl_trace_kernel per 64-ray segment (rocprofv3 --pmc $CTR; last dispatch of a 64-batch launch)"
for CFG in demo-1080p ablate-noprisms ablate-fixed7 ablate-allgrey glass-720p replicated-1080p; do
  ARGS="--steps 1 --warmup 1 --launches-per-step 2 --batches-per-launch 64 --no-cpu-baseline --no-others --no-live-counters --config $CFG"
  rocprofv3 --pmc $CTR -f csv -d $OUT/lds_$CFG -o p -- python bench.py $ARGS > $OUT/lds_$CFG.log 2>&1
  python - $CFG $OUT <<'PY'
import csv, json, collections, sys
cfg, out = sys.argv[1], sys.argv[2]
b = json.loads([l for l in open("%s/lds_%s.log" % (out, cfg)) if l.startswith("{")][-1])
d = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("%s/lds_%s/p_counter_collection.csv" % (out, cfg))):
    if "rl_trace" in r["Kernel_Name"]:
        d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
c = d[sorted(d, key=int)[-1]]
s = b["roofline"]["rays_per_launch"] / 64.0
cu_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 256.0
print("%-18s LDS instructions %6.1f  LDS-array cycles %7.1f  of which bank conflicts %6.1f (%4.1f %%)  address conflicts %5.1f   |  LDS array busy %4.1f %% of CU time, conflicts %4.1f %% of CU time, waves waiting to issue an LDS instruction %4.1f %% of wave time"
      % (cfg, c["SQ_INSTS_LDS"] / s, c["SQ_LDS_IDX_ACTIVE"] / s, c["SQ_LDS_BANK_CONFLICT"] / s, 100.0 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
         c["SQ_LDS_ADDR_CONFLICT"] / s, 100.0 * c["SQ_LDS_IDX_ACTIVE"] / cu_cycles, 100.0 * c["SQ_LDS_BANK_CONFLICT"] / cu_cycles, 100.0 * c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"]))
PY
done
echo
echo "tools/lds_conflict_microbench.hip: run time"
tools/lds_mb
echo
echo "the same under rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVES -f csv -d $OUT/lds_mb_pmc -o p -- tools/lds_mb > $OUT/lds_mb_pmc.log 2>&1
python - $OUT <<'PY'
import csv, collections, sys
d = collections.defaultdict(lambda: collections.defaultdict(float)); last = {}
for r in csv.DictReader(open(sys.argv[1] + "/lds_mb_pmc/p_counter_collection.csv")):
    last[r["Kernel_Name"]] = r["Dispatch_Id"]
    d[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
for k, disp in last.items():
    c = d[(k, disp)]
    n = c["SQ_INSTS_LDS"]
    print("%-34s per LDS wave-instruction: LDS-array cycles %5.2f  bank-conflict cycles %5.2f  address-conflict cycles %4.2f" % (
        k.split("(")[0], c["SQ_LDS_IDX_ACTIVE"] / n, c["SQ_LDS_BANK_CONFLICT"] / n, c["SQ_LDS_ADDR_CONFLICT"] / n))
PY
} > $OUT/lds_conflicts.txt 2>&1
cat $OUT/lds_conflicts.txt
