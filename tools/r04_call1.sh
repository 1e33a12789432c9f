#!/bin/bash
# Round-4 GPU call 1: parity of the new build, A/B against HEAD and against the f64 libm, LDS-conflict attribution.
set -u
OUT=gpurun_out/c1; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
for i in 1 2; do
  for which in head f64 new; do
    if [ $which != new ]; then export RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_$which.so; else unset RL_LIBRARY; fi
    timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/ab_$which$i.json 2> $OUT/ab_$which$i.err
    python - $which $OUT/ab_$which$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "demo", round(d["value"]), " ".join("%s %d" % (o["config"].split("-")[0] + ("-global" if "global" in o["workload"] else ""), round(o["value"])) for o in d["config"].get("others", [])[:4]))
PY
  done
done
unset RL_LIBRARY
for CFG in demo-1080p ablate-fixed7 ablate-noprisms ablate-allgrey; do
  ARGS="--steps 1 --warmup 1 --launches-per-step 2 --batches-per-launch 64 --no-cpu-baseline --no-others --config $CFG"
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES -f csv -d $OUT/lds_$CFG -o p -- python bench.py $ARGS > $OUT/lds_$CFG.log 2>&1
  bash tools/pmc_quick.sh c1 $CFG lds
  python - $CFG $OUT <<'PY'
import csv, json, collections, sys
cfg, out = sys.argv[1], sys.argv[2]
b = json.loads([l for l in open("%s/lds_%s.log" % (out, cfg)) if l.startswith("{")][-1])
d = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("%s/lds_%s/p_counter_collection.csv" % (out, cfg))):
    if "rl_trace" in r["Kernel_Name"]:
        d[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
c = d[sorted(d, key=int)[-1]]
segs64 = b["roofline"]["rays_per_launch"] / 64.0
print(cfg, "per 64-ray segment:", " ".join("%s %.1f" % (k.replace("SQ_", ""), v / segs64) for k, v in sorted(c.items())))
PY
done
tools/lds_mb > $OUT/lds_mb.txt 2>&1; cat $OUT/lds_mb.txt
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
    print("%-34s per LDS wave-instruction: idx_active %.2f bank_conflict %.2f addr_conflict %.2f active_inst_lds %.2f wait_inst_lds %.2f" % (
        k.split("(")[0], c["SQ_LDS_IDX_ACTIVE"] / n, c["SQ_LDS_BANK_CONFLICT"] / n, c["SQ_LDS_ADDR_CONFLICT"] / n, c["SQ_ACTIVE_INST_LDS"] / n, c["SQ_WAIT_INST_LDS"] / n))
PY
for s in demo; do RL_LIBRARY=$PWD/robigo_luculenta_amd/librobigo_luculenta_stats.so timeout 120 python tools/kernel_stats.py 64 $s; done > $OUT/kernel_stats.txt 2>&1
tail -40 $OUT/kernel_stats.txt
