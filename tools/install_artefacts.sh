#!/bin/bash
# Copies the summaries of gpurun_out/<tag>/ (tools/artefact_round.sh) into profiles/ under the tag's name.
set -eu
TAG=$1
SRC=gpurun_out/$TAG
cp $SRC/trace/t_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
cp $SRC/bench.json profiles/${TAG}_bench.json
cp $SRC/configs.txt profiles/${TAG}_configs.txt
cp $SRC/image_parity.txt profiles/${TAG}_image_parity.txt
cp $SRC/kernel_stats.txt profiles/${TAG}_kernel_event_stats.txt
cp $SRC/app.txt profiles/${TAG}_app.txt
cp $SRC/big_parity.txt profiles/${TAG}_big_parity.txt
tail -4 gpurun_out/${TAG}_console.txt > profiles/${TAG}_pmc_summary.txt
python - "$TAG" <<'PY'
import json, re, sys
tag = sys.argv[1]
d = json.load(open("profiles/%s_bench.json" % tag))
s = open("profiles/%s_pmc_summary.txt" % tag).read()
g = lambda k: float(re.search(k + r"=([0-9.e+]+)", s).group(1))
rays = d["value"] * 1e6 * d["ms_per_step"] * 1e-3
print("Mrays/s %.1f  Mpaths/s %.1f  batches/s %.0f  ms/step %.2f  frac %.3f  HBM GB/s %.0f  cpu %.2f Mrays/s"
      % (d["value"], d["mpaths_per_s"], d["batches_per_s"], d["ms_per_step"], d["roofline"]["frac"],
         d["roofline"]["hbm"]["achieved"], d["cpu_baseline"]["value"]))
print("VALU instructions per 64-ray segment %.0f  active lanes %.1f %%  cycles per VALU instruction per SIMD %.2f"
      % (g("SQ_INSTS_VALU") / (rays / 64), 100 * g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64),
         (g("GRBM_GUI_ACTIVE") / 8) / (g("SQ_INSTS_VALU") / 1024)))
PY
