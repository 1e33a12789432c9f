#!/bin/bash
# Copies the summaries of gpurun_out/<tag>/ (tools/artefact_round.sh) into profiles/ under the tag's name.
set -eu
TAG=$1
SRC=gpurun_out/$TAG
cp $SRC/trace/t_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
cp $SRC/trace_app/t_kernel_stats.csv profiles/${TAG}_app_kernel_stats.csv
cp $SRC/app_overlap.txt profiles/${TAG}_app_overlap.txt
cp $SRC/bench.json profiles/${TAG}_bench.json
cp $SRC/${TAG}_pmc.json profiles/${TAG}_pmc.json
cp $SRC/${TAG}_pmc_summary.txt profiles/${TAG}_pmc_summary.txt
cp $SRC/${TAG}_*_pmc.json profiles/ 2>/dev/null || true   # the other BASELINE configs (profile_round.sh)
cp $SRC/image_parity.txt profiles/${TAG}_image_parity.txt
cp $SRC/kernel_stats.txt profiles/${TAG}_kernel_event_stats.txt
cp $SRC/app.txt profiles/${TAG}_app.txt
[ -f $SRC/spill_ab.txt ] && cp $SRC/spill_ab.txt profiles/${TAG}_spill_ab_final.txt
cp $SRC/big_parity.txt profiles/${TAG}_big_parity.txt
cp $SRC/bench_2ranks_gloo.txt profiles/${TAG}_bench_2ranks_gloo.txt
cp $SRC/valu_microbench.txt profiles/${TAG}_valu_microbench.txt
cp $SRC/open_launch_stress.txt profiles/${TAG}_open_launch_stress.txt
[ -f $SRC/lds_conflicts.txt ] && cp $SRC/lds_conflicts.txt profiles/${TAG}_lds_conflicts.txt
[ -f $SRC/instruction_mix.txt ] && cp $SRC/instruction_mix.txt profiles/${TAG}_instruction_mix.txt
[ -f $SRC/app_soak.txt ] && cp $SRC/app_soak.txt profiles/${TAG}_app_soak.txt
[ -f $SRC/wave_time.txt ] && cp $SRC/wave_time.txt profiles/${TAG}_wave_time.txt
[ -f $SRC/issue_cost_probes.txt ] && cp $SRC/issue_cost_probes.txt profiles/${TAG}_issue_cost_probes.txt
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open("profiles/%s_bench.json" % tag))
e = d["roofline"]["executed"]
print("frac_executed", d["roofline"].get("frac_executed"), "valu_busy", d["roofline"].get("valu_busy"))
print("Mrays/s %.1f  Mpaths/s %.1f  batches/s %.0f  ms/step %.2f  frac (executed lane-slots) %.3f  cpu %.2f Mrays/s (%d cores)"
      % (d["value"], d["mpaths_per_s"], d["batches_per_s"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
print("executed:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items()})
print("others:", [(o["config"], round(o["value"]), (o.get("executed") or {}).get("active_lanes")) for o in d["config"]["others"]])
PY
