#!/bin/bash
# Alternating short bench runs of the tree's library and alternative builds (robigo_luculenta_amd/librl_alt_<name>.so), no
# parity run first (tools/ab2.sh does that).  Usage (through gpurun): [N=2] bash tools/abq.sh name...
set -u
N=${N:-2}
mkdir -p gpurun_out/ab
for i in $(seq $N); do
for which in tree "$@"; do
  if [ $which != tree ]; then export RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_$which.so; else unset RL_LIBRARY; fi
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-live-counters > gpurun_out/ab/$which$i.json 2> gpurun_out/ab/$which$i.err
  python - $which gpurun_out/ab/$which$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("%-10s" % sys.argv[1], "demo", round(d["value"]), " ".join("%s %d" % (o["config"].split("-")[0] + ("-global" if "global" in o["workload"] else ""), round(o["value"])) for o in d["config"].get("others", [])[:5]))
PY
done
done
