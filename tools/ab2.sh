#!/bin/bash
# A/B of alternative builds (robigo_luculenta_amd/librl_alt_<name>.so) against the tree's library: a quick parity check of
# each alternative, then N alternating short bench runs.  Usage (through gpurun): [N=2] bash tools/ab2.sh name...
set -u
N=${N:-2}
mkdir -p gpurun_out/ab
for a in "$@"; do
  RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_$a.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_independent.py -x -q -m gpu -k "bit_exact or matrix or independent or random_scenes" 2>&1 | tail -1 | sed "s/^/$a: /"
done
for i in $(seq $N); do
  for which in tree "$@"; do
    if [ $which != tree ]; then export RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_$which.so; else unset RL_LIBRARY; fi
    timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/ab/$which$i.json 2> gpurun_out/ab/$which$i.err
    python - $which gpurun_out/ab/$which$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("%-10s" % sys.argv[1], "demo", round(d["value"]), " ".join("%s %d" % (o["config"].split("-")[0] + ("-global" if "global" in o["workload"] else ""), round(o["value"])) for o in d["config"].get("others", [])[:4]))
PY
  done
done
