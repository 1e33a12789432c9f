#!/bin/bash
# A/B of alternative builds (robigo_luculenta_amd/librl_alt_<name>.so) against the tree's library, like ab2.sh, with the parity
# check only for the alternatives named in PARITY (diagnostic builds -- another occupancy, an ablation -- are timed only).
# Usage (through gpurun): [N=2] [PARITY="a b"] [BENCH_ARGS=...] bash tools/ab3.sh name...
set -u
N=${N:-2}
PARITY=${PARITY:-}
BENCH_ARGS=${BENCH_ARGS:---steps 4 --warmup 1 --no-cpu-baseline --no-live-counters}
mkdir -p gpurun_out/ab
for a in $PARITY; do
  RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_$a.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_independent.py -x -q -m gpu -k "bit_exact or matrix or independent or random_scenes" 2>&1 | tail -1 | sed "s/^/$a: /"
done
for i in $(seq $N); do
  for which in tree "$@"; do
    if [ $which != tree ]; then export RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_$which.so; else unset RL_LIBRARY; fi
    timeout 300 python bench.py $BENCH_ARGS > gpurun_out/ab/$which$i.json 2> gpurun_out/ab/$which$i.err
    python - $which gpurun_out/ab/$which$i.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("%-10s" % sys.argv[1], "demo %.0f" % d["value"], " ".join("%s %d" % (o["config"].split("-")[0] + ("-global" if "global" in o["workload"] else ""), round(o["value"])) for o in d["config"].get("others", [])[:5]))
except Exception as e:
    print("%-10s" % sys.argv[1], "FAILED", e)
PY
  done
done
