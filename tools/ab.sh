#!/bin/bash
# A/B of builds of the library on the GPU box: [N=rounds] tools/ab.sh <alt .so>...   (through gpurun; an alternative is
# selected with RL_LIBRARY).  Prints the bench value and config.others of the tree's build ("old") and of each alternative, alternating.
set -u
N=${N:-2}
mkdir -p gpurun_out/ab
for ALT in "$@"; do RL_LIBRARY=$ALT timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_independent.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -1; done
for i in $(seq $N); do
  for which in old "$@"; do
    if [ $which != old ]; then export RL_LIBRARY=$which; else unset RL_LIBRARY; fi
    tag=$(basename $which .so)
    timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/ab/$tag$i.json 2> gpurun_out/ab/$tag$i.err
    python - $tag gpurun_out/ab/$tag$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "demo", round(d["value"]), " ".join("%s %d" % (o["config"].split("-")[0] + ("-global" if "global" in o["workload"] else ""), round(o["value"])) for o in d["config"].get("others", [])[:4]))
PY
  done
done
