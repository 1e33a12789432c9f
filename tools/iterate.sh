#!/bin/bash
# One build -> measure cycle on the GPU box (through gpurun): the bit-exact parity tests, the wave-level event counts and
# region timers of the three BASELINE scenes, a short bench line.  Output: gpurun_out/it/.
# Usage: gpurun --timeout 900 -- 'bash tools/iterate.sh [tag] [quick]'
set -u
TAG=${1:-it}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_independent.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for s in demo glass replicated; do timeout 120 python tools/kernel_stats.py 64 $s; done > $OUT/kernel_stats.txt 2>&1
cat $OUT/kernel_stats.txt
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ${2:+--no-others} > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("bench", round(d["value"]), d["unit"], "launch ms", round(d["roofline"]["kernel_ms_per_launch"], 3))
for o in d["config"].get("others", []):
    print("  ", o["config"], round(o["value"]), o["workload"][-60:])
PY
