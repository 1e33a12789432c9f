#!/bin/bash
# rl_app_run rows of interest with the tree's library and an alternative build, alternating.  Usage (through gpurun): bash tools/exp_app_ab.sh <name>
set -u
for rep in 1 2; do
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_$lib.so; else unset RL_LIBRARY; fi
  python - "${lib:-tree}" <<'PY'
import sys
import robigo_luculenta_amd as R
out = []
for fused, depth, threads, blocking in ((False, 16, 16, False), (False, 16, 4, False), (False, 64, 2, False), (True, 16, 16, False), (False, 16, 16, True), (True, 16, 16, True)):
    rgb, st = R.app_run(1280, 720, 4096, concurrency=depth, threads=threads, photons_per_batch=524288, fused=fused, blocking_trace=blocking, verbose=False)
    out.append("%s%s %d/%d: %d" % ("F" if fused else "U", "b" if blocking else "", depth, threads, round(st["segments"] / st["seconds"] / 1e6)))
print("%-6s" % sys.argv[1], "  ".join(out), flush=True)
PY
done
done
