#!/bin/bash
# A/B of the un-fused App seam between the tree's library and an alternative build (tools/build_alt.sh <name>).
# Usage (through gpurun): bash tools/exp_unfused_ab.sh <name>
set -u
for lib in "" "$@"; do
  echo "lib=${lib:-tree}"
  if [ -n "$lib" ]; then export RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_$lib.so; else unset RL_LIBRARY; fi
  python - <<'PY'
import robigo_luculenta_amd as R
for depth, threads, blocking in ((32, 2, False), (64, 2, False), (64, 4, False), (16, 16, True)):
    for rep in range(2):
        rgb, st = R.app_run(1280, 720, 4096, concurrency=depth, threads=threads, photons_per_batch=524288, fused=False, blocking_trace=blocking, verbose=False)
        print(" depth", depth, "threads", threads, "blocking" if blocking else "", round(st["segments"] / st["seconds"] / 1e6), "Mrays/s", flush=True)
PY
done
