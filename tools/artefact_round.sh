#!/bin/bash
# Everything the judged artefacts of one round need, on the GPU box, into gpurun_out/<tag>/ :
#   profile_round.sh (kernel trace stats of bench + un-fused App, the four PMC passes merged into <tag>_pmc.json, the
#   default bench line quoting that profile), image-level parity, wave-level event counts + region timers (needs
#   `make -C robigo_luculenta_amd/csrc stats`), the long bit-exact parity sweeps, the App table, bench.py's N > 1 branch
#   with two ranks sharing the GPU, the VALU microbenchmark.  ~8 minutes.
# Usage (through gpurun): bash tools/artefact_round.sh r02   -> then tools/install_artefacts.sh r02 here.
set -u
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_console.txt 2>&1
(timeout 300 python tools/image_parity.py 1280 720 32 demo; timeout 300 python tools/image_parity.py 1920 1080 16 demo
 timeout 300 python tools/image_parity.py 1280 720 16 glass) > $OUT/image_parity.txt 2>&1
for s in demo glass replicated spill; do timeout 120 python tools/kernel_stats.py 64 $s; done > $OUT/kernel_stats.txt 2>&1
for s in spill2500 random5k random20k; do timeout 200 python tools/kernel_stats.py 8 $s; done >> $OUT/kernel_stats.txt 2>&1
(timeout 1200 python tools/big_parity.py 256 demo; timeout 500 python tools/big_parity.py 96 glass
 timeout 500 python tools/big_parity.py 128 replicated; timeout 900 python tools/random_scene_sweep.py 1000
 timeout 900 python tools/random_scene_sweep.py 100 32768 big) > $OUT/big_parity.txt 2>&1
(timeout 900 python tools/app_table.py; echo; echo "What the DEVICE does without the worker pool (tools/unfused_ceiling.py):"; timeout 200 python tools/unfused_ceiling.py
 python - <<'PY'
import robigo_luculenta_amd as R
rgb, st = R.app_run(1280, 720, 1024, concurrency=4, photons_per_batch=524288, fused=True, devices=[0, 0], verbose=False)
print("fused, two ranks on one GPU (devices = [0, 0]), depth 4", round(st["seconds"], 3), "s", round(st["segments"] / st["seconds"] / 1e6), "Mrays/s", st["paths"], "paths")
PY
) > $OUT/app.txt 2>&1
# scenes beyond LDS: the default, without the third table level, and what each staged
(export SPILL_QUICK=1; echo "== default"; timeout 400 python tools/spill_ab.py 2>&1 | grep -E "objects|Grays"; echo "== RL_SUPER_MIN=99999 (two-level table)"; RL_SUPER_MIN=99999 timeout 400 python tools/spill_ab.py 2>&1 | grep objects) > $OUT/spill_ab.txt 2>&1
(echo '$ python bench.py --gpus 2 --dist-backend gloo --steps 4 --warmup 1 --launches-per-step 2 --batches-per-launch 64   # two ranks share GPU 0'
 timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 4 --warmup 1 --launches-per-step 2 --batches-per-launch 64) > $OUT/bench_2ranks_gloo.txt 2>&1
(echo '$ python bench.py --gpus 1 --dist-backend rccl  (one rank; for comparison)'; timeout 300 python bench.py --steps 4 --warmup 1 --launches-per-step 2 --batches-per-launch 64 --no-others --no-cpu-baseline --no-live-counters) >> $OUT/bench_2ranks_gloo.txt 2>&1
timeout 300 python tools/open_launch_stress.py > $OUT/open_launch_stress.txt 2>&1
timeout 900 tools/valu_mb > $OUT/valu_microbench.txt 2>&1
bash tools/lds_conflicts.sh $TAG > /dev/null 2>&1
bash tools/pmc_stalls.sh $TAG demo-1080p > $OUT/instruction_mix.txt 2>&1; bash tools/pmc_stalls.sh $TAG glass-720p >> $OUT/instruction_mix.txt 2>&1
# every kind of instruction per 64-ray segment and the split of a wave's time (round 5: what the kernel's time is made of)
for CF in "demo-1080p lds" "glass-720p lds" "replicated-1080p lds" "replicated-1080p global"; do set -- $CF; bash tools/pmc_mix.sh $TAG $1 $2 | grep -v "^\[" ; done > $OUT/wave_time.txt 2>&1
cat $OUT/app.txt $OUT/spill_ab.txt $OUT/big_parity.txt $OUT/image_parity.txt
tail -12 gpurun_out/${TAG}_console.txt | cut -c1-400
