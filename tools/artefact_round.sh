#!/bin/bash
# Everything the judged artefacts of one round need, on the GPU box, into gpurun_out/<tag>/ :
#   profile_round.sh (bench line, rocprofv3 kernel stats, PMC passes), the other BASELINE configs and the
#   global-fetch mode, image-level parity, wave-level event counts (needs `make -C robigo_luculenta_amd/csrc stats`),
#   the long bit-exact parity sweep, and the App table.  ~10 minutes.
# Usage (through gpurun): bash tools/artefact_round.sh r02a   -> then tools/install_artefacts.sh r02a here.
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_console.txt 2>&1
line='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["config"], round(d["value"],1), "Mrays/s", round(d["ms_per_step"],2), "ms/step", round(d["roofline"]["frac"],3))'
for c in demo-720p glass-720p replicated-1080p; do
    timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --config $c 2>/dev/null | python -c "$line"
done > $OUT/configs.txt
timeout 100 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --fetch global 2>/dev/null \
    | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("demo-1080p global-fetch", round(d["value"],1))' >> $OUT/configs.txt
(timeout 300 python tools/image_parity.py 1280 720 32 demo; timeout 300 python tools/image_parity.py 1920 1080 16 demo
 timeout 300 python tools/image_parity.py 1280 720 16 glass) > $OUT/image_parity.txt 2>&1
for s in demo glass replicated; do timeout 120 python tools/kernel_stats.py 64 $s; done > $OUT/kernel_stats.txt 2>&1
(timeout 1200 python tools/big_parity.py 512 demo; timeout 500 python tools/big_parity.py 96 glass
 timeout 500 python tools/big_parity.py 128 replicated) > $OUT/big_parity.txt 2>&1
python - > $OUT/app.txt <<'PY'
import robigo_luculenta_amd as R
for fused in (False, True):
    for c in (1, 8, 16):
        rgb, st = R.app_run(1280, 720, 4096, concurrency=c, photons_per_batch=524288, fused=fused, verbose=False)
        print("fused" if fused else "un-fused", "workers", c, round(st["seconds"], 3), "s", round(st["segments"] / st["seconds"] / 1e6),
              "Mrays/s", round(st["paths"] / 524288 / st["seconds"]), "batches/s")
rgb, st = R.app_run(1280, 720, 96, concurrency=2, photons_per_batch=64 * 524288, fused=True, verbose=False)
print("fused, 64-batch tasks, workers 2", round(st["seconds"], 3), "s", round(st["segments"] / st["seconds"] / 1e6), "Mrays/s")
PY
cat $OUT/configs.txt $OUT/big_parity.txt $OUT/app.txt
tail -5 gpurun_out/${TAG}_console.txt | cut -c1-330
