#!/bin/bash
# The bit-exact sweeps of tools/artefact_round.sh run longer (~20 minutes): usage (through gpurun) bash tools/long_parity.sh <tag>
set -u
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_big_parity_long.txt
echo "\$ tools/big_parity.py 768 demo; 256 glass; 256 replicated; tools/random_scene_sweep.py 2500; tools/random_scene_sweep.py 200 32768 big   (build $(python -c 'import robigo_luculenta_amd as R; print(R.build_id())'), one MI355X)" > $OUT
(timeout 1800 python tools/big_parity.py 768 demo; timeout 900 python tools/big_parity.py 256 glass
 timeout 900 python tools/big_parity.py 256 replicated; timeout 1200 python tools/random_scene_sweep.py 2500
 timeout 900 python tools/random_scene_sweep.py 200 32768 big) >> $OUT 2>&1
cat $OUT
