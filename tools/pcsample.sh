#!/bin/bash
# PC sampling (rocprofv3, stochastic, gfx950) of a short bench run: where the trace kernel's waves are, instruction by instruction,
# and why they are not issuing.  Raw samples stay on the box; what comes back is the per-instruction aggregate
# (tools/pcsample_summary.py).  Usage (through gpurun): bash tools/pcsample.sh <tag> [config] [fetch] [interval]
set -u
TAG=${1:-pcs}; CFG=${2:-demo-1080p}; FETCH=${3:-lds}; IV=${4:-1048576}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
RAW=/tmp/pcs_$TAG
rm -rf $RAW
ARGS="--steps 1 --warmup 0 --launches-per-step 2 --batches-per-launch 512 --no-cpu-baseline --no-others --no-live-counters --config $CFG --fetch $FETCH"
timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval $IV \
    --kernel-trace -f csv -d $RAW -o p -- python bench.py $ARGS > $OUT/pcs_$CFG-$FETCH.log 2>&1
echo "rocprofv3 rc $?"
find $RAW -type f | xargs ls -la
for f in $(find $RAW -name '*pc_sampling*.csv'); do echo "== $f"; head -5 $f; done
python tools/pcsample_summary.py $RAW $OUT/pcs_$CFG-$FETCH 2>&1 | tail -40
