# round 6 experiment: does a fifth wave per SIMD pay?  Global-fetch variant (no scene in LDS), workgroups of 256 threads: 5 per CU
# (96 VGPRs) against 4 per CU (128 VGPRs), emitter queues in global memory in both; and the cost of those queues on the standard kernel.
set -u
mkdir -p gpurun_out/w5
B="--steps 4 --warmup 1 --no-cpu-baseline --no-live-counters --no-others"
run() { # name library fetch
  if [ "$2" != tree ]; then export RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_$2.so; else unset RL_LIBRARY; fi
  RL_DEBUG_LAUNCH=1 python bench.py $B --fetch $3 > gpurun_out/w5/$1.json 2> gpurun_out/w5/$1.err
  python -c "import json;d=json.load(open('gpurun_out/w5/$1.json'));print('$1', round(d['value']), 'Mrays/s')"
  grep "trace launch" gpurun_out/w5/$1.err | sort | uniq -c | head -2
}
for i in 1 2; do
  run tree_lds$i tree lds; run e4_lds$i e4 lds
  run tree_glob$i tree global; run g4_glob$i g4 global; run g5_glob$i g5 global
done
for a in e4 g5; do
RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_$a.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact_demo or matrix or fused" 2>&1 | tail -2
done
