timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_independent.py -x -q -m gpu 2>&1 | tail -1
RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_noieee.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_independent.py -x -q -m gpu 2>&1 | tail -1
N=2 BENCH_ARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-live-counters" bash tools/ab3.sh noieee nounroll
bash tools/plan_ab.sh
