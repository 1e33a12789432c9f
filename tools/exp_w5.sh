set -u
mkdir -p gpurun_out/w5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_independent.py -x -q -m gpu 2>&1 | tail -2
B="--steps 4 --warmup 1 --no-cpu-baseline --no-live-counters --no-others"
for i in 1 2; do
python bench.py $B > gpurun_out/w5/tree_lds$i.json 2>gpurun_out/w5/err.txt; python -c "import json;d=json.load(open('gpurun_out/w5/tree_lds$i.json'));print('tree lds', round(d['value']))"
python bench.py $B --fetch global > gpurun_out/w5/tree_glob$i.json 2>>gpurun_out/w5/err.txt; python -c "import json;d=json.load(open('gpurun_out/w5/tree_glob$i.json'));print('tree global', round(d['value']))"
RL_DEBUG_LAUNCH=1 RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_w5.so python bench.py $B --fetch global > gpurun_out/w5/w5_glob$i.json 2>gpurun_out/w5/w5err.txt; python -c "import json;d=json.load(open('gpurun_out/w5/w5_glob$i.json'));print('w5 global', round(d['value']))"
grep "trace launch" gpurun_out/w5/w5err.txt | sort | uniq -c | head -3
done
RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_w5.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact_demo or matrix" 2>&1 | tail -2
