mkdir -p gpurun_out/w5
for a in g4 g5; do
RL_LIBRARY=$PWD/robigo_luculenta_amd/librl_alt_$a.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-others --fetch global > gpurun_out/w5/${a}_live.json 2> gpurun_out/w5/${a}_live.err
python - <<PY
import json
d=json.load(open("gpurun_out/w5/${a}_live.json"))
l=d["roofline"]["executed_live"]
print("$a", round(d["value"]), {k:(round(v,3) if isinstance(v,float) else v) for k,v in l.items() if k in ("resident_waves_per_simd","cycles_per_valu_inst_per_simd","active_lanes","useful_lane_slots_vs_2cyc","insts_per_64ray_segment","valu_insts_per_64ray_segment","wave_time","skipped")})
PY
done
