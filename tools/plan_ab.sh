#!/bin/bash
# Forces each of the planner's candidate plans (cluster size x clusters per group) in turn and times the bench config: does the cost
# model of rl_scene.cpp (fit in round 3) still rank the plans the way the kernel as it is now runs them?  Through gpurun.
set -u
mkdir -p gpurun_out/plan
B="--steps 3 --warmup 1 --no-cpu-baseline --no-live-counters --no-others"
for cfg in demo-1080p replicated-1080p; do
  RL_PLAN_VERBOSE=1 python bench.py $B --config $cfg 2>&1 >/dev/null | grep "rl: plan" | sort -u
  for plan in auto 10,3 10,4 14,3 14,4; do
    if [ $plan = auto ]; then unset RL_PLAN; else export RL_PLAN=$plan; fi
    python bench.py $B --config $cfg > gpurun_out/plan/$cfg-$plan.json 2>/dev/null
    python -c "import json;d=json.load(open('gpurun_out/plan/$cfg-$plan.json'));print('$cfg plan $plan', round(d['value']), 'Mrays/s')"
  done
done
