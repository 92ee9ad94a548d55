#!/bin/bash
# Bench lines of the SURVEY 8f rows + configs[0] -> gpurun_out/next/*.json
mkdir -p gpurun_out/next
for w in td3 ddpg redq dsac qrdqn c51 rainbow npg trpo ppo_discrete; do
  timeout 170 python bench.py --workload $w > gpurun_out/next/bench_$w.json 2> gpurun_out/next/bench_$w.err || echo "$w failed rc=$?"
  tail -c 600 gpurun_out/next/bench_$w.err | tail -3
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/next/bench_$w.json").read().strip().splitlines()[-1])
    print("$w", round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 3),
          "launches", d["roofline"]["launches_per_update"], "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"], 2))
except Exception as e:
    print("$w: no result", e)
PY
done
