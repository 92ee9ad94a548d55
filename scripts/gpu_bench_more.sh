#!/bin/bash
mkdir -p gpurun_out/more
for w in rainbow npg trpo; do
  timeout 200 python bench.py --workload $w > gpurun_out/more/bench_$w.json 2> gpurun_out/more/bench_$w.err || echo "$w failed rc=$?"
  tail -c 400 gpurun_out/more/bench_$w.err | tail -2
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/more/bench_$w.json").read().strip().splitlines()[-1])
    print("$w", round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 3),
          "launches", d["roofline"]["launches_per_update"], "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"], 3), d.get("final_stats"))
except Exception as e:
    print("$w: no result", e)
PY
done
