#!/bin/bash
# round 2: fused three-layer MLP forward (ts_mlp.hip) for the SAC family: parity, then A/B against the per-layer path
mkdir -p gpurun_out/r2mlp
python -m pytest tests/test_gpu_sac.py tests/test_gpu_td3.py tests/test_gpu_redq.py tests/test_gpu_hooks.py -x -q -m gpu > gpurun_out/r2mlp/tests.log 2>&1
grep -E "passed|failed" gpurun_out/r2mlp/tests.log; grep -E "^E " gpurun_out/r2mlp/tests.log | head -20
for v in fused perlayer; do
  if [ $v = perlayer ]; then export TS_MLP_PER_LAYER=1; else unset TS_MLP_PER_LAYER; fi
  for w in sac td3 ddpg redq; do
    python bench.py --workload $w --no-cpu-baseline > gpurun_out/r2mlp/bench_${w}_$v.json 2> gpurun_out/r2mlp/bench_${w}_$v.err
    python - <<PY
import json
d = json.load(open("gpurun_out/r2mlp/bench_${w}_$v.json"))
r = d["roofline"]
print("$v", "$w", round(d["value"], 1), d["unit"], "frac", round(r["frac"], 3), "fwd us", round(r.get("kernel_us_per_update", {}).get("conv_fwd", 0)), "launches", r.get("launches_per_update"))
PY
  done
done
