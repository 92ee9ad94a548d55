#!/bin/bash
# SAC family after the fused chains + elementwise work: parity, then the bench lines
mkdir -p gpurun_out/r2mlp
python -m pytest tests/test_gpu_sac.py tests/test_gpu_td3.py tests/test_gpu_redq.py tests/test_gpu_dsac.py tests/test_gpu_hooks.py tests/test_gpu_dqn.py tests/test_gpu_npg.py -x -q -m gpu > gpurun_out/r2mlp/tests2.log 2>&1
grep -E "passed|failed" gpurun_out/r2mlp/tests2.log; grep -E "^E " gpurun_out/r2mlp/tests2.log | head -20
for w in sac td3 ddpg redq dsac; do
  python bench.py --workload $w --no-cpu-baseline > gpurun_out/r2mlp/bench2_$w.json 2> gpurun_out/r2mlp/bench2_$w.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r2mlp/bench2_$w.json"))
r = d["roofline"]
print("$w", round(d["value"], 1), d["unit"], "frac", round(r["frac"], 3), r.get("kernel_us_per_update"), "launches", r.get("launches_per_update"))
PY
done
