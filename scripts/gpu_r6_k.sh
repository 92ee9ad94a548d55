#!/bin/bash
# A/B: XCD-aware tile order of the second-generation weight-gradient kernel (TS_WGRAD2_XCD=0: natural grid order)
O=$GRAFT_REPO_ROOT/gpurun_out/r6k; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv2.py tests/test_gpu_ppo_cnn.py -x -q -m gpu > $O/pytest.txt 2>&1
grep -v amdgpu.ids $O/pytest.txt | grep -E "passed|failed" | tail -2
for x in 0 1; do
  echo "== TS_WGRAD2_XCD=$x"
  TS_WGRAD2_XCD=$x PYTHONPATH=. timeout 600 python scripts/gpu_conv2_check.py bench 65536 2>> $O/err.txt | tee $O/layers_xcd$x.txt | tail -12
  for rep in 1 2; do
  TS_WGRAD2_XCD=$x timeout 600 python bench.py --workload ppo_atari --no-cpu-baseline > $O/tmp.json 2>> $O/err.txt
  python - <<PY
import json
d = json.loads(open("$O/tmp.json").read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("ppo_atari xcd=$x", round(d["value"], 2), d.get("unit"), "frac", round(r.get("frac") or 0, 4), r.get("kernel_ms_per_step") or r.get("kernel_us_per_update") or "")
PY
  done
done
grep -v amdgpu.ids $O/err.txt | tail -5
