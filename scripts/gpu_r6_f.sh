#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r6f; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sac.py tests/test_gpu_dqn.py tests/test_gpu_td3.py tests/test_gpu_conv2.py -x -q -m gpu > $O/pytest.txt 2>&1
grep -v amdgpu.ids $O/pytest.txt | tail -3
for v in "" "TS_CONV_V2=1" "TS_CONV_V2=1 TS_WGRAD2_PLAN=1" "TS_CONV_V2=1 TS_WGRAD2_PLAN=2" "TS_CONV_V2=1 TS_WGRAD2_PLAN=3"; do
  for wl in sac; do
    env $v timeout 300 python bench.py --workload $wl --no-cpu-baseline > $O/tmp.json 2>> $O/err.txt
    python - <<PY
import json
d = json.loads(open("$O/tmp.json").read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("$wl [$v]", round(d["value"], 1), d.get("unit"), "frac", r.get("frac"), (r.get("kernel_us_per_update") or ""), r.get("launches_per_update"))
PY
  done
done
grep -v amdgpu.ids $O/err.txt | tail -5
