#!/bin/bash
# A/B: 128 x 128 weight-gradient tiles (TS_WGRAD_TILE128=0: 128 x 64 as before)
O=$GRAFT_REPO_ROOT/gpurun_out/r6g; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sac.py tests/test_gpu_dqn.py tests/test_gpu_td3.py tests/test_gpu_conv2.py tests/test_gpu_distq.py -x -q -m gpu > $O/pytest.txt 2>&1
grep -v amdgpu.ids $O/pytest.txt | tail -3
for rep in 1 2; do
  for x in 0 1; do
    for wl in sac dqn td3 redq dsac qrdqn drqn; do
      TS_WGRAD_TILE128=$x timeout 300 python bench.py --workload $wl --no-cpu-baseline > $O/tmp.json 2>> $O/err.txt
      python - <<PY
import json
d = json.loads(open("$O/tmp.json").read().strip().splitlines()[-1])
r = d.get("roofline") or {}
k = r.get("kernel_us_per_update") or {kk: round(v["us_per_update"], 1) for kk, v in (d.get("roofline_by_kind") or {}).items()}
print("$wl tile128=$x rep $rep", round(d["value"], 1), d.get("unit"), "frac", round(r.get("frac") or 0, 4), k)
PY
    done
  done
done
grep -v amdgpu.ids $O/err.txt | tail -5
