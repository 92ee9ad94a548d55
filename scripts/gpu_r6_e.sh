#!/bin/bash
# A/B: XCD-aware block -> (tile, split) map of the weight-gradient kernels (TS_WGRAD_XCD=0: plain order)
O=$GRAFT_REPO_ROOT/gpurun_out/r6e; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sac.py tests/test_gpu_dqn.py tests/test_gpu_conv.py -x -q -m gpu > $O/pytest.txt 2>&1
grep -v amdgpu.ids $O/pytest.txt | tail -3
for rep in 1 2; do
  for x in 0 1; do
    for wl in sac dqn td3 redq dsac; do
      TS_WGRAD_XCD=$x timeout 300 python bench.py --workload $wl --no-cpu-baseline > $O/${wl}_xcd${x}_$rep.json 2>> $O/err.txt
      python - <<PY
import json
d = json.loads(open("$O/${wl}_xcd${x}_$rep.json").read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("$wl xcd=$x rep $rep", round(d["value"], 1), d.get("unit"), "frac", r.get("frac"), (r.get("kernel_us_per_update") or d.get("roofline_by_kind", {})))
PY
    done
  done
done
grep -v amdgpu.ids $O/err.txt | tail -5
