#!/bin/bash
# Probe: two four-wave teams as ONE 512-thread workgroup (TS_PPO_WG8=1) against two 256-thread workgroups per CU
O=$GRAFT_REPO_ROOT/gpurun_out/r6m; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
TS_PPO_WG8=1 timeout 600 python -m pytest tests/test_gpu_ppo.py -x -q -m gpu -k "golden or linearity or larger" > $O/pytest.txt 2>&1
grep -v amdgpu.ids $O/pytest.txt | grep -E "passed|failed" | tail -2
for rep in 1 2 3; do
  for m in 0 1; do
    TS_PPO_WG8=$m timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 > $O/tmp.json 2>> $O/err.txt
    python - <<PY
import json
d = json.loads(open("$O/tmp.json").read().strip().splitlines()[-1])
print("wg8=$m rep $rep", round(d["value"]), d["ms_per_step"], round(d["roofline"]["frac"], 4), {k: round(v, 2) for k, v in d["kernel_us"].items() if v})
PY
  done
done
grep -v amdgpu.ids $O/err.txt | tail -5
