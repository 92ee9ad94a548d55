#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r6j; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_hooks.py tests/test_gpu_policy.py -x -q -m gpu > $O/pytest.txt 2>&1
grep -v amdgpu.ids $O/pytest.txt | tail -3
for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 > $O/tmp.json 2>> $O/err.txt
    python - <<PY
import json
d = json.loads(open("$O/tmp.json").read().strip().splitlines()[-1])
print("rep $rep", round(d["value"]), d["ms_per_step"], round(d["roofline"]["frac"], 4), {k: round(v, 2) for k, v in d["kernel_us"].items() if v}, d["roofline"]["traffic"], d["roofline_gae"]["traffic"])
PY
done
grep -v amdgpu.ids $O/err.txt | tail -5
