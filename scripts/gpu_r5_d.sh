#!/bin/bash
# round 5, call D: PPO-family GPU tests with the row-count dispatch in place + the bench line
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ppo_stepq.py tests/test_gpu_ppo.py tests/test_gpu_hooks.py tests/test_gpu_reinforce.py tests/test_gpu_npg.py tests/test_gpu_collective.py -x -q > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.json 2>> $O/err.txt
python - <<'PY'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5d/bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"],1), "roof", d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
for e in d["strong_scaling_projection"]["by_world_size"]: print("    ", {k:(round(v,2) if isinstance(v,float) else v) for k,v in e.items()})
PY
