#!/bin/bash
# Round 6, first GPU call: the two workgroups of a CU taking the two networks in opposite order (TS_PPO_ORDER=1|2|3,
# ppo_step2x_kernel) against the shipped order -- parity first, then the driver-equivalent line of each.
O=$GRAFT_REPO_ROOT/gpurun_out/r6a; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
for m in 1 2 3; do
  TS_PPO_ORDER=$m timeout 600 python -m pytest tests/test_gpu_ppo.py -x -q -m gpu > $O/pytest_order$m.txt 2>&1
  tail -2 $O/pytest_order$m.txt
done
for rep in 1 2; do
  for m in 0 1 2 3; do
    TS_PPO_ORDER=$m timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 > $O/bench_order${m}_$rep.json 2>> $O/err.txt
    python - <<PY
import json
d = json.loads(open("$O/bench_order${m}_$rep.json").read().strip().splitlines()[-1])
print("order $m rep $rep", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("kernel_us"))
PY
  done
done
grep -v amdgpu.ids $O/err.txt | tail -5
