#!/bin/bash
# A/B: the uniform `bounded` branch of the step kernel (default build) against a build without it (-DTS_NO_MU_BOUND)
O=$GRAFT_REPO_ROOT/gpurun_out/r6i; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in "" "$GRAFT_REPO_ROOT/tianshou_amd/lib/libtsengine_nobound.so"; do
    TS_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 > $O/tmp.json 2>> $O/err.txt
    python - <<PY
import json
d = json.loads(open("$O/tmp.json").read().strip().splitlines()[-1])
print("lib=[$lib]"[-28:], "rep $rep", round(d["value"]), d["ms_per_step"], round(d["roofline"]["frac"], 4), {k: round(v, 2) for k, v in d["kernel_us"].items() if v})
PY
  done
done
grep -v amdgpu.ids $O/err.txt | tail -5
