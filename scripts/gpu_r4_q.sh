#!/bin/bash
# block-parallel loss kernels: tests + benches of the engines that use them
O=$GRAFT_REPO_ROOT/gpurun_out/r4q; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_npg.py tests/test_gpu_td3.py tests/test_gpu_redq.py tests/test_gpu_ppo_wide.py tests/test_gpu_reinforce.py tests/test_gpu_hooks.py -m gpu -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for w in npg trpo redq td3 ddpg; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/$w.json 2>> $O/err.txt
done
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4q"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],4), "host", d.get("host_enqueue_ms_per_step"))
    except Exception as e: print(f,"ERR",e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
