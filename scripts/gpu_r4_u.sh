#!/bin/bash
# DRQN: W_ih / W_hh weight gradients on the two side streams, fc1's ahead of the join -- recurrent tests + A/B vs the committed library
# (build the previous commit's library into gpurun_keep/libtsengine_prev.so first: git stash; build; cp; git stash pop; build)
O=$GRAFT_REPO_ROOT/gpurun_out/r4u; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_drqn.py tests/test_gpu_recurrent_nets.py tests/test_gpu_hooks.py -m gpu -q > $O/pytest.txt 2>&1; grep -n "passed\|failed" $O/pytest.txt | tail -2
for i in 1 2; do
  timeout 60 python bench.py --workload drqn --no-cpu-baseline > $O/drqn_new_$i.json 2>> $O/err.txt
  TS_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_keep/libtsengine_prev.so timeout 60 python bench.py --workload drqn --no-cpu-baseline > $O/drqn_prev_$i.json 2>> $O/err.txt
done
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4u"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],4), "host", d.get("host_enqueue_ms_per_step"), "loss", d.get("final_loss"))
    except Exception as e: print(f,"ERR",e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
