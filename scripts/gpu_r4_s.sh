#!/bin/bash
# DRQN: one-launch stacked pair gather, returns in the target passes' last kernel, next batch on a replay stream
O=$GRAFT_REPO_ROOT/gpurun_out/r4s; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_drqn.py tests/test_gpu_hooks.py tests/test_gpu_dqn.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
TS_DRQN_NO_LEARN_STEP=1 timeout 200 python bench.py --workload drqn --no-cpu-baseline > $O/drqn_pycalls_replay.json 2>> $O/err.txt
for i in 1 2; do
  timeout 200 python bench.py --workload drqn --no-cpu-baseline > $O/drqn_onecall_$i.json 2>> $O/err.txt
  TS_DRQN_NO_LEARN_STEP=1 TS_DRQN_NO_REPLAY_STREAM=1 timeout 200 python bench.py --workload drqn --no-cpu-baseline > $O/drqn_seq_$i.json 2>> $O/err.txt
done
TS_DRQN_NO_LEARN_STEP=1 TS_DRQN_NO_REPLAY_STREAM=1 TS_DRQN_NO_PAIR=1 timeout 200 python bench.py --workload drqn --no-cpu-baseline > $O/drqn_seq_nopair.json 2>> $O/err.txt
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4s"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],4), "host", d.get("host_enqueue_ms_per_step"), "loss", d.get("final_loss"))
    except Exception as e: print(f,"ERR",e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
