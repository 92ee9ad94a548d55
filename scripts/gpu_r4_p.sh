#!/bin/bash
# host share of the launch-bound updates
O=$GRAFT_REPO_ROOT/gpurun_out/r4p; mkdir -p $O
cd $GRAFT_REPO_ROOT
for w in drqn ppo_discrete reinforce dqn sac; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/$w.json 2>> $O/err.txt
done
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4p"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],4), "host", d.get("host_enqueue_ms_per_step"), d["steps"])
    except Exception as e: print(f,"ERR",e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
