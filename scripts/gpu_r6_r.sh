#!/bin/bash
# round 6: bench lines of every workload at HEAD (no CPU baselines) -> gpurun_out/r6r/
O=$GRAFT_REPO_ROOT/gpurun_out/r6r; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
for w in sac dqn ppo_atari td3 ddpg redq dsac qrdqn c51 rainbow npg trpo ppo_discrete drqn reinforce; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>> $O/err.txt
done
timeout 600 python bench.py --no-cpu-baseline > $O/bench_ppo.json 2>> $O/err.txt
python - <<'PY'
import json,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6r/"
for f in sorted(glob.glob(O+"bench_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["value"],1), d.get("unit"), (d.get("roofline") or {}).get("frac"))
    except Exception as e: print(os.path.basename(f), "unreadable", e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
