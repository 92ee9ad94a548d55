#!/bin/bash
# round 5, call E: 32-row workgroups for the twin-critic launches of the SAC family
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sac.py tests/test_gpu_td3.py tests/test_gpu_redq.py tests/test_gpu_dsac.py tests/test_gpu_ppo_stepq.py -x -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for w in sac td3 ddpg redq dsac; do
  for rb in 1 0; do
    if [ $rb = 0 ]; then unset TS_MLP_RB; else export TS_MLP_RB=$rb; fi
    timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/bench_${w}_rb$rb.json 2>> $O/err.txt
  done
done
unset TS_MLP_RB
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5e/bench_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(os.path.basename(f), round(d["value"],1), d.get("unit"), "ms", round(d.get("ms_per_step",0),4))
PY
grep -v amdgpu.ids $O/err.txt | tail -5
