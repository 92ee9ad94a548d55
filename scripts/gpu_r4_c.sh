#!/bin/bash
# Round-4 call 3: one-launch CartPole PPO update, bench-scale Atari parity, PPO suite on the no-sign tanh build
O=$GRAFT_REPO_ROOT/gpurun_out/r4c; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ppo_discrete.py tests/test_gpu_ppo_cnn.py tests/test_gpu_returns.py tests/test_gpu_ppo.py -m gpu -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt; grep -n "bench-scale layer" $O/pytest.txt
TS_LIB_PATH=$GRAFT_REPO_ROOT/tianshou_amd/lib/libtsengine_nosign.so timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_hooks.py -m gpu -q > $O/pytest_nosign.txt 2>&1; tail -5 $O/pytest_nosign.txt
timeout 300 python bench.py --workload ppo_discrete --no-cpu-baseline > $O/bench_ppo_discrete.json 2>> $O/err.txt
TS_MLP_PPO_PER_STEP=1 timeout 300 python bench.py --workload ppo_discrete --no-cpu-baseline > $O/bench_ppo_discrete_per_step.json 2>> $O/err.txt
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4c"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), d["unit"], round(d["ms_per_step"],3), (d.get("roofline") or {}).get("frac"))
    except Exception as e: print(f,"ERR",e)
PY
tail -5 $O/err.txt
