#!/bin/bash
# Round-4 call 5: DQN prefetch A/B, Atari PPO with the vector gather, tests of the touched paths
O=$GRAFT_REPO_ROOT/gpurun_out/r4e; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dqn.py tests/test_gpu_hooks.py tests/test_gpu_sac.py tests/test_gpu_redq.py tests/test_gpu_ppo_wide.py tests/test_gpu_ppo_discrete.py tests/test_gpu_ppo_cnn.py tests/test_gpu_index_segtree.py -m gpu -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
for rep in 1 2; do
  timeout 200 python bench.py --workload dqn --no-cpu-baseline > $O/dqn_prefetch_$rep.json 2>> $O/err.txt
  TS_DQN_NO_PREFETCH=1 timeout 200 python bench.py --workload dqn --no-cpu-baseline > $O/dqn_plain_$rep.json 2>> $O/err.txt
done
timeout 300 python bench.py --workload ppo_atari --no-cpu-baseline > $O/ppo_atari.json 2>> $O/err.txt
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4e"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],2), d["unit"], round(d["ms_per_step"],3), (d.get("roofline") or {}).get("frac"), d.get("whole_update_mfma_frac"))
    except Exception as e: print(f,"ERR",e)
PY
tail -5 $O/err.txt
