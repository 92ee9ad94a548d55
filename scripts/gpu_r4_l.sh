#!/bin/bash
# DQN replay stream A/B + tests
O=$GRAFT_REPO_ROOT/gpurun_out/r4l; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dqn.py tests/test_gpu_index_segtree.py tests/test_gpu_hooks.py tests/test_gpu_distq.py -m gpu -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for rep in 1 2; do
  timeout 200 python bench.py --workload dqn --no-cpu-baseline > $O/dqn_replay_$rep.json 2>> $O/err.txt
  TS_DQN_NO_REPLAY_STREAM=1 timeout 200 python bench.py --workload dqn --no-cpu-baseline > $O/dqn_seq_$rep.json 2>> $O/err.txt
done
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4l"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), d["unit"], round(d["ms_per_step"],4), (d.get("roofline") or {}).get("frac"), d.get("final_loss"))
    except Exception as e: print(f,"ERR",e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
