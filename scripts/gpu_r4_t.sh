#!/bin/bash
# Reinforce / NPG / TRPO: one-network step kernel (ts_ppo_hparams.nets), device-resident return statistics, keyed permutations
O=$GRAFT_REPO_ROOT/gpurun_out/r4t; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_reinforce.py tests/test_gpu_npg.py tests/test_gpu_hooks.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for w in reinforce npg trpo; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/${w}_one_net.json 2>> $O/err.txt
  TS_REINFORCE_BOTH_NETS=1 TS_NPG_BOTH_NETS=1 timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/${w}_both_nets.json 2>> $O/err.txt
done
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4t"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],4), "host", d.get("host_enqueue_ms_per_step"), "loss", d.get("final_loss"), (d.get("roofline") or {}).get("avg_launch_us"))
    except Exception as e: print(f,"ERR",e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
