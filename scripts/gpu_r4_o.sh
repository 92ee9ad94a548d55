#!/bin/bash
# DRQN: LSTM layer kernels A/B + tests
O=$GRAFT_REPO_ROOT/gpurun_out/r4o; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_drqn.py tests/test_gpu_recurrent_nets.py tests/test_gpu_hooks.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
for rep in 1 2; do
  timeout 200 python bench.py --workload drqn --no-cpu-baseline > $O/drqn_layer_$rep.json 2>> $O/err.txt
  TS_DRQN_NO_PREFETCH=1 timeout 200 python bench.py --workload drqn --no-cpu-baseline > $O/drqn_noprefetch_$rep.json 2>> $O/err.txt
done
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4o"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), d["unit"], round(d["ms_per_step"],4), (d.get("roofline") or {}).get("frac"), d.get("final_loss"))
    except Exception as e: print(f,"ERR",e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
