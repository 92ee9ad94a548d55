#!/bin/bash
# Round-4 call: permuted-record copies on the low-priority stream (TS_PPO_PREGATHER), A/B + PPO tests
O=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_hooks.py tests/test_gpu_collective.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for rep in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/pre_$rep.json 2>> $O/err.txt
  TS_PPO_PREGATHER=0 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/rows_$rep.json 2>> $O/err.txt
done
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4g"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), {k:(round(v,2) if v else v) for k,v in d["kernel_us"].items() if k.startswith("ppo")}, d["final_losses"][:2])
    except Exception as e: print(f,"ERR",e)
PY
tail -3 $O/err.txt
