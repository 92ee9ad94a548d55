#!/bin/bash
# SAC rows / seeded sampling A/B + tests; PPO tests on the TS_PK=2 default
O=$GRAFT_REPO_ROOT/gpurun_out/r4j; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sac.py tests/test_gpu_index_segtree.py tests/test_gpu_hooks.py tests/test_gpu_ppo.py tests/test_gpu_td3.py tests/test_gpu_redq.py tests/test_gpu_dsac.py -m gpu -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for rep in 1 2; do
  timeout 200 python bench.py --workload sac --no-cpu-baseline > $O/sac_rows_$rep.json 2>> $O/err.txt
  TS_SAC_NO_ROWS=1 timeout 200 python bench.py --workload sac --no-cpu-baseline > $O/sac_gather_$rep.json 2>> $O/err.txt
done
timeout 200 python bench.py --workload td3 --no-cpu-baseline > $O/td3.json 2>> $O/err.txt
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/ppo.json 2>> $O/err.txt
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4j"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), d["unit"], round(d["ms_per_step"],4), (d.get("roofline") or {}).get("frac"))
    except Exception as e: print(f,"ERR",e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
