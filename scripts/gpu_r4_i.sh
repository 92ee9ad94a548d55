#!/bin/bash
# packed-VALU levels of the step kernel (TS_PK builds): A/B + parity tests on the default build
O=$GRAFT_REPO_ROOT/gpurun_out/r4i; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_hooks.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/pk3_$rep.json 2>> $O/err.txt
  for v in 0 1 2; do
    TS_LIB_PATH=$GRAFT_REPO_ROOT/tianshou_amd/lib/libtsengine_pk$v.so timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/pk${v}_$rep.json 2>> $O/err.txt
  done
done
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4i"
for f in sorted(glob.glob(O+"/*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), round(d["value"],1), round(d["ms_per_step"],3), {k:(round(v,2) if v else v) for k,v in d["kernel_us"].items() if k.startswith("ppo")}, d["final_losses"][:2])
PY
