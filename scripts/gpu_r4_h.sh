#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  TS_PPO_PREGATHER=inline timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/inline_$rep.json 2>> $O/err.txt
  TS_PPO_PREGATHER=0 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/rows_$rep.json 2>> $O/err.txt
done
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4h"
for f in sorted(glob.glob(O+"/*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), round(d["value"],1), round(d["ms_per_step"],3), {k:(round(v,2) if v else v) for k,v in d["kernel_us"].items() if k.startswith("ppo")})
PY
