#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r5m; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
for c in "768 768" "512 768" "512 512" "512 256" "768 256" "256 256" "512 384"; do
  set -- $c
  TS_NPG_FVP_WGS=$1 TS_NPG_CRITIC_WGS=$2 timeout 200 python bench.py --workload npg --no-cpu-baseline > $O/b_npg_$1_$2.json 2>> $O/err.txt
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5m/b_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"unreadable"); continue
    print(os.path.basename(f), round(d["value"],1))
PY
