#!/bin/bash
# round 5, call J: the one-launch Fisher-vector product (ts_npg_q.h) -- NPG / TRPO parity + bench A/B + kernel table
O=$GRAFT_REPO_ROOT/gpurun_out/r5j; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_npg.py -x -q > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
for w in npg trpo; do
  for v in 0 1; do
    TS_NPG_FVP=$v timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/b_${w}_fvp$v.json 2>> $O/err.txt
  done
done
for g in 256 768; do
  TS_NPG_FVP_WGS=$g timeout 200 python bench.py --workload npg --no-cpu-baseline > $O/b_npg_wgs$g.json 2>> $O/err.txt
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5j/b_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"unreadable"); continue
    print(os.path.basename(f), round(d["value"],1))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o npg -- python $GRAFT_REPO_ROOT/bench.py --workload npg --no-cpu-baseline --steps 10 --warmup 2 > $O/prof_log.txt 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_top.py $O/prof/npg_results.db $O/npg_stats.csv 2>&1 | head -30 | tee $O/npg_top.txt
grep -v amdgpu.ids $O/err.txt | tail -5
