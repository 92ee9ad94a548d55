#!/bin/bash
# round 5, call K: SAC update with fewer launches (slab sums + Adam + Polyak + alpha fused, one packed copy of the observations)
O=$GRAFT_REPO_ROOT/gpurun_out/r5k; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sac.py tests/test_gpu_td3.py tests/test_gpu_redq.py tests/test_gpu_dsac.py tests/test_gpu_npg.py tests/test_gpu_index_segtree.py -x -q > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
for v in 0 1; do
  if [ $v = 1 ]; then export TS_SAC_SPLIT_ADAM=1; else unset TS_SAC_SPLIT_ADAM; fi
  timeout 200 python bench.py --workload sac --no-cpu-baseline > $O/b_sac_split$v.json 2>> $O/err.txt
done
unset TS_SAC_SPLIT_ADAM
for w in td3 ddpg redq dsac npg trpo; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/b_$w.json 2>> $O/err.txt
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5k/b_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"unreadable"); continue
    print(os.path.basename(f), round(d["value"],1), round(d["roofline"]["frac"],3))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o sac -- python $GRAFT_REPO_ROOT/bench.py --workload sac --no-cpu-baseline --steps 20 --warmup 5 > $O/prof_log.txt 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_top.py $O/prof/sac_results.db $O/sac_stats.csv 2>&1 | head -30 | tee $O/sac_top.txt
grep -v amdgpu.ids $O/err.txt | tail -5
PYTHONPATH=. python scripts/gpu_sample_ubench.py 2>&1 | grep -v amdgpu.ids | tee $O/sample_ubench.txt | grep "bs   4096"
