#!/bin/bash
# Round-6 evidence run (under gpurun): GPU suite, smoke, TCC / SQ counter passes (-> the JSONs the bench lines read), the driver's
# bench line, every other workload's line, the 2-rank dry run, rocprofv3 kernel statistics of the driver's command and of the
# C3 / C5 benches -> gpurun_out/r6final/
O=$GRAFT_REPO_ROOT/gpurun_out/r6final; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; grep -v amdgpu.ids $O/pytest_gpu.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
bash scripts/gpu_r6_pmc.sh > $O/pmc_log.txt 2>&1
P=$GRAFT_REPO_ROOT/gpurun_out/r6pmc
for f in pmc_hbm_traffic.json pmc_sac.json pmc_dqn.json; do [ -s $P/$f ] && cp $P/$f profiles/r06_$f && cp $P/$f $O/$f; done
for f in ppo_traffic_summary.txt sac_traffic_summary.txt dqn_traffic_summary.txt pmc_sac_sq.txt pmc_sac_tcc.txt; do [ -s $P/$f ] && cp $P/$f $O/$f; done
( time python bench.py ) > $O/bench_n1.json 2> $O/bench_n1.time; tail -3 $O/bench_n1.time
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 > $O/bench_n1_dispatch.json 2>> $O/err.txt
for w in sac dqn; do
  timeout 400 python bench.py --workload $w > $O/bench_$w.json 2>> $O/err.txt
done
for w in ppo_atari td3 ddpg redq dsac qrdqn c51 rainbow npg trpo ppo_discrete drqn reinforce; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>> $O/err.txt
done
TS_BENCH_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_dryrun_2ranks_one_gpu.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_ppo -o ppo -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprofv3.json 2>> $O/err.txt
cd $GRAFT_REPO_ROOT
python scripts/rocprof_top.py $O/prof_ppo/ppo_results.db $O/rocprofv3_kernel_stats.csv > $O/ppo_top.txt 2>&1
rm -rf $O/prof_ppo
for w in sac dqn; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o $w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>> $O/err.txt
  cd $GRAFT_REPO_ROOT
  python scripts/rocprof_top.py $O/prof_$w/${w}_results.db $O/${w}_rocprofv3_kernel_stats.csv 2>&1 | head -40 > $O/${w}_top_kernels.txt
  rm -rf $O/prof_$w
done
ls $O; head -c 700 $O/bench_n1.json; echo; head -12 $O/ppo_top.txt
python - <<'PY'
import json,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6final/"
for f in sorted(glob.glob(O+"bench_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["value"],1), d.get("unit"), (d.get("roofline") or {}).get("frac"))
    except Exception as e: print(os.path.basename(f), "unreadable", e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
