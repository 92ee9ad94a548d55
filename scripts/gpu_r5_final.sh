#!/bin/bash
# Round-5 evidence run (under gpurun): GPU suite, smoke, the driver's bench line, rocprofv3 kernel stats of the same command, SQ / TCC
# counters of the step kernels (separate --pmc passes, kernel-trace only), bench lines of the other workloads -> gpurun_out/r5final/
O=$GRAFT_REPO_ROOT/gpurun_out/r5final; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time python bench.py ) > $O/bench_n1.json 2> $O/bench_n1.time; tail -3 $O/bench_n1.time
for w in sac dqn ppo_atari td3 ddpg redq dsac qrdqn c51 rainbow npg trpo ppo_discrete drqn reinforce; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>> $O/err.txt
done
TS_BENCH_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_dryrun_2ranks_one_gpu.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_ppo -o ppo -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprofv3.json 2>> $O/err.txt
cd $GRAFT_REPO_ROOT
python scripts/rocprof_top.py $O/prof_ppo/ppo_results.db $O/rocprofv3_kernel_stats.csv > $O/ppo_top.txt 2>&1
rm -rf $O/prof_ppo
# NPG / SAC: kernel tables of one bench run each (the one-launch actor passes; the SAC update's small kernels)
for w in npg trpo sac; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o $w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>> $O/err.txt
  cd $GRAFT_REPO_ROOT
  python scripts/rocprof_top.py $O/prof_$w/${w}_results.db $O/${w}_kernel_stats.csv 2>&1 | head -32 > $O/${w}_top_kernels.txt
  rm -rf $O/prof_$w
done
PYTHONPATH=. python scripts/gpu_sample_ubench.py 2>&1 | grep -v amdgpu.ids > $O/sample_kernel_by_phases.txt
bash scripts/gpu_r2_pmc.sh > $O/pmc_step_log.txt 2>&1
cp gpurun_out/pmc/pmc_step_mode2.txt $O/pmc_ppo_step.txt 2>/dev/null
bash scripts/gpu_pmc_traffic.sh > $O/pmc_traffic_log.txt 2>&1
cp gpurun_out/traffic/traffic_summary.txt $O/pmc_hbm_traffic.txt 2>/dev/null
cp gpurun_out/traffic/pmc_hbm_traffic.json $O/pmc_hbm_traffic.json 2>/dev/null
ls $O; head -c 600 $O/bench_n1.json; echo; head -12 $O/ppo_top.txt
python - <<'PY'
import json,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5final/"
for f in sorted(glob.glob(O+"bench_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["value"],1), d.get("unit"))
    except Exception as e: print(os.path.basename(f), "unreadable", e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
