#!/bin/bash
# Round-2 evidence run (under gpurun): GPU suite, smoke, bench lines + rocprofv3 kernel stats for every workload -> gpurun_out/r2final/
O=$GRAFT_REPO_ROOT/gpurun_out/r2final2; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time python bench.py ) > $O/bench_n1.json 2> $O/bench_n1.time
python bench.py --workload dqn > $O/bench_dqn.json 2>> $O/err.txt
python bench.py --workload sac > $O/bench_sac.json 2>> $O/err.txt
python bench.py --workload ppo_atari > $O/bench_ppo_atari.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_ppo -o ppo -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprofv3.json 2>> $O/err.txt
rocprofv3 --kernel-trace --stats -d $O/prof_dqn -o dqn -- python $GRAFT_REPO_ROOT/bench_dqn.py --steps 20 --no-cpu-baseline > $O/dqn_under_rocprofv3.json 2>> $O/err.txt
rocprofv3 --kernel-trace --stats -d $O/prof_sac -o sac -- python $GRAFT_REPO_ROOT/bench_sac.py --steps 20 --no-cpu-baseline > $O/sac_under_rocprofv3.json 2>> $O/err.txt
rocprofv3 --kernel-trace --stats -d $O/prof_atari -o atari -- python $GRAFT_REPO_ROOT/bench_ppo_cnn.py --repeat 1 --no-cpu-baseline > $O/atari_under_rocprofv3.json 2>> $O/err.txt
cd $GRAFT_REPO_ROOT
for w in td3 ddpg redq dsac qrdqn c51 rainbow npg trpo ppo_discrete drqn reinforce; do
  timeout 170 python bench.py --workload $w > $O/bench_$w.json 2>> $O/err.txt
done
PYTHONPATH=. timeout 120 python scripts/gpu_gae_sweep.py > $O/gae_sweep.txt 2>> $O/err.txt
for w in ppo dqn sac atari; do
  db=$O/prof_$w/${w}_results.db
  python scripts/rocprof_top.py $db $O/${w}_rocprofv3_kernel_stats.csv > $O/${w}_top.txt 2>&1
  rm -rf $O/prof_$w
done
ls -la $O; tail -3 $O/bench_n1.time
