#!/bin/bash
# Round-3 evidence run (under gpurun): GPU suite, smoke, bench lines + rocprofv3 kernel stats for every workload -> gpurun_out/r3final/
O=$GRAFT_REPO_ROOT/gpurun_out/r3final; mkdir -p $O
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
# second-generation conv layers at the Atari-shape minibatch: layer sweep + SQ counters (separate --pmc passes)
timeout 300 python scripts/gpu_conv2_check.py bench 65536 16384 4096 512 > $O/conv2_layers.txt 2>> $O/err.txt
bash scripts/gpu_conv2_pmc.sh 65536 conv1u8,conv2,conv3,fc1 > $O/conv2_pmc_log.txt 2>&1
cp gpurun_out/pmc_conv2/pmc_conv2.txt $O/pmc_conv2.txt 2>/dev/null
# SQ counters + HBM traffic of the PPO step kernel at HEAD
bash scripts/gpu_r2_pmc.sh > $O/pmc_step_log.txt 2>&1
cp gpurun_out/pmc/pmc_step_mode2.txt $O/pmc_ppo_step2.txt 2>/dev/null
bash scripts/gpu_pmc_traffic.sh > $O/pmc_traffic_log.txt 2>&1
ls gpurun_out/ | head -30
