#!/bin/bash
# full GPU suite + rocprofv3 kernel stats of the PPO bench
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -m gpu -q -x ) > $O/gputests.log 2>&1; echo rc=$? >> $O/gputests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_ppo -o ppo -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprofv3.json 2>> $O/err.txt
cd $GRAFT_REPO_ROOT
python scripts/rocprof_top.py $O/prof_ppo/ppo_results.db $O/ppo_rocprofv3_kernel_stats.csv > $O/ppo_top.txt 2>&1
rm -rf $O/prof_ppo
tail -6 $O/gputests.log; head -14 $O/ppo_top.txt
