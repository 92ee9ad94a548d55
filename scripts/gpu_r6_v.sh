#!/bin/bash
# round 6: rocprofv3 kernel statistics of the driver's bench command, step kernels by launch geometry (agreement with bench.py's HIP events)
O=$GRAFT_REPO_ROOT/gpurun_out/r6v; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_ppo -o ppo -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprofv3.json 2> $O/err.txt
cd $GRAFT_REPO_ROOT
python scripts/rocprof_top.py $O/prof_ppo/ppo_results.db $O/rocprofv3_kernel_stats.csv ppo_step > $O/ppo_top.txt 2>&1
rm -rf $O/prof_ppo
tail -12 $O/ppo_top.txt
python -c "
import json; d=json.loads(open('$O/bench_under_rocprofv3.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'], d['kernel_us'])"
