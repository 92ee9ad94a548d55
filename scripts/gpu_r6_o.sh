#!/bin/bash
# round 6: per-dispatch timeline of one C3 update through ts_dqn_learn_step + HIP API time of the call
O=$GRAFT_REPO_ROOT/gpurun_out/r6o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o dqn -- python $GRAFT_REPO_ROOT/bench_dqn.py --steps 12 --warmup 5 --no-cpu-baseline > $O/dqn_tl.json 2> $O/dqn_tl.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name '*.db' | head -1)
python scripts/rocprof_timeline.py $DB "adam_kernel(" 14 > $O/dqn_timeline.txt 2>&1
rm -rf $O/prof
cd /tmp
rocprofv3 --hip-trace --stats -d $O/hip -o dqn -- python $GRAFT_REPO_ROOT/bench_dqn.py --steps 200 --warmup 5 --no-cpu-baseline > $O/dqn_hip.json 2> $O/dqn_hip.err
cd $GRAFT_REPO_ROOT
find $O/hip -name '*stats*' | head
F=$(find $O/hip -name '*hip_api_stats.csv' | head -1)
[ -n "$F" ] && cp $F $O/hip_api_stats.csv
rm -rf $O/hip
cat $O/dqn_timeline.txt
head -30 $O/hip_api_stats.csv
