#!/bin/bash
# per-dispatch timeline of one DRQN update -> gpurun_out/r4tl/drqn_timeline.txt
O=$GRAFT_REPO_ROOT/gpurun_out/r4tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof_drqn -o drqn -- python $GRAFT_REPO_ROOT/bench_next.py drqn --steps 12 --warmup 5 --no-cpu-baseline > $O/drqn_tl.json 2> $O/drqn_tl.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof_drqn -name '*.db' | head -1)
python scripts/rocprof_timeline.py $DB "adam_kernel(" 14 > $O/drqn_timeline.txt 2>&1
rm -rf $O/prof_drqn
tail -120 $O/drqn_timeline.txt | cut -c1-128
tail -3 $O/drqn_tl.err
