#!/bin/bash
# kernel stats of the NPG update -> gpurun_out/r4npg/
O=$GRAFT_REPO_ROOT/gpurun_out/r4npg; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o npg -- python $GRAFT_REPO_ROOT/bench_next.py npg --steps 3 --warmup 1 --no-cpu-baseline > $O/npg.json 2> $O/npg.err
cd $GRAFT_REPO_ROOT
db=$(find $O/prof -name '*.db' | head -1)
python scripts/rocprof_top.py $db $O/npg_kernel_stats.csv > $O/npg_top.txt 2>&1
rm -rf $O/prof
head -34 $O/npg_top.txt | cut -c1-170
tail -2 $O/npg.json | cut -c1-300
