#!/bin/bash
# round 6: per-dispatch timeline of one C5 SAC update at HEAD -> gpurun_out/r6t/sac_timeline.txt
O=$GRAFT_REPO_ROOT/gpurun_out/r6t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof_sac -o sac -- python $GRAFT_REPO_ROOT/bench_sac.py --steps 12 --no-cpu-baseline > $O/sac_tl.json 2> $O/sac_tl.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof_sac -name '*.db' | head -1)
python scripts/rocprof_timeline.py $DB "slab_adam_kernel<true>" 12 > $O/sac_timeline.txt 2>&1
rm -rf $O/prof_sac
tail -70 $O/sac_timeline.txt
