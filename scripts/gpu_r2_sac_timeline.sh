#!/bin/bash
# per-dispatch timeline of one SAC update (two streams as in production) -> gpurun_out/r2mlp/sac_timeline.txt
O=$GRAFT_REPO_ROOT/gpurun_out/r2mlp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof_sac -o sac -- python $GRAFT_REPO_ROOT/bench_sac.py --steps 12 --no-cpu-baseline > $O/sac_tl.json 2> $O/sac_tl.err
cd $GRAFT_REPO_ROOT
python scripts/rocprof_timeline.py $O/prof_sac/sac_results.db polyak2_kernel 12 > $O/sac_timeline.txt 2>&1
rm -rf $O/prof_sac
tail -3 $O/sac_timeline.txt
