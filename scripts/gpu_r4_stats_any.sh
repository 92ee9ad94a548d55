#!/bin/bash
# kernel stats of bench_next workloads: gpu_r4_stats_any.sh redq td3 ...  -> gpurun_out/r4stats/<w>_top.txt
O=$GRAFT_REPO_ROOT/gpurun_out/r4stats; mkdir -p $O
for w in "$@"; do
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $O/prof_$w -o $w -- python $GRAFT_REPO_ROOT/bench_next.py $w --steps 10 --warmup 3 --no-cpu-baseline > $O/$w.json 2> $O/$w.err
  cd $GRAFT_REPO_ROOT
  db=$(find $O/prof_$w -name '*.db' | head -1)
  python scripts/rocprof_top.py $db $O/${w}_kernel_stats.csv > $O/${w}_top.txt 2>&1
  rm -rf $O/prof_$w
  echo "== $w"; head -${TOPN:-16} $O/${w}_top.txt | cut -c1-150
done
