#!/bin/bash
cd $GRAFT_REPO_ROOT
for st in 0 100 200 300 400; do
TS_PPO_STAGGER=$st python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/b_st_$st.log 2>&1
done
