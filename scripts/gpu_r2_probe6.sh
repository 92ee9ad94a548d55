#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --workload sac --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/sac_base.log 2>&1
TS_CONV_SPLIT_TILES=300 TS_CONV_SPLIT_TARGET=512 python bench.py --workload sac --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/sac_split2.log 2>&1
TS_CONV_SPLIT_TILES=300 TS_CONV_SPLIT_TARGET=1024 python bench.py --workload sac --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/sac_split4.log 2>&1
python bench.py --workload td3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/td3_base.log 2>&1
TS_CONV_SPLIT_TILES=300 TS_CONV_SPLIT_TARGET=512 python bench.py --workload td3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/td3_split2.log 2>&1
