#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ppo.py tests/test_gpu_hooks.py -m gpu -q -x > gpurun_out/t7.log 2>&1; tail -3 gpurun_out/t7.log
TS_PPO_FUSED_ADAM=0 python -m pytest tests/test_gpu_ppo.py -m gpu -q -x > gpurun_out/t7b.log 2>&1; tail -2 gpurun_out/t7b.log
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b_fused_$i.log 2>&1
TS_PPO_FUSED_ADAM=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b_unfused_$i.log 2>&1
done
