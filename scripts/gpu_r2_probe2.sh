#!/bin/bash
# round-2 probe: v2 step kernel parity + phase timing + bench, v1 beside it
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ppo.py tests/test_gpu_hooks.py -m gpu -q > gpurun_out/t3.log 2>&1; echo rc=$? >> gpurun_out/t3.log
python scripts/gpu_step_phases.py > gpurun_out/phases_v2.txt 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b_v2.log 2>&1
TS_PPO_STEP_V1=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b_v1.log 2>&1
tail -5 gpurun_out/t3.log
