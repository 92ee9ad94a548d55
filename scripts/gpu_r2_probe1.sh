#!/bin/bash
# round-2 probe: hook tests, phase timing of the step kernel at 2 and 1 workgroups per CU
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_hooks.py tests/test_gpu_ppo.py -m gpu -x -q > gpurun_out/t2.log 2>&1; echo rc=$? >> gpurun_out/t2.log
python scripts/gpu_step_phases.py > gpurun_out/phases_2wg.txt 2>&1
TS_PPO_WG_PER_CU=1 python scripts/gpu_step_phases.py > gpurun_out/phases_1wg.txt 2>&1
TS_PPO_WG_PER_CU=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b_1wg.log 2>&1
tail -3 gpurun_out/t2.log
