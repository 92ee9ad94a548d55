#!/bin/bash
# round 6: any depth for the SAC family (ts_mlp_set_trunk): engine- and hook-level suites of SAC / TD3 / DDPG / DiscreteSAC / REDQ
mkdir -p gpurun_out/r6z
timeout 1500 python -m pytest tests/test_gpu_sac.py tests/test_gpu_td3.py tests/test_gpu_dsac.py tests/test_gpu_redq.py tests/test_gpu_policy.py -q -m gpu > gpurun_out/r6z/engines.txt 2>&1; tail -25 gpurun_out/r6z/engines.txt
timeout 1500 python -m pytest tests/test_gpu_hooks.py -q -m gpu > gpurun_out/r6z/hooks.txt 2>&1; tail -25 gpurun_out/r6z/hooks.txt
