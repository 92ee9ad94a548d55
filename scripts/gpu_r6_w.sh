#!/bin/bash
# round 6: two-hidden-layer networks of any widths (zero-padding embedding) -- SAC / TD3 / DDPG / REDQ / DiscreteSAC / NPG / TRPO
mkdir -p gpurun_out/r6w
timeout 1500 python -m pytest tests/test_gpu_sac.py tests/test_gpu_td3.py tests/test_gpu_redq.py tests/test_gpu_dsac.py tests/test_gpu_npg.py tests/test_gpu_policy.py tests/test_gpu_reinforce.py tests/test_gpu_ppo_wide.py -q -m gpu > gpurun_out/r6w/engine.txt 2>&1; tail -15 gpurun_out/r6w/engine.txt
timeout 1500 python -m pytest tests/test_gpu_hooks.py -q -m gpu > gpurun_out/r6w/hooks.txt 2>&1; tail -15 gpurun_out/r6w/hooks.txt
