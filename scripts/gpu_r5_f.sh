#!/bin/bash
# round 5, call F: generic Net trunks + obs_next fallback + LDS pitch change
O=$GRAFT_REPO_ROOT/gpurun_out/r5f; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ppo_net.py tests/test_gpu_sac.py tests/test_gpu_ppo_stepq.py tests/test_gpu_ppo_wide.py -x -q > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt
timeout 300 python scripts/gpu_stepq_check.py time 2>&1 | grep -v amdgpu | grep "^ *8192\|^ *16384\|^ *4096\|rows" > $O/time.txt; cat $O/time.txt
