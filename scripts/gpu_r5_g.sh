#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r5g; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_hooks.py tests/test_gpu_ppo_net.py -x -q > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt
