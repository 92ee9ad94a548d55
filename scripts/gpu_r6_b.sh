#!/bin/bash
# Round 6: parity of the bounded actor / RMSprop / weight-decay paths against the reference-written fixtures.
O=$GRAFT_REPO_ROOT/gpurun_out/r6b; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_ppo_net.py tests/test_gpu_ppo_stepq.py -x -q -m gpu > $O/pytest.txt 2>&1
grep -v "amdgpu.ids" $O/pytest.txt | tail -40
