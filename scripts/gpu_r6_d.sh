#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r6d; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_policy.py -q -m gpu > $O/pytest.txt 2>&1
grep -v "amdgpu.ids" $O/pytest.txt | tail -60
