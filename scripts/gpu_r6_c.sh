#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r6c; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_hooks.py -x -q -m gpu -k "replay or a2c or scheduler" > $O/pytest.txt 2>&1
grep -v "amdgpu.ids" $O/pytest.txt | tail -40
