#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r6l; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_hooks.py -q -m gpu -x -k "reinforce" > $O/pytest.txt 2>&1
grep -v amdgpu.ids $O/pytest.txt | grep -E "^E  |passed|failed|Error|error" | head -30
