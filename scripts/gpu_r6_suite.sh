#!/bin/bash
# Round 6: the whole GPU suite + smoke at HEAD
O=$GRAFT_REPO_ROOT/gpurun_out/r6suite; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1
grep -v amdgpu.ids $O/pytest_gpu.txt | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
