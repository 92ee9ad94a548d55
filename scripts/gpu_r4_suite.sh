#!/bin/bash
# whole GPU suite + smoke at HEAD
O=$GRAFT_REPO_ROOT/gpurun_out/r4suite; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt; grep -n "bench-scale layer" $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
