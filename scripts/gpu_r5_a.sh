#!/bin/bash
# round 5, call A: feature-split step kernel -- gradients / losses against the oracle and the 128-sample kernel, then timings
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 500 python scripts/gpu_stepq_check.py check > $O/check.txt 2>&1; echo "check rc=$?" >> $O/check.txt
timeout 300 python scripts/gpu_stepq_check.py time > $O/time.txt 2>&1; echo "time rc=$?" >> $O/time.txt
timeout 600 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_hooks.py tests/test_gpu_reinforce.py tests/test_gpu_npg.py -x -q > $O/pytest_ppo.txt 2>&1
tail -40 $O/check.txt; cat $O/time.txt; tail -15 $O/pytest_ppo.txt
