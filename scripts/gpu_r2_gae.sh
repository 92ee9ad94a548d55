#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_returns.py tests/test_gpu_index_segtree.py -m gpu -q -x > gpurun_out/t6.log 2>&1; tail -2 gpurun_out/t6.log
PYTHONPATH=. python scripts/gpu_gae_sweep.py 20 22 24 26 > gpurun_out/gae_sweep_r2.txt 2>&1; cat gpurun_out/gae_sweep_r2.txt
