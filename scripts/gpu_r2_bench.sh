#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_n1.json 2> $O/bench_n1.err
tail -4 $O/bench_n1.err
python -m pytest tests/test_gpu_npg.py tests/test_gpu_hooks.py -m gpu -q -x > $O/t5.log 2>&1; tail -2 $O/t5.log
bash scripts/gpu_pmc_traffic.sh > $O/traffic.log 2>&1; tail -30 $O/traffic.log
