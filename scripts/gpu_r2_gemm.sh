#!/bin/bash
# round 2: the restructured row-GEMM kernel (ts_conv.hip): parity, micro timings, the workloads built on it
mkdir -p gpurun_out/r2gemm
python -m pytest tests/test_gpu_dqn.py tests/test_gpu_sac.py tests/test_gpu_npg.py tests/test_gpu_ppo_wide.py tests/test_gpu_ppo_cnn.py -x -q -m gpu > gpurun_out/r2gemm/tests.log 2>&1
tail -3 gpurun_out/r2gemm/tests.log
python scripts/gpu_conv_micro.py new > gpurun_out/r2gemm/micro.log 2>&1; cat gpurun_out/r2gemm/micro.log
for w in sac dqn td3; do
  python bench.py --workload $w --no-cpu-baseline > gpurun_out/r2gemm/bench_$w.json 2> gpurun_out/r2gemm/bench_$w.err
  cat gpurun_out/r2gemm/bench_$w.json
done
