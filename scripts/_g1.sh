mkdir -p gpurun_out/g1
export PYTHONPATH=.
timeout 200 python scripts/gpu_mlp_marks.py > gpurun_out/g1/marks.txt 2>&1
tail -n 9 gpurun_out/g1/marks.txt | cut -c1-230
timeout 600 python -m pytest tests/test_gpu_sac.py tests/test_gpu_td3.py tests/test_gpu_redq.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --workload sac > gpurun_out/g1/bench_sac.txt 2>&1
tail -n 1 gpurun_out/g1/bench_sac.txt | cut -c1-200
