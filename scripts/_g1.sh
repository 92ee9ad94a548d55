mkdir -p gpurun_out/g1
export PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_sac.py tests/test_gpu_td3.py tests/test_gpu_redq.py -x -q -m gpu > gpurun_out/g1/pytest.txt 2>&1
timeout 200 python scripts/gpu_mlp_marks.py > gpurun_out/g1/marks.txt 2>&1
timeout 200 python scripts/gpu_mlp_rows.py > gpurun_out/g1/rows.txt 2>&1
timeout 300 python bench.py --workload sac > gpurun_out/g1/bench_sac.txt 2>&1
timeout 300 python bench.py --workload td3 > gpurun_out/g1/bench_td3.txt 2>&1
for f in gpurun_out/g1/*.txt; do echo "== $f"; tail -n 12 $f | cut -c1-420; done
