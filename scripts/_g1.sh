mkdir -p gpurun_out/g1
export PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_collective.py -x -q -m gpu > gpurun_out/g1/pytest.txt 2>&1
tail -n 30 gpurun_out/g1/pytest.txt
