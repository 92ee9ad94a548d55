mkdir -p gpurun_out/g1
export PYTHONPATH=.
timeout 1200 python -m pytest tests/test_gpu_sac.py tests/test_gpu_td3.py tests/test_gpu_redq.py tests/test_gpu_dsac.py tests/test_gpu_dqn.py tests/test_gpu_hooks.py -x -q -m gpu > gpurun_out/g1/pytest.txt 2>&1
tail -n 5 gpurun_out/g1/pytest.txt
