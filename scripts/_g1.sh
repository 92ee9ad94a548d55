export PYTHONPATH=.
timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_hooks.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -3
