PYTHONPATH=. python scripts/gpu_gae_sweep.py 20 24 2>&1 | grep "envs=  8192\|envs=   512"
python -m pytest tests/test_gpu_returns.py tests/test_gpu_ppo.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E " | head -5
