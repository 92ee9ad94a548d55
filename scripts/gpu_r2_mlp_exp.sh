export TS_GAE_DIRECT_MAX=1024
echo "== tickets, one cache line per shard"; PYTHONPATH=. python scripts/gpu_gae_sweep.py 22 24 26 2>&1 | grep "envs=  8192"
python -m pytest tests/test_gpu_returns.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E " | head -5
