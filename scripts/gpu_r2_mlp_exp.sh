python -m pytest tests/test_gpu_td3.py tests/test_gpu_sac.py tests/test_gpu_hooks.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E " | head
for i in 1 2; do
for w in td3 ddpg; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', round(d['value'],1), round(d['ms_per_step'],3))"
done
done
