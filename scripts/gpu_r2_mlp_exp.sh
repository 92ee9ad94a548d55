for i in 1 2 3 4 5 6; do
  python bench.py --workload ddpg --no-cpu-baseline --steps 300 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ddpg steps=300', round(d['value'],1), round(d['ms_per_step'],3))"
done
