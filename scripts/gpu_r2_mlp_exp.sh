for v in "" "TS_MLP_PER_LAYER=1"; do
  env $v python bench.py --workload dsac --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v]', 'dsac', round(d['value'],1), round(d['ms_per_step'],3), d['roofline'].get('kernel_us_per_update'), d['config'])"
done
