#!/bin/bash
# round 6: slab_sum_few_kernel A/B on C3 (and C5, which sums MLP weight-gradient slabs through the same entry)
mkdir -p gpurun_out/r6q
for i in 1 2; do
  timeout 300 python bench.py --workload dqn --steps 300 --warmup 20 > gpurun_out/r6q/dqn_few_$i.json 2> gpurun_out/r6q/err.txt
  TS_SLAB_SUM_FEW=0 timeout 300 python bench.py --workload dqn --steps 300 --warmup 20 > gpurun_out/r6q/dqn_old_$i.json 2>> gpurun_out/r6q/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6q/dqn_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value'],1), 'host', round(d['host_enqueue_ms_per_step'],3), 'loss', d['final_loss'])
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 600 python -m pytest tests/test_gpu_dqn.py tests/test_gpu_distq.py tests/test_gpu_rainbow.py tests/test_gpu_ppo_cnn.py -q -m gpu 2>&1 | tail -3
