#!/bin/bash
# correctness of the conv-based rows + DQN / SAC bench lines (no CPU baseline); run under gpurun
timeout 600 python -m pytest tests/test_gpu_dqn.py tests/test_gpu_sac.py -q -m gpu -x 2>&1 | grep -E "^E  |passed|failed" | head -12
timeout 300 python bench_dqn.py --slots 262144 --steps 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('DQN', round(d['value'],1), 'upd/s', {k:(round(v['us_per_update'],1), round(v['frac'],3)) for k,v in d['roofline_by_kind'].items()})"
timeout 300 python bench_sac.py --slots 262144 --steps 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('SAC', round(d['value'],1), 'upd/s', {k:round(v,1) for k,v in d['roofline']['kernel_us_per_update'].items()}, round(d['roofline']['frac'],3))"
