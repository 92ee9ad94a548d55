#!/bin/bash
# round 6: ts_dqn_learn_step -- parity with the separate calls, C3 bench A/B (one call vs the Python-driven cycle)
mkdir -p gpurun_out/r6n
timeout 900 python -m pytest tests/test_gpu_dqn.py -q -m gpu -k "learn_step or uniform_draws or replay_stream" > gpurun_out/r6n/pytest.txt 2>&1
tail -5 gpurun_out/r6n/pytest.txt
for i in 1 2; do
  timeout 300 python bench.py --workload dqn --steps 300 --warmup 20 > gpurun_out/r6n/dqn_learn_$i.json 2> gpurun_out/r6n/dqn_learn_$i.err
  TS_DQN_NO_LEARN_STEP=1 timeout 300 python bench.py --workload dqn --steps 300 --warmup 20 > gpurun_out/r6n/dqn_separate_$i.json 2> gpurun_out/r6n/dqn_separate_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6n/dqn_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value'],1), 'host', round(d['host_enqueue_ms_per_step'],3), d.get('update_path'))
    except Exception as e:
        print(f, 'ERR', e)
PY
