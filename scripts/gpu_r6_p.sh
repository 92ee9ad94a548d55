#!/bin/bash
# round 6: ts_dqn_learn_step replayed from captured HIP graphs -- parity + C3 bench A/B (graph / streams / separate calls)
mkdir -p gpurun_out/r6p
TS_DQN_GRAPH_VERBOSE=1 timeout 900 python -m pytest tests/test_gpu_dqn.py -q -m gpu -k "learn_step or uniform_draws" > gpurun_out/r6p/pytest.txt 2>&1
tail -15 gpurun_out/r6p/pytest.txt
for i in 1 2; do
  TS_DQN_GRAPH_VERBOSE=1 timeout 300 python bench.py --workload dqn --steps 300 --warmup 20 > gpurun_out/r6p/dqn_graph_$i.json 2> gpurun_out/r6p/dqn_graph_$i.err
  TS_DQN_GRAPH=0 timeout 300 python bench.py --workload dqn --steps 300 --warmup 20 > gpurun_out/r6p/dqn_streams_$i.json 2> gpurun_out/r6p/dqn_streams_$i.err
done
TS_DQN_NO_LEARN_STEP=1 timeout 300 python bench.py --workload dqn --steps 300 --warmup 20 > gpurun_out/r6p/dqn_separate_1.json 2> gpurun_out/r6p/dqn_separate_1.err
grep -h "capture" gpurun_out/r6p/*.err | head
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6p/dqn_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value'],1), 'host', round(d['host_enqueue_ms_per_step'],3), d.get('update_path'), 'loss', d['final_loss'])
    except Exception as e:
        print(f, 'ERR', e)
PY
