#!/bin/bash
# round 6: side streams at lower / higher priority than the caller's stream (C3 and the distributional engines)
mkdir -p gpurun_out/r6u
for w in dqn qrdqn rainbow drqn; do
for p in default low high; do
  if [ $p = default ]; then unset TS_SIDE_PRIORITY; else export TS_SIDE_PRIORITY=$p; fi
  timeout 300 python bench.py --workload $w --steps 30 --warmup 2 --no-cpu-baseline > gpurun_out/r6u/${w}_$p.json 2>/dev/null
done
done
unset TS_SIDE_PRIORITY
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6u/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],1), d.get('final_loss'))
    except Exception as e: print(f,'ERR',e)
PY
