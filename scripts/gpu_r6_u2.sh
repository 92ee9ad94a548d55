#!/bin/bash
mkdir -p gpurun_out/r6u2
for i in 1 2; do
for w in dqn drqn c51 sac td3 redq dsac npg ppo_discrete; do
for p in default high; do
  if [ $p = default ]; then unset TS_SIDE_PRIORITY; else export TS_SIDE_PRIORITY=$p; fi
  timeout 300 python bench.py --workload $w --steps 30 --warmup 2 --no-cpu-baseline > gpurun_out/r6u2/${w}_${p}_$i.json 2>/dev/null
done
done
done
unset TS_SIDE_PRIORITY
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6u2/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],1))
    except Exception as e: print(f,'ERR',e)
PY
