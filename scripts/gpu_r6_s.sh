#!/bin/bash
# round 6: wgrad2 plan 4 (256 x 128 tiles) against plan 1 at the Atari-shape minibatch; check first
mkdir -p gpurun_out/r6s
timeout 600 python scripts/gpu_conv2_check.py check > gpurun_out/r6s/check_default.txt 2>&1; tail -2 gpurun_out/r6s/check_default.txt
TS_WGRAD2_PLAN=4 timeout 600 python scripts/gpu_conv2_check.py check > gpurun_out/r6s/check_plan4.txt 2>&1; tail -2 gpurun_out/r6s/check_plan4.txt
timeout 600 python scripts/gpu_conv2_check.py bench 65536 2>&1 | grep -v amdgpu > gpurun_out/r6s/bench_default.txt
TS_WGRAD2_PLAN=4 timeout 600 python scripts/gpu_conv2_check.py bench 65536 2>&1 | grep -v amdgpu > gpurun_out/r6s/bench_plan4.txt
grep "fc1\|total" gpurun_out/r6s/bench_default.txt gpurun_out/r6s/bench_plan4.txt
for i in 1 2; do
timeout 300 python bench.py --workload ppo_atari --no-cpu-baseline > gpurun_out/r6s/atari_default_$i.json 2>/dev/null
TS_WGRAD2_PLAN=4 timeout 300 python bench.py --workload ppo_atari --no-cpu-baseline > gpurun_out/r6s/atari_plan4_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6s/atari_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],2), d['roofline']['frac'])
PY
