#!/bin/bash
# round 6: HipSAC hook path -- lazy write-back / index-only update / engine noise: tests + hook-level rates
mkdir -p gpurun_out/r6x
timeout 1200 python -m pytest tests/test_gpu_hooks.py tests/test_gpu_policy.py -q -m gpu -x > gpurun_out/r6x/hooks.txt 2>&1; tail -25 gpurun_out/r6x/hooks.txt
timeout 600 python - > gpurun_out/r6x/hook_level.txt 2>&1 <<'PY'
import json, bench_sac
print(json.dumps(bench_sac.hook_level(), indent=1))
PY
grep -v amdgpu gpurun_out/r6x/hook_level.txt | tail -20
