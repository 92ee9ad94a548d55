#!/bin/bash
# round 6: HipSAC hook path after the train()-mode shortcut: hook tests + hook-level rates + host profile
mkdir -p gpurun_out/r6y
timeout 1200 python -m pytest tests/test_gpu_hooks.py tests/test_gpu_policy.py tests/test_gpu_sac.py -q -m gpu -x > gpurun_out/r6y/hooks.txt 2>&1; tail -5 gpurun_out/r6y/hooks.txt
timeout 600 python - > gpurun_out/r6y/hook_level.txt 2>&1 <<'PY'
import json, bench_sac
print(json.dumps(bench_sac.hook_level(), indent=1))
PY
grep -v amdgpu gpurun_out/r6y/hook_level.txt | tail -20
timeout 600 python scripts/gpu_hook_offpolicy.py > gpurun_out/r6y/profile.txt 2>&1; grep -v amdgpu gpurun_out/r6y/profile.txt | head -30
