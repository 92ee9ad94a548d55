# HipDQN at hook level: the default-mode tests, the DQN hook / engine tests, and the hook-level rates (default vs reference-exact mode)
mkdir -p gpurun_out/r6hb
python -m pytest tests/test_gpu_hooks.py tests/test_gpu_policy.py tests/test_gpu_dqn.py -q -x -k "dqn or q_policy" 2>&1 | tail -15
python - <<'P' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6hb/dqn_hook_level.json
import json, bench_dqn
print(json.dumps(bench_dqn.hook_level(), indent=1))
P
TS_DQN_TWO_CALLS=1 python - <<'P' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6hb/dqn_hook_level_two_calls.json
import json, bench_dqn
print(json.dumps(bench_dqn.hook_level(), indent=1))
P
python scripts/gpu_hook_offpolicy.py dqn 2>&1 | grep -v amdgpu.ids > gpurun_out/r6hb/dqn_hook_profile.txt
