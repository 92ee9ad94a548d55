# HipDQN at hook level: the new default-mode test, the DQN hook tests, and the hook-level rates (default vs reference-exact mode)
mkdir -p gpurun_out/r6hb
python -m pytest tests/test_gpu_hooks.py tests/test_gpu_policy.py -q -x -k "dqn or q_policy" 2>&1 | tail -15
python scripts/gpu_hook_offpolicy.py dqn 2>&1 | grep -v amdgpu.ids > gpurun_out/r6hb/dqn_hook_profile.txt
