# conv2's input gradient in pixel-shuffle form (Rows2Args.ps) against one GEMM per stride parity (TS_DGRAD_PS=0): parity, then the
# Atari-shape PPO bench alternating on one box, then the per-layer timings of scripts/gpu_conv2_check.py at 65,536 rows.
python -m pytest tests/test_gpu_conv2.py tests/test_gpu_ppo_cnn.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -3
for i in 1 2; do
  for ps in 1 0; do
    TS_DGRAD_PS=$ps python bench.py --workload ppo_atari --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); r = d['roofline']
print('ps=$ps', round(d['value'], 2), 'steps/s; one step', round(r['one_step_wall_ms'], 2), 'ms; GEMM ms', {k: round(v, 2) for k, v in r['kernel_ms_per_step'].items()}, 'frac', round(r['frac'], 4))"
  done
done
