# C3 DQN: conv2's input gradient at B = 512 on the second-generation kernel in pixel-shuffle form (TS_DGRAD_PS_MIN_ROWS=100000)
# against the first-generation kernel (one launch, grid.z = parity), alternating on one box; DQN parity suites under it first.
TS_DGRAD_PS_MIN_ROWS=100000 python -m pytest tests/test_gpu_dqn.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -2
for i in 1 2 3; do
  for m in 100000 ""; do
    TS_DGRAD_PS_MIN_ROWS=$m python bench.py --workload dqn --no-cpu-baseline --steps 150 --warmup 30 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('ps_min_rows=${m:-off}', round(d['value'], 1), 'updates/s', round(d['roofline']['frac'], 4))"
  done
done
