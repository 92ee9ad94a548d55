# SAC C5: the one-call update (ts_sac_learn_rows) against the two entry points of rounds 2-5, alternating on one box; parity first.
python -m pytest tests/test_gpu_sac.py tests/test_gpu_td3.py tests/test_gpu_dsac.py tests/test_gpu_redq.py -m gpu -x -q > gpurun_out/r6_sac_learn_test.txt 2>&1; tail -3 gpurun_out/r6_sac_learn_test.txt
for i in 1 2 3; do
python bench.py --workload sac --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one_call', d['value'], d.get('host_enqueue_ms_per_step'))"
TS_SAC_TWO_CALLS=1 python bench.py --workload sac --steps 300 --warmup 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two_calls', d['value'], d.get('host_enqueue_ms_per_step'))"
done
