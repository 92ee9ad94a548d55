#!/bin/bash
# mid-round check: full GPU suite, SAC-family benches (noise kernel + clock warm-up), repeated for run-to-run spread
O=$GRAFT_REPO_ROOT/gpurun_out/mid; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
for i in 1 2 3; do
  for w in sac td3; do
    python bench.py --workload $w --no-cpu-baseline > $O/bench_${w}_$i.json 2>> $O/err.txt
    python -c "import json;d=json.loads(open('$O/bench_${w}_$i.json').read().strip().splitlines()[-1]);print('$w run $i', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['frac'],3))"
  done
done
python bench.py --workload redq --no-cpu-baseline > $O/bench_redq.json 2>> $O/err.txt
python bench.py --workload ddpg --no-cpu-baseline > $O/bench_ddpg.json 2>> $O/err.txt
python -c "
import json
for w in ('redq','ddpg'):
    d=json.loads(open('$O/bench_%s.json'%w).read().strip().splitlines()[-1]);print(w, round(d['value'],1), round(d['ms_per_step'],4))"
