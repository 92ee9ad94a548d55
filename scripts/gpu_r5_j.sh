#!/bin/bash
# round 5, call J: the one-launch actor passes of NPG / TRPO (ts_npg_q.h) -- parity + bench A/B (per-layer passes; one / two streams) + kernel table
O=$GRAFT_REPO_ROOT/gpurun_out/r5j; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_npg.py tests/test_gpu_hooks.py tests/test_gpu_index_segtree.py -x -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for w in npg trpo; do
  TS_NPG_FVP=0 timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/b_${w}_per_layer.json 2>> $O/err.txt
  TS_NPG_ONE_STREAM=1 timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/b_${w}_one_stream.json 2>> $O/err.txt
  timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/b_${w}.json 2>> $O/err.txt
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5j/b_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"unreadable"); continue
    print(os.path.basename(f), round(d["value"],1), round(d["roofline"]["frac"],3))
PY
grep -v amdgpu.ids $O/err.txt | tail -5
