#!/bin/bash
# replay stream for QRDQN / C51 / Rainbow + Rainbow's backward pass on the workspace's streams: tests, then A/B benches
O=$GRAFT_REPO_ROOT/gpurun_out/r4r; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_distq.py tests/test_gpu_rainbow.py tests/test_gpu_dqn.py tests/test_gpu_hooks.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for w in qrdqn c51 rainbow; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/${w}_replay.json 2>> $O/err.txt
  TS_DISTQ_NO_REPLAY_STREAM=1 timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/${w}_seq.json 2>> $O/err.txt
done
TS_NO_SIDE_STREAM=1 TS_DISTQ_NO_REPLAY_STREAM=1 timeout 200 python bench.py --workload rainbow --no-cpu-baseline > $O/rainbow_seq_onestream.json 2>> $O/err.txt
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4r"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],4), "host", d.get("host_enqueue_ms_per_step"), "loss", d.get("final_loss"))
    except Exception as e: print(f,"ERR",e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
