#!/bin/bash
# Evidence at HEAD after the last changes of round 4: whole GPU suite, smoke, bench lines of the workloads that changed
O=$GRAFT_REPO_ROOT/gpurun_out/r4fin2; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
for w in drqn rainbow c51 qrdqn reinforce npg trpo ppo_discrete; do
  timeout 80 python bench.py --workload $w > $O/bench_$w.json 2>> $O/err.txt
done
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4fin2"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],4), "host", d.get("host_enqueue_ms_per_step"), "frac", (d.get("roofline") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(f,"ERR",e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
