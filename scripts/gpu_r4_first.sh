#!/bin/bash
# Round-4 first GPU call: the new tests, the whole GPU suite, the N = 1 bench line, SAC / DQN lines (LDS overlay applied)
O=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
( time timeout 600 python bench.py ) > $O/bench_n1.json 2> $O/bench_n1.time; tail -4 $O/bench_n1.time
timeout 200 python bench.py --workload sac --no-cpu-baseline > $O/bench_sac.json 2>> $O/err.txt
timeout 200 python bench.py --workload dqn --no-cpu-baseline > $O/bench_dqn.json 2>> $O/err.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4a"
for f in ("bench_n1","bench_sac","bench_dqn"):
    try:
        d=json.loads([l for l in open(f"{O}/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d.get("ms_per_step"), (d.get("roofline") or {}).get("frac"), (d.get("kernel_us") or ""))
        if f=="bench_n1":
            print({k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.get("other_workloads",{}).items()})
            print("recompute", d.get("recompute_advantage"))
    except Exception as e: print(f,"ERR",e)
PY
