#!/bin/bash
# round 5, call C: the driver's bench line under the three step-kernel choices
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in 0 2 2p384; do
  unset TS_PPO_STEPQ_PAIRS
  if [ $v = 2p384 ]; then export TS_PPO_STEPQ=2 TS_PPO_STEPQ_PAIRS=384; else export TS_PPO_STEPQ=$v; fi
  timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$v.json 2>> $O/err.txt
done
unset TS_PPO_STEPQ TS_PPO_STEPQ_PAIRS
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5c/bench_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(os.path.basename(f), "value", round(d["value"],1), "ms", round(d["ms_per_step"],3), "roof", d["roofline"]["kernel"], round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_us"],2), {k: round(v,2) for k,v in d["kernel_us"].items() if v and k.startswith("ppo")})
    for e in d.get("strong_scaling_projection",{}).get("by_world_size",[]): print("    ", {k:(round(v,2) if isinstance(v,float) else v) for k,v in e.items()})
PY
grep -v amdgpu.ids $O/err.txt | tail
