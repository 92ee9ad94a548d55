#!/bin/bash
# round 5, call H: split of the weight-gradient reduction (SAC family) + the 2-rank dry run with the exchange self-test
O=$GRAFT_REPO_ROOT/gpurun_out/r5h; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
TS_BENCH_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --selftest-rounds 200 > $O/dry2.json 2>> $O/err.txt
python - <<'PY'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5h/dry2.json").read().strip().splitlines()[-1])
print("dry run 2 ranks:", round(d["value"],1), d["config"]["exchange"], json.dumps(d.get("exchange_selftest"))[:600])
PY
for cfg in "768 4" "768 6" "768 8" "512 4" "512 8" "384 8" "1024 4" "640 6"; do
  set -- $cfg
  export TS_WGRAD_TARGET=$1 TS_WGRAD_MIN_CHUNKS=$2
  for w in sac td3; do
    timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/b_${w}_$1_$2.json 2>> $O/err.txt
  done
done
unset TS_WGRAD_TARGET TS_WGRAD_MIN_CHUNKS
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5h/b_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"unreadable"); continue
    ku=d["roofline"].get("kernel_us_per_update",{})
    print(os.path.basename(f), round(d["value"],1), {k:round(v,1) for k,v in ku.items()})
PY
grep -v amdgpu.ids $O/err.txt | tail -5
