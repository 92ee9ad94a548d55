#!/bin/bash
# round 5, call I: lin_wgrad_kernel (Linear-layer weight gradients at replay-batch sizes) -- parity suites + bench A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r5i; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sac.py tests/test_gpu_td3.py tests/test_gpu_redq.py tests/test_gpu_dsac.py tests/test_gpu_conv2.py tests/test_gpu_ppo_wide.py tests/test_gpu_ppo_net.py -x -q > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
for w in sac td3 ddpg redq dsac; do
  for v in 0 1; do
    export TS_LIN_WGRAD=$v
    timeout 200 python bench.py --workload $w --no-cpu-baseline > $O/b_${w}_lin$v.json 2>> $O/err.txt
  done
done
for r in 64 96 192 256; do
  TS_LIN_WGRAD=1 TS_LIN_WGRAD_ROWS=$r timeout 200 python bench.py --workload sac --no-cpu-baseline > $O/b_sac_rows$r.json 2>> $O/err.txt
done
unset TS_LIN_WGRAD
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5i/b_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"unreadable"); continue
    ku=d["roofline"].get("kernel_us_per_update",{})
    print(os.path.basename(f), round(d["value"],1), {k:round(v,1) for k,v in ku.items()})
PY
grep -v amdgpu.ids $O/err.txt | tail -5
