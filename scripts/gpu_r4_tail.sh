#!/bin/bash
# Round-4: fused tail (ppo_tail_kernel) against the split tail, and the no-sign tanh variant; PPO tests on the fused tail
O=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_hooks.py tests/test_gpu_ppo_cnn.py tests/test_gpu_sac.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/fused_$rep.json 2>> $O/err.txt
  TS_PPO_TAIL=split timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/split_$rep.json 2>> $O/err.txt
  TS_LIB_PATH=$GRAFT_REPO_ROOT/tianshou_amd/lib/libtsengine_nosign.so timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/nosign_$rep.json 2>> $O/err.txt
done
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4b"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), {k:(round(v,2) if v else v) for k,v in d["kernel_us"].items() if k.startswith("ppo")}, d["final_losses"])
    except Exception as e: print(f,"ERR",e)
PY
tail -5 $O/err.txt
