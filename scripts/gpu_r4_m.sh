#!/bin/bash
# DQN: hardware queue count A/B (five streams on the default four queues share one)
O=$GRAFT_REPO_ROOT/gpurun_out/r4m; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  timeout 200 python bench.py --workload dqn --no-cpu-baseline > $O/dqn_q4_$rep.json 2>> $O/err.txt
  GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py --workload dqn --no-cpu-baseline > $O/dqn_q8_$rep.json 2>> $O/err.txt
done
GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py --workload sac --no-cpu-baseline > $O/sac_q8.json 2>> $O/err.txt
timeout 200 python bench.py --workload sac --no-cpu-baseline > $O/sac_q4.json 2>> $O/err.txt
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/ppo_q8.json 2>> $O/err.txt
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4m"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"],1), d["unit"], round(d["ms_per_step"],4), (d.get("roofline") or {}).get("frac"), d.get("final_loss"))
    except Exception as e: print(f,"ERR",e)
PY
grep -v amdgpu.ids $O/err.txt | tail -5
cd /tmp && export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace -d $O/prof -o dqn -- python $GRAFT_REPO_ROOT/bench_dqn.py --steps 12 --warmup 5 --no-cpu-baseline > $O/dqn_tl.json 2> $O/dqn_tl.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name '*.db' | head -1)
python scripts/rocprof_timeline.py $DB "adam_kernel(" 14 > $O/dqn_timeline_q8.txt 2>&1
rm -rf $O/prof
tail -30 $O/dqn_timeline_q8.txt
