#!/bin/bash
# feature-major LDS tile pitch of the feature-split kernels (ts_ppo_q.h PF = 36 / 40 / 44 / 52; -DTS_Q_PF builds): NPG bench on one
# stream + LDS bank-conflict counters of the Fisher-vector-product kernel
O=$GRAFT_REPO_ROOT/gpurun_out/r5pf; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TS_NPG_ONE_STREAM=1
for pf in 40 36 44 52; do
  if [ $pf = 40 ]; then unset TS_LIB_PATH; else export TS_LIB_PATH=$GRAFT_REPO_ROOT/tianshou_amd/lib/libtsengine_pf$pf.so; fi
  for i in 1 2; do timeout 200 python bench.py --workload npg --no-cpu-baseline > $O/b_pf${pf}_$i.json 2>> $O/err.txt; done
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/p$pf -o t -- python $GRAFT_REPO_ROOT/bench.py --workload npg --steps 2 --warmup 1 --no-cpu-baseline > $O/log$pf.txt 2>&1
  cd $GRAFT_REPO_ROOT
  python scripts/rocprof_pmc.py $O/p$pf/t_results.db --match npg_fvp > $O/pmc_pf$pf.txt 2>&1; rm -rf $O/p$pf
  echo "PF $pf"; grep -A1 npg_fvp $O/pmc_pf$pf.txt | tail -1
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5pf/b_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"unreadable"); continue
    print(os.path.basename(f), round(d["value"],1), round(d["roofline"]["kernel_us_per_update"]["ppo_step"],1))
PY
