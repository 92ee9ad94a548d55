#!/bin/bash
# effective shader clock of the two step-kernel generations: GRBM_GUI_ACTIVE and SQ_BUSY_CYCLES per launch next to the
# kernel's duration from the same rocprofv3 run (MI355X_MICROARCH.md, "DVFS give-back")
O=$GRAFT_REPO_ROOT/gpurun_out/pmcclk; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 2 3; do
  TS_PPO_STEP=$m rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES -d $O/m$m -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/log$m.txt 2>&1
  echo "== TS_PPO_STEP=$m"
  python $GRAFT_REPO_ROOT/scripts/rocprof_pmc.py $O/m$m/t_results.db --match ppo_step
  python $GRAFT_REPO_ROOT/scripts/rocprof_top.py $O/m$m/t_results.db /dev/null 2>&1 | grep -i "ppo_step\|reduce\|adam" | head -4
  rm -rf $O/m$m
done
