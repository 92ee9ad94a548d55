#!/bin/bash
# round 5, call B: parity check + timings + phase stamps of the feature-split step kernel
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 500 python scripts/gpu_stepq_check.py check > $O/check.txt 2>&1; echo "check rc=$?" >> $O/check.txt
timeout 300 python scripts/gpu_stepq_check.py time > $O/time.txt 2>&1
for cfg in ${PHASE_CFGS:-"8192 2 0" "65536 2 0" "65536 2 384"}; do
  set -- $cfg
  if [ "$3" = "0" ]; then unset TS_PPO_STEPQ_PAIRS; else export TS_PPO_STEPQ_PAIRS=$3; fi
  echo "=== rows $1 variant $2 pairs $3" >> $O/phases.txt
  NROWS=$1 TS_PPO_STEPQ=$2 PYTHONPATH=. timeout 200 python scripts/gpu_stepq_phases.py 2>> $O/err.txt | grep -v "^trial 1" | tail -24 >> $O/phases.txt
done
unset TS_PPO_STEPQ_PAIRS
grep -c MISMATCH $O/check.txt; grep "CHECK\|check rc" $O/check.txt; cat $O/time.txt; cat $O/phases.txt; grep -v amdgpu.ids $O/err.txt | tail -5
