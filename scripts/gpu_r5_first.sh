#!/bin/bash
# First GPU call of the next round: where the actor's half of the step kernel spends its extra ~11 us.
# Phase marks (shader cycles of workgroup 0 / wave 0 + per-workgroup start / end) of the two-network kernel and of the
# one-network kernels on the actor / the critic alone, then the launch durations of the three by HIP events.
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
for n in 0 1 2; do
  NETS=$n PYTHONPATH=. timeout 200 python scripts/gpu_step_phases.py > $O/phases_nets$n.txt 2>> $O/err.txt
done
TS_REINFORCE_BOTH_NETS=1 timeout 100 python bench.py --workload reinforce --no-cpu-baseline > $O/reinforce_both.json 2>> $O/err.txt
timeout 100 python bench.py --workload reinforce --no-cpu-baseline > $O/reinforce_actor.json 2>> $O/err.txt
timeout 100 python bench.py --workload npg --no-cpu-baseline > $O/npg_critic.json 2>> $O/err.txt
head -24 $O/phases_nets0.txt; head -24 $O/phases_nets1.txt; head -24 $O/phases_nets2.txt
grep -v amdgpu.ids $O/err.txt | tail -5
