#!/bin/bash
# HBM traffic counters of the PPO bench kernels: two separate PMC passes (FETCH_SIZE, WRITE_SIZE), summarised on the box
# (text per kernel and grid + a JSON with the mean KiB per launch of every kernel name)
O=$GRAFT_REPO_ROOT/gpurun_out/traffic; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/p_$c -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/log_$c.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python scripts/rocprof_pmc.py $O/p_FETCH_SIZE/t_results.db $O/p_WRITE_SIZE/t_results.db --json $O/pmc_hbm_traffic.json > $O/traffic_summary.txt 2>&1
# (copy $O/pmc_hbm_traffic.json to profiles/r05_pmc_hbm_traffic.json: bench.py's roofline.traffic fields read it)
rm -rf $O/p_FETCH_SIZE $O/p_WRITE_SIZE
grep -A1 -E "ppo_step|gae_single|reduce_slabs|ppo_adam|ppo_infer|ppo_pack" $O/traffic_summary.txt | head -40
