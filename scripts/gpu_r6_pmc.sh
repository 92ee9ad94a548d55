#!/bin/bash
# Round 6 counter evidence: TCC FETCH_SIZE / WRITE_SIZE (separate --pmc passes, kernel-trace only)
# of the three BASELINE workloads' own bench commands -> JSONs keyed by (kernel, grid) that the bench lines read for
# `roofline.traffic`; SQ counters (three passes) of the C5 SAC update's GEMM kernels at B = 4096; TCC hit / miss of the same.
#   gpurun_out/r6pmc/{pmc_hbm_traffic.json, pmc_sac.json, pmc_dqn.json, *_summary.txt, pmc_sac_sq.txt, pmc_sac_tcc.txt}
O=$GRAFT_REPO_ROOT/gpurun_out/r6pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in ppo sac dqn; do
  if [ $wl = ppo ]; then B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"; else B="python $R/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline"; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/p_${wl}_$c -o t -- $B > $O/log_${wl}_$c.txt 2>&1
  done
  ( cd $R && python scripts/rocprof_pmc.py $O/p_${wl}_FETCH_SIZE/t_results.db $O/p_${wl}_WRITE_SIZE/t_results.db --json $O/pmc_$wl.json > $O/${wl}_traffic_summary.txt 2>&1 )
  rm -rf $O/p_${wl}_FETCH_SIZE $O/p_${wl}_WRITE_SIZE
done
mv $O/pmc_ppo.json $O/pmc_hbm_traffic.json
B="python $R/bench.py --workload sac --steps 10 --warmup 3 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU -d $O/s1 -o t -- $B > $O/log_s1.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVE_CYCLES -d $O/s2 -o t -- $B > $O/log_s2.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_VMEM SQ_WAVE_CYCLES -d $O/s3 -o t -- $B > $O/log_s3.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE -d $O/s4 -o t -- $B > $O/log_s4.txt 2>&1
cd $R
python scripts/rocprof_pmc.py $O/s1/t_results.db $O/s2/t_results.db $O/s3/t_results.db --match mlp3 > $O/pmc_sac_sq.txt 2>&1
python scripts/rocprof_pmc.py $O/s1/t_results.db $O/s2/t_results.db $O/s3/t_results.db --match conv_wgrad >> $O/pmc_sac_sq.txt 2>&1
python scripts/rocprof_pmc.py $O/s4/t_results.db > $O/pmc_sac_tcc_all.txt 2>&1
grep -A1 -E "mlp3|conv_wgrad|slab_adam|sac_" $O/pmc_sac_tcc_all.txt > $O/pmc_sac_tcc.txt
rm -rf $O/s1 $O/s2 $O/s3 $O/s4 $O/pmc_sac_tcc_all.txt
grep -A1 -E "ppo_step|gae_single|reduce_slabs|ppo_adam" $O/ppo_traffic_summary.txt | head -30
grep -A1 -E "mlp3|conv_wgrad" $O/sac_traffic_summary.txt | head -30
grep -A1 -E "conv_rows|conv_wgrad" $O/dqn_traffic_summary.txt | head -40
head -60 $O/pmc_sac_sq.txt; cat $O/pmc_sac_tcc.txt | head -40
tail -3 $O/log_s4.txt
