#!/bin/bash
# SQ counters (three separate --pmc passes, kernel-trace only) of the C3 DQN update's kernels and of the LSTM layer kernels
# -> gpurun_out/pmc2/pmc_dqn.txt, pmc_drqn.txt
O=$GRAFT_REPO_ROOT/gpurun_out/pmc2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVE_CYCLES"
P3="SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_VMEM SQ_WAVE_CYCLES"
run() {   # tag, command
  tag=$1; shift
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $P -d $O/${tag}_p$i -o t -- "$@" > $O/${tag}_log$i.txt 2>&1
  done
}
run dqn python $GRAFT_REPO_ROOT/bench_dqn.py --steps 6 --warmup 3 --no-cpu-baseline
run drqn python $GRAFT_REPO_ROOT/bench_next.py drqn --steps 6 --warmup 3 --no-cpu-baseline
cd $GRAFT_REPO_ROOT
for tag in dqn drqn; do
  python scripts/rocprof_pmc.py $(find $O/${tag}_p1 $O/${tag}_p2 $O/${tag}_p3 -name '*.db') > $O/pmc_$tag.txt 2>&1
  rm -rf $O/${tag}_p1 $O/${tag}_p2 $O/${tag}_p3
done
grep -A3 "conv_rows_kernel<true\|conv_wgrad_kernel\|lstm_layer" $O/pmc_dqn.txt $O/pmc_drqn.txt | head -40
