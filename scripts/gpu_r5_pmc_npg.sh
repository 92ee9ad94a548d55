#!/bin/bash
# SQ counters of the one-launch NPG passes (ts_npg_q.h), mean per launch (three separate --pmc passes, kernel-trace only)
# -> gpurun_out/pmc_npg/pmc_npg_kernels.txt
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_npg; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload npg --steps 2 --warmup 1 --no-cpu-baseline"
export TS_NPG_ONE_STREAM=1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU -d $O/p1 -o t -- $B > $O/log1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVE_CYCLES -d $O/p2 -o t -- $B > $O/log2.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_VMEM SQ_WAVE_CYCLES -d $O/p3 -o t -- $B > $O/log3.txt 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_pmc.py $O/p1/t_results.db $O/p2/t_results.db $O/p3/t_results.db --match npg_ > $O/pmc_npg_kernels.txt 2>&1
rm -rf $O/p1 $O/p2 $O/p3
cat $O/pmc_npg_kernels.txt
