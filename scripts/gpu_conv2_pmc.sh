#!/bin/bash
# SQ counters of the second-generation conv kernels, mean per launch (three separate --pmc passes, kernel-trace only):
#   bash scripts/gpu_conv2_pmc.sh [batch] [layers]   -> gpurun_out/pmc_conv2/pmc_conv2.txt
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/scripts/gpu_conv2_one.py ${1:-65536} ${2:-}"
cd $GRAFT_REPO_ROOT
R="rocprofv3 --kernel-trace"
( cd /tmp; $R --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU -d $O/p1 -o t -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/log1.txt 2>&1 )
( cd /tmp; $R --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVE_CYCLES -d $O/p2 -o t -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/log2.txt 2>&1 )
( cd /tmp; $R --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_VMEM SQ_WAVE_CYCLES -d $O/p3 -o t -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/log3.txt 2>&1 )
( cd /tmp; $R --pmc GRBM_GUI_ACTIVE -d $O/p4 -o t -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/log4.txt 2>&1 )
( cd /tmp; $R --stats -d $O/p5 -o t -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/log5.txt 2>&1 )
python scripts/rocprof_pmc.py $O/p1/t_results.db $O/p2/t_results.db $O/p3/t_results.db $O/p4/t_results.db --match conv_ > $O/pmc_conv2.txt 2>&1
find $O/p5 -name "*kernel_stats*" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
cat $O/pmc_conv2.txt; head -12 $O/kernel_stats.csv | cut -c1-200
