#!/bin/bash
# split-bf16 forward conv kernel (TS_CONV_SPLIT=1): error against fp64, timing at the C3 / C5 / Atari-PPO shapes, A/B of the workloads
cd $GRAFT_REPO_ROOT; O=gpurun_out/convsplit; mkdir -p $O
for v in 0 1; do
  echo "== TS_CONV_SPLIT=$v"
  TS_CONV_SPLIT=$v python scripts/gpu_conv_check.py 2>&1 | grep -v amdgpu.ids
  TS_CONV_SPLIT=$v python scripts/gpu_conv_micro.py split$v 2>&1 | grep -v amdgpu.ids
done
