#!/bin/bash
# prints VGPR / scratch / occupancy of selected kernels in a .hip file
f=${1:-/root/repo/tianshou_amd/csrc/ts_ppo.hip}; pat=${2:-"ppo_step_kernelILi9E|ppo_infer_kernelILi9E"}
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/_chk.o 2>&1 \
 | grep -E -A12 "Function Name: _ZN.*($pat)" | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy" | sed 's/.*remark: *//'
