#!/bin/bash
# A/B of the PPO step kernels through the C ABI, no torch (see scripts/step3_check.cpp)
cd $GRAFT_REPO_ROOT
hipcc -O2 scripts/step3_check.cpp -Iinclude -Ltianshou_amd/lib -ltsengine -Wl,-rpath,$PWD/tianshou_amd/lib -o /tmp/s3c || exit 1
CASES=${CASES:-"17 6 65536 1|17 6 70001 0|3 1 4096 1|31 8 65536 1|11 3 33000 0"}
IFS='|'; for args in $CASES; do
  echo "== obs act rows adv_norm: $args"; IFS=' ' ; timeout 120 /tmp/s3c $args 2>&1 | grep -v amdgpu.ids
done
