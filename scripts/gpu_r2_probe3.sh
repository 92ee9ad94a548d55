#!/bin/bash
# A/B: write-through (sc1) slab stores vs plain stores
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ppo.py -m gpu -q -x > gpurun_out/t4.log 2>&1; echo rc=$? >> gpurun_out/t4.log
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b_sc1_$i.log 2>&1
TS_LIB_PATH=$GRAFT_REPO_ROOT/tianshou_amd/lib/libtsengine_plain.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b_plain_$i.log 2>&1
done
tail -3 gpurun_out/t4.log
