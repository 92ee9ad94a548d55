#!/bin/bash
mkdir -p gpurun_out/r6p
TS_DQN_GRAPH_VERBOSE=1 timeout 300 python bench.py --workload dqn --steps 30 --warmup 5 > gpurun_out/r6p/dbg.json 2> gpurun_out/r6p/dbg.err
grep -v "amdgpu.ids" gpurun_out/r6p/dbg.err | head -20
