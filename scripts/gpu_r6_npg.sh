#!/bin/bash
# round 6: NPG / TRPO on generic trunks (NetNPGEngine)
mkdir -p gpurun_out/r6npg
timeout 1500 python -m pytest tests/test_gpu_npg.py -q -m gpu -x -k "generic or golden or bad" > gpurun_out/r6npg/engine.txt 2>&1; tail -30 gpurun_out/r6npg/engine.txt
timeout 1500 python -m pytest tests/test_gpu_hooks.py -q -m gpu -x -k "natural" > gpurun_out/r6npg/hooks.txt 2>&1; tail -30 gpurun_out/r6npg/hooks.txt
