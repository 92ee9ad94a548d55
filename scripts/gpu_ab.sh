#!/bin/bash
# A/B of library variants in one session: scripts/gpu_ab.sh libA.so libB.so ...   (interleaved rounds)
for round in 1 2 3; do for lib in "$@"; do echo -n "$(basename $lib) "; TS_LIB_PATH=$lib PYTHONPATH=. timeout 120 python scripts/gpu_step_modes.py 2>&1 | tail -1; done; done
