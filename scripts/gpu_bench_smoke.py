"""Runs every secondary bench runner for a few steps without the CPU baseline (a quick end-to-end check of the bench
scripts on a GPU box): python scripts/gpu_bench_smoke.py [workload ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench_next  # noqa: E402

names = sys.argv[1:] or ["drqn", "reinforce", "ppo_discrete", "npg", "trpo", "td3", "ddpg", "dsac", "redq", "qrdqn", "c51", "rainbow"]
for w in names:
    t0 = time.perf_counter()
    try:
        r = bench_next.run(w, 3, 1, with_cpu=False)
        print(json.dumps({"workload": w, "value": round(r["value"], 1), "unit": r["unit"], "frac": round(r["roofline"]["frac"], 3),
                          "wall_s": round(time.perf_counter() - t0, 1)}), flush=True)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"workload": w, "error": repr(e)[:300]}), flush=True)
if not sys.argv[1:]:
    import bench_sac

    r = bench_sac.run(5, 1, with_cpu=False)
    print(json.dumps({"workload": "sac", "value": round(r["value"], 1), "unit": r["unit"]}), flush=True)
