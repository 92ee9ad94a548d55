"""Host-side (Python) profile of a bench_next.py workload's update loop: where the enqueue time of a host-bound update goes.
    python scripts/gpu_host_profile.py redq"""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench_next as BN

orig_time = BN._time


def fake_time(update, steps, warmup):
    for _ in range(30):
        update()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        update()
    pr.disable()
    torch.cuda.synchronize()
    for key, n in (("tottime", 30), ("cumulative", 26)):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
        print(s.getvalue())
    return orig_time(update, 20, 5)


BN._time = fake_time
w = sys.argv[1] if len(sys.argv) > 1 else "redq"
print(BN.RUNNERS[w](20, 5, False)["value"])
