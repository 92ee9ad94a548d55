"""Host-side (Python) profile of the DRQN update loop: where the enqueue time of a launch-bound update goes."""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import bench_next as BN

calls = []
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
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue())
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(24)
    print(s.getvalue())
    return orig_time(update, 20, 5)


BN._time = fake_time
BN.run_drqn(20, 5, False)
