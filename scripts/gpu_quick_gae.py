"""Quick GPU timing of the GAE scan (scratch script for gpurun; bench.py is the real contract)."""
import time
import numpy as np
import torch
from tianshou_amd import returns as R

for logn in (20, 22, 24, 26):
    n = 1 << logn
    E = 512
    g = torch.Generator(device="cuda").manual_seed(0)
    v_s = torch.randn(n, device="cuda", generator=g)
    v_n = torch.randn(n, device="cuda", generator=g)
    rew = torch.randn(n, device="cuda", generator=g)
    term = (torch.rand(n, device="cuda", generator=g) < 0.005).to(torch.uint8)
    trunc = torch.zeros(n, dtype=torch.uint8, device="cuda")
    cut = (torch.arange(E, device="cuda") + 1) * (n // E) - 1
    for rew_t in (rew, rew.double()):
        for _ in range(5):
            R.gae_scan(v_s, v_n, rew_t, term, trunc, cut)
        torch.cuda.synchronize()
        iters = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(iters):
            R.gae_scan(v_s, v_n, rew_t, term, trunc, cut)
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / iters
        dt = e0.elapsed_time(e1) / iters * 1e-3
        bpt = 22 if rew_t.dtype == torch.float32 else 26
        print(f"N=2^{logn} rew={rew_t.dtype} gpu {dt*1e6:9.1f} us  wall {wall*1e6:9.1f} us  "
              f"{n/dt/1e9:8.2f} Gtrans/s  {n*bpt/dt/1e12:6.3f} TB/s algorithmic", flush=True)
