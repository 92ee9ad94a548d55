"""GAE scan bandwidth vs size and number of sub-buffer cuts (HIP events on the launch stream; the bitmask pre-pass of
long cut lists is outside the bracket, the wall-clock column includes it)."""
import sys
import time

import torch

from tianshou_amd import _lib
from tianshou_amd.returns import gae_scan

dev = torch.device("cuda")
ws = _lib.default_workspace(0)
g = torch.Generator(device=dev).manual_seed(1)
sizes = [int(x) for x in sys.argv[1:]] or [20, 22, 24, 26]
for log2n in sizes:
    n = 1 << log2n
    v, vn = torch.randn(n, generator=g, device=dev), torch.randn(n, generator=g, device=dev)
    rew = torch.randn(n, generator=g, device=dev).double()
    term = torch.rand(n, generator=g, device=dev) < 0.002
    trunc = torch.zeros(n, dtype=torch.bool, device=dev)
    for envs in (512, 8192, 65536, 512):
        T = n // envs
        if T < 16:
            continue
        cut = torch.arange(T - 1, n, T, device=dev, dtype=torch.int64)
        for _ in range(3):
            gae_scan(v, vn, rew, term, trunc, cut)
        torch.cuda.synchronize()
        ws.profile_begin()
        for _ in range(20):
            gae_scan(v, vn, rew, term, trunc, cut)
        prof = ws.profile_end()
        us = (prof["gae_maps"][0] + prof["gae_apply"][0]) / 20 * 1e3
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            gae_scan(v, vn, rew, term, trunc, cut)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 50 * 1e6
        print(f"n=2^{log2n} envs={envs:6d} kernel {us:8.1f} us {26 * n / us / 1e3:7.1f} GB/s {n / us / 1e3:6.1f} G trans/s | "
              f"wall {wall:8.1f} us", flush=True)
    del v, vn, rew, term, trunc
