"""GAE scan timed with the library's HIP events (kernel launches only), several sizes."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from tianshou_amd import _lib
from tianshou_amd.returns import gae_scan
ws = _lib.default_workspace(0)
for logn in (20, 22, 24):
    n = 1 << logn
    g = torch.Generator(device="cuda").manual_seed(0)
    v = torch.randn(n, device="cuda", generator=g); vn = torch.randn(n, device="cuda", generator=g)
    rew = torch.randn(n, device="cuda", generator=g).double()
    term = (torch.rand(n, device="cuda", generator=g) < 0.005).to(torch.uint8)
    trunc = torch.zeros(n, dtype=torch.uint8, device="cuda")
    cut = (torch.arange(512, device="cuda") + 1) * (n // 512) - 1
    for _ in range(5): gae_scan(v, vn, rew, term, trunc, cut)
    torch.cuda.synchronize()
    ws.profile_begin()
    for _ in range(50): gae_scan(v, vn, rew, term, trunc, cut)
    p = ws.profile_end()
    us = (p["gae_maps"][0] + p["gae_apply"][0]) / 50 * 1e3
    print(f"N=2^{logn} mode={'two-pass' if os.environ.get('TS_GAE_TWO_PASS')=='1' else 'single'} {us:8.2f} us  {n/us/1e3:7.1f} Gtrans/s  {26*n/us/1e6:6.3f} TB/s alg(26B)")
