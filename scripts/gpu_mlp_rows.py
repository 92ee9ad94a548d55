"""Duration of one fused-MLP forward launch (actor shape 384 -> 256 -> 256 -> 64) against the number of rows: a launch bound by
the weight stream out of L2 scales with the workgroup count, one bound by per-workgroup latency does not."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from tianshou_amd import sac as S, _lib

cfg = S.SACConfig()
lay = S.layout(376, 17)
g = torch.Generator().manual_seed(0)
eng = S.SACEngine(376, 17, (torch.randn(lay["actor_count"], generator=g) * 0.05).cuda(), (torch.randn(lay["critic_count"], generator=g) * 0.05).cuda(),
                  (torch.randn(lay["critic_count"], generator=g) * 0.05).cuda(), cfg)
ws = eng._ws
for B in (256, 1024, 2048, 4096, 8192, 16384):
    obs = torch.randn(B, 376, device="cuda")
    noise = torch.randn(B, 17, device="cuda")
    for _ in range(5):
        eng.policy_forward(obs, noise)
    torch.cuda.synchronize()
    ws.profile_begin()
    for _ in range(20):
        eng.policy_forward(obs, noise)
    prof = ws.profile_end()
    ms, n = prof["conv_fwd"]
    print(f"B={B:6d} workgroups={B // 16:5d}  fused forward {ms * 1e3 / max(n, 1):7.1f} us per launch ({n} launches)")
