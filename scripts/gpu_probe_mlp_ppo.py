"""Runs ts_mlp_ppo_step on a grid of shapes, each in its own process (a GPU fault aborts the process)."""
import subprocess
import sys

CASES = [(4, 64, 2, 64), (4, 64, 2, 4096), (4, 64, 2, 16384), (4, 64, 2, 65536), (17, 128, 6, 65536), (4, 32, 2, 65536),
         (40, 64, 2, 65536)]
CODE = r'''
import sys, torch
from tianshou_amd import ppo_discrete as PD
from tianshou_amd.ppo import PPOConfig
obs_dim, hidden, A, B, apply = (int(x) for x in sys.argv[1:6])
n = PD.layout(obs_dim, hidden, A)["count"]
eng = PD.DiscretePPOEngine(obs_dim, hidden, A, torch.randn(n, device="cuda") * 0.1, PPOConfig(max_grad_norm=0.5))
g = torch.Generator().manual_seed(0)
obs = torch.randn(B, obs_dim, generator=g); act = torch.randint(0, A, (B,), generator=g)
z = lambda: torch.randn(B, generator=g)
grad = torch.empty(n, device="cuda")
out = eng.step(obs, act, z(), z(), z() - 1, z(), grad_out=grad, apply=bool(apply))
torch.cuda.synchronize()
print("ok", out.cpu().tolist()[:2], float(grad.abs().max()))
'''
for case in CASES:
    for apply in (0, 1):
        r = subprocess.run([sys.executable, "-c", CODE, *map(str, case), str(apply)], capture_output=True, text=True)
        tail = (r.stdout.strip().splitlines() or [""])[-1]
        err = [ln for ln in r.stderr.splitlines() if "rror" in ln or "fault" in ln.lower() or "HSA" in ln][:3]
        print(case, "apply" if apply else "grad ", "rc", r.returncode, tail, err, flush=True)
