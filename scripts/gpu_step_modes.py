"""Times the step kernel under TS_PPO_DBG_MODE (diagnostic experiments)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
dev = torch.device("cuda", 0)
L = bench.Learner(dev, 0, 1)
b = L.preprocess()
perms = [L.rng.permutation(bench.N_TRANS) for _ in range(2)]
L.eng.cfg.lr = 0.0
for _ in range(2):
    L.eng.update(b, bench.MINIBATCH, 2, perms)
torch.cuda.synchronize()
L.ws.profile_begin()
L.eng.update(b, bench.MINIBATCH, 2, perms)
prof = L.ws.profile_end()
print("mode", os.environ.get("TS_PPO_DBG_MODE", "0"), {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in prof.items() if v[1]})
