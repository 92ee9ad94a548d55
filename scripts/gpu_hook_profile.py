"""Where a steady-state HipPPO.update() (C2 shape, host buffer) spends its 11.6 ms: wall time of the two hooks with a device
synchronisation after each, and a cProfile of the host side."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from torch import nn
from tests import standin as SI
from tianshou_amd.integration import make_hip_ppo

dev = torch.device("cuda", 0)
HipPPO = make_hip_ppo("ppo", ref=SI)
torch.manual_seed(0)
actor = SI.ContinuousActorProbabilistic(SI.Net(bench.OBS, [64, 64], nn.Tanh), bench.ACT, unbounded=True)
critic = SI.ContinuousCritic(SI.Net(bench.OBS, [64, 64], nn.Tanh))
c = bench.mujoco_cfg()
algo = HipPPO(policy=SI.Policy(actor), critic=critic, device="cuda", lr=c.lr, eps_clip=c.eps_clip, value_clip=True,
              advantage_normalization=False, vf_coef=c.vf_coef, ent_coef=c.ent_coef, max_grad_norm=c.max_grad_norm,
              return_scaling=True, gamma=c.gamma, gae_lambda=c.gae_lambda).to(dev)
N, E, T = bench.N_TRANS, bench.N_ENV, bench.T_STEPS
buf = SI.VectorReplayBuffer(N, E, obs_shape=(bench.OBS,), act_shape=(bench.ACT,))
rng = np.random.default_rng(3)
buf.obs[:] = rng.standard_normal((N, bench.OBS), dtype=np.float32)
buf.obs_next[:] = rng.standard_normal((N, bench.OBS), dtype=np.float32)
buf.act[:] = rng.standard_normal((N, bench.ACT), dtype=np.float32)
buf.rew[:] = rng.standard_normal(N, dtype=np.float32)
buf.terminated[:] = rng.random(N) < 0.005
buf.done[:] = buf.terminated
for e, sb in enumerate(buf.buffers):
    sb._size, sb._insertion_idx = T, 0
    buf._lengths[e] = T
    buf.last_index[e] = (e + 1) * T - 1
algo.policy.is_within_training_step = True
for _ in range(2):
    algo.update(buf, bench.MINIBATCH, bench.REPEAT)
torch.cuda.synchronize()

marks = {}
orig_pre, orig_upd = algo._preprocess_batch, algo._update_with_batch
def pre(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = orig_pre(*a, **k); torch.cuda.synchronize(); marks["preprocess"] = time.perf_counter() - t; return r
def upd(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = orig_upd(*a, **k); torch.cuda.synchronize(); marks["update_with_batch"] = time.perf_counter() - t; return r
algo._preprocess_batch, algo._update_with_batch = pre, upd
t0 = time.perf_counter(); algo.update(buf, bench.MINIBATCH, bench.REPEAT); torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f"update {tot * 1e3:.2f} ms: " + ", ".join(f"{k} {v * 1e3:.2f} ms" for k, v in marks.items()), f"| rest {(tot - sum(marks.values())) * 1e3:.2f} ms")
algo._preprocess_batch, algo._update_with_batch = orig_pre, orig_upd
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    algo.update(buf, bench.MINIBATCH, bench.REPEAT)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
