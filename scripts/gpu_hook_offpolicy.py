"""Hook-level rate of the off-policy subclasses at the bench shapes: HipSAC.update(buffer, 4096) on the C5 shape over a host-filled
VectorReplayBuffer stand-in (tests/standin.py), beside the engine-level figure of bench_sac.py; cProfile of the host side.
    python scripts/gpu_hook_offpolicy.py [sac | dqn]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from torch import nn

from tests import standin as SI
from tianshou_amd.integration import make_hip_sac

WHICH = sys.argv[1] if len(sys.argv) > 1 else "sac"
NAME = "HipDQN" if WHICH == "dqn" else "HipSAC"
if WHICH == "dqn":
    from tianshou_amd.integration import make_hip_dqn

    C_, H_, W_, A_, B, E, SLOTS = 4, 84, 84, 6, 512, 16, 1 << 12
    torch.manual_seed(0)
    algo = make_hip_dqn(ref=SI)(policy=SI.DiscreteQLearningPolicy(SI.DQNet(C_, H_, W_, A_)), lr=1e-4, gamma=0.99, n_step_return_horizon=3,
                                target_update_freq=500, is_double=True, huber_loss_delta=1.0, device="cuda").to("cuda")
    N = E * SLOTS
    buf = SI.PrioritizedVectorReplayBuffer(N, E, obs_shape=(H_, W_), act_shape=(), obs_dtype=np.uint8, act_dtype=np.int64, stack_num=C_,
                                           alpha=0.6, beta=0.4)
    rng = np.random.default_rng(0)
    buf.obs[:] = rng.integers(0, 256, (N, H_, W_), dtype=np.uint8)
    buf.obs_next[:] = rng.integers(0, 256, (N, H_, W_), dtype=np.uint8)
    buf.act[:] = rng.integers(0, A_, N)
    buf.rew[:] = rng.standard_normal(N)
    buf.terminated[:] = rng.random(N) < 0.002
    buf.done[:] = buf.terminated
    buf.prio[:] = 1.0
    buf._meta = SI._Meta(("obs", "act", "rew", "terminated", "truncated", "done"))   # ignore_obs_next=True (atari_dqn.py)
    for e, sb in enumerate(buf.buffers):
        sb._size, sb._insertion_idx = SLOTS, 0
        buf._lengths[e] = SLOTS
        buf.last_index[e] = (e + 1) * SLOTS - 1
else:
    OBS, ACT, B, E, SLOTS = 376, 17, 4096, 16, 1 << 14
    torch.manual_seed(0)
    actor = SI.ContinuousActorProbabilistic(SI.Net(OBS, [256, 256], nn.ReLU), ACT, unbounded=True, conditioned_sigma=True)
    c1 = SI.ContinuousCritic(SI.Net(OBS + ACT, [256, 256], nn.ReLU))
    c2 = SI.ContinuousCritic(SI.Net(OBS + ACT, [256, 256], nn.ReLU))
    algo = make_hip_sac(ref=SI)(policy=SI.Policy(actor), critic=c1, critic2=c2, lr=1e-3, critic_lr=1e-3, tau=0.005, gamma=0.99,
                                alpha=SI.AutoAlpha(-float(ACT), 0.0, 3e-4), n_step_return_horizon=1, device="cuda").to("cuda")
    N = E * SLOTS
    buf = SI.VectorReplayBuffer(N, E, obs_shape=(OBS,), act_shape=(ACT,))
    rng = np.random.default_rng(0)
    buf.obs[:] = rng.standard_normal((N, OBS), dtype=np.float32)
    buf.obs_next[:] = rng.standard_normal((N, OBS), dtype=np.float32)
    buf.act[:] = rng.uniform(-1, 1, (N, ACT)).astype(np.float32)
    buf.rew[:] = rng.standard_normal(N, dtype=np.float32)
    buf.terminated[:] = rng.random(N) < 0.001
    buf.done[:] = buf.terminated
    for e, sb in enumerate(buf.buffers):
        sb._size, sb._insertion_idx = SLOTS, 0
        buf._lengths[e] = SLOTS
        buf.last_index[e] = (e + 1) * SLOTS - 1
algo.policy.is_within_training_step = True
for _ in range(20):
    algo.update(buf, B)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    algo.update(buf, B)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{NAME}.update(): {n / dt:.1f} updates/s ({dt / n * 1e3:.3f} ms per update)")
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    algo.update(buf, B)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
print(s.getvalue()[:7000])
