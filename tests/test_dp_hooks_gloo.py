"""CPU, world_size 2 (gloo), reference mounted: the hook-level data-parallel path a Tianshou user reaches through
`algorithm.update()` -- `HipPPO(data_parallel=True)` over the unmodified reference PPO.

Under test is what ships on the host: each rank mirrors only ITS sub-buffers of the shared VectorReplayBuffer
(`DeviceReplayBuffer.from_tianshou(env_range=shard_envs(...))`, incremental sync of that range only), the hooks route
through `DataParallelPPO` (shard-local preprocess with the global return-statistics hook, one exchange per step), and
the replicas end identical.  The engine and the two device steps are CPU doubles; the arithmetic of the exchange itself
is pinned against the union-batch oracle in tests/test_dp_gloo.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference not mounted")

N_ENV, T = 5, 12           # uneven shards: 3 + 2 sub-buffers


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ref_shim.install()
        import gymnasium as gym
        from torch import nn
        from torch.distributions import Independent, Normal

        from tianshou.algorithm.modelfree.reinforce import ProbabilisticActorPolicy
        from tianshou.algorithm.optim import AdamOptimizerFactory
        from tianshou.data import Batch, VectorReplayBuffer
        from tianshou.utils.net.common import Net
        from tianshou.utils.net.continuous import ContinuousActorProbabilistic, ContinuousCritic
        from tianshou.utils.torch_utils import policy_within_training_step
        import tianshou_amd.buffer as B
        import tianshou_amd.distributed as DD
        import tianshou_amd.integration as I
        import tianshou_amd.returns as R

        seen = {"n": [], "stats": [], "cut": []}

        class FakeEngine:
            """Contract of PPOEngine as the hooks and DataParallelPPO use it."""
            def __init__(self, obs_dim, act_dim, flat, cfg):
                self.obs_dim, self.act_dim, self.cfg, self.P = obs_dim, act_dim, cfg, flat.numel()
                self.params, self.adam_m, self.adam_v, self.adam_step = flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat), 0
                self.ret_rms = [0.0, 1.0, 0.0]
                self.device = torch.device("cpu")

            def preprocess(self, obs, obs_next, act, rew, term, trunc, cut, d_n, reduce_stats=None):
                n = obs.shape[0]
                s1, s2, cnt = float(rew.sum()), float((rew * rew).sum()), float(n)
                g = reduce_stats(s1, s2, cnt) if reduce_stats else (s1, s2, cnt)      # the global-moments hook
                seen["n"].append(n); seen["stats"].append(g)
                seen["cut"].append(sorted(int(c) for c in cut[: int(d_n)]))
                self.ret_rms = [g[0] / g[2], 1.0, g[2]]
                z = torch.zeros(n)
                return {"obs": obs, "act": act, "v_s": z, "returns": z, "adv": rew.float(), "logp_old": z}

            def check(self):
                pass

        class FakeDP(DD.DataParallelPPO):
            """The real minibatch line-up / exchange / loss fix-up; the two device steps replaced."""
            def _pack(self, b):
                return b

            def _begin_update(self):
                pass

            def _local_grad(self, rec, rows, global_batch, adv_stats, out):
                out.zero_()
                out[: self.eng.P] = rec["adv"][rows].sum() / global_batch      # "gradient": sum of local advantages / B
                out[self.eng.P + 1] = rows.numel() / global_batch               # clip part: this rank's share of the batch

            def _apply(self, grad):
                self.eng.adam_step += 1
                self.eng.params = self.eng.params - grad[: self.eng.P]

        def cpu_sample_all(self, batch_size):
            return torch.as_tensor(np.concatenate([
                self.h_offset[e] + (np.arange(self.h_lengths[e]) if self.h_lengths[e] < (self.h_offset[e + 1] - self.h_offset[e])
                                    else (self.h_insertion[e] + np.arange(self.h_lengths[e])) % self.h_lengths[e])
                for e in range(self.buffer_num)]).astype(np.int64))

        def cpu_cuts(m, idx):
            unf = [int(m.h_last_index[e]) for e in range(m.buffer_num) if m.h_lengths[e] > 0 and not bool(m.done[m.h_last_index[e]])]
            pos = np.nonzero(np.isin(idx.numpy(), unf))[0]
            return torch.as_tensor(pos), torch.tensor([len(pos)])

        I._require_gpu = lambda device, who: None
        I.PPOEngine = FakeEngine
        DD.DataParallelPPO = FakeDP
        B.gather_rows = lambda src, idx: src[idx]
        B.gather_rows_multi = lambda srcs, idx: [s[idx] for s in srcs]
        B.DeviceReplayBuffer.sample_indices = cpu_sample_all
        R.cut_positions = cpu_cuts

        torch.manual_seed(0)                         # identical replicas to start with
        actor = ContinuousActorProbabilistic(preprocess_net=Net(state_shape=(17,), hidden_sizes=[64, 64], activation=nn.Tanh),
                                             action_shape=(6,), unbounded=True)
        critic = ContinuousCritic(preprocess_net=Net(state_shape=(17,), hidden_sizes=[64, 64], activation=nn.Tanh))
        policy = ProbabilisticActorPolicy(actor=actor, dist_fn=lambda ls: Independent(Normal(*ls), 1), action_scaling=True,
                                          action_bound_method="clip", action_space=gym.spaces.Box(low=-1.0, high=1.0, shape=(6,)))
        algo = I.make_hip_ppo()(policy=policy, critic=critic, optim=AdamOptimizerFactory(lr=3e-4), eps_clip=0.2,
                                 value_clip=True, vf_coef=0.25, ent_coef=0.0, max_grad_norm=0.5, return_scaling=True,
                                 advantage_normalization=False, dual_clip=None, device="cpu", permutations="host",
                                 data_parallel=True)
        buf = VectorReplayBuffer(N_ENV * T, N_ENV)   # the same shared rollout on every rank (same seed)
        rng = np.random.default_rng(5)
        for t in range(T):
            term = rng.random(N_ENV) < 0.1
            buf.add(Batch(obs=rng.normal(size=(N_ENV, 17)).astype(np.float32), act=rng.normal(size=(N_ENV, 6)).astype(np.float32),
                          rew=rng.normal(size=N_ENV), terminated=term, truncated=np.zeros(N_ENV, bool),
                          obs_next=rng.normal(size=(N_ENV, 17)).astype(np.float32)))
        np.random.seed(10 + rank)
        with policy_within_training_step(algo.policy):
            s0 = algo.update(buffer=buf, batch_size=8, repeat=2)
            # second rollout: only env 1 and env 4 advance -> each rank copies just its own new slots
            for t in range(3):
                buf.add(Batch(obs=np.ones((2, 17), np.float32), act=np.ones((2, 6), np.float32), rew=np.ones(2),
                              terminated=np.zeros(2, bool), truncated=np.zeros(2, bool), obs_next=np.ones((2, 17), np.float32)),
                        buffer_ids=[1, 4])
            m = algo._hip_mirror
            copied = m.sync_from_tianshou(buf)
        lo, hi = DD.shard_envs(N_ENV, rank, world)
        rew = np.asarray(buf.rew)
        w1 = actor.preprocess.model.model[0].weight.detach().clone()
        q.put(dict(rank=rank, n=seen["n"], stats=seen["stats"], env_range=m.env_range, base=m.base, copied=copied,
                   mirror_ok=bool(np.array_equal(m.rew.numpy(), rew[lo * T:hi * T])),
                   steps=s0.gradient_steps, clip=float(s0.actor_loss.mean), w_sum=float(w1.sum()),
                   rms=(algo.ret_rms.mean, algo.ret_rms.count)))
    finally:
        dist.destroy_process_group()


def test_hip_ppo_data_parallel_hooks_world_2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=300)
        res[r["rank"]] = r
    for p in procs:
        p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    a, b = res[0], res[1]
    assert a["env_range"] == (0, 3) and b["env_range"] == (3, 5) and (a["base"], b["base"]) == (0, 3 * T)
    assert a["n"] == [3 * T] and b["n"] == [2 * T]                        # shard-local preprocess
    assert a["stats"] == b["stats"] and a["stats"][0][2] == N_ENV * T      # global return moments on both ranks
    assert a["rms"] == b["rms"] and a["rms"][1] == N_ENV * T
    assert a["mirror_ok"] and b["mirror_ok"]
    assert (a["copied"], b["copied"]) == (3, 3)                            # env 1 -> rank 0, env 4 -> rank 1: own slots only
    assert a["steps"] == b["steps"] == 2 * 4                                # largest shard 36 rows / 8 -> 4 chunks, 2 repeats
    assert a["clip"] == pytest.approx(1.0) and b["clip"] == pytest.approx(1.0)   # shares of every global minibatch add to 1
    assert a["w_sum"] == pytest.approx(b["w_sum"], rel=0, abs=0)            # identical replicas after the update
