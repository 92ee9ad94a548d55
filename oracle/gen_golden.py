"""TEST INFRASTRUCTURE ONLY - generates tests/golden/*.npz by running the UNMODIFIED reference.

Run in the authoring container (where /root/reference is mounted):

    python -m oracle.gen_golden            # writes tests/golden/*.npz

The fixtures are small, committed, and are what pins `oracle/` (and through it the HIP
engine) to the reference.  /root/reference does not exist on the GPU box, so nothing at test
time regenerates them.  Each fixture stores inputs AND the reference's outputs; fixtures
derived from the reference's own tests also store the literal expected vectors of those
tests (file:line cited next to each).
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402
from torch import nn  # noqa: E402
from torch.distributions import Independent, Normal  # noqa: E402

from tianshou.algorithm import Algorithm  # noqa: E402
from tianshou.algorithm.modelfree.a2c import A2C  # noqa: E402
from tianshou.algorithm.modelfree.ppo import PPO  # noqa: E402
from tianshou.algorithm.modelfree.reinforce import ProbabilisticActorPolicy  # noqa: E402
from tianshou.algorithm.optim import AdamOptimizerFactory  # noqa: E402
from tianshou.data import (  # noqa: E402
    Batch,
    PrioritizedVectorReplayBuffer,
    ReplayBuffer,
    SegmentTree,
    VectorReplayBuffer,
)
from tianshou.data.stats import SequenceSummaryStats  # noqa: E402
from tianshou.utils.net.common import ActorCritic, Net  # noqa: E402
from tianshou.utils.net.continuous import ContinuousActorProbabilistic, ContinuousCritic  # noqa: E402
from tianshou.utils.torch_utils import policy_within_training_step  # noqa: E402
import gymnasium as gym  # noqa: E402  (shim stub)

OUT = os.environ.get("TS_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")     # (env: scratch dir of the regeneration test)


def manager_state(buf) -> dict:
    """Raw index state of a ReplayBuffer / ReplayBufferManager."""
    if hasattr(buf, "buffers"):
        return {
            "offset": np.array(buf._extend_offset, np.int64),
            "last_index": np.array(buf.last_index, np.int64),
            "lengths": np.array(buf._lengths, np.int64),
            "insertion": np.asarray([b._insertion_idx for b in buf.buffers], np.int64),
        }
    return {
        "offset": np.asarray([0, buf.maxsize], np.int64),
        "last_index": np.array(buf.last_index, np.int64),
        "lengths": np.asarray([len(buf)], np.int64),
        "insertion": np.asarray([buf._insertion_idx], np.int64),
    }


# ---------------------------------------------------------------------------------------------
def gen_returns_kat() -> None:
    """GAE / n-step known-answer cases of test/base/test_returns.py."""
    out: dict[str, np.ndarray] = {}
    fn = Algorithm.compute_episodic_return
    # (terminated, truncated, rew, v, gamma, lambda, literal answer, cite)
    cases = [
        ([1, 0, 0, 1, 0, 0, 0, 1.0], [0, 0, 0, 0, 0, 1, 0, 0], [0, 1, 2, 3, 4, 5, 6, 7.0], None, 0.1, 1.0,
         [0, 1.23, 2.3, 3, 4.5, 5, 6.7, 7]),                                    # test_returns.py:27-46
        ([0, 1, 0, 1, 0, 1, 0.0], [0] * 7, [7, 6, 1, 2, 3, 4, 5.0], None, 0.1, 1.0,
         [7.6, 6, 1.2, 2, 3.4, 4, 5]),                                          # :48-61
        ([0, 1, 0, 1, 0, 0, 1.0], [0] * 7, [7, 6, 1, 2, 3, 4, 5.0], None, 0.1, 1.0,
         [7.6, 6, 1.2, 2, 3.45, 4.5, 5]),                                       # :63-76
        ([0, 0, 0, 1.0, 0, 0, 0, 1, 0, 0, 0, 1], [0] * 12,
         [101, 102, 103.0, 200, 104, 105, 106, 201, 107, 108, 109, 202],
         [2.0, 3.0, 4, -1, 5.0, 6.0, 7, -2, 8.0, 9.0, 10, -3], 0.99, 0.95,
         [454.8344, 376.1143, 291.298, 200.0, 464.5610, 383.1085, 295.387, 201.0, 474.2876,
          390.1027, 299.476, 202.0]),                                           # :78-108
        ([0] * 11 + [1], [0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0],
         [101, 102, 103.0, 200, 104, 105, 106, 201, 107, 108, 109, 202],
         [2.0, 3.0, 4, -1, 5.0, 6.0, 7, -2, 8.0, 9.0, 10, -3], 0.99, 0.95,
         [454.0109, 375.2386, 290.3669, 199.01, 462.9138, 381.3571, 293.5248, 199.02, 474.2876,
          390.1027, 299.476, 202.0]),                                           # :110-159
    ]
    buf = ReplayBuffer(20)
    for c, (term, trunc, rew, v, gamma, lam, ans) in enumerate(cases):
        buf.reset()
        batch = Batch(terminated=np.array(term, float), truncated=np.array(trunc, float),
                      rew=np.array(rew, float))
        for b in iter(batch):
            b.obs = b.act = 1
            buf.add(b)
        indices = buf.sample_indices(0)
        vt = None if v is None else np.array(v, float)
        returns, adv = fn(batch, buf, indices, vt, gamma=gamma, gae_lambda=lam)
        assert np.allclose(returns, ans)
        out[f"gae{c}_terminated"] = np.array(term, bool)
        out[f"gae{c}_truncated"] = np.array(trunc, bool)
        out[f"gae{c}_rew"] = np.array(rew, float)
        out[f"gae{c}_has_v"] = np.array(v is not None)
        out[f"gae{c}_v_next"] = np.zeros(len(rew)) if v is None else np.array(v, float)
        out[f"gae{c}_indices"] = indices
        out[f"gae{c}_unfinished"] = buf.unfinished_index()
        out[f"gae{c}_gamma_lambda"] = np.array([gamma, lam])
        out[f"gae{c}_ref_returns"] = returns
        out[f"gae{c}_ref_adv"] = adv
        out[f"gae{c}_literal"] = np.array(ans, float)
    out["n_gae"] = np.array(len(cases))

    # n-step: test_returns.py:197-275 (plain) and :278-357 (time-limit truncation)
    def target_q_fn(buffer, indices):  # test_returns.py:162-165
        indices = buffer.next(indices)
        return torch.tensor(-buffer.rew[indices], dtype=torch.float32)

    def target_q_fn_multidim(buffer, indices):  # :167-168
        return target_q_fn(buffer, indices).unsqueeze(1).repeat(1, 51)

    literal = {
        (0, 1): [2.6, 4, 4.4, 5.3, 6.2, 8, 8, 8.9, 9.8, 12],                 # :222
        (0, 2): [3.4, 4, 5.53, 6.62, 7.8, 8, 9.89, 10.98, 12.2, 12],          # :242
        (0, 10): [3.4, 4, 5.678, 6.78, 7.8, 8, 10.122, 11.22, 12.2, 12],      # :262
        (1, 1): [2.6, 3.6, 4.4, 5.3, 6.2, 8, 8, 8.9, 9.8, 12],               # :304
        (1, 2): [3.36, 3.6, 5.53, 6.62, 7.8, 8, 9.89, 10.98, 12.2, 12],       # :324
        (1, 10): [3.36, 3.6, 5.678, 6.78, 7.8, 8, 10.122, 11.22, 12.2, 12],   # :344
    }
    for variant in (0, 1):
        buf = ReplayBuffer(10)
        for i in range(12):
            if variant == 0:
                b = Batch(obs=0, act=0, rew=i + 1, terminated=i % 4 == 3, truncated=False)
            else:
                b = Batch(obs=0, act=0, rew=i + 1, terminated=i % 4 == 3 and i != 3,
                          truncated=i == 3, info={"TimeLimit.truncated": i == 3})
            buf.add(b)
        batch, indices = buf.sample(0)
        st = manager_state(buf)
        pre = f"nstep{variant}_"
        for k, v in st.items():
            out[pre + k] = v
        out[pre + "rew"] = np.asarray(buf.rew, float)
        out[pre + "terminated"] = np.asarray(buf.terminated, bool)
        out[pre + "truncated"] = np.asarray(buf.truncated, bool)
        out[pre + "done"] = np.asarray(buf.done, bool)
        out[pre + "indices"] = indices
        for n in (1, 2, 10):
            r = Algorithm.compute_nstep_return(batch, buf, indices, target_q_fn, gamma=0.1,
                                               n_step=n).pop("returns")
            r = r.numpy()
            assert np.allclose(r.reshape(-1), literal[(variant, n)])
            rm = Algorithm.compute_nstep_return(batch, buf, indices, target_q_fn_multidim,
                                                gamma=0.1, n_step=n).pop("returns").numpy()
            out[pre + f"n{n}_ref"] = r
            out[pre + f"n{n}_ref_multidim"] = rm
            out[pre + f"n{n}_literal"] = np.array(literal[(variant, n)], float)
    np.savez_compressed(os.path.join(OUT, "returns_kat.npz"), **out)


# ---------------------------------------------------------------------------------------------
def gen_buffer_index() -> None:
    """Random VectorReplayBuffer histories -> next/prev/unfinished/sample_indices(0) vectors.
    Also replays the hand-written scenario of test/base/test_buffer.py:740-880 (4 sub-buffers
    of 5 slots, ragged buffer_ids) whose literal index vectors the reference test pins."""
    out: dict[str, np.ndarray] = {}
    rng = np.random.default_rng(1234)
    scen = 0

    def snapshot(buf, tag):
        st = manager_state(buf)
        for k, v in st.items():
            out[f"{tag}_{k}"] = v
        B = buf.maxsize
        out[f"{tag}_done"] = np.asarray(buf.done, bool).copy()
        q = np.concatenate([np.arange(B), rng.integers(-2 * B, 3 * B, size=64)]).astype(np.int64)
        out[f"{tag}_query"] = q
        out[f"{tag}_next"] = np.asarray(buf.next(q), np.int64)
        out[f"{tag}_prev"] = np.asarray(buf.prev(q), np.int64)
        out[f"{tag}_unfinished"] = np.asarray(buf.unfinished_index(), np.int64)
        out[f"{tag}_sample0"] = np.asarray(buf.sample_indices(0), np.int64)

    for (total, E, steps, p_done) in [(20, 4, 3, 0.3), (20, 4, 11, 0.3), (64, 8, 37, 0.1),
                                      (12, 3, 3, 0.0), (35, 5, 50, 0.5), (16, 1, 23, 0.2)]:
        buf = VectorReplayBuffer(total, E)
        for t in range(steps):
            k = int(rng.integers(1, E + 1))
            ids = np.sort(rng.choice(E, size=k, replace=False))
            term = rng.random(k) < p_done
            trunc = (rng.random(k) < p_done / 3) & ~term
            buf.add(Batch(obs=np.zeros(k), act=np.zeros(k), rew=rng.normal(size=k),
                          terminated=term, truncated=trunc, obs_next=np.zeros(k)),
                    buffer_ids=ids)
        snapshot(buf, f"s{scen}")
        scen += 1
    # literal scenario of test/base/test_buffer.py:740-961: 4 x ReplayBuffer(5), ragged adds
    buf = VectorReplayBuffer(20, 4)
    batch = Batch(obs=[1, 2, 3], act=[1, 2, 3], rew=[1, 2, 3], terminated=[0, 0, 1],
                  truncated=[0, 0, 0])
    buf.add(batch, buffer_ids=[0, 1, 2])                                       # :752
    buf.add(Batch(obs=[4], act=[4], rew=[4], terminated=[1], truncated=[0]), buffer_ids=[3])  # :770
    data = np.array([0, 0, 0, 0])
    buf.add(Batch(obs=data, act=data, rew=data, terminated=data, truncated=data),
            buffer_ids=[0, 1, 2, 3])                                           # :786
    buf.add(Batch(obs=data, act=data, rew=data, terminated=1 - data, truncated=data),
            buffer_ids=[0, 1, 2, 3])                                           # :793
    buf.add(Batch(obs=data, act=data, rew=data, terminated=data, truncated=data),
            buffer_ids=[0, 1, 2, 3])                                           # :801
    buf.add(Batch(obs=data, act=data, rew=data, terminated=[0, 1, 0, 1], truncated=data),
            buffer_ids=[0, 1, 2, 3])                                           # :808
    snapshot(buf, f"s{scen}")
    out["litA_scen"] = np.array(scen)
    out["litA_done"] = np.array([0, 0, 1, 0, 0, 0, 0, 1, 0, 1, 1, 0, 1, 0, 0, 1, 0, 1, 0, 1], bool)  # :822-846
    out["litA_prev"] = np.array([0, 0, 1, 3, 3, 5, 5, 6, 8, 8, 10, 11, 11, 13, 13, 15, 16, 16, 18, 18], np.int64)  # :847-871
    out["litA_next"] = np.array([1, 2, 2, 4, 4, 6, 7, 7, 9, 9, 10, 12, 12, 14, 14, 15, 17, 17, 19, 19], np.int64)  # :872-896
    out["litA_unfinished"] = np.array([4, 14], np.int64)                        # :897
    assert np.array_equal(out[f"s{scen}_next"][:20], out["litA_next"])
    assert np.array_equal(out[f"s{scen}_prev"][:20], out["litA_prev"])
    scen += 1
    buf.add(Batch(obs=[1], act=[1], rew=[1], terminated=[1], truncated=[0]), buffer_ids=[2])  # :898
    snapshot(buf, f"s{scen}")
    out["litB_scen"] = np.array(scen)
    out["litB_prev"] = np.array([0, 0, 1, 3, 3, 5, 5, 6, 8, 8, 14, 11, 11, 13, 13, 15, 16, 16, 18, 18], np.int64)  # :912-936
    out["litB_next"] = np.array([1, 2, 2, 4, 4, 6, 7, 7, 9, 9, 10, 12, 12, 14, 10, 15, 17, 17, 19, 19], np.int64)  # :937-961
    out["litB_unfinished"] = np.array([4], np.int64)                            # :909
    assert np.array_equal(out[f"s{scen}_next"][:20], out["litB_next"])
    assert np.array_equal(out[f"s{scen}_prev"][:20], out["litB_prev"])
    scen += 1
    out["n_scen"] = np.array(scen)
    np.savez_compressed(os.path.join(OUT, "buffer_index.npz"), **out)


# ---------------------------------------------------------------------------------------------
def gen_segtree_per() -> None:
    out: dict[str, np.ndarray] = {}
    rng = np.random.default_rng(7)
    for c, size in enumerate([1, 6, 100, 1000]):
        tree = SegmentTree(size)
        bound = tree._bound
        out[f"t{c}_size_bound"] = np.array([size, bound])
        for r in range(3):
            k = int(rng.integers(1, max(2, size)))
            idx = rng.integers(0, size, size=k)
            val = rng.random(k) * 3.0
            out[f"t{c}_r{r}_tree_before"] = tree._value.copy()
            tree[idx] = val
            out[f"t{c}_r{r}_idx"] = idx.astype(np.int64)
            out[f"t{c}_r{r}_val"] = val
            out[f"t{c}_r{r}_tree_after"] = tree._value.copy()
        total = tree.reduce()
        q = rng.random(257) * total
        q[0] = 0.0
        out[f"t{c}_query"] = q.copy()
        out[f"t{c}_prefix_idx"] = np.asarray(tree.get_prefix_sum_idx(q.copy()), np.int64)
        lo = rng.integers(0, size, size=16)
        hi = np.minimum(lo + rng.integers(1, size + 1, size=16), size)
        out[f"t{c}_range"] = np.stack([lo, hi]).astype(np.int64)
        out[f"t{c}_range_sum"] = np.array([tree.reduce(int(a), int(b)) for a, b in zip(lo, hi)])
    out["n_tree"] = np.array(4)

    # PER: PrioritizedVectorReplayBuffer(64, 4, alpha .6, beta .4)
    buf = PrioritizedVectorReplayBuffer(64, 4, alpha=0.6, beta=0.4)
    for t in range(23):
        buf.add(Batch(obs=np.zeros(4), act=np.zeros(4), rew=rng.normal(size=4),
                      terminated=rng.random(4) < 0.1, truncated=np.zeros(4, bool),
                      obs_next=np.zeros(4)))
    idx = buf.sample_indices(0)
    out["per_tree0"] = buf.weight._value.copy()
    out["per_bound"] = np.array(buf.weight._bound)
    out["per_alpha_beta"] = np.array([0.6, 0.4])
    td = torch.tensor(rng.normal(size=len(idx)).astype(np.float32))
    out["per_upd_idx"] = idx.astype(np.int64)
    out["per_upd_td"] = td.numpy()
    out["per_prio_before"] = np.array([buf._max_prio, buf._min_prio])
    buf.update_weight(idx, td)
    out["per_tree1"] = buf.weight._value.copy()
    out["per_prio_after"] = np.array([float(buf._max_prio), float(buf._min_prio)])
    np.random.seed(5)
    u = np.random.rand(32)
    out["per_uniform"] = u
    np.random.seed(5)
    sidx = buf.sample_indices(32)
    out["per_sample_idx"] = sidx.astype(np.int64)
    out["per_is_weight"] = np.asarray(buf[sidx].weight, np.float64)
    np.savez_compressed(os.path.join(OUT, "segtree_per.npz"), **out)


# ---------------------------------------------------------------------------------------------
def _flat_from_modules(actor, critic) -> np.ndarray:
    """Actor trunk (w, b)*, mu (w, b), sigma_param | critic trunk (w, b)*, last (w, b) -- Net trunks of any depth (the Linears of
    the Sequential in order)."""
    sd_a, sd_c = actor.state_dict(), critic.state_dict()
    trunk = lambda sd: [sd[k] for k in sd if k.startswith("preprocess.model.model.")]      # noqa: E731 (weight, bias per Linear, in order)
    parts = trunk(sd_a) + [sd_a["mu.model.0.weight"], sd_a["mu.model.0.bias"], sd_a["sigma_param"]] + \
        trunk(sd_c) + [sd_c["last.model.0.weight"], sd_c["last.model.0.bias"]]
    return torch.cat([p.detach().reshape(-1) for p in parts]).numpy().astype(np.float32)


def _flat_adam(algorithm, actor, critic, key: str) -> np.ndarray:
    opt = algorithm.optim._optim
    named = dict(ActorCritic(actor, critic).named_parameters())
    order = [
        "actor.preprocess.model.model.0.weight", "actor.preprocess.model.model.0.bias",
        "actor.preprocess.model.model.2.weight", "actor.preprocess.model.model.2.bias",
        "actor.mu.model.0.weight", "actor.mu.model.0.bias", "actor.sigma_param",
        "critic.preprocess.model.model.0.weight", "critic.preprocess.model.model.0.bias",
        "critic.preprocess.model.model.2.weight", "critic.preprocess.model.model.2.bias",
        "critic.last.model.0.weight", "critic.last.model.0.bias",
    ]
    return torch.cat([opt.state[named[k]][key].detach().reshape(-1) for k in order]).numpy()


def gen_ppo(tag: str, *, E: int, T: int, obs_dim: int, act_dim: int, batch_size: int, repeat: int,
            seed: int, n_updates: int = 1, algo: str = "ppo", lr_decay: tuple[int, int, int] | None = None,
            max_action: float | None = None, optim: tuple[str, dict] | None = None, **ppo_kwargs) -> None:
    """Runs the reference PPO.update() on a synthetic VectorReplayBuffer and dumps every
    intermediate the engine has to reproduce.  `lr_decay` = (max_epochs, epoch_num_steps,
    collection_step_num_env_steps) attaches `LRSchedulerFactoryLinear` exactly like
    examples/mujoco/mujoco_ppo.py:124-131 (stepped by Algorithm._update after every update(),
    algorithm_base.py:628-629); the learning rate in force during update u is recorded as `u{u}_lr`.
    `max_action` (round 6): ContinuousActorProbabilistic(unbounded=False, max_action=...) -- the constructor default
    (utils/net/continuous.py:194, 230-231) -- with a mu head large enough for tanh to matter.  `optim` = ("rmsprop" | "adam",
    factory kwargs): RMSpropOptimizerFactory / AdamOptimizerFactory of tianshou/algorithm/optim.py:89-140 (the state vectors
    are recorded under the names u{u}_adam_m / u{u}_adam_v: momentum buffer or grad_avg / square_avg for RMSprop)."""
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    N = E * T
    net_a = Net(state_shape=(obs_dim,), hidden_sizes=[64, 64], activation=nn.Tanh)
    if max_action is None:
        actor = ContinuousActorProbabilistic(preprocess_net=net_a, action_shape=(act_dim,), unbounded=True)
    else:
        actor = ContinuousActorProbabilistic(preprocess_net=net_a, action_shape=(act_dim,), max_action=max_action)
        assert not actor._unbounded
    net_c = Net(state_shape=(obs_dim,), hidden_sizes=[64, 64], activation=nn.Tanh)
    critic = ContinuousCritic(preprocess_net=net_c)
    torch.nn.init.constant_(actor.sigma_param, -0.5)
    for m in ActorCritic(actor, critic).modules():
        if isinstance(m, nn.Linear):
            nn.init.orthogonal_(m.weight, gain=np.sqrt(2))
            nn.init.zeros_(m.bias)
    for m in actor.mu.modules():
        if isinstance(m, nn.Linear):
            nn.init.zeros_(m.bias)
            # (bounded actor: a head whose outputs reach into tanh's curved range, so that the bound and its derivative matter)
            m.weight.data.copy_((0.01 if max_action is None else 0.6) * m.weight.data)

    def dist(loc_scale):
        loc, scale = loc_scale
        return Independent(Normal(loc, scale), 1)

    space = gym.spaces.Box(low=-1.0, high=1.0, shape=(act_dim,))
    policy = ProbabilisticActorPolicy(actor=actor, dist_fn=dist, action_scaling=True,
                                      action_bound_method="clip", action_space=space)
    lr = ppo_kwargs.pop("lr", 3e-4)
    cls = PPO if algo == "ppo" else A2C
    if optim is None:
        optim_factory = AdamOptimizerFactory(lr=lr)
    elif optim[0] == "rmsprop":
        from tianshou.algorithm.optim import RMSpropOptimizerFactory

        optim_factory = RMSpropOptimizerFactory(lr=lr, **optim[1])
    else:
        optim_factory = AdamOptimizerFactory(lr=lr, **optim[1])
    if lr_decay is not None:
        from tianshou.algorithm.optim import LRSchedulerFactoryLinear

        optim_factory.with_lr_scheduler_factory(LRSchedulerFactoryLinear(
            max_epochs=lr_decay[0], epoch_num_steps=lr_decay[1], collection_step_num_env_steps=lr_decay[2]))
    algorithm = cls(policy=policy, critic=critic, optim=optim_factory, **ppo_kwargs)
    assert (len(algorithm.lr_schedulers) == 1) == (lr_decay is not None)

    out: dict[str, np.ndarray] = {}
    out["flat_params0"] = _flat_from_modules(actor, critic)
    out["dims"] = np.array([E, T, obs_dim, act_dim, batch_size, repeat, n_updates])

    # capture the global-np.random permutations Batch.split draws and per-step loss sequences
    perms: list[np.ndarray] = []
    orig_perm = np.random.permutation

    def rec_perm(n):
        p = orig_perm(n)
        perms.append(np.asarray(p, np.int64))
        return p

    seqs: list[np.ndarray] = []
    orig_from = SequenceSummaryStats.from_sequence.__func__

    def rec_from(cls, seq):
        seqs.append(np.asarray(seq, np.float64))
        return orig_from(cls, seq)

    pre_dump: dict[str, np.ndarray] = {}
    orig_pre = cls._preprocess_batch

    def rec_pre(self, batch, buffer, indices):
        b = orig_pre(self, batch, buffer, indices)
        if not pre_dump:
            pre_dump["v_s"] = b.v_s.numpy().copy()
            pre_dump["returns"] = b.returns.numpy().copy()
            pre_dump["adv"] = b.adv.numpy().copy()
            pre_dump["logp_old"] = b.logp_old.numpy().copy() if algo == "ppo" else np.zeros(len(indices), np.float32)
            pre_dump["indices"] = np.asarray(indices, np.int64)
            pre_dump["unfinished"] = np.asarray(buffer.unfinished_index(), np.int64)
        return b

    np.random.permutation = rec_perm
    SequenceSummaryStats.from_sequence = classmethod(rec_from)
    cls._preprocess_batch = rec_pre
    try:
        for u in range(n_updates):
            buf = VectorReplayBuffer(N, E)
            obs = rng.normal(size=(T + 1, E, obs_dim)).astype(np.float32)
            act = rng.normal(size=(T, E, act_dim)).astype(np.float32)
            rew = rng.normal(size=(T, E)).astype(np.float32)
            term = rng.random((T, E)) < 0.02
            trunc = np.zeros((T, E), bool)
            trunc[T // 2 - 1 :: T // 2] = True       # a time-limit style truncation pattern
            trunc &= ~term
            for t in range(T):
                buf.add(Batch(obs=obs[t], act=act[t], rew=rew[t], terminated=term[t],
                              truncated=trunc[t], obs_next=obs[t + 1]))
            if u == 0:
                out["obs"] = np.asarray(buf.obs, np.float32)
                out["obs_next"] = np.asarray(buf.obs_next, np.float32)
                out["act"] = np.asarray(buf.act, np.float32)
                out["rew"] = np.asarray(buf.rew, np.float64)
                out["terminated"] = np.asarray(buf.terminated, bool)
                out["truncated"] = np.asarray(buf.truncated, bool)
                for k, v in manager_state(buf).items():
                    out["buf_" + k] = v
            else:
                out[f"u{u}_obs"] = np.asarray(buf.obs, np.float32)
                out[f"u{u}_obs_next"] = np.asarray(buf.obs_next, np.float32)
                out[f"u{u}_act"] = np.asarray(buf.act, np.float32)
                out[f"u{u}_rew"] = np.asarray(buf.rew, np.float64)
                out[f"u{u}_terminated"] = np.asarray(buf.terminated, bool)
                out[f"u{u}_truncated"] = np.asarray(buf.truncated, bool)
            np.random.seed(seed + 100 + u)
            n_perm0, n_seq0 = len(perms), len(seqs)
            out[f"u{u}_lr"] = np.array(algorithm.optim._optim.param_groups[0]["lr"], np.float64)
            with policy_within_training_step(algorithm.policy):
                stats = algorithm.update(buffer=buf, batch_size=batch_size, repeat=repeat)
            p = perms[n_perm0:]
            assert len(p) == repeat, (len(p), repeat)
            out[f"u{u}_perms"] = np.stack(p)
            s = seqs[n_seq0:]
            assert len(s) == 4
            # order of construction in ppo.py:218-222: loss, actor(clip) loss, vf loss, ent loss
            out[f"u{u}_losses"] = np.stack(s, axis=1)
            out[f"u{u}_gradient_steps"] = np.array(stats.gradient_steps)
            out[f"u{u}_flat_params"] = _flat_from_modules(actor, critic)
            if optim is not None and optim[0] == "rmsprop":
                aux = "momentum_buffer" if optim[1].get("momentum", 0) > 0 else ("grad_avg" if optim[1].get("centered") else None)
                out[f"u{u}_adam_m"] = (_flat_adam(algorithm, actor, critic, aux) if aux
                                       else np.zeros_like(out[f"u{u}_flat_params"]))
                out[f"u{u}_adam_v"] = _flat_adam(algorithm, actor, critic, "square_avg")
            else:
                out[f"u{u}_adam_m"] = _flat_adam(algorithm, actor, critic, "exp_avg")
                out[f"u{u}_adam_v"] = _flat_adam(algorithm, actor, critic, "exp_avg_sq")
            out[f"u{u}_ret_rms"] = np.array([float(algorithm.ret_rms.mean),
                                             float(algorithm.ret_rms.var),
                                             float(algorithm.ret_rms.count)])
    finally:
        np.random.permutation = orig_perm
        SequenceSummaryStats.from_sequence = classmethod(orig_from)
        cls._preprocess_batch = orig_pre
    for k, v in pre_dump.items():
        out["pre_" + k] = v
    cfg = dict(gamma=algorithm.gamma, gae_lambda=algorithm.gae_lambda,
               eps_clip=getattr(algorithm, "eps_clip", 0.0),
               dual_clip=getattr(algorithm, "dual_clip", None) or 0.0,
               value_clip=float(getattr(algorithm, "value_clip", False)),
               advantage_normalization=float(getattr(algorithm, "advantage_normalization", False)),
               recompute_advantage=float(getattr(algorithm, "recompute_adv", False)), vf_coef=algorithm.vf_coef,
               ent_coef=algorithm.ent_coef,
               max_grad_norm=algorithm.optim._max_grad_norm or 0.0,
               return_scaling=float(algorithm.return_scaling), lr=lr,
               max_batchsize=float(algorithm.max_batchsize))
    cfg["is_a2c"] = float(algo == "a2c")
    g0 = algorithm.optim._optim.param_groups[0]
    cfg["max_action"] = float(max_action or 0.0)                                  # 0: unbounded
    cfg["opt_rmsprop"] = float(type(algorithm.optim._optim).__name__ == "RMSprop")
    cfg["weight_decay"] = float(g0.get("weight_decay", 0.0))
    cfg["opt_eps"] = float(g0["eps"])
    cfg["rms_alpha"] = float(g0.get("alpha", 0.99))
    cfg["rms_momentum"] = float(g0.get("momentum", 0.0))
    cfg["rms_centered"] = float(g0.get("centered", False))
    out["cfg_keys"] = np.array(list(cfg.keys()))
    out["cfg_vals"] = np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"ppo_{tag}.npz"), **out)


def gen_ppo_net(tag: str, *, E: int, T: int, obs_dim: int, act_dim: int, hidden_a: list[int], hidden_c: list[int], activation,
                batch_size: int, repeat: int, seed: int, algo: str = "ppo", conditioned_sigma: bool = False,
                max_action: float | None = None, optim: tuple[str, dict] | None = None, layer_norm: bool = False,
                norm_args: dict | None = None, **ppo_kwargs) -> None:
    """The reference PPO / A2C update() for actor-critics whose trunks are Net(hidden_sizes=..., activation=...) of any depth
    (utils/net/common.py:90-178, 246-369; `activation` = nn.Tanh, nn.ReLU or None): inputs, Batch.split's permutations,
    per-step losses and the parameters / Adam moments after the update, as lists of tensors in module order
    (trunk (w, b)*, head w, head b[, sigma_param]).  Replayed by tests/test_gpu_ppo_net.py on the engine's per-layer path.
    layer_norm: Net(norm_layer=nn.LayerNorm, norm_args=...) -- Linear -> LayerNorm -> activation per hidden layer
    (common.py:25-39); the trunk's tensors are then (w, b, gamma, beta)* and gamma / beta start away from (1, 0)."""
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    N = E * T
    nkw = dict(norm_layer=nn.LayerNorm, norm_args=norm_args) if layer_norm else {}
    net_a = Net(state_shape=(obs_dim,), hidden_sizes=hidden_a, activation=activation, **nkw)
    if max_action is None:
        actor = ContinuousActorProbabilistic(preprocess_net=net_a, action_shape=(act_dim,), unbounded=True,
                                             conditioned_sigma=conditioned_sigma)
    else:       # the constructor default: mu = max_action * tanh(Linear(h)) (continuous.py:194, 230-231)
        actor = ContinuousActorProbabilistic(preprocess_net=net_a, action_shape=(act_dim,), max_action=max_action,
                                             conditioned_sigma=conditioned_sigma)
    net_c = Net(state_shape=(obs_dim,), hidden_sizes=hidden_c, activation=activation, **nkw)
    critic = ContinuousCritic(preprocess_net=net_c)
    if not conditioned_sigma:
        torch.nn.init.constant_(actor.sigma_param, -0.5)
    for m in ActorCritic(actor, critic).modules():
        if isinstance(m, nn.Linear):
            nn.init.orthogonal_(m.weight, gain=np.sqrt(2))
            nn.init.normal_(m.bias, std=0.1)
        elif isinstance(m, nn.LayerNorm):
            nn.init.uniform_(m.weight, 0.5, 1.5)
            nn.init.normal_(m.bias, std=0.2)
    if conditioned_sigma:                       # a sigma head whose outputs straddle the upper clamp (SIGMA_MAX = 2)
        with torch.no_grad():
            actor.sigma.model[0].weight.mul_(0.3)
            actor.sigma.model[0].bias.add_(1.8)

    def dist(loc_scale):
        loc, scale = loc_scale
        return Independent(Normal(loc, scale), 1)

    policy = ProbabilisticActorPolicy(actor=actor, dist_fn=dist, action_scaling=True, action_bound_method="clip",
                                      action_space=gym.spaces.Box(low=-1.0, high=1.0, shape=(act_dim,)))
    lr = ppo_kwargs.pop("lr", 3e-4)
    cls = PPO if algo == "ppo" else A2C
    if optim is not None and optim[0] == "rmsprop":
        from tianshou.algorithm.optim import RMSpropOptimizerFactory

        optim_factory = RMSpropOptimizerFactory(lr=lr, **optim[1])
    else:
        optim_factory = AdamOptimizerFactory(lr=lr, **(optim[1] if optim else {}))
    algorithm = cls(policy=policy, critic=critic, optim=optim_factory, **ppo_kwargs)

    def tensors(mod, head):
        lin = [m for m in mod.preprocess.model.model if isinstance(m, (nn.Linear, nn.LayerNorm))]
        out = []
        for m in lin + [m for m in head.modules() if isinstance(m, nn.Linear)]:
            out += [m.weight, m.bias]
        return out

    a_par = tensors(actor, actor.mu) + ([actor.sigma.model[0].weight, actor.sigma.model[0].bias] if conditioned_sigma
                                        else [actor.sigma_param])
    c_par = tensors(critic, critic.last)
    out: dict[str, np.ndarray] = {"dims": np.array([E, T, obs_dim, act_dim, batch_size, repeat]),
                                  "hidden_a": np.array(hidden_a), "hidden_c": np.array(hidden_c),
                                  "activation": np.array({nn.Tanh: 0, nn.ReLU: 1, None: 2}[activation]),
                                  "conditioned_sigma": np.array(int(conditioned_sigma)),
                                  "max_action": np.array(float(max_action or 0.0))}
    if layer_norm:                                              # (only then: the other fixtures keep their key set)
        out["layer_norm"] = np.array(1)
        out["ln_eps"] = np.array(float((norm_args or {}).get("eps", 1e-5)))
    for i, t in enumerate(a_par):
        out[f"a{i}_0"] = t.detach().numpy().copy()
    for i, t in enumerate(c_par):
        out[f"c{i}_0"] = t.detach().numpy().copy()
    perms: list[np.ndarray] = []
    orig_perm = np.random.permutation

    def rec_perm(n):
        q = orig_perm(n)
        perms.append(np.asarray(q, np.int64))
        return q

    seqs: list[np.ndarray] = []
    orig_from = SequenceSummaryStats.from_sequence.__func__

    def rec_from(c_, seq):
        seqs.append(np.asarray(seq, np.float64))
        return orig_from(c_, seq)

    pre_dump: dict[str, np.ndarray] = {}
    orig_pre = cls._preprocess_batch

    def rec_pre(self, batch, buffer, indices):
        b = orig_pre(self, batch, buffer, indices)
        pre_dump["v_s"] = b.v_s.numpy().copy()
        pre_dump["returns"] = b.returns.numpy().copy()
        pre_dump["adv"] = b.adv.numpy().copy()
        pre_dump["logp_old"] = b.logp_old.numpy().copy() if algo == "ppo" else np.zeros(len(indices), np.float32)
        pre_dump["indices"] = np.asarray(indices, np.int64)
        pre_dump["unfinished"] = np.asarray(buffer.unfinished_index(), np.int64)
        return b

    np.random.permutation = rec_perm
    SequenceSummaryStats.from_sequence = classmethod(rec_from)
    cls._preprocess_batch = rec_pre
    try:
        buf = VectorReplayBuffer(N, E)
        obs = rng.normal(size=(T + 1, E, obs_dim)).astype(np.float32)
        act = rng.normal(size=(T, E, act_dim)).astype(np.float32)
        rew = rng.normal(size=(T, E)).astype(np.float32)
        term = rng.random((T, E)) < 0.03
        trunc = np.zeros((T, E), bool)
        for t in range(T):
            buf.add(Batch(obs=obs[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t], obs_next=obs[t + 1]))
        for k in ("obs", "obs_next", "act"):
            out[k] = np.asarray(getattr(buf, k), np.float32)
        out["rew"] = np.asarray(buf.rew, np.float64)
        out["terminated"], out["truncated"] = np.asarray(buf.terminated, bool), np.asarray(buf.truncated, bool)
        np.random.seed(seed + 100)
        with policy_within_training_step(algorithm.policy):
            stats = algorithm.update(buffer=buf, batch_size=batch_size, repeat=repeat)
        assert len(perms) == repeat and len(seqs) == 4
        out["perms"] = np.stack(perms)
        out["losses"] = np.stack(seqs, axis=1)              # loss, clip / actor loss, vf loss, entropy (ppo.py:218-222)
        out["gradient_steps"] = np.array(stats.gradient_steps)
    finally:
        np.random.permutation = orig_perm
        SequenceSummaryStats.from_sequence = classmethod(orig_from)
        cls._preprocess_batch = orig_pre
    opt = algorithm.optim._optim
    rms = type(opt).__name__ == "RMSprop"
    g0 = opt.param_groups[0]
    k_m = ("momentum_buffer" if g0.get("momentum", 0) > 0 else ("grad_avg" if g0.get("centered") else None)) if rms else "exp_avg"
    k_v = "square_avg" if rms else "exp_avg_sq"
    for i, t in enumerate(a_par):
        out[f"a{i}_1"] = t.detach().numpy().copy()
        out[f"a{i}_m"] = opt.state[t][k_m].numpy().copy() if k_m else np.zeros_like(out[f"a{i}_1"])
        out[f"a{i}_v"] = opt.state[t][k_v].numpy().copy()
    for i, t in enumerate(c_par):
        out[f"c{i}_1"] = t.detach().numpy().copy()
        out[f"c{i}_m"] = opt.state[t][k_m].numpy().copy() if k_m else np.zeros_like(out[f"c{i}_1"])
        out[f"c{i}_v"] = opt.state[t][k_v].numpy().copy()
    for k, v in pre_dump.items():
        out["pre_" + k] = v
    cfg = dict(gamma=algorithm.gamma, gae_lambda=algorithm.gae_lambda, eps_clip=getattr(algorithm, "eps_clip", 0.0),
               dual_clip=getattr(algorithm, "dual_clip", None) or 0.0, value_clip=float(getattr(algorithm, "value_clip", False)),
               advantage_normalization=float(getattr(algorithm, "advantage_normalization", False)), vf_coef=algorithm.vf_coef,
               ent_coef=algorithm.ent_coef, max_grad_norm=algorithm.optim._max_grad_norm or 0.0,
               return_scaling=float(algorithm.return_scaling), lr=lr, is_a2c=float(algo == "a2c"),
               opt_rmsprop=float(rms), weight_decay=float(g0.get("weight_decay", 0.0)), opt_eps=float(g0["eps"]),
               rms_alpha=float(g0.get("alpha", 0.99)), rms_momentum=float(g0.get("momentum", 0.0)),
               rms_centered=float(g0.get("centered", False)))
    out["cfg_keys"] = np.array(list(cfg.keys()))
    out["cfg_vals"] = np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"ppo_net_{tag}.npz"), **out)


def gen_ppo_net_all() -> None:
    # a three-layer ReLU trunk with unequal widths (none a multiple of 32) and different actor / critic trunks
    gen_ppo_net("relu3", E=4, T=50, obs_dim=11, act_dim=3, hidden_a=[96, 72, 40], hidden_c=[64, 48], activation=nn.ReLU,
                batch_size=64, repeat=2, seed=11, eps_clip=0.2, vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5, value_clip=True,
                advantage_normalization=True, return_scaling=False, gae_lambda=0.95, gamma=0.99)
    # one hidden layer, tanh; A2C
    gen_ppo_net("tanh1_a2c", algo="a2c", E=3, T=40, obs_dim=5, act_dim=2, hidden_a=[48], hidden_c=[48], activation=nn.Tanh,
                batch_size=60, repeat=1, seed=12, vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5, return_scaling=False,
                gae_lambda=0.9, gamma=0.99)
    # conditioned sigma (a second linear head, clamped to [-20, 2] before exp), two tanh layers
    gen_ppo_net("csigma", E=4, T=48, obs_dim=9, act_dim=4, hidden_a=[64, 64], hidden_c=[64, 64], activation=nn.Tanh,
                conditioned_sigma=True, batch_size=64, repeat=2, seed=14, eps_clip=0.2, vf_coef=0.5, ent_coef=0.02,
                max_grad_norm=0.5, value_clip=True, advantage_normalization=True, return_scaling=False, gae_lambda=0.95, gamma=0.99)
    # no activation (a linear trunk), four layers
    gen_ppo_net("linear4", E=2, T=64, obs_dim=20, act_dim=5, hidden_a=[32, 32, 32, 32], hidden_c=[33, 17, 9, 5], activation=None,
                batch_size=128, repeat=1, seed=13, eps_clip=0.1, dual_clip=2.0, vf_coef=0.25, ent_coef=0.0, max_grad_norm=None,
                value_clip=False, advantage_normalization=False, return_scaling=True, gae_lambda=0.95, gamma=0.99)


def gen_ppo_round6() -> None:
    """Round 6: the reference's DEFAULT Gaussian actor (unbounded=False: mu = max_action * tanh(.), continuous.py:194,
    230-231) and the other optimizer factories of tianshou/algorithm/optim.py (RMSprop: the optimizer of
    examples/mujoco/mujoco_a2c.py:117; Adam with weight decay, optim.py:95-109)."""
    # PPO, MuJoCo nets, bounded actor with the default max_action = 1
    gen_ppo("bounded", E=6, T=64, obs_dim=17, act_dim=6, batch_size=96, repeat=2, seed=21, n_updates=2, max_action=1.0,
            gamma=0.99, gae_lambda=0.95, max_grad_norm=0.5, vf_coef=0.25, ent_coef=0.01, return_scaling=True, eps_clip=0.2,
            value_clip=True, dual_clip=None, advantage_normalization=True, recompute_advantage=False, max_batchsize=256)
    # A2C exactly as examples/mujoco/mujoco_a2c.py:117-121 builds its optimizer: RMSprop(lr=7e-4, eps=1e-5, alpha=0.99)
    gen_ppo("a2c_rmsprop", algo="a2c", E=4, T=60, obs_dim=17, act_dim=6, batch_size=64, repeat=2, seed=22, n_updates=2,
            optim=("rmsprop", dict(eps=1e-5, alpha=0.99)), vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5, gae_lambda=0.95,
            gamma=0.99, return_scaling=True, lr=7e-4, max_batchsize=256)
    # PPO with Adam(weight_decay) and a bounded actor with max_action = 2
    gen_ppo("adam_wd", E=4, T=48, obs_dim=17, act_dim=6, batch_size=64, repeat=2, seed=23, n_updates=2, max_action=2.0,
            optim=("adam", dict(weight_decay=1e-2)), gamma=0.99, gae_lambda=0.95, max_grad_norm=0.5, vf_coef=0.25, ent_coef=0.0,
            return_scaling=False, eps_clip=0.2, value_clip=False, dual_clip=None, advantage_normalization=True,
            recompute_advantage=False, max_batchsize=256, lr=1e-3)
    # RMSprop with momentum and weight decay; centered RMSprop (two short A2C runs)
    gen_ppo("rms_momentum", algo="a2c", E=3, T=40, obs_dim=17, act_dim=6, batch_size=60, repeat=1, seed=24, n_updates=2,
            optim=("rmsprop", dict(eps=1e-5, alpha=0.95, momentum=0.9, weight_decay=1e-3)), vf_coef=0.5, ent_coef=0.0,
            max_grad_norm=None, gae_lambda=0.95, gamma=0.99, return_scaling=False, lr=5e-4, max_batchsize=256)
    gen_ppo("rms_centered", algo="a2c", E=3, T=40, obs_dim=17, act_dim=6, batch_size=60, repeat=1, seed=25, n_updates=2,
            optim=("rmsprop", dict(eps=1e-5, alpha=0.9, centered=True)), vf_coef=0.5, ent_coef=0.0,
            max_grad_norm=0.5, gae_lambda=0.95, gamma=0.99, return_scaling=False, lr=5e-4, max_batchsize=256)
    # per-layer engine: bounded actor over a three-layer ReLU trunk (max_action 1.5), RMSprop
    gen_ppo_net("bounded_relu3", E=4, T=50, obs_dim=11, act_dim=3, hidden_a=[96, 72, 40], hidden_c=[64, 48], activation=nn.ReLU,
                max_action=1.5, optim=("rmsprop", dict(eps=1e-5, alpha=0.99)), batch_size=64, repeat=2, seed=26, eps_clip=0.2,
                vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5, value_clip=True, advantage_normalization=True, return_scaling=False,
                gae_lambda=0.95, gamma=0.99)
    # per-layer engine: bounded actor with conditioned sigma, Net[128, 128] tanh (the "wide" shape), Adam + weight decay
    gen_ppo_net("bounded_cs", E=4, T=48, obs_dim=9, act_dim=4, hidden_a=[128, 128], hidden_c=[128, 128], activation=nn.Tanh,
                conditioned_sigma=True, max_action=1.0, optim=("adam", dict(weight_decay=5e-3)), batch_size=64, repeat=2, seed=27,
                eps_clip=0.2, vf_coef=0.5, ent_coef=0.02, max_grad_norm=0.5, value_clip=True, advantage_normalization=True,
                return_scaling=False, gae_lambda=0.95, gamma=0.99)


def gen_ppo_layernorm() -> None:
    """Round 6: MLP(norm_layer=nn.LayerNorm) trunks (utils/net/common.py:25-39, 99-137) under PPO / A2C on the per-layer engine."""
    # three ReLU layers with LayerNorm, unequal widths (none a multiple of 32), different actor / critic trunks, PPO
    gen_ppo_net("ln_relu3", E=4, T=50, obs_dim=11, act_dim=3, hidden_a=[96, 72, 40], hidden_c=[64, 48], activation=nn.ReLU,
                layer_norm=True, batch_size=64, repeat=2, seed=31, eps_clip=0.2, vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5,
                value_clip=True, advantage_normalization=True, return_scaling=False, gae_lambda=0.95, gamma=0.99)
    # one tanh layer with LayerNorm(eps=1e-3), the bounded actor (the class default), A2C with RMSprop + weight decay
    gen_ppo_net("ln_tanh1_a2c", algo="a2c", E=3, T=40, obs_dim=5, act_dim=2, hidden_a=[48], hidden_c=[80], activation=nn.Tanh,
                layer_norm=True, norm_args=dict(eps=1e-3), max_action=2.0, optim=("rmsprop", dict(eps=1e-5, alpha=0.99, weight_decay=1e-3)),
                batch_size=60, repeat=1, seed=32, vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5, return_scaling=False,
                gae_lambda=0.9, gamma=0.99)


def gen_policy_forward() -> None:
    """what the Collector computes once per vector step (data/collector.py:735-744) -- `policy(batch)` =
    Policy.forward, then `policy.map_action(act)` (algorithm_base.py:254-287) -- run through the UNMODIFIED reference for the
    three policy families the engine overrides:
      * `ProbabilisticActorPolicy` over ContinuousActorProbabilistic (reinforce.py:167-192): every action_bound_method
        (clip / tanh / None) x action_scaling into an asymmetric Box, the unbounded actor and the reference's default bounded
        one (max_action * tanh), the MuJoCo nets and a three-layer ReLU trunk, sampling (within a training step) and
        dist.mode (deterministic_eval outside of one);
      * `SACPolicy` (sac.py:108-131): rsample, tanh squashing, corrected log-probability, scaling;
      * `DiscreteQLearningPolicy` over DQNet (dqn.py:101-143): logits, greedy action, with and without an action mask.
    dist.sample() / rsample() consume torch's CPU generator through N(0, 1) draws that are scaled and shifted
    (torch.normal(mean, std) / _standard_normal): the draws are recovered by re-running `normal_()` from the saved
    generator state and stored as `noise` (asserted to reproduce the reference's action)."""
    from tianshou.algorithm.modelfree.dqn import DiscreteQLearningPolicy
    from tianshou.algorithm.modelfree.sac import SACPolicy
    from tianshou.env.atari.atari_network import DQNet

    out: dict[str, np.ndarray] = {}

    def dist(loc_scale):
        loc, scale = loc_scale
        return Independent(Normal(loc, scale), 1)

    def run_gauss(tag, *, obs_dim, act_dim, hidden, activation, max_action, bound, scaling, n, seed, training, det_eval):
        rng = np.random.default_rng(seed)
        torch.manual_seed(seed)
        net = Net(state_shape=(obs_dim,), hidden_sizes=hidden, activation=activation)
        if max_action is None:
            actor = ContinuousActorProbabilistic(preprocess_net=net, action_shape=(act_dim,), unbounded=True)
        else:
            actor = ContinuousActorProbabilistic(preprocess_net=net, action_shape=(act_dim,), max_action=max_action)
        torch.nn.init.normal_(actor.sigma_param, mean=-0.7, std=0.3)
        low, high = np.linspace(-2.0, -0.5, act_dim).astype(np.float32), np.linspace(0.4, 3.0, act_dim).astype(np.float32)
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            policy = ProbabilisticActorPolicy(actor=actor, dist_fn=dist, action_scaling=scaling, action_bound_method=bound,
                                              action_space=gym.spaces.Box(low=low, high=high, shape=(act_dim,)),
                                              deterministic_eval=det_eval)
        obs = (rng.normal(size=(n, obs_dim)) * 1.5).astype(np.float32)
        st = torch.get_rng_state()
        ctx = policy_within_training_step(policy) if training else contextlib.nullcontext()
        with ctx, torch.no_grad():
            res = policy(Batch(obs=obs, info={}), None)
        mu, sigma = (t.numpy().copy() for t in res.logits)
        act = res.act.numpy().copy()
        sampled = training or not det_eval
        if sampled:
            torch.set_rng_state(st)
            noise = torch.empty(n, act_dim).normal_().numpy()
            assert np.allclose(mu + sigma * noise, act, rtol=0, atol=1e-6), "noise stream of dist.sample() not reproduced"
        else:
            noise = np.zeros((0, act_dim), np.float32)
            assert np.array_equal(act, mu)
        mapped = policy.map_action(act.copy())
        sa = actor.state_dict()
        keys = [k for k in sa if k != "sigma_param"] + ["sigma_param"]
        for i, k in enumerate(keys):
            out[f"{tag}_p{i}"] = sa[k].numpy().copy()
        out[f"{tag}_keys"] = np.array(keys)
        out[f"{tag}_obs"], out[f"{tag}_noise"], out[f"{tag}_mu"], out[f"{tag}_sigma"] = obs, noise, mu, sigma
        out[f"{tag}_act"], out[f"{tag}_mapped"] = act, np.asarray(mapped, np.float32)
        out[f"{tag}_low"], out[f"{tag}_high"] = low, high
        out[f"{tag}_cfg"] = np.array([obs_dim, act_dim, {None: 0, "clip": 1, "tanh": 2}[bound], int(scaling), max_action or 0.0,
                                      {nn.Tanh: 0, nn.ReLU: 1, None: 2}[activation], seed, int(training), int(det_eval), int(sampled)],
                                     np.float64)
        out[f"{tag}_hidden"] = np.array(hidden)

    import contextlib

    mj = dict(obs_dim=17, act_dim=6, hidden=[64, 64], activation=nn.Tanh, n=96)
    run_gauss("g_clip", max_action=None, bound="clip", scaling=True, seed=31, training=True, det_eval=False, **mj)
    run_gauss("g_tanh", max_action=None, bound="tanh", scaling=True, seed=32, training=True, det_eval=True, **mj)
    run_gauss("g_none", max_action=None, bound=None, scaling=False, seed=33, training=False, det_eval=True, **mj)      # dist.mode
    run_gauss("g_bounded", max_action=1.0, bound="clip", scaling=True, seed=34, training=True, det_eval=False, **mj)  # default actor
    run_gauss("g_bounded2", max_action=2.5, bound=None, scaling=False, seed=35, training=False, det_eval=False, **mj)
    run_gauss("g_net", obs_dim=11, act_dim=3, hidden=[96, 72, 40], activation=nn.ReLU, n=70, max_action=1.5, bound="tanh",
              scaling=True, seed=36, training=True, det_eval=False)
    run_gauss("g_wide", obs_dim=40, act_dim=12, hidden=[128, 128], activation=nn.Tanh, n=65, max_action=None, bound="clip",
              scaling=True, seed=37, training=True, det_eval=False)

    # ---- SACPolicy (nets of examples/mujoco/mujoco_sac.py:82-104)
    for tag, obs_dim, act_dim, hid, n, seed, training in (("s_train", 23, 5, 256, 80, 41, True), ("s_eval", 23, 5, 256, 33, 42, False),
                                                         ("s_h128", 376, 17, 128, 48, 43, True)):
        rng = np.random.default_rng(seed)
        torch.manual_seed(seed)
        actor = ContinuousActorProbabilistic(preprocess_net=Net(state_shape=(obs_dim,), hidden_sizes=[hid, hid]),
                                             action_shape=(act_dim,), unbounded=True, conditioned_sigma=True)
        low, high = np.linspace(-3.0, -1.0, act_dim).astype(np.float32), np.linspace(0.5, 2.0, act_dim).astype(np.float32)
        policy = SACPolicy(actor=actor, action_space=gym.spaces.Box(low=low, high=high, shape=(act_dim,)))
        obs = rng.normal(size=(n, obs_dim)).astype(np.float32)
        st = torch.get_rng_state()
        ctx = policy_within_training_step(policy) if training else contextlib.nullcontext()
        with ctx, torch.no_grad():
            res = policy(Batch(obs=obs, info={}), None)
        mu, sigma = (t.numpy().copy() for t in res.logits)
        act = res.act.numpy().copy()
        if training:
            torch.set_rng_state(st)
            noise = torch.empty(n, act_dim).normal_().numpy()
            assert np.allclose(np.tanh(mu + sigma * noise), act, rtol=0, atol=1e-6), "noise stream of rsample() not reproduced"
        else:
            noise = np.zeros((0, act_dim), np.float32)
        sa = actor.state_dict()
        for i, (k, v) in enumerate(sa.items()):
            out[f"{tag}_p{i}"] = v.numpy().copy()
        out[f"{tag}_obs"], out[f"{tag}_noise"], out[f"{tag}_mu"], out[f"{tag}_sigma"] = obs, noise, mu, sigma
        out[f"{tag}_act"], out[f"{tag}_logp"] = act, res.log_prob.numpy().copy()
        out[f"{tag}_mapped"] = np.asarray(policy.map_action(act.copy()), np.float32)
        out[f"{tag}_low"], out[f"{tag}_high"] = low, high
        out[f"{tag}_cfg"] = np.array([obs_dim, act_dim, hid, seed, int(training)], np.float64)

    # ---- DiscreteQLearningPolicy over DQNet (uint8 frames as the Atari wrappers deliver them: [n, c, h, w])
    for tag, c, h, w, n_act, n, seed, masked in (("q_plain", 4, 44, 44, 6, 24, 51, False), ("q_mask", 2, 44, 36, 5, 17, 52, True)):
        rng = np.random.default_rng(seed)
        torch.manual_seed(seed)
        model = DQNet(c=c, h=h, w=w, action_shape=n_act)
        policy = DiscreteQLearningPolicy(model=model, action_space=gym.spaces.Discrete(n_act))
        frames = rng.integers(0, 256, size=(n, c, h, w), dtype=np.uint8)
        frames = np.where(rng.random(frames.shape) < 0.2, frames, 0).astype(np.uint8)
        if masked:
            mask = rng.random((n, n_act)) < 0.6
            mask[np.arange(n), rng.integers(0, n_act, n)] = True
            b = Batch(obs=Batch(obs=frames, mask=mask), info={})
            out[f"{tag}_mask"] = mask
        else:
            b = Batch(obs=frames, info={})
        with torch.no_grad():
            res = policy(b, None)
        for i, (k, v) in enumerate(model.state_dict().items()):
            out[f"{tag}_p{i}"] = v.numpy().copy()
        out[f"{tag}_obs"] = frames
        out[f"{tag}_logits"] = res.logits.numpy().copy()
        out[f"{tag}_act"] = np.asarray(res.act, np.int64)
        out[f"{tag}_cfg"] = np.array([c, h, w, n_act, seed, int(masked)], np.float64)
    np.savez_compressed(os.path.join(OUT, "policy_forward.npz"), **out)


def gen_ppo_sched() -> None:
    """The mujoco_ppo.py configuration WITH its default linear learning-rate decay (lr_decay=True,
    examples/mujoco/mujoco_ppo.py:48,124-131): 3 epochs x 2 collects -> max_update_num 6, so four updates run at
    lr x (1, 5/6, 4/6, 3/6)."""
    gen_ppo("sched", E=4, T=48, obs_dim=17, act_dim=6, batch_size=64, repeat=2, seed=5, n_updates=4,
            lr_decay=(3, 384, 192), gamma=0.99, gae_lambda=0.95, max_grad_norm=0.5, vf_coef=0.25, ent_coef=0.0,
            return_scaling=True, eps_clip=0.2, value_clip=True, dual_clip=None, advantage_normalization=False,
            recompute_advantage=False, max_batchsize=256)


def gen_sample_random() -> None:
    """ReplayBufferManager.sample_indices(batch_size > 0) (manager.py:216-234) on unevenly filled / wrapped
    VectorReplayBuffers, with the draws the reference consumed recovered from copies of its RandomStates (the manager's
    for RandomState.choice(E, bs, p), every sub-buffer's own for choice(len, n)) so that the engine can replay them."""
    import copy

    rng = np.random.default_rng(11)
    out: dict[str, np.ndarray] = {}
    cases = [(60, 4, 10), (64, 8, 37), (4096, 512, 4096), (30, 3, 1)]
    for c, (size, E, bs) in enumerate(cases):
        buf = VectorReplayBuffer(size, E)
        steps = int(rng.integers(3, 3 * size // E))
        for t in range(steps):
            ids = np.flatnonzero(rng.random(E) < 0.7)
            if t == 0 or ids.size == 0:
                ids = np.arange(E)
            k = ids.size
            buf.add(Batch(obs=rng.normal(size=(k, 3)), act=rng.normal(size=(k, 1)), rew=rng.normal(size=k),
                          terminated=rng.random(k) < 0.1, truncated=np.zeros(k, bool), obs_next=rng.normal(size=(k, 3))),
                    buffer_ids=ids)
        for rep in range(2):                                  # consecutive calls advance the generators
            st = copy.deepcopy(buf._random_state.get_state())
            child = [copy.deepcopy(b._random_state.get_state()) for b in buf.buffers]
            res = buf.sample_indices(bs)
            rs = np.random.RandomState()
            rs.set_state(st)
            u = rs.random_sample(bs)
            L = np.asarray(buf._lengths)
            p = L / L.sum()
            cdf = p.cumsum()
            cdf /= cdf[-1]
            cnt = np.bincount(cdf.searchsorted(u, side="right"), minlength=E)
            within = []
            for e in range(E):
                if cnt[e]:
                    r = np.random.RandomState()
                    r.set_state(child[e])
                    within.append(r.choice(int(L[e]), int(cnt[e])))
            within = np.concatenate(within)
            key = f"c{c}_r{rep}_"
            out[key + "offset"] = np.asarray(buf._extend_offset, np.int64)
            out[key + "lengths"] = L.astype(np.int64)
            out[key + "u"], out[key + "within"] = u, within.astype(np.int64)
            out[key + "result"] = np.asarray(res, np.int64)
    out["n_cases"] = np.array([len(cases), 2])
    np.savez_compressed(os.path.join(OUT, "sample_random.npz"), **out)


def gen_sample_stack() -> None:
    """ReplayBufferManager.sample_indices with frame stacking (manager.py:205-216 -> buffer_base.py:518-545,
    `stack_num > 1 and sample_avail`): only indices with stack_num - 1 predecessors in their episode are available.
    batch_size = 0 (all of them) and batch_size > 0 (`RandomState.choice(all_indices, bs)`; the positions the reference
    drew are recovered from the result, the available indices being distinct)."""
    rng = np.random.default_rng(23)
    out: dict[str, np.ndarray] = {}
    cases = [(40, 4, 2, 9), (48, 3, 4, 16), (4096, 64, 4, 256), (24, 2, 3, 5)]
    for c, (size, E, stack, bs) in enumerate(cases):
        buf = VectorReplayBuffer(size, E, stack_num=stack, sample_avail=True)
        steps = int(rng.integers(size // E // 2, 3 * size // E))
        for t in range(steps):
            ids = np.flatnonzero(rng.random(E) < 0.8)
            if t == 0 or ids.size == 0:
                ids = np.arange(E)
            k = ids.size
            buf.add(Batch(obs=rng.normal(size=(k, 3)), act=rng.normal(size=(k, 1)), rew=rng.normal(size=k),
                          terminated=rng.random(k) < 0.15, truncated=np.zeros(k, bool), obs_next=rng.normal(size=(k, 3))),
                    buffer_ids=ids)
        all_idx = np.asarray(buf.sample_indices(0), np.int64)
        res = np.asarray(buf.sample_indices(bs), np.int64)
        pos = np.searchsorted(np.sort(all_idx), res)
        order = np.argsort(all_idx, kind="stable")
        positions = order[pos].astype(np.int64)
        assert np.array_equal(all_idx[positions], res)
        key = f"c{c}_"
        out[key + "offset"] = np.asarray(buf._extend_offset, np.int64)
        out[key + "lengths"] = np.asarray(buf._lengths, np.int64)
        out[key + "insertion"] = np.asarray([b._insertion_idx for b in buf.buffers], np.int64)
        out[key + "last_index"] = np.asarray(buf.last_index, np.int64)
        out[key + "done"] = np.asarray(buf.done, np.uint8)
        out[key + "stack"] = np.array([stack], np.int64)
        out[key + "all"], out[key + "positions"], out[key + "result"] = all_idx, positions, res
    out["n_cases"] = np.array([len(cases)])
    np.savez_compressed(os.path.join(OUT, "sample_stack.npz"), **out)


def gen_npg(tag: str, *, algo: str, E: int, T: int, obs_dim: int, act_dim: int, batch_size: int, repeat: int, seed: int,
            lr: float = 1e-3, hidden_a=(64, 64), hidden_c=(64, 64), activation=nn.Tanh, **kwargs) -> None:
    """Runs the reference NPG.update() / TRPO.update() on the MuJoCo actor-critic (examples/mujoco/mujoco_npg.py:103-128,
    the PPO nets) over a synthetic VectorReplayBuffer and dumps every intermediate.  Round 6: `hidden_a` / `hidden_c` of any
    length and `activation` (nn.Tanh, nn.ReLU or None) = any Net trunk (utils/net/common.py:90-178)."""
    from tianshou.algorithm.modelfree.npg import NPG
    from tianshou.algorithm.modelfree.trpo import TRPO

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    N = E * T
    net_a = Net(state_shape=(obs_dim,), hidden_sizes=list(hidden_a), activation=activation)
    actor = ContinuousActorProbabilistic(preprocess_net=net_a, action_shape=(act_dim,), unbounded=True)
    net_c = Net(state_shape=(obs_dim,), hidden_sizes=list(hidden_c), activation=activation)
    critic = ContinuousCritic(preprocess_net=net_c)
    torch.nn.init.constant_(actor.sigma_param, -0.5)
    for m in ActorCritic(actor, critic).modules():
        if isinstance(m, nn.Linear):
            nn.init.orthogonal_(m.weight, gain=np.sqrt(2))
            nn.init.zeros_(m.bias)
    for m in actor.mu.modules():
        if isinstance(m, nn.Linear):
            nn.init.zeros_(m.bias)
            m.weight.data.copy_(0.01 * m.weight.data)

    def dist(loc_scale):
        loc, scale = loc_scale
        return Independent(Normal(loc, scale), 1)

    space = gym.spaces.Box(low=-1.0, high=1.0, shape=(act_dim,))
    policy = ProbabilisticActorPolicy(actor=actor, dist_fn=dist, action_scaling=True, action_bound_method="clip",
                                      action_space=space)
    cls = NPG if algo == "npg" else TRPO
    algorithm = cls(policy=policy, critic=critic, optim=AdamOptimizerFactory(lr=lr), **kwargs)
    assert [n for n, _ in actor.named_parameters()][0] == "sigma_param"
    out: dict[str, np.ndarray] = {"flat_params0": _flat_from_modules(actor, critic),
                                  "dims": np.array([E, T, obs_dim, act_dim, batch_size, repeat, int(algo == "trpo")])}
    if len(hidden_a) != 2 or len(hidden_c) != 2 or activation is not nn.Tanh:
        out["hidden_a"], out["hidden_c"] = np.array(list(hidden_a), np.int64), np.array(list(hidden_c), np.int64)
        out["activation"] = np.array({nn.Tanh: 0, nn.ReLU: 1, None: 2}[activation])
        out["seed"] = np.array(seed)
    elif tuple(hidden_a) != (64, 64) or tuple(hidden_c) != (64, 64):
        out["hidden"] = np.array(list(hidden_a) + list(hidden_c), np.int64)
        out["seed"] = np.array(seed)
    buf = VectorReplayBuffer(N, E)
    obs = rng.normal(size=(T + 1, E, obs_dim)).astype(np.float32)
    act = rng.normal(size=(T, E, act_dim)).astype(np.float32) * 0.7
    rew = rng.normal(size=(T, E)).astype(np.float32)
    term = rng.random((T, E)) < 0.02
    trunc = np.zeros((T, E), bool)
    trunc[T // 2 - 1:: T // 2] = True
    trunc &= ~term
    for t in range(T):
        buf.add(Batch(obs=obs[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t], obs_next=obs[t + 1]))
    out["obs"], out["obs_next"] = np.asarray(buf.obs, np.float32), np.asarray(buf.obs_next, np.float32)
    out["act"], out["rew"] = np.asarray(buf.act, np.float32), np.asarray(buf.rew, np.float64)
    out["terminated"], out["truncated"] = np.asarray(buf.terminated, bool), np.asarray(buf.truncated, bool)
    for k, v in manager_state(buf).items():
        out["buf_" + k] = v

    perms, seqs, pre_dump = [], [], {}
    orig_perm, orig_from, orig_pre = np.random.permutation, SequenceSummaryStats.from_sequence.__func__, cls._preprocess_batch

    def rec_perm(n):
        p = orig_perm(n)
        perms.append(np.asarray(p, np.int64))
        return p

    def rec_from(c_, seq):
        seqs.append(np.asarray(seq, np.float64))
        return orig_from(c_, seq)

    def rec_pre(self, batch, buffer, indices):
        b = orig_pre(self, batch, buffer, indices)
        pre_dump.update(v_s=b.v_s.numpy().copy(), returns=b.returns.numpy().copy(), adv=b.adv.numpy().copy(),
                        logp_old=b.logp_old.numpy().copy(), indices=np.asarray(indices, np.int64),
                        unfinished=np.asarray(buffer.unfinished_index(), np.int64))
        return b

    np.random.permutation, cls._preprocess_batch = rec_perm, rec_pre
    SequenceSummaryStats.from_sequence = classmethod(rec_from)
    try:
        np.random.seed(seed + 100)
        with policy_within_training_step(algorithm.policy):
            algorithm.update(buffer=buf, batch_size=batch_size, repeat=repeat)
    finally:
        np.random.permutation, cls._preprocess_batch = orig_perm, orig_pre
        SequenceSummaryStats.from_sequence = classmethod(orig_from)
    assert len(perms) == repeat and len(seqs) == (3 if algo == "npg" else 4)
    # npg.py:189-193 order: actor_loss, vf_loss, kl; trpo.py:204-207: actor_loss, vf_loss, kl, step_size
    out["perms"], out["stats"] = np.stack(perms), np.stack(seqs, axis=1)
    out["flat_params"] = _flat_from_modules(actor, critic)
    for k, v in pre_dump.items():
        out["pre_" + k] = v
    cfg = dict(gamma=algorithm.gamma, gae_lambda=algorithm.gae_lambda, optim_critic_iters=algorithm.optim_critic_iters,
               trust_region_size=getattr(algorithm, "trust_region_size", 0.5),
               advantage_normalization=float(algorithm.advantage_normalization),
               return_scaling=float(algorithm.return_scaling), max_batchsize=float(algorithm.max_batchsize),
               damping=algorithm._damping, max_kl=getattr(algorithm, "max_kl", 0.01),
               backtrack_coeff=getattr(algorithm, "backtrack_coeff", 0.8), max_backtracks=getattr(algorithm, "max_backtracks", 10),
               lr=lr)
    out["cfg_keys"], out["cfg_vals"] = np.array(list(cfg.keys())), np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"npg_{tag}.npz"), **out)


def gen_npg_all() -> None:
    gen_npg("npg", algo="npg", E=4, T=64, obs_dim=17, act_dim=6, batch_size=128, repeat=2, seed=41, optim_critic_iters=3,
            trust_region_size=0.1, advantage_normalization=True, gae_lambda=0.95, gamma=0.99, return_scaling=True,
            max_batchsize=64)
    gen_npg("trpo", algo="trpo", E=4, T=64, obs_dim=17, act_dim=6, batch_size=128, repeat=2, seed=43, optim_critic_iters=2,
            max_kl=0.01, backtrack_coeff=0.8, max_backtracks=10, advantage_normalization=True, gae_lambda=0.95, gamma=0.99,
            return_scaling=False, max_batchsize=256)


def gen_reinforce(tag: str, *, E: int, T: int, obs_dim: int, act_dim: int, batch_size: int, repeat: int, seed: int,
                  n_updates: int, lr: float = 1e-3, hidden=(64, 64), activation=nn.Tanh, max_action: float | None = None,
                  optim: tuple[str, dict] | None = None, norm_args: dict | None = None, layer_norm: bool = False, **kwargs) -> None:
    """Runs the reference Reinforce.update() (actor of examples/mujoco/mujoco_reinforce.py:84-103) on synthetic rollouts.
    Round 6: `hidden` / `activation` = any Net trunk (utils/net/common.py:90-178), `max_action` = the bounded (default) actor,
    `optim` = ("rmsprop" | "adam", factory kwargs) -- the fixtures of the per-layer engine path (keys `a{i}_0 / a{i}_{u}`)."""
    from tianshou.algorithm.modelfree.reinforce import Reinforce

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    N = E * T
    hidden = list(hidden)
    generic = hidden != [64, 64] or activation is not nn.Tanh or max_action is not None or optim is not None or layer_norm
    nkw = dict(norm_layer=nn.LayerNorm, norm_args=norm_args) if layer_norm else {}        # common.py:25-39 (round 6)
    net_a = Net(state_shape=(obs_dim,), hidden_sizes=hidden, activation=activation, **nkw)
    if max_action is None:
        actor = ContinuousActorProbabilistic(preprocess_net=net_a, action_shape=(act_dim,), unbounded=True)
    else:
        actor = ContinuousActorProbabilistic(preprocess_net=net_a, action_shape=(act_dim,), max_action=max_action)
    torch.nn.init.constant_(actor.sigma_param, -0.5)
    for m in actor.modules():
        if isinstance(m, nn.Linear):
            nn.init.orthogonal_(m.weight, gain=np.sqrt(2))
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.uniform_(m.weight, 0.5, 1.5)
            nn.init.normal_(m.bias, std=0.2)
    for m in actor.mu.modules():
        if isinstance(m, nn.Linear):
            m.weight.data.copy_(0.3 * m.weight.data)

    def dist(loc_scale):
        loc, scale = loc_scale
        return Independent(Normal(loc, scale), 1)

    policy = ProbabilisticActorPolicy(actor=actor, dist_fn=dist, action_scaling=True, action_bound_method="tanh",
                                      action_space=gym.spaces.Box(low=-1.0, high=1.0, shape=(act_dim,)))
    if optim is not None and optim[0] == "rmsprop":
        from tianshou.algorithm.optim import RMSpropOptimizerFactory

        optim_factory = RMSpropOptimizerFactory(lr=lr, **optim[1])
    else:
        optim_factory = AdamOptimizerFactory(lr=lr, **(optim[1] if optim else {}))
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        algorithm = Reinforce(policy=policy, optim=optim_factory, **kwargs)
    keys = [k for k in actor.state_dict() if k != "sigma_param"] + ["sigma_param"]
    if not generic:
        assert keys == ["preprocess.model.model.0.weight", "preprocess.model.model.0.bias", "preprocess.model.model.2.weight",
                        "preprocess.model.model.2.bias", "mu.model.0.weight", "mu.model.0.bias", "sigma_param"]
    flat = lambda: torch.cat([actor.state_dict()[k].reshape(-1) for k in keys]).numpy().copy()  # noqa: E731
    out: dict[str, np.ndarray] = {"actor0": flat(), "dims": np.array([E, T, obs_dim, act_dim, batch_size or 0, repeat, n_updates])}
    if generic:
        out["hidden"], out["activation"] = np.array(hidden), np.array({nn.Tanh: 0, nn.ReLU: 1, None: 2}[activation])
        out["max_action"], out["keys"] = np.array(float(max_action or 0.0)), np.array(keys)
        if layer_norm:
            out["ln_eps"] = np.array(float((norm_args or {}).get("eps", 1e-5)))
        for i, k in enumerate(keys):
            out[f"a{i}_0"] = actor.state_dict()[k].numpy().copy()
    perms, seqs, rets = [], [], []
    orig_perm, orig_from, orig_pre = np.random.permutation, SequenceSummaryStats.from_sequence.__func__, Reinforce._preprocess_batch

    def rec_perm(n):
        p = orig_perm(n)
        perms.append(np.asarray(p, np.int64))
        return p

    def rec_from(c_, seq):
        seqs.append(np.asarray(seq, np.float64))
        return orig_from(c_, seq)

    def rec_pre(self, batch, buffer, indices):
        b = orig_pre(self, batch, buffer, indices)
        rets.append((np.asarray(b.returns, np.float64).copy(), np.asarray(indices, np.int64),
                     np.asarray(buffer.unfinished_index(), np.int64)))
        return b

    np.random.permutation, Reinforce._preprocess_batch = rec_perm, rec_pre
    SequenceSummaryStats.from_sequence = classmethod(rec_from)
    try:
        for u in range(n_updates):
            buf = VectorReplayBuffer(N, E)
            obs = rng.normal(size=(T + 1, E, obs_dim)).astype(np.float32)
            act = rng.normal(size=(T, E, act_dim)).astype(np.float32) * 0.7
            rew = rng.normal(size=(T, E)).astype(np.float32) + 0.3
            term = rng.random((T, E)) < 0.03
            trunc = np.zeros((T, E), bool)
            trunc[T // 2 - 1:: T // 2] = True
            trunc &= ~term
            for t in range(T):
                buf.add(Batch(obs=obs[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t], obs_next=obs[t + 1]))
            for k, v in (("obs", buf.obs), ("act", buf.act)):
                out[f"u{u}_{k}"] = np.asarray(v, np.float32)
            out[f"u{u}_rew"] = np.asarray(buf.rew, np.float64)
            out[f"u{u}_terminated"], out[f"u{u}_truncated"] = np.asarray(buf.terminated, bool), np.asarray(buf.truncated, bool)
            np.random.seed(seed + 100 + u)
            p0, s0 = len(perms), len(seqs)
            with policy_within_training_step(algorithm.policy):
                algorithm.update(buffer=buf, batch_size=batch_size, repeat=repeat)
            assert len(perms) - p0 == repeat and len(seqs) - s0 == 1
            out[f"u{u}_perms"], out[f"u{u}_losses"] = np.stack(perms[p0:]), seqs[s0]
            out[f"u{u}_returns"], out[f"u{u}_indices"], out[f"u{u}_unfinished"] = rets[-1]
            out[f"u{u}_actor"] = flat()
            if generic:
                opt = algorithm.optim._optim
                rmsp = type(opt).__name__ == "RMSprop"
                named = dict(actor.named_parameters())
                for i, k in enumerate(keys):
                    out[f"u{u}_a{i}"] = actor.state_dict()[k].numpy().copy()
                    out[f"u{u}_a{i}_v"] = opt.state[named[k]]["square_avg" if rmsp else "exp_avg_sq"].numpy().copy()
            rms = algorithm.discounted_return_computation.ret_rms
            out[f"u{u}_ret_rms"] = np.array([float(rms.mean), float(rms.var), float(rms.count)])
    finally:
        np.random.permutation, Reinforce._preprocess_batch = orig_perm, orig_pre
        SequenceSummaryStats.from_sequence = classmethod(orig_from)
    cfg = dict(gamma=algorithm.discounted_return_computation.gamma,
               return_standardization=float(algorithm.discounted_return_computation.return_standardization), lr=lr)
    if generic:
        g0 = algorithm.optim._optim.param_groups[0]
        cfg.update(max_grad_norm=float(algorithm.optim._max_grad_norm or 0.0),
                   opt_rmsprop=float(type(algorithm.optim._optim).__name__ == "RMSprop"), weight_decay=float(g0.get("weight_decay", 0.0)),
                   opt_eps=float(g0["eps"]), rms_alpha=float(g0.get("alpha", 0.99)), rms_momentum=float(g0.get("momentum", 0.0)),
                   rms_centered=float(g0.get("centered", False)))
    out["cfg_keys"], out["cfg_vals"] = np.array(list(cfg.keys())), np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"reinforce_{tag}.npz"), **out)


def gen_reinforce_net() -> None:
    """Round 6: Reinforce outside Net[h, h] tanh -- a three-layer ReLU trunk under the reference's default bounded actor with
    RMSprop; a single wide tanh layer with Adam + weight decay, unbounded."""
    gen_reinforce("net_relu3", E=4, T=48, obs_dim=11, act_dim=3, batch_size=64, repeat=2, seed=55, n_updates=2, gamma=0.97,
                  return_standardization=True, hidden=(96, 72, 40), activation=nn.ReLU, max_action=1.5,
                  optim=("rmsprop", dict(eps=1e-5, alpha=0.99)), lr=7e-4)
    gen_reinforce("net_tanh1", E=3, T=40, obs_dim=20, act_dim=5, batch_size=None, repeat=1, seed=56, n_updates=2, gamma=0.99,
                  return_standardization=False, hidden=(200,), activation=nn.Tanh, optim=("adam", dict(weight_decay=1e-2)), lr=1e-3)


def gen_reinforce_layernorm() -> None:
    """Round 6: Reinforce over Net(norm_layer=nn.LayerNorm) (common.py:25-39): two ReLU layers with LayerNorm(eps=1e-4), Adam."""
    gen_reinforce("net_ln_relu2", E=4, T=48, obs_dim=11, act_dim=3, batch_size=64, repeat=2, seed=57, n_updates=2, gamma=0.97,
                  return_standardization=True, hidden=(72, 40), activation=nn.ReLU, layer_norm=True, norm_args=dict(eps=1e-4), lr=7e-4)


def gen_reinforce_all() -> None:
    gen_reinforce("std", E=4, T=48, obs_dim=17, act_dim=6, batch_size=64, repeat=2, seed=51, n_updates=2, gamma=0.97,
                  return_standardization=True)
    gen_reinforce("plain", E=3, T=40, obs_dim=11, act_dim=3, batch_size=None, repeat=1, seed=53, n_updates=1, gamma=0.99,
                  return_standardization=False)


def gen_drqn(tag: str, *, E: int, slots: int, steps: int, obs_dim: int, hidden: int, layers: int, n_act: int, stack_num: int,
             batch: int, n_updates: int, seed: int, per: bool, lr: float = 1e-3, **dqn_kwargs) -> None:
    """Runs the reference DQN.update() with the Recurrent Q network on a stacked, obs_next-free buffer
    (test/discrete/test_drqn.py:79-108) and dumps indices, n-step returns, TD errors, losses and parameters of every update."""
    from tianshou.algorithm.modelfree.dqn import DQN, DiscreteQLearningPolicy
    from tianshou.utils.net.common import Recurrent
    from oracle import oracle_drqn as ORQ

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    net = Recurrent(layer_num=layers, state_shape=(obs_dim,), action_shape=n_act, hidden_layer_size=hidden)
    keys = ORQ.param_keys(layers)
    assert list(net.state_dict().keys()) == keys, list(net.state_dict().keys())
    policy = DiscreteQLearningPolicy(model=net, action_space=gym.spaces.Discrete(n_act))
    algorithm = DQN(policy=policy, optim=AdamOptimizerFactory(lr=lr), **dqn_kwargs)
    kw = dict(stack_num=stack_num, ignore_obs_next=True)
    buf = PrioritizedVectorReplayBuffer(E * slots, E, alpha=0.6, beta=0.4, **kw) if per else VectorReplayBuffer(E * slots, E, **kw)
    obs = rng.normal(size=(steps + 1, E, obs_dim)).astype(np.float32)
    act = rng.integers(0, n_act, size=(steps, E))
    rew = rng.normal(size=(steps, E)).astype(np.float32)
    term = rng.random((steps, E)) < 0.08
    trunc = (rng.random((steps, E)) < 0.04) & ~term
    for t in range(steps):
        buf.add(Batch(obs=obs[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t], obs_next=obs[t + 1]))
    flat = lambda: torch.cat([net.state_dict()[k].reshape(-1) for k in keys]).numpy().copy()  # noqa: E731
    out: dict[str, np.ndarray] = {"dims": np.array([E, slots, steps, obs_dim, hidden, layers, n_act, stack_num, batch, n_updates,
                                                    seed, int(per)]), "params0": flat()}
    out["obs_rows"] = np.asarray(buf.obs, np.float32)
    out["act"], out["rew"] = np.asarray(buf.act, np.int64), np.asarray(buf.rew, np.float64)
    out["terminated"], out["truncated"] = np.asarray(buf.terminated, bool), np.asarray(buf.truncated, bool)
    for k, v in manager_state(buf).items():
        out["buf_" + k] = v
    rec: list[dict] = []
    orig_pre, orig_upd = DQN._preprocess_batch, DQN._update_with_batch

    def rec_pre(self, batch, buffer, indices):
        r = {"indices": np.array(indices, np.int64), "obs": np.array(batch.obs)}
        if hasattr(batch, "weight"):
            r["is_weight"] = np.array(batch.weight, np.float64)
        b = orig_pre(self, batch, buffer, indices)
        r["returns"] = b.returns.numpy().copy().reshape(-1)
        rec.append(r)
        return b

    def rec_upd(self, batch):
        stats = orig_upd(self, batch)
        rec[-1]["td"] = batch.weight.detach().numpy().copy()
        rec[-1]["loss"] = np.array(stats.loss)
        return stats

    DQN._preprocess_batch, DQN._update_with_batch = rec_pre, rec_upd
    try:
        np.random.seed(seed + 7)
        for u in range(n_updates):
            with policy_within_training_step(algorithm.policy):
                algorithm.update(buffer=buf, sample_size=batch)
            r = rec[-1]
            assert r["obs"].shape == (batch, stack_num, obs_dim)
            for k in ("indices", "returns", "td", "loss"):
                out[f"u{u}_{k}"] = r[k]
            if "is_weight" in r:
                out[f"u{u}_is_weight"] = r["is_weight"]
            if u == 0:
                out["u0_obs"] = r["obs"]
            out[f"u{u}_params_strided"] = flat()[::17].copy()           # + the small tensors in full
            sd = net.state_dict()
            out[f"u{u}_small"] = torch.cat([sd[k].reshape(-1) for k in keys if "weight_" not in k]).numpy().copy()
    finally:
        DQN._preprocess_batch, DQN._update_with_batch = orig_pre, orig_upd
    # evaluation-mode steps with a carried state (Recurrent.forward with obs [B, dim], common.py:419-452)
    with torch.no_grad():
        o1, o2 = obs[0, :, :], obs[1, :, :]
        q1, s1 = net(o1)
        q2, s2 = net(o2, state=s1)
    out["eval_obs"] = np.stack([o1, o2])
    out["eval_q"] = np.stack([q1.numpy(), q2.numpy()])
    out["eval_hidden"], out["eval_cell"] = s2["hidden"].numpy().copy(), s2["cell"].numpy().copy()      # [B, L, H]
    cfg = dict(gamma=algorithm.gamma, n_step=algorithm.n_step, target_update_freq=algorithm.target_update_freq,
               is_double=float(algorithm.is_double),
               huber_delta=-1.0 if algorithm.huber_loss_delta is None else algorithm.huber_loss_delta, lr=lr)
    out["cfg_keys"], out["cfg_vals"] = np.array(list(cfg.keys())), np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"drqn_{tag}.npz"), **out)


def gen_drqn_all() -> None:
    # the test_drqn.py setup: CartPole observations, 2 LSTM layers of 128, stack 4, n-step 3, double-Q with a lagged net
    gen_drqn("cartpole", E=4, slots=40, steps=60, obs_dim=4, hidden=128, layers=2, n_act=2, stack_num=4, batch=32, n_updates=3,
             seed=61, per=False, gamma=0.95, n_step_return_horizon=3, target_update_freq=2, is_double=True)
    # one layer of 64, a wider observation, PER weights with the MSE loss, Huber off, vanilla max-Q target without a lagged net
    gen_drqn("per", E=3, slots=30, steps=40, obs_dim=37, hidden=64, layers=1, n_act=5, stack_num=3, batch=24, n_updates=2,
             seed=63, per=True, lr=3e-4, gamma=0.9, n_step_return_horizon=1, target_update_freq=0, is_double=False)


def gen_dqn(tag: str, *, E: int, slots: int, steps: int, c: int, h: int, w: int, n_act: int, batch: int,
            n_updates: int, seed: int, per: bool, stack: bool, lr: float = 1e-4, **dqn_kwargs) -> None:
    """Runs the reference DQN.update() (DQNet + DiscreteQLearningPolicy, dqn.py) on a synthetic
    (Prioritized)VectorReplayBuffer and dumps the sampled indices, n-step returns, TD errors, losses
    and parameter samples of every update.  `stack`: atari layout (stack_num=c, save_only_last_obs,
    ignore_obs_next; examples/atari/atari_dqn.py:137-142), else full [c,h,w] observations."""
    from tianshou.algorithm.modelfree.dqn import DQN, DiscreteQLearningPolicy
    from tianshou.env.atari.atari_network import DQNet
    from oracle import oracle_dqn as OD

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    net = DQNet(c=c, h=h, w=w, action_shape=n_act)
    p0 = OD.init_params(c, h, w, n_act, seed)
    sd = net.state_dict()
    for k_ref, k in zip(OD.TIANSHOU_KEYS, OD.PARAM_ORDER):
        assert torch.equal(sd[k_ref], p0[k]), f"oracle init differs from DQNet at {k}"
    policy = DiscreteQLearningPolicy(model=net, action_space=gym.spaces.Discrete(n_act))
    algorithm = DQN(policy=policy, optim=AdamOptimizerFactory(lr=lr), **dqn_kwargs)

    kw = dict(stack_num=c, ignore_obs_next=True, save_only_last_obs=True) if stack else dict(ignore_obs_next=True)
    if per:
        buf = PrioritizedVectorReplayBuffer(E * slots, E, alpha=0.6, beta=0.4, **kw)
    else:
        buf = VectorReplayBuffer(E * slots, E, **kw)
    frames = rng.integers(0, 256, size=(steps + 1, E, c, h, w), dtype=np.uint8)
    # a sparse image so that the fixture compresses: ~6% non-zero pixels
    frames = np.where(rng.random(frames.shape) < 0.06, frames, 0).astype(np.uint8)
    act = rng.integers(0, n_act, size=(steps, E))
    rew = rng.normal(size=(steps, E)).astype(np.float32)
    term = rng.random((steps, E)) < 0.08
    trunc = (rng.random((steps, E)) < 0.04) & ~term
    for t in range(steps):
        buf.add(Batch(obs=frames[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t],
                      obs_next=frames[t + 1]))
    out: dict[str, np.ndarray] = {}
    out["dims"] = np.array([E, slots, steps, c, h, w, n_act, batch, n_updates, seed, int(per), int(stack)])
    out["frames"] = np.asarray(buf.obs, np.uint8)          # [B, h, w] (stack) or [B, c, h, w]
    out["act"] = np.asarray(buf.act, np.int64)
    out["rew"] = np.asarray(buf.rew, np.float64)
    out["terminated"] = np.asarray(buf.terminated, bool)
    out["truncated"] = np.asarray(buf.truncated, bool)
    for k, v in manager_state(buf).items():
        out["buf_" + k] = v
    if per:
        out["tree0"] = np.asarray(buf.weight._value, np.float64).copy()

    rec: list[dict] = []
    orig_pre, orig_upd = DQN._preprocess_batch, DQN._update_with_batch

    def rec_pre(self, batch, buffer, indices):
        r = {"indices": np.array(indices, np.int64)}
        r["obs"] = np.array(batch.obs)
        if hasattr(batch, "weight"):
            r["is_weight"] = np.array(batch.weight, np.float64)
        b = orig_pre(self, batch, buffer, indices)
        r["returns"] = b.returns.numpy().copy().reshape(-1)
        rec.append(r)
        return b

    def rec_upd(self, batch):
        stats = orig_upd(self, batch)
        rec[-1]["td"] = batch.weight.detach().numpy().copy()
        rec[-1]["loss"] = np.array(stats.loss)
        return stats

    DQN._preprocess_batch, DQN._update_with_batch = rec_pre, rec_upd
    try:
        np.random.seed(seed + 7)
        for u in range(n_updates):
            with policy_within_training_step(algorithm.policy):
                algorithm.update(buffer=buf, sample_size=batch)
            r = rec[-1]
            flat = torch.cat([net.state_dict()[k].reshape(-1) for k in OD.TIANSHOU_KEYS]).numpy()
            for k in ("indices", "returns", "td", "loss"):
                out[f"u{u}_{k}"] = r[k]
            if "is_weight" in r:
                out[f"u{u}_is_weight"] = r["is_weight"]
            if u == 0:
                out["u0_obs_sample"] = r["obs"][:2]
            out[f"u{u}_params_strided"] = flat[::61].copy()
            out[f"u{u}_conv1_w"] = net.state_dict()["net.0.0.weight"].numpy().copy()
            out[f"u{u}_fc2_w"] = net.state_dict()["net.3.weight"].numpy().copy()
            out[f"u{u}_biases"] = torch.cat([net.state_dict()[k].reshape(-1) for k in OD.TIANSHOU_KEYS
                                             if k.endswith("bias")]).numpy().copy()
            if per:
                out[f"u{u}_tree"] = np.asarray(buf.weight._value, np.float64).copy()
                out[f"u{u}_prio_minmax"] = np.array([buf._min_prio, buf._max_prio])
    finally:
        DQN._preprocess_batch, DQN._update_with_batch = orig_pre, orig_upd
    cfg = dict(gamma=algorithm.gamma, n_step=algorithm.n_step, target_update_freq=algorithm.target_update_freq,
               is_double=float(algorithm.is_double),
               huber_delta=-1.0 if algorithm.huber_loss_delta is None else algorithm.huber_loss_delta, lr=lr)
    out["cfg_keys"] = np.array(list(cfg.keys()))
    out["cfg_vals"] = np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"dqn_{tag}.npz"), **out)



def gen_recurrent() -> None:
    """RecurrentActorProb / RecurrentCritic (utils/net/continuous.py:241-380) as the reference's own test builds them
    (test/base/test_utils.py:99-112: 3 layers on a vector observation), plus a bounded single-layer actor with a carried
    state: parameters, inputs, outputs and -- through autograd on the reference modules -- parameter gradients of a fixed
    linear functional of the outputs."""
    from oracle import oracle_recurrent as OR
    from tianshou.utils.net.continuous import RecurrentActorProb, RecurrentCritic

    out = {}
    for tag, (obs_dim, act_dim, hidden, layers, B, T, max_action, unbounded) in {
        "utils": (6, 5, 32, 3, 7, 4, 1.0, False),               # test_utils.py:99-112: layer_num=3 (hidden 32 keeps the file small)
        "small": (11, 3, 64, 1, 33, 1, 2.5, False),
        "free": (37, 17, 32, 2, 9, 6, 1.0, True),
    }.items():
        torch.manual_seed(sum(tag.encode()))        # a literal function of the tag (str hashes are randomised per process)
        rng = np.random.default_rng(len(tag))
        actor = RecurrentActorProb(layer_num=layers, state_shape=(obs_dim,), action_shape=(act_dim,), hidden_layer_size=hidden,
                                   max_action=max_action, unbounded=unbounded)
        critic = RecurrentCritic(layer_num=layers, state_shape=(obs_dim,), action_shape=(act_dim,), hidden_layer_size=hidden)
        with torch.no_grad():
            actor.sigma_param.copy_(torch.from_numpy(rng.normal(size=(act_dim, 1)).astype(np.float32)) * 0.3)
        assert list(actor.state_dict().keys()) == OR.actor_keys(layers), list(actor.state_dict().keys())
        assert list(critic.state_dict().keys()) == OR.critic_keys(layers), list(critic.state_dict().keys())
        obs = rng.normal(size=(B, T, obs_dim)).astype(np.float32)
        act = rng.normal(size=(B, act_dim)).astype(np.float32)
        state = {"hidden": torch.from_numpy(rng.normal(size=(B, layers, hidden)).astype(np.float32)) * 0.5,
                 "cell": torch.from_numpy(rng.normal(size=(B, layers, hidden)).astype(np.float32)) * 0.5}
        w_mu = torch.from_numpy(rng.normal(size=(B, act_dim)).astype(np.float32))
        w_v = torch.from_numpy(rng.normal(size=(B, 1)).astype(np.float32))
        (mu, sigma), st = actor(obs)
        (mu_s, sigma_s), st_s = actor(obs[:, -1], state=state)       # evaluation mode: [B, dim] with a carried state
        actor.zero_grad()
        (mu * w_mu).sum().backward()
        a_grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in actor.named_parameters()}
        v = critic(obs, act)
        critic.zero_grad()
        (v * w_v).sum().backward()
        c_grads = {k: p.grad for k, p in critic.named_parameters()}
        pre = f"{tag}_"
        out[pre + "dims"] = np.array([obs_dim, act_dim, hidden, layers, B, T], np.int64)
        out[pre + "max_action"], out[pre + "unbounded"] = np.float64(max_action), np.int64(unbounded)
        for k, t in actor.state_dict().items():
            out[pre + "actor." + k] = t.detach().numpy().copy()
        for k, t in critic.state_dict().items():
            out[pre + "critic." + k] = t.detach().numpy().copy()
        for k, t in a_grads.items():
            out[pre + "actor_grad." + k] = t.detach().numpy().copy()
        for k, t in c_grads.items():
            out[pre + "critic_grad." + k] = t.detach().numpy().copy()
        out[pre + "obs"], out[pre + "act"], out[pre + "w_mu"], out[pre + "w_v"] = obs, act, w_mu.numpy(), w_v.numpy()
        out[pre + "state_hidden"], out[pre + "state_cell"] = state["hidden"].numpy(), state["cell"].numpy()
        out[pre + "mu"], out[pre + "sigma"] = mu.detach().numpy(), sigma.detach().numpy()
        out[pre + "hidden"], out[pre + "cell"] = st["hidden"].numpy(), st["cell"].numpy()
        out[pre + "mu_s"], out[pre + "hidden_s"], out[pre + "cell_s"] = mu_s.detach().numpy(), st_s["hidden"].numpy(), st_s["cell"].numpy()
        out[pre + "value"] = v.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "recurrent_nets.npz"), **out)


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "recurrent":
        gen_recurrent()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "dqn":
        gen_dqn_all()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ppo_net":
        gen_ppo_net_all()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ppo_layernorm":
        gen_ppo_layernorm()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "rainbow":
        gen_rainbow_all()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "drqn":
        gen_drqn_all()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "reinforce_net":
        gen_reinforce_net()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "reinforce_layernorm":
        gen_reinforce_layernorm()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "reinforce":
        gen_reinforce_all()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "npg":
        gen_npg_all()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "distq":
        gen_distq_all()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "buffer_add":
        gen_buffer_add()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ppo_cnn":
        # atari_ppo.py defaults: eps 0.1, vf 0.25, ent 0.01, max_grad_norm 0.5, value_clip, adv norm, no return scaling
        gen_ppo_cnn(gamma=0.99, gae_lambda=0.95, max_grad_norm=0.5, vf_coef=0.25, ent_coef=0.01, return_scaling=False,
                    eps_clip=0.1, value_clip=True, dual_clip=None, advantage_normalization=True,
                    recompute_advantage=False, max_batchsize=32)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ppo_discrete":
        gen_ppo_discrete_all()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "depth":
        gen_depth()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "widths":
        gen_widths()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "policy_forward":
        gen_policy_forward()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ppo_round6":
        gen_ppo_round6()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ppo_sched":
        gen_ppo_sched()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "sample_stack":
        gen_sample_stack()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "sample_random":
        gen_sample_random()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "redq":
        gen_redq_all()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "dsac":
        gen_dsac_all()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "td3":
        gen_td3_all()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "sac":
        gen_sac_all()
        return
    only_ppo = len(sys.argv) > 1 and sys.argv[1] == "ppo"
    if not only_ppo:
        gen_returns_kat()
        gen_buffer_index()
        gen_segtree_per()
    # mujoco-example style (examples/mujoco/mujoco_ppo.py:28-62) without recompute
    gen_ppo("mujoco", E=8, T=64, obs_dim=17, act_dim=6, batch_size=128, repeat=3, seed=0,
            n_updates=2, gamma=0.99, gae_lambda=0.95, max_grad_norm=0.5, vf_coef=0.25,
            ent_coef=0.0, return_scaling=True, eps_clip=0.2, value_clip=True, dual_clip=None,
            advantage_normalization=False, recompute_advantage=False, max_batchsize=256)
    # library defaults (ppo.py:24-36) + dual clip + recompute_advantage, ragged last minibatch
    gen_ppo("defaults", E=4, T=75, obs_dim=17, act_dim=6, batch_size=128, repeat=2, seed=1,
            n_updates=1, dual_clip=3.0, recompute_advantage=True, lr=1e-3, max_batchsize=64)
    # A2C (a2c.py:163-290) with the same nets: vf 0.5, ent 0.01, max_grad_norm 0.5, return scaling
    gen_ppo("a2c", algo="a2c", E=4, T=60, obs_dim=17, act_dim=6, batch_size=64, repeat=2, seed=2,
            n_updates=2, vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5, gae_lambda=0.95, gamma=0.99,
            return_scaling=True, lr=7e-4, max_batchsize=256)
    if only_ppo:
        return
    gen_ppo_sched()
    gen_sample_random()
    gen_sample_stack()
    gen_buffer_add()
    gen_dqn_all()
    gen_sac_all()
    gen_td3_all()
    gen_ppo_cnn(gamma=0.99, gae_lambda=0.95, max_grad_norm=0.5, vf_coef=0.25, ent_coef=0.01, return_scaling=False,
                eps_clip=0.1, value_clip=True, dual_clip=None, advantage_normalization=True,
                recompute_advantage=False, max_batchsize=32)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


def gen_sac(tag: str, *, E: int, slots: int, steps: int, obs_dim: int, act_dim: int, batch: int, n_updates: int,
            seed: int, auto_alpha: bool, alpha: float = 0.2, n_step: int = 1, tau: float = 0.005,
            gamma: float = 0.99, actor_lr: float = 1e-3, critic_lr: float = 1e-3, alpha_lr: float = 3e-4, hidden=256,
            max_action: float = 0.0, activation=nn.ReLU) -> None:
    """Runs the reference SAC.update() (nets as in examples/mujoco/mujoco_sac.py:82-104) on a synthetic
    VectorReplayBuffer, recording the rsample() noise of every policy call and the outputs of every update.
    hidden: int, (h1, h2) or (actor h1, actor h2, critic h1, critic h2) -- Net(hidden_sizes=...) takes any widths.
    max_action > 0: the actor is built with the class default `unbounded=False` (mu = max_action * tanh(mu)) instead of the
    examples' `unbounded=True`."""
    import torch.distributions.normal as tdn
    from tianshou.algorithm.modelfree.sac import SAC, AutoAlpha, SACPolicy
    from oracle import oracle_sac as OS

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    sa_, sc_ = OS.layer_sizes(hidden)              # (any depth since round 6: a nested (actor sizes, critic sizes) pair)
    AK, CK = OS.trunk_keys(len(sa_), ("mu", "sigma")), OS.trunk_keys(len(sc_), ("last",))
    net_a = Net(state_shape=(obs_dim,), hidden_sizes=list(sa_), activation=activation)
    if max_action > 0.0:
        actor = ContinuousActorProbabilistic(preprocess_net=net_a, action_shape=(act_dim,), max_action=max_action,
                                             conditioned_sigma=True)           # unbounded=False: the class default
        assert not actor._unbounded
    else:
        actor = ContinuousActorProbabilistic(preprocess_net=net_a, action_shape=(act_dim,), unbounded=True,
                                             conditioned_sigma=True)
    net_c1 = Net(state_shape=(obs_dim,), action_shape=(act_dim,), hidden_sizes=list(sc_), concat=True, activation=activation)
    net_c2 = Net(state_shape=(obs_dim,), action_shape=(act_dim,), hidden_sizes=list(sc_), concat=True, activation=activation)
    critic1, critic2 = ContinuousCritic(preprocess_net=net_c1), ContinuousCritic(preprocess_net=net_c2)
    space = gym.spaces.Box(low=-1.0, high=1.0, shape=(act_dim,))
    policy = SACPolicy(actor=actor, action_space=space)
    al = AutoAlpha(float(-act_dim), 0.0, AdamOptimizerFactory(lr=alpha_lr)) if auto_alpha else alpha
    algorithm = SAC(policy=policy, policy_optim=AdamOptimizerFactory(lr=actor_lr), critic=critic1,
                    critic_optim=AdamOptimizerFactory(lr=critic_lr), critic2=critic2,
                    critic2_optim=AdamOptimizerFactory(lr=critic_lr), tau=tau, gamma=gamma, alpha=al,
                    n_step_return_horizon=n_step)
    out: dict[str, np.ndarray] = {}
    out["dims"] = np.array([E, slots, steps, obs_dim, act_dim, batch, n_updates, seed, int(auto_alpha), n_step])
    if len(sa_) == 2 and len(sc_) == 2:
        out["hidden"] = np.array(sa_ + sc_, np.int64)
    else:
        out["hidden_actor"], out["hidden_critic"] = np.array(sa_, np.int64), np.array(sc_, np.int64)
    p0 = OS.init_sac_params(obs_dim, act_dim, seed, (sa_, sc_))
    for pd, order, mod, keys in ((p0[0], OS.actor_order(len(sa_)), actor, AK),
                                 (p0[1], OS.critic_order(len(sc_)), critic1, CK),
                                 (p0[2], OS.critic_order(len(sc_)), critic2, CK)):
        sd = mod.state_dict()
        assert list(sd.keys()) == keys, list(sd.keys())
        for k_ref, k in zip(keys, order):       # the fixture stores the seed only: the oracle re-creates the init
            assert torch.equal(sd[k_ref], pd[k]), f"oracle init differs from the reference at {k}"

    buf = VectorReplayBuffer(E * slots, E)
    obs = rng.normal(size=(steps + 1, E, obs_dim)).astype(np.float32)
    act = rng.uniform(-1, 1, size=(steps, E, act_dim)).astype(np.float32)
    rew = rng.normal(size=(steps, E)).astype(np.float32)
    term = rng.random((steps, E)) < 0.05
    trunc = (rng.random((steps, E)) < 0.03) & ~term
    for t in range(steps):
        buf.add(Batch(obs=obs[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t],
                      obs_next=obs[t + 1]))
    out["obs"] = np.asarray(buf.obs, np.float32)
    out["obs_next"] = np.asarray(buf.obs_next, np.float32)
    out["act"] = np.asarray(buf.act, np.float32)
    out["rew"] = np.asarray(buf.rew, np.float64)
    out["terminated"] = np.asarray(buf.terminated, bool)
    out["truncated"] = np.asarray(buf.truncated, bool)
    for k, v in manager_state(buf).items():
        out["buf_" + k] = v

    noises: list[np.ndarray] = []
    orig_sn = tdn._standard_normal

    def rec_sn(shape, dtype, device):
        e = orig_sn(shape, dtype, device)
        noises.append(e.numpy().copy())
        return e

    rec: list[dict] = []
    orig_pre = SAC._preprocess_batch

    def rec_pre(self, batch, buffer, indices):
        b = orig_pre(self, batch, buffer, indices)
        rec.append({"indices": np.array(indices, np.int64), "returns": b.returns.numpy().copy().reshape(-1)})
        return b

    tdn._standard_normal = rec_sn
    SAC._preprocess_batch = rec_pre
    try:
        for u in range(n_updates):
            n0 = len(noises)
            with policy_within_training_step(algorithm.policy):
                stats = algorithm.update(buffer=buf, sample_size=batch)
            assert len(noises) - n0 == 2, len(noises) - n0          # target action, actor-loss action
            out[f"u{u}_noise_target"], out[f"u{u}_noise_actor"] = noises[n0], noises[n0 + 1]
            out[f"u{u}_indices"], out[f"u{u}_returns"] = rec[-1]["indices"], rec[-1]["returns"]
            out[f"u{u}_stats"] = np.array([stats.actor_loss, stats.critic1_loss, stats.critic2_loss,
                                           stats.alpha if stats.alpha is not None else np.nan,
                                           stats.alpha_loss if stats.alpha_loss is not None else np.nan])
            for name, mod, keys in (("actor", actor, AK), ("critic1", critic1, CK), ("critic2", critic2, CK),
                                    ("critic1_old", algorithm.critic_old.module, CK),
                                    ("critic2_old", algorithm.critic2_old.module, CK)):
                sd = mod.state_dict()
                out[f"u{u}_{name}"] = torch.cat([sd[k].reshape(-1) for k in keys]).numpy()[::61].copy()
    finally:
        tdn._standard_normal = orig_sn
        SAC._preprocess_batch = orig_pre
    cfg = dict(gamma=gamma, tau=tau, n_step=n_step, alpha=alpha, auto_alpha=float(auto_alpha),
               target_entropy=float(-act_dim), log_alpha0=0.0, actor_lr=actor_lr, critic_lr=critic_lr,
               alpha_lr=alpha_lr, **({"max_action": max_action} if max_action > 0.0 else {}),
               **({"tanh_trunks": 1.0} if activation is nn.Tanh else {}))
    out["cfg_keys"] = np.array(list(cfg.keys()))
    out["cfg_vals"] = np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"sac_{tag}.npz"), **out)


def gen_td3(tag: str, *, twin: bool, E: int, slots: int, steps: int, obs_dim: int, act_dim: int, batch: int,
            n_updates: int, seed: int, max_action: float = 1.0, n_step: int = 1, hidden=256, activation=nn.ReLU, **kw) -> None:
    """Runs the reference TD3.update() (twin) or DDPG.update() (nets of examples/mujoco/mujoco_td3.py:85-103 /
    mujoco_ddpg.py) on a synthetic VectorReplayBuffer; TD3's torch.randn smoothing noise is recorded."""
    from tianshou.algorithm.modelfree import td3 as td3_mod
    from tianshou.algorithm.modelfree.ddpg import DDPG, ContinuousDeterministicPolicy
    from tianshou.algorithm.modelfree.td3 import TD3
    from tianshou.utils.net.continuous import ContinuousActorDeterministic
    from oracle import oracle_sac as OS

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    sa_, sc_ = OS.layer_sizes(hidden)              # (any depth since round 6: a nested (actor sizes, critic sizes) pair)
    AK, CK = OS.trunk_keys(len(sa_), ("last",)), OS.trunk_keys(len(sc_), ("last",))
    AO, CO = OS.det_actor_order(len(sa_)), OS.critic_order(len(sc_))
    actor = ContinuousActorDeterministic(preprocess_net=Net(state_shape=(obs_dim,), hidden_sizes=list(sa_), activation=activation),
                                         action_shape=(act_dim,), max_action=max_action)
    mk_net = lambda: Net(state_shape=(obs_dim,), action_shape=(act_dim,), hidden_sizes=list(sc_), concat=True, activation=activation)  # noqa: E731
    if twin:
        n1, n2 = mk_net(), mk_net()
        critic1, critic2 = ContinuousCritic(preprocess_net=n1), ContinuousCritic(preprocess_net=n2)
    else:
        critic1, critic2 = ContinuousCritic(preprocess_net=mk_net()), None
    p0 = OS.init_td3_params(obs_dim, act_dim, seed, twin, (sa_, sc_))
    checks = [(p0[0], AO, actor, AK), (p0[1], CO, critic1, CK)]
    if twin:
        checks.append((p0[2], CO, critic2, CK))
    for pd, order, mod, keys in checks:
        sd = mod.state_dict()
        assert list(sd.keys()) == keys, list(sd.keys())
        for k_ref, k in zip(keys, order):
            assert torch.equal(sd[k_ref], pd[k]), f"oracle init differs from the reference at {k}"
    space = gym.spaces.Box(low=-max_action, high=max_action, shape=(act_dim,))
    policy = ContinuousDeterministicPolicy(actor=actor, action_space=space, exploration_noise=None)
    common = dict(policy=policy, policy_optim=AdamOptimizerFactory(lr=kw.get("actor_lr", 1e-3)), critic=critic1,
                  critic_optim=AdamOptimizerFactory(lr=kw.get("critic_lr", 1e-3)), tau=kw.get("tau", 0.005),
                  gamma=kw.get("gamma", 0.99), n_step_return_horizon=n_step)
    if twin:
        algorithm = TD3(critic2=critic2, critic2_optim=AdamOptimizerFactory(lr=kw.get("critic_lr", 1e-3)),
                        policy_noise=kw.get("policy_noise", 0.2), noise_clip=kw.get("noise_clip", 0.5),
                        update_actor_freq=kw.get("update_actor_freq", 2), **common)
    else:
        algorithm = DDPG(**common)
    buf = VectorReplayBuffer(E * slots, E)
    obs = rng.normal(size=(steps + 1, E, obs_dim)).astype(np.float32)
    act = rng.uniform(-max_action, max_action, size=(steps, E, act_dim)).astype(np.float32)
    rew = rng.normal(size=(steps, E)).astype(np.float32)
    term = rng.random((steps, E)) < 0.05
    trunc = (rng.random((steps, E)) < 0.03) & ~term
    for t in range(steps):
        buf.add(Batch(obs=obs[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t], obs_next=obs[t + 1]))
    out: dict[str, np.ndarray] = {"dims": np.array([E, slots, steps, obs_dim, act_dim, batch, n_updates, seed, int(twin), n_step]),
                                  **({"hidden": np.array(sa_ + sc_, np.int64)} if len(sa_) == 2 and len(sc_) == 2 else
                                     {"hidden_actor": np.array(sa_, np.int64), "hidden_critic": np.array(sc_, np.int64)})}
    for k2 in ("obs", "obs_next", "act"):
        out[k2] = np.asarray(getattr(buf, k2), np.float32)
    out["rew"], out["terminated"], out["truncated"] = np.asarray(buf.rew, np.float64), np.asarray(buf.terminated, bool), np.asarray(buf.truncated, bool)
    for k2, v in manager_state(buf).items():
        out["buf_" + k2] = v
    noises, rec = [], []
    orig_randn, cls = torch.randn, type(algorithm)
    orig_pre = cls._preprocess_batch

    def rec_randn(*a, **k):
        e = orig_randn(*a, **k)
        if "size" in k:
            noises.append(e.numpy().copy())
        return e

    def rec_pre(self, batch, buffer, indices):
        b = orig_pre(self, batch, buffer, indices)
        rec.append({"indices": np.array(indices, np.int64), "returns": b.returns.numpy().copy().reshape(-1)})
        return b

    torch.randn, cls._preprocess_batch = rec_randn, rec_pre
    try:
        for u in range(n_updates):
            n0 = len(noises)
            with policy_within_training_step(algorithm.policy):
                stats = algorithm.update(buffer=buf, sample_size=batch)
            if twin:
                assert len(noises) - n0 == 1
                out[f"u{u}_noise"] = noises[n0]
            out[f"u{u}_indices"], out[f"u{u}_returns"] = rec[-1]["indices"], rec[-1]["returns"]
            out[f"u{u}_stats"] = np.array([stats.actor_loss, stats.critic1_loss, stats.critic2_loss] if twin
                                          else [stats.actor_loss, stats.critic_loss])
            mods = [("actor", actor, AK), ("critic1", critic1, CK), ("actor_old", algorithm.actor_old.module, AK),
                    ("critic1_old", algorithm.critic_old.module, CK)]
            if twin:
                mods += [("critic2", critic2, CK), ("critic2_old", algorithm.critic2_old.module, CK)]
            for name, mod, keys in mods:
                sd = mod.state_dict()
                out[f"u{u}_{name}"] = torch.cat([sd[k2].reshape(-1) for k2 in keys]).numpy()[::61].copy()
    finally:
        torch.randn, cls._preprocess_batch = orig_randn, orig_pre
    cfg = dict(gamma=common["gamma"], tau=common["tau"], n_step=n_step, twin=float(twin),
               policy_noise=kw.get("policy_noise", 0.2), noise_clip=kw.get("noise_clip", 0.5),
               update_actor_freq=kw.get("update_actor_freq", 2), max_action=max_action,
               actor_lr=kw.get("actor_lr", 1e-3), critic_lr=kw.get("critic_lr", 1e-3),
               **({"tanh_trunks": 1.0} if activation is nn.Tanh else {}))
    out["cfg_keys"], out["cfg_vals"] = np.array(list(cfg.keys())), np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"td3_{tag}.npz"), **out)


def gen_dsac(tag: str, *, E: int, slots: int, steps: int, obs_dim: int, n_act: int, hidden, batch: int,
             n_updates: int, seed: int, auto_alpha: bool, alpha: float = 0.05, n_step: int = 1, tau: float = 0.005,
             gamma: float = 0.95, actor_lr: float = 1e-3, critic_lr: float = 1e-3, alpha_lr: float = 3e-4) -> None:
    """Runs the reference DiscreteSAC.update() (nets as in test/discrete/test_discrete_sac.py:88-97) on a synthetic
    PrioritizedVectorReplayBuffer and records the outputs of every update."""
    from tianshou.algorithm.modelfree.discrete_sac import DiscreteSAC, DiscreteSACPolicy
    from tianshou.algorithm.modelfree.sac import AutoAlpha
    from tianshou.utils.net.discrete import DiscreteActor, DiscreteCritic
    from oracle import oracle_dsac as ODS

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    from oracle import oracle_sac as OS_

    sa_, sc_ = OS_.layer_sizes(hidden)      # int, (h1, h2), (actor h1, actor h2, critic h1, critic h2) or a nested pair of any depth
    hw = tuple(sa_) + tuple(sc_)
    DK = OS_.trunk_keys(len(sa_), ("last",))
    assert len(sa_) == len(sc_)
    actor = DiscreteActor(preprocess_net=Net(state_shape=(obs_dim,), hidden_sizes=list(sa_)),
                          action_shape=n_act, softmax_output=False)
    critic1 = DiscreteCritic(preprocess_net=Net(state_shape=(obs_dim,), hidden_sizes=list(sc_)), last_size=n_act)
    critic2 = DiscreteCritic(preprocess_net=Net(state_shape=(obs_dim,), hidden_sizes=list(sc_)), last_size=n_act)
    policy = DiscreteSACPolicy(actor=actor, action_space=gym.spaces.Discrete(n_act))
    target_entropy = 0.98 * float(np.log(n_act))
    al = AutoAlpha(target_entropy, 0.0, AdamOptimizerFactory(lr=alpha_lr)) if auto_alpha else alpha
    algorithm = DiscreteSAC(policy=policy, policy_optim=AdamOptimizerFactory(lr=actor_lr), critic=critic1,
                            critic_optim=AdamOptimizerFactory(lr=critic_lr), critic2=critic2,
                            critic2_optim=AdamOptimizerFactory(lr=critic_lr), tau=tau, gamma=gamma, alpha=al,
                            n_step_return_horizon=n_step)
    out: dict[str, np.ndarray] = {}
    out["dims"] = np.array([E, slots, steps, obs_dim, n_act, max(hw), batch, n_updates, seed, int(auto_alpha), n_step])
    if len(sa_) != 2:
        out["hidden_actor"], out["hidden_critic"] = np.array(sa_, np.int64), np.array(sc_, np.int64)
    elif len(set(hw)) > 1:
        out["hidden"] = np.array(hw, np.int64)
    p0 = ODS.init_params(obs_dim, n_act, (sa_, sc_), seed)
    for pd, mod in zip(p0, (actor, critic1, critic2)):
        sd = mod.state_dict()
        assert list(sd.keys()) == DK, list(sd.keys())
        for k_ref, k in zip(DK, ODS.net_order(len(sa_))):
            assert torch.equal(sd[k_ref], pd[k]), f"oracle init differs from the reference at {k}"

    buf = PrioritizedVectorReplayBuffer(E * slots, E, alpha=0.6, beta=0.4)
    obs = rng.normal(size=(steps + 1, E, obs_dim)).astype(np.float32)
    act = rng.integers(0, n_act, size=(steps, E))
    rew = rng.normal(size=(steps, E)).astype(np.float32)
    term = rng.random((steps, E)) < 0.05
    trunc = (rng.random((steps, E)) < 0.03) & ~term
    for t in range(steps):
        buf.add(Batch(obs=obs[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t],
                      obs_next=obs[t + 1]))
    out["obs"] = np.asarray(buf.obs, np.float32)
    out["obs_next"] = np.asarray(buf.obs_next, np.float32)
    out["act"] = np.asarray(buf.act, np.int64)
    out["rew"] = np.asarray(buf.rew, np.float64)
    out["terminated"] = np.asarray(buf.terminated, bool)
    out["truncated"] = np.asarray(buf.truncated, bool)
    for k, v in manager_state(buf).items():
        out["buf_" + k] = v

    rec: list[dict] = []
    orig_pre, orig_upd = DiscreteSAC._preprocess_batch, DiscreteSAC._update_with_batch

    def rec_pre(self, batch, buffer, indices):
        r = {"indices": np.array(indices, np.int64), "is_weight": np.array(batch.weight, np.float64)}
        b = orig_pre(self, batch, buffer, indices)
        r["returns"] = b.returns.numpy().copy().reshape(-1)
        rec.append(r)
        return b

    def rec_upd(self, batch):
        stats = orig_upd(self, batch)
        rec[-1]["new_weight"] = batch.weight.detach().numpy().copy()
        return stats

    DiscreteSAC._preprocess_batch, DiscreteSAC._update_with_batch = rec_pre, rec_upd
    try:
        np.random.seed(seed + 7)
        for u in range(n_updates):
            with policy_within_training_step(algorithm.policy):
                stats = algorithm.update(buffer=buf, sample_size=batch)
            for k in ("indices", "returns", "is_weight", "new_weight"):
                out[f"u{u}_{k}"] = rec[-1][k]
            out[f"u{u}_stats"] = np.array([stats.actor_loss, stats.critic1_loss, stats.critic2_loss,
                                           stats.alpha if stats.alpha is not None else np.nan,
                                           stats.alpha_loss if stats.alpha_loss is not None else np.nan])
            for name, mod in (("actor", actor), ("critic1", critic1), ("critic2", critic2),
                              ("critic1_old", algorithm.critic_old.module), ("critic2_old", algorithm.critic2_old.module)):
                sd = mod.state_dict()
                out[f"u{u}_{name}"] = torch.cat([sd[k].reshape(-1) for k in DK]).numpy()[::5].copy()
    finally:
        DiscreteSAC._preprocess_batch, DiscreteSAC._update_with_batch = orig_pre, orig_upd
    cfg = dict(gamma=gamma, tau=tau, n_step=n_step, alpha=alpha, auto_alpha=float(auto_alpha),
               target_entropy=target_entropy, log_alpha0=0.0, actor_lr=actor_lr, critic_lr=critic_lr, alpha_lr=alpha_lr)
    out["cfg_keys"] = np.array(list(cfg.keys()))
    out["cfg_vals"] = np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"dsac_{tag}.npz"), **out)


def gen_dsac_all() -> None:
    gen_dsac("auto", E=3, slots=40, steps=60, obs_dim=11, n_act=5, hidden=64, batch=48, n_updates=3, seed=21,
             auto_alpha=True, n_step=3)
    gen_dsac("fixed", E=2, slots=40, steps=50, obs_dim=40, n_act=3, hidden=96, batch=32, n_updates=2, seed=23,
             auto_alpha=False, alpha=0.05, n_step=1, tau=0.01)


def gen_redq(tag: str, *, E: int, slots: int, steps: int, obs_dim: int, act_dim: int, batch: int, n_updates: int, seed: int,
             ensemble: int, subset: int, actor_delay: int, target_mode: str, auto_alpha: bool, alpha: float = 0.2,
             n_step: int = 1, tau: float = 0.005, gamma: float = 0.99, actor_lr: float = 1e-3, critic_lr: float = 1e-3,
             alpha_lr: float = 3e-4, hidden=256) -> None:
    """Runs the reference REDQ.update() (nets as in test/continuous/test_redq.py:86-107, hidden [256, 256]) on a synthetic
    VectorReplayBuffer, recording the rsample() noise, the np.random.choice subsets and the outputs of every update."""
    import torch.distributions.normal as tdn
    from tianshou.algorithm.modelfree.redq import REDQ, REDQPolicy
    from tianshou.algorithm.modelfree.sac import AutoAlpha
    from tianshou.utils.net.common import EnsembleLinear
    from oracle import oracle_redq as OR
    from oracle import oracle_sac as OS

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    sa_, sc_ = OS.layer_sizes(hidden)       # (any depth since round 6: a nested (actor sizes, critic sizes) pair)
    hw = tuple(sa_) + tuple(sc_)
    AK = OS.trunk_keys(len(sa_), ("mu", "sigma"))
    CK = [k.replace(".bias", ".bias_weights") for k in OS.trunk_keys(len(sc_), ("last",))]
    net_a = Net(state_shape=(obs_dim,), hidden_sizes=list(sa_))
    actor = ContinuousActorProbabilistic(preprocess_net=net_a, action_shape=(act_dim,), unbounded=True,
                                         conditioned_sigma=True)

    def linear(x: int, y: int):
        return EnsembleLinear(ensemble, x, y)

    net_c = Net(state_shape=(obs_dim,), action_shape=(act_dim,), hidden_sizes=list(sc_), concat=True, linear_layer=linear)
    critic = ContinuousCritic(preprocess_net=net_c, linear_layer=linear, flatten_input=False)
    space = gym.spaces.Box(low=-1.0, high=1.0, shape=(act_dim,))
    policy = REDQPolicy(actor=actor, action_space=space)
    al = AutoAlpha(float(-act_dim), 0.0, AdamOptimizerFactory(lr=alpha_lr)) if auto_alpha else alpha
    algorithm = REDQ(policy=policy, policy_optim=AdamOptimizerFactory(lr=actor_lr), critic=critic,
                     critic_optim=AdamOptimizerFactory(lr=critic_lr), ensemble_size=ensemble, subset_size=subset,
                     tau=tau, gamma=gamma, alpha=al, n_step_return_horizon=n_step, actor_delay=actor_delay,
                     target_mode=target_mode)
    out: dict[str, np.ndarray] = {}
    out["dims"] = np.array([E, slots, steps, obs_dim, act_dim, batch, n_updates, seed, int(auto_alpha), n_step, ensemble,
                            subset, actor_delay, int(target_mode == "mean")])
    if len(sa_) != 2 or len(sc_) != 2:
        out["hidden_actor"], out["hidden_critic"] = np.array(sa_, np.int64), np.array(sc_, np.int64)
    elif hidden != 256:
        out["hidden"] = np.array(hw, np.int64)
    a0, c0 = OR.init_params(obs_dim, act_dim, ensemble, seed, (sa_, sc_))
    sa, sc = actor.state_dict(), critic.state_dict()
    assert list(sc.keys()) == CK, list(sc.keys())
    for k_ref, k in zip(AK, OS.actor_order(len(sa_))):
        assert torch.equal(sa[k_ref], a0[k]), f"oracle actor init differs at {k}"
    for k_ref, k in zip(CK, OS.critic_order(len(sc_))):
        assert torch.equal(sc[k_ref], c0[k]), f"oracle critic init differs at {k}"

    buf = VectorReplayBuffer(E * slots, E)
    obs = rng.normal(size=(steps + 1, E, obs_dim)).astype(np.float32)
    act = rng.uniform(-1, 1, size=(steps, E, act_dim)).astype(np.float32)
    rew = rng.normal(size=(steps, E)).astype(np.float32)
    term = rng.random((steps, E)) < 0.05
    trunc = (rng.random((steps, E)) < 0.03) & ~term
    for t in range(steps):
        buf.add(Batch(obs=obs[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t], obs_next=obs[t + 1]))
    out["obs"], out["obs_next"] = np.asarray(buf.obs, np.float32), np.asarray(buf.obs_next, np.float32)
    out["act"], out["rew"] = np.asarray(buf.act, np.float32), np.asarray(buf.rew, np.float64)
    out["terminated"], out["truncated"] = np.asarray(buf.terminated, bool), np.asarray(buf.truncated, bool)
    for k, v in manager_state(buf).items():
        out["buf_" + k] = v

    noises: list[np.ndarray] = []
    subsets: list[np.ndarray] = []
    orig_sn, orig_choice, orig_pre = tdn._standard_normal, np.random.choice, REDQ._preprocess_batch

    def rec_sn(shape, dtype, device):
        e = orig_sn(shape, dtype, device)
        noises.append(e.numpy().copy())
        return e

    def rec_choice(*a, **k):
        r = orig_choice(*a, **k)
        if k.get("replace", True) is False and len(a) == 2 and a[0] == ensemble:
            subsets.append(np.asarray(r, np.int64).copy())
        return r

    rec: list[dict] = []

    def rec_pre(self, batch, buffer, indices):
        b = orig_pre(self, batch, buffer, indices)
        rec.append({"indices": np.array(indices, np.int64), "returns": b.returns.numpy().copy().reshape(-1)})
        return b

    tdn._standard_normal, np.random.choice, REDQ._preprocess_batch = rec_sn, rec_choice, rec_pre
    try:
        np.random.seed(seed + 3)
        for u in range(n_updates):
            n0, s0 = len(noises), len(subsets)
            with policy_within_training_step(algorithm.policy):
                stats = algorithm.update(buffer=buf, sample_size=batch)
            did_actor = (u + 1) % actor_delay == 0
            assert len(noises) - n0 == (2 if did_actor else 1) and len(subsets) - s0 == 1
            out[f"u{u}_noise_target"] = noises[n0]
            if did_actor:
                out[f"u{u}_noise_actor"] = noises[n0 + 1]
            out[f"u{u}_subset"] = subsets[s0]
            out[f"u{u}_indices"], out[f"u{u}_returns"] = rec[-1]["indices"], rec[-1]["returns"]
            out[f"u{u}_stats"] = np.array([stats.actor_loss, stats.critic_loss, stats.alpha,
                                           stats.alpha_loss if stats.alpha_loss is not None else np.nan])
            sa, sc, so = actor.state_dict(), critic.state_dict(), algorithm.critic_old.module.state_dict()
            out[f"u{u}_actor"] = torch.cat([sa[k].reshape(-1) for k in AK]).numpy()[::61].copy()
            out[f"u{u}_critic"] = torch.cat([sc[k].reshape(-1) for k in CK]).numpy()[::61].copy()
            out[f"u{u}_critic_old"] = torch.cat([so[k].reshape(-1) for k in CK]).numpy()[::61].copy()
    finally:
        tdn._standard_normal, np.random.choice, REDQ._preprocess_batch = orig_sn, orig_choice, orig_pre
    cfg = dict(gamma=gamma, tau=tau, n_step=n_step, alpha=alpha, auto_alpha=float(auto_alpha),
               target_entropy=float(-act_dim), log_alpha0=0.0, actor_lr=actor_lr, critic_lr=critic_lr, alpha_lr=alpha_lr)
    out["cfg_keys"] = np.array(list(cfg.keys()))
    out["cfg_vals"] = np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"redq_{tag}.npz"), **out)


def gen_redq_all() -> None:
    gen_redq("min", E=3, slots=30, steps=40, obs_dim=11, act_dim=3, batch=32, n_updates=4, seed=31, ensemble=4, subset=2,
             actor_delay=2, target_mode="min", auto_alpha=True, n_step=3)
    gen_redq("mean", E=2, slots=30, steps=40, obs_dim=17, act_dim=6, batch=24, n_updates=3, seed=33, ensemble=3, subset=3,
             actor_delay=1, target_mode="mean", auto_alpha=False, alpha=0.1, n_step=1, tau=0.01)


def gen_td3_all() -> None:
    gen_td3("twin", twin=True, E=4, slots=32, steps=30, obs_dim=23, act_dim=5, batch=64, n_updates=4, seed=12,
            max_action=1.0, actor_lr=3e-4, critic_lr=1e-3)
    gen_td3("ddpg", twin=False, E=2, slots=40, steps=40, obs_dim=11, act_dim=3, batch=48, n_updates=3, seed=13,
            max_action=2.0, n_step=2, tau=0.01, gamma=0.97)


def gen_widths() -> None:
    """Round 6: two-hidden-layer networks of unequal widths / widths that are no multiple of 32 (the engines embed them by zero
    padding, tianshou_amd/widths.py): SAC with actor [48, 80] and critics [72, 40]; TD3 with the papers' [400, 300]; DDPG with
    [24, 56]."""
    gen_sac("widths", E=4, slots=32, steps=30, obs_dim=23, act_dim=5, batch=64, n_updates=3, seed=21, auto_alpha=True,
            hidden=(48, 80, 72, 40))
    gen_td3("widths", twin=True, E=4, slots=32, steps=30, obs_dim=17, act_dim=6, batch=64, n_updates=4, seed=22,
            max_action=1.0, actor_lr=3e-4, critic_lr=1e-3, hidden=(400, 300))
    gen_td3("ddpg_widths", twin=False, E=2, slots=40, steps=40, obs_dim=11, act_dim=3, batch=48, n_updates=3, seed=23,
            max_action=2.0, n_step=2, tau=0.01, gamma=0.97, hidden=(24, 56, 40, 24))
    gen_dsac("widths", E=3, slots=30, steps=40, obs_dim=13, n_act=5, hidden=(40, 72, 56, 24), batch=32, n_updates=3, seed=24,
             auto_alpha=True, n_step=2)
    gen_redq("widths", E=3, slots=30, steps=40, obs_dim=11, act_dim=3, batch=32, n_updates=4, seed=25, ensemble=4, subset=2,
             actor_delay=2, target_mode="min", auto_alpha=True, n_step=2, hidden=(48, 80, 72, 40))
    gen_npg("npg_widths", algo="npg", E=4, T=64, obs_dim=17, act_dim=6, batch_size=128, repeat=2, seed=26, optim_critic_iters=3,
            trust_region_size=0.1, advantage_normalization=True, gae_lambda=0.95, gamma=0.99, return_scaling=True,
            max_batchsize=64, hidden_a=(48, 80), hidden_c=(40, 56))
    gen_npg("trpo_widths", algo="trpo", E=4, T=64, obs_dim=17, act_dim=6, batch_size=128, repeat=2, seed=27, optim_critic_iters=2,
            max_kl=0.01, backtrack_coeff=0.8, max_backtracks=10, advantage_normalization=True, gae_lambda=0.95, gamma=0.99,
            return_scaling=False, max_batchsize=256, hidden_a=(100, 60), hidden_c=(60, 100))


def gen_depth() -> None:
    """Round 6: trunks of other depths than two hidden layers (Net(hidden_sizes=[...]) takes any list; the engines run them layer
    by layer on the GEMM kernels, `ts_mlp_set_trunk`): SAC with a three-layer actor [64, 48, 32] and three-layer critics
    [40, 56, 24] (unequal widths, none but one a multiple of 32: embedded by zero padding) and SAC with ONE hidden layer [96];
    TD3 with four layers; DDPG with one."""
    gen_sac("depth3", E=4, slots=32, steps=30, obs_dim=23, act_dim=5, batch=64, n_updates=3, seed=31, auto_alpha=True,
            hidden=((64, 48, 32), (40, 56, 24)))
    gen_sac("depth1", E=3, slots=30, steps=30, obs_dim=11, act_dim=3, batch=48, n_updates=3, seed=32, auto_alpha=False, alpha=0.15,
            n_step=2, hidden=((96,), (96,)))
    gen_td3("depth4", twin=True, E=4, slots=32, steps=30, obs_dim=17, act_dim=6, batch=64, n_updates=4, seed=33,
            hidden=((64, 64, 32, 32), (48, 64, 64, 40)), max_action=1.5)
    gen_td3("ddpg_depth1", twin=False, E=2, slots=40, steps=40, obs_dim=11, act_dim=3, batch=48, n_updates=3, seed=34,
            hidden=((128,), (64,)))
    # NPG on three ReLU layers (actor [64, 48, 32], critic [40, 56]: different depths too); TRPO on one tanh layer [96] / [80]
    gen_npg("npg_relu3", algo="npg", E=4, T=64, obs_dim=17, act_dim=6, batch_size=128, repeat=2, seed=28, optim_critic_iters=3,
            trust_region_size=0.1, advantage_normalization=True, gae_lambda=0.95, gamma=0.99, return_scaling=True, max_batchsize=64,
            hidden_a=(64, 48, 32), hidden_c=(40, 56), activation=nn.ReLU)
    gen_npg("trpo_tanh1", algo="trpo", E=4, T=64, obs_dim=11, act_dim=3, batch_size=128, repeat=2, seed=29, optim_critic_iters=2,
            max_kl=0.01, backtrack_coeff=0.8, max_backtracks=10, advantage_normalization=True, gae_lambda=0.95, gamma=0.99,
            return_scaling=False, max_batchsize=256, hidden_a=(96,), hidden_c=(80,))
    # SAC with the class-default BOUNDED actor (unbounded=False, max_action 1.5) on a two-layer and a three-layer trunk
    gen_sac("bounded", E=4, slots=32, steps=30, obs_dim=23, act_dim=5, batch=64, n_updates=3, seed=37, auto_alpha=True, max_action=1.5)
    gen_sac("bounded_depth3", E=3, slots=30, steps=30, obs_dim=11, act_dim=3, batch=48, n_updates=3, seed=38, auto_alpha=False,
            alpha=0.1, hidden=((48, 64, 40), (64, 32, 32)), max_action=0.8)
    # nn.Tanh trunks (Net's default is nn.ReLU): SAC on actor [64, 48] / critics [40, 72]; TD3 on three layers
    gen_sac("tanh", E=4, slots=32, steps=30, obs_dim=23, act_dim=5, batch=64, n_updates=3, seed=39, auto_alpha=True,
            hidden=(64, 48, 40, 72), activation=nn.Tanh)
    gen_td3("tanh3", twin=True, E=4, slots=32, steps=30, obs_dim=17, act_dim=6, batch=64, n_updates=4, seed=40,
            hidden=((64, 32, 32), (48, 64, 40)), activation=nn.Tanh)
    # DiscreteSAC with three hidden layers (actor [40, 72, 24], critics [56, 24, 48]); REDQ with one (actor [64], ensemble [48])
    gen_dsac("depth3", E=3, slots=30, steps=40, obs_dim=13, n_act=5, hidden=((40, 72, 24), (56, 24, 48)), batch=32, n_updates=3, seed=35,
             auto_alpha=True)
    gen_redq("depth1", E=3, slots=30, steps=40, obs_dim=11, act_dim=3, batch=32, n_updates=4, seed=36, ensemble=4, subset=2,
             actor_delay=2, target_mode="mean", auto_alpha=True, hidden=((64,), (48,)))


def gen_sac_all() -> None:
    gen_sac("auto", E=4, slots=32, steps=30, obs_dim=23, act_dim=5, batch=64, n_updates=3, seed=4, auto_alpha=True)
    gen_sac("fixed", E=2, slots=40, steps=40, obs_dim=376, act_dim=17, batch=48, n_updates=2, seed=6,
            auto_alpha=False, alpha=0.2, n_step=3, tau=0.01, gamma=0.97, actor_lr=3e-4)


def gen_ppo_cnn(tag: str = "cnn", *, E: int = 3, T: int = 20, c: int = 2, h: int = 44, w: int = 36, n_act: int = 4,
                batch_size: int = 16, repeat: int = 2, seed: int = 8, **ppo_kwargs) -> None:
    """Runs the reference PPO.update() with the Atari actor-critic of examples/atari/atari_ppo.py:106-118 (shared
    DQNet feature trunk, DiscreteActor / DiscreteCritic, Categorical policy) on a synthetic VectorReplayBuffer."""
    from tianshou.algorithm.modelfree.reinforce import DiscreteActorPolicy
    from tianshou.env.atari.atari_network import DQNet
    from tianshou.utils.net.discrete import DiscreteActor, DiscreteCritic
    from oracle import oracle_ppo_cnn as OC

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    net = DQNet(c=c, h=h, w=w, action_shape=n_act, features_only=True, output_dim_added_layer=512)
    actor = DiscreteActor(preprocess_net=net, action_shape=n_act, softmax_output=False)
    critic = DiscreteCritic(preprocess_net=net)
    p0 = OC.init_params(c, h, w, n_act, seed)
    sa, sc = actor.state_dict(), critic.state_dict()
    ref = [sa[k] for k in OC.TRUNK_KEYS] + [sa[k] for k in OC.HEAD_KEYS] + [sc[k] for k in OC.HEAD_KEYS]
    for t, k in zip(ref, OC.PARAM_ORDER):
        assert torch.equal(t, p0[k]), f"oracle init differs from the reference at {k}"
    policy = DiscreteActorPolicy(actor=actor, action_space=gym.spaces.Discrete(n_act))
    lr = ppo_kwargs.pop("lr", 2.5e-4)
    algorithm = PPO(policy=policy, critic=critic, optim=AdamOptimizerFactory(lr=lr, eps=1e-5), **ppo_kwargs)

    def flat():
        sa, sc = actor.state_dict(), critic.state_dict()
        ts = [sa[k] for k in OC.TRUNK_KEYS] + [sa[k] for k in OC.HEAD_KEYS] + [sc[k] for k in OC.HEAD_KEYS]
        return torch.cat([t.reshape(-1) for t in ts]).numpy().copy()

    N = E * T
    buf = VectorReplayBuffer(N, E)
    frames = rng.integers(0, 256, size=(T + 1, E, c, h, w), dtype=np.uint8)
    frames = np.where(rng.random(frames.shape) < 0.08, frames, 0).astype(np.uint8)
    act = rng.integers(0, n_act, size=(T, E))
    rew = rng.normal(size=(T, E)).astype(np.float32)
    term = rng.random((T, E)) < 0.06
    trunc = (rng.random((T, E)) < 0.04) & ~term
    for t in range(T):
        buf.add(Batch(obs=frames[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t],
                      obs_next=frames[t + 1]))
    out: dict[str, np.ndarray] = {"dims": np.array([E, T, c, h, w, n_act, batch_size, repeat, seed])}
    out["obs"], out["obs_next"] = np.asarray(buf.obs, np.uint8), np.asarray(buf.obs_next, np.uint8)
    out["act"], out["rew"] = np.asarray(buf.act, np.int64), np.asarray(buf.rew, np.float64)
    out["terminated"], out["truncated"] = np.asarray(buf.terminated, bool), np.asarray(buf.truncated, bool)
    for k, v in manager_state(buf).items():
        out["buf_" + k] = v

    perms, seqs, pre_dump = [], [], {}
    orig_perm, orig_from, orig_pre = np.random.permutation, SequenceSummaryStats.from_sequence.__func__, PPO._preprocess_batch

    def rec_perm(n):
        p = orig_perm(n)
        perms.append(np.asarray(p, np.int64))
        return p

    def rec_from(cls, seq):
        seqs.append(np.asarray(seq, np.float64))
        return orig_from(cls, seq)

    def rec_pre(self, batch, buffer, indices):
        b = orig_pre(self, batch, buffer, indices)
        pre_dump.update(v_s=b.v_s.numpy().copy(), returns=b.returns.numpy().copy(), adv=b.adv.numpy().copy(),
                        logp_old=b.logp_old.numpy().copy(), indices=np.asarray(indices, np.int64),
                        unfinished=np.asarray(buffer.unfinished_index(), np.int64))
        return b

    np.random.permutation, PPO._preprocess_batch = rec_perm, rec_pre
    SequenceSummaryStats.from_sequence = classmethod(rec_from)
    try:
        np.random.seed(seed + 100)
        with policy_within_training_step(algorithm.policy):
            stats = algorithm.update(buffer=buf, batch_size=batch_size, repeat=repeat)
    finally:
        np.random.permutation, PPO._preprocess_batch = orig_perm, orig_pre
        SequenceSummaryStats.from_sequence = classmethod(orig_from)
    assert len(perms) == repeat and len(seqs) == 4
    out["perms"], out["losses"] = np.stack(perms), np.stack(seqs, axis=1)
    out["gradient_steps"] = np.array(stats.gradient_steps)
    out["params_strided"] = flat()[::17]
    out["ret_rms"] = np.array([float(algorithm.ret_rms.mean), float(algorithm.ret_rms.var), float(algorithm.ret_rms.count)])
    for k, v in pre_dump.items():
        out["pre_" + k] = v
    cfg = dict(gamma=algorithm.gamma, gae_lambda=algorithm.gae_lambda, eps_clip=algorithm.eps_clip,
               dual_clip=algorithm.dual_clip or 0.0, value_clip=float(algorithm.value_clip),
               advantage_normalization=float(algorithm.advantage_normalization), vf_coef=algorithm.vf_coef,
               ent_coef=algorithm.ent_coef, max_grad_norm=algorithm.optim._max_grad_norm or 0.0,
               return_scaling=float(algorithm.return_scaling), lr=lr, adam_eps=1e-5,
               max_batchsize=float(algorithm.max_batchsize))
    out["cfg_keys"], out["cfg_vals"] = np.array(list(cfg.keys())), np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"ppo_{tag}.npz"), **out)


def gen_ppo_discrete(tag: str, *, E: int, T: int, obs_dim: int, hidden: int, n_act: int, batch_size: int, repeat: int,
                     seed: int, softmax_output: bool, lr: float = 3e-4, algo: str = "ppo", **ppo_kwargs) -> None:
    """Runs the reference PPO.update() with the CartPole-shape networks of test/discrete/test_ppo_discrete.py:88-127
    (BASELINE.json configs[0]): Net(obs, [h, h]) shared by DiscreteActor and DiscreteCritic, orthogonal init."""
    from tianshou.algorithm.modelfree.reinforce import DiscreteActorPolicy
    from tianshou.utils.net.common import ActorCritic
    from tianshou.utils.net.discrete import DiscreteActor, DiscreteCritic
    from oracle import oracle_ppo_discrete as OD

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    net = Net(state_shape=(obs_dim,), hidden_sizes=[hidden, hidden])
    actor = DiscreteActor(preprocess_net=net, action_shape=n_act, softmax_output=softmax_output)
    critic = DiscreteCritic(preprocess_net=net)
    for m in ActorCritic(actor, critic).modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)

    def tensors():
        sa, sc = actor.state_dict(), critic.state_dict()
        return [sa[k] for k in OD.TRUNK_KEYS] + [sa[k] for k in OD.HEAD_KEYS] + [sc[k] for k in OD.HEAD_KEYS]

    p0 = OD.init_params(obs_dim, hidden, n_act, seed)
    for t, k in zip(tensors(), OD.PARAM_ORDER):
        assert torch.equal(t, p0[k]), f"oracle init differs from the reference at {k}"
    if softmax_output:                # test_ppo_discrete.py:104-110: Categorical receives the probabilities
        policy = DiscreteActorPolicy(actor=actor, dist_fn=torch.distributions.Categorical,
                                     action_space=gym.spaces.Discrete(n_act))
    else:
        policy = DiscreteActorPolicy(actor=actor, action_space=gym.spaces.Discrete(n_act))
    from tianshou.algorithm.modelfree.a2c import A2C

    cls = PPO if algo == "ppo" else A2C
    algorithm = cls(policy=policy, critic=critic, optim=AdamOptimizerFactory(lr=lr), **ppo_kwargs)

    N = E * T
    buf = VectorReplayBuffer(N, E)
    obs = rng.normal(size=(T + 1, E, obs_dim)).astype(np.float32)
    act = rng.integers(0, n_act, size=(T, E))
    rew = rng.normal(size=(T, E)).astype(np.float32)
    term = rng.random((T, E)) < 0.06
    trunc = (rng.random((T, E)) < 0.04) & ~term
    for t in range(T):
        buf.add(Batch(obs=obs[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t], obs_next=obs[t + 1]))
    out: dict[str, np.ndarray] = {"dims": np.array([E, T, obs_dim, hidden, n_act, batch_size, repeat, seed,
                                                    int(softmax_output)])}
    out["obs"], out["obs_next"] = np.asarray(buf.obs, np.float32), np.asarray(buf.obs_next, np.float32)
    out["act"], out["rew"] = np.asarray(buf.act, np.int64), np.asarray(buf.rew, np.float64)
    out["terminated"], out["truncated"] = np.asarray(buf.terminated, bool), np.asarray(buf.truncated, bool)
    for k, v in manager_state(buf).items():
        out["buf_" + k] = v

    perms, seqs, pre_dump = [], [], {}
    orig_perm, orig_from, orig_pre = np.random.permutation, SequenceSummaryStats.from_sequence.__func__, cls._preprocess_batch

    def rec_perm(n):
        p = orig_perm(n)
        perms.append(np.asarray(p, np.int64))
        return p

    def rec_from(cls, seq):
        seqs.append(np.asarray(seq, np.float64))
        return orig_from(cls, seq)

    def rec_pre(self, batch, buffer, indices):
        b = orig_pre(self, batch, buffer, indices)
        pre_dump.update(v_s=b.v_s.numpy().copy(), returns=b.returns.numpy().copy(), adv=b.adv.numpy().copy(),
                        indices=np.asarray(indices, np.int64), unfinished=np.asarray(buffer.unfinished_index(), np.int64))
        if algo == "ppo":
            pre_dump["logp_old"] = b.logp_old.numpy().copy()
        return b

    np.random.permutation, cls._preprocess_batch = rec_perm, rec_pre
    SequenceSummaryStats.from_sequence = classmethod(rec_from)
    try:
        np.random.seed(seed + 100)
        with policy_within_training_step(algorithm.policy):
            stats = algorithm.update(buffer=buf, batch_size=batch_size, repeat=repeat)
    finally:
        np.random.permutation, cls._preprocess_batch = orig_perm, orig_pre
        SequenceSummaryStats.from_sequence = classmethod(orig_from)
    assert len(perms) == repeat and len(seqs) == 4
    out["perms"], out["losses"] = np.stack(perms), np.stack(seqs, axis=1)
    out["gradient_steps"] = np.array(stats.gradient_steps)
    out["params"] = torch.cat([t.reshape(-1) for t in tensors()]).numpy().copy()
    out["ret_rms"] = np.array([float(algorithm.ret_rms.mean), float(algorithm.ret_rms.var), float(algorithm.ret_rms.count)])
    for k, v in pre_dump.items():
        out["pre_" + k] = v
    cfg = dict(gamma=algorithm.gamma, gae_lambda=algorithm.gae_lambda, eps_clip=getattr(algorithm, "eps_clip", 0.2),
               dual_clip=getattr(algorithm, "dual_clip", None) or 0.0, value_clip=float(getattr(algorithm, "value_clip", False)),
               advantage_normalization=float(getattr(algorithm, "advantage_normalization", False)), vf_coef=algorithm.vf_coef,
               ent_coef=algorithm.ent_coef, max_grad_norm=algorithm.optim._max_grad_norm or 0.0,
               return_scaling=float(algorithm.return_scaling), lr=lr, adam_eps=1e-8,
               max_batchsize=float(algorithm.max_batchsize), a2c=float(algo == "a2c"))
    out["cfg_keys"], out["cfg_vals"] = np.array(list(cfg.keys())), np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"ppo_discrete_{tag}.npz"), **out)


def gen_ppo_discrete_all() -> None:
    # BASELINE.json configs[0] (test/discrete/test_ppo_discrete.py defaults): CartPole shape, MLP[64, 64], batch 64
    gen_ppo_discrete("c1", E=4, T=60, obs_dim=4, hidden=64, n_act=2, batch_size=64, repeat=3, seed=1626,
                     softmax_output=True, gamma=0.99, gae_lambda=0.95, max_grad_norm=0.5, vf_coef=0.5, ent_coef=0.0,
                     eps_clip=0.2, return_scaling=False, value_clip=False, dual_clip=None,
                     advantage_normalization=False, recompute_advantage=False)
    # every option on, logits actor, other widths
    gen_ppo_discrete("opts", E=3, T=50, obs_dim=9, hidden=96, n_act=5, batch_size=40, repeat=2, seed=7,
                     softmax_output=False, gamma=0.97, gae_lambda=0.9, max_grad_norm=0.7, vf_coef=0.25, ent_coef=0.01,
                     eps_clip=0.15, return_scaling=True, value_clip=True, dual_clip=3.0,
                     advantage_normalization=True, recompute_advantage=False, max_batchsize=64)
    # A2C (a2c.py:249-290) on the same networks
    gen_ppo_discrete("a2c", E=3, T=40, obs_dim=6, hidden=64, n_act=3, batch_size=32, repeat=2, seed=9, algo="a2c",
                     softmax_output=True, gamma=0.98, gae_lambda=0.92, max_grad_norm=0.5, vf_coef=0.5, ent_coef=0.02,
                     return_scaling=True, max_batchsize=64)


def gen_buffer_add() -> None:
    """Random VectorReplayBuffer.add histories (ragged buffer_ids, ring wrap, episode ends): the tuple every
    add() returns (manager.py:193-198) and the final buffer contents."""
    out: dict[str, np.ndarray] = {}
    rng = np.random.default_rng(99)
    for s, (total, E, steps, p_done, obs_dim) in enumerate([(20, 4, 17, 0.3, 3), (64, 8, 60, 0.1, 5), (15, 3, 9, 0.0, 2),
                                                           (36, 6, 80, 0.45, 4), (8, 1, 30, 0.2, 1)]):
        buf = VectorReplayBuffer(total, E)
        ids_l, cnt, rets = [], [], []
        cols = {k: [] for k in ("rew", "term", "trunc", "obs", "act", "obs_next")}
        for t in range(steps):
            k = E if t % 3 == 0 else int(rng.integers(1, E + 1))
            ids = np.arange(E) if k == E else np.sort(rng.choice(E, size=k, replace=False))
            term = rng.random(k) < p_done
            trunc = (rng.random(k) < p_done / 3) & ~term
            b = Batch(obs=rng.normal(size=(k, obs_dim)).astype(np.float32), act=rng.integers(0, 5, size=k),
                      rew=rng.normal(size=k), terminated=term, truncated=trunc,
                      obs_next=rng.normal(size=(k, obs_dim)).astype(np.float32))
            r = buf.add(b, buffer_ids=None if (k == E and t % 2 == 0) else ids)
            ids_l.append(ids); cnt.append(k)
            rets.append(np.stack([np.asarray(x, np.float64) for x in r], axis=1))
            for key, v in zip(cols, (b.rew, term, trunc, b.obs, b.act, b.obs_next)):
                cols[key].append(np.asarray(v))
        out[f"s{s}_dims"] = np.array([total, E, steps, obs_dim])
        out[f"s{s}_counts"] = np.array(cnt)
        out[f"s{s}_ids"] = np.concatenate(ids_l)
        out[f"s{s}_returned"] = np.concatenate(rets)          # [sum k, 4]: ptr, ep_rew, ep_len, ep_idx
        for key, v in cols.items():
            out[f"s{s}_in_{key}"] = np.concatenate(v)
        for k2, v in manager_state(buf).items():
            out[f"s{s}_final_{k2}"] = v
        for key in ("obs", "act", "rew", "terminated", "truncated", "done", "obs_next"):
            out[f"s{s}_final_{key}"] = np.asarray(getattr(buf, key))
    out["n_scen"] = np.array(5)
    np.savez_compressed(os.path.join(OUT, "buffer_add.npz"), **out)


def gen_distq(kind: str, *, E: int, slots: int, steps: int, c: int, h: int, w: int, n_act: int, n_atoms: int,
              batch: int, n_updates: int, seed: int, lr: float, v_min: float = -2.0, v_max: float = 3.0,
              **algo_kwargs) -> None:
    """Runs the reference QRDQN.update() / C51.update() (QRDQNet / C51Net, qrdqn.py, c51.py) on a synthetic
    PrioritizedVectorReplayBuffer that stores obs and obs_next, and dumps indices, n-step returns
    [B, n_atoms], the new priorities, losses and parameter samples of every update."""
    from tianshou.algorithm.modelfree.c51 import C51, C51Policy
    from tianshou.algorithm.modelfree.qrdqn import QRDQN, QRDQNPolicy
    from tianshou.env.atari.atari_network import C51Net, QRDQNet
    from oracle import oracle_distq as OQ
    from oracle import oracle_dqn as OD

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    if kind == "qr":
        net = QRDQNet(c=c, h=h, w=w, action_shape=[n_act], num_quantiles=n_atoms)
        policy = QRDQNPolicy(model=net, action_space=gym.spaces.Discrete(n_act))
        algorithm = QRDQN(policy=policy, optim=AdamOptimizerFactory(lr=lr), num_quantiles=n_atoms, **algo_kwargs)
        cls = QRDQN
    else:
        net = C51Net(c=c, h=h, w=w, action_shape=[n_act], num_atoms=n_atoms)
        policy = C51Policy(model=net, action_space=gym.spaces.Discrete(n_act), num_atoms=n_atoms, v_min=v_min,
                           v_max=v_max)
        algorithm = C51(policy=policy, optim=AdamOptimizerFactory(lr=lr), **algo_kwargs)
        cls = C51
    p0 = OQ.init_params(c, h, w, n_act, n_atoms, seed)
    sd = net.state_dict()
    for k_ref, k in zip(OD.TIANSHOU_KEYS, OD.PARAM_ORDER):
        assert torch.equal(sd[k_ref], p0[k]), f"oracle init differs from the reference net at {k}"

    buf = PrioritizedVectorReplayBuffer(E * slots, E, alpha=0.6, beta=0.4)
    frames = rng.integers(0, 256, size=(steps + 1, E, c, h, w), dtype=np.uint8)
    frames = np.where(rng.random(frames.shape) < 0.06, frames, 0).astype(np.uint8)
    act = rng.integers(0, n_act, size=(steps, E))
    rew = rng.normal(size=(steps, E)).astype(np.float32)
    term = rng.random((steps, E)) < 0.08
    trunc = (rng.random((steps, E)) < 0.04) & ~term
    for t in range(steps):
        buf.add(Batch(obs=frames[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t],
                      obs_next=frames[t + 1]))
    out: dict[str, np.ndarray] = {}
    out["dims"] = np.array([E, slots, steps, c, h, w, n_act, n_atoms, batch, n_updates, seed])
    out["frames"] = np.asarray(buf.obs, np.uint8)
    out["frames_next"] = np.asarray(buf.obs_next, np.uint8)
    out["act"] = np.asarray(buf.act, np.int64)
    out["rew"] = np.asarray(buf.rew, np.float64)
    out["terminated"] = np.asarray(buf.terminated, bool)
    out["truncated"] = np.asarray(buf.truncated, bool)
    for k, v in manager_state(buf).items():
        out["buf_" + k] = v
    out["tree0"] = np.asarray(buf.weight._value, np.float64).copy()

    rec: list[dict] = []
    orig_pre, orig_upd = cls._preprocess_batch, cls._update_with_batch

    def rec_pre(self, batch, buffer, indices):
        r = {"indices": np.array(indices, np.int64), "is_weight": np.array(batch.weight, np.float64)}
        b = orig_pre(self, batch, buffer, indices)
        r["returns"] = b.returns.numpy().copy()
        rec.append(r)
        return b

    def rec_upd(self, batch):
        stats = orig_upd(self, batch)
        rec[-1]["prio"] = batch.weight.detach().numpy().copy()
        loss = stats.loss
        rec[-1]["loss"] = np.array(loss.mean if hasattr(loss, "mean") and not isinstance(loss, float) else loss)
        return stats

    cls._preprocess_batch, cls._update_with_batch = rec_pre, rec_upd
    try:
        np.random.seed(seed + 7)
        for u in range(n_updates):
            with policy_within_training_step(algorithm.policy):
                algorithm.update(buffer=buf, sample_size=batch)
            r = rec[-1]
            flat = torch.cat([net.state_dict()[k].reshape(-1) for k in OD.TIANSHOU_KEYS]).numpy()
            for k in ("indices", "returns", "prio", "loss", "is_weight"):
                out[f"u{u}_{k}"] = r[k]
            out[f"u{u}_params_strided"] = flat[::61].copy()
            out[f"u{u}_conv1_w"] = net.state_dict()["net.0.0.weight"].numpy().copy()
            out[f"u{u}_fc2_w_strided"] = net.state_dict()["net.3.weight"].numpy().reshape(-1)[::7].copy()
            out[f"u{u}_biases"] = torch.cat([net.state_dict()[k].reshape(-1) for k in OD.TIANSHOU_KEYS
                                             if k.endswith("bias")]).numpy().copy()
            out[f"u{u}_tree"] = np.asarray(buf.weight._value, np.float64).copy()
    finally:
        cls._preprocess_batch, cls._update_with_batch = orig_pre, orig_upd
    cfg = dict(gamma=algorithm.gamma, n_step=algorithm.n_step, target_update_freq=algorithm.target_update_freq,
               lr=lr, v_min=v_min, v_max=v_max)
    out["cfg_keys"] = np.array(list(cfg.keys()))
    out["cfg_vals"] = np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, "qrdqn.npz" if kind == "qr" else "c51.npz"), **out)


def gen_distq_all() -> None:
    gen_distq("qr", E=3, slots=24, steps=30, c=2, h=44, w=36, n_act=3, n_atoms=20, batch=24, n_updates=3, seed=11,
              lr=3e-4, gamma=0.95, n_step_return_horizon=3, target_update_freq=2)
    gen_distq("c51", E=3, slots=24, steps=30, c=2, h=44, w=36, n_act=4, n_atoms=11, batch=24, n_updates=3, seed=13,
              lr=3e-4, gamma=0.9, n_step_return_horizon=2, target_update_freq=0)


def gen_rainbow(tag: str, *, E: int, slots: int, steps: int, c: int, h: int, w: int, n_act: int, n_atoms: int, batch: int,
                n_updates: int, seed: int, lr: float, v_min: float, v_max: float, **algo_kwargs) -> None:
    """Runs the reference RainbowDQN.update() (RainbowNet: noisy, dueling; rainbow.py, c51.py) on a synthetic
    PrioritizedVectorReplayBuffer and dumps, per update: the noise drawn for both networks, indices, n-step returns, new
    priorities, loss and parameter samples."""
    from tianshou.algorithm.modelfree.c51 import C51Policy
    from tianshou.algorithm.modelfree.rainbow import RainbowDQN
    from tianshou.env.atari.atari_network import RainbowNet
    from oracle import oracle_rainbow as ORB

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    net = RainbowNet(c=c, h=h, w=w, action_shape=[n_act], num_atoms=n_atoms)
    p0, n0 = ORB.init_params(c, h, w, n_act, n_atoms, seed)
    sd = net.state_dict()
    for k_ref, k in zip(ORB.TIANSHOU_KEYS, ORB.PARAM_ORDER):
        assert torch.equal(sd[k_ref], p0[k]), f"oracle init differs from RainbowNet at {k}"
    for k_ref, (L, t) in zip(ORB.NOISE_KEYS, [(L, t) for L in ORB.NOISY for t in ("eps_p", "eps_q")]):
        assert torch.equal(sd[k_ref], n0[f"{L}.{t}"]), f"oracle noise differs at {k_ref}"
    policy = C51Policy(model=net, action_space=gym.spaces.Discrete(n_act), num_atoms=n_atoms, v_min=v_min, v_max=v_max)
    algorithm = RainbowDQN(policy=policy, optim=AdamOptimizerFactory(lr=lr), **algo_kwargs)

    buf = PrioritizedVectorReplayBuffer(E * slots, E, alpha=0.6, beta=0.4)
    frames = rng.integers(0, 256, size=(steps + 1, E, c, h, w), dtype=np.uint8)
    frames = np.where(rng.random(frames.shape) < 0.06, frames, 0).astype(np.uint8)
    act = rng.integers(0, n_act, size=(steps, E))
    rew = rng.normal(size=(steps, E)).astype(np.float32)
    term = rng.random((steps, E)) < 0.08
    trunc = (rng.random((steps, E)) < 0.04) & ~term
    for t in range(steps):
        buf.add(Batch(obs=frames[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t], obs_next=frames[t + 1]))
    out: dict[str, np.ndarray] = {}
    out["dims"] = np.array([E, slots, steps, c, h, w, n_act, n_atoms, batch, n_updates, seed])
    out["frames"], out["frames_next"] = np.asarray(buf.obs, np.uint8), np.asarray(buf.obs_next, np.uint8)
    out["act"], out["rew"] = np.asarray(buf.act, np.int64), np.asarray(buf.rew, np.float64)
    out["terminated"], out["truncated"] = np.asarray(buf.terminated, bool), np.asarray(buf.truncated, bool)
    for k, v in manager_state(buf).items():
        out["buf_" + k] = v

    rec: list[dict] = []
    orig_pre, orig_upd, orig_sample = RainbowDQN._preprocess_batch, RainbowDQN._update_with_batch, RainbowDQN._sample_noise
    drawn: list[dict] = []

    def rec_sample(model):
        r = orig_sample(model)
        sdm = model.state_dict()
        drawn.append({f"{L}.{t}": sdm[k].numpy().copy()
                      for k, (L, t) in zip(ORB.NOISE_KEYS, [(L, t) for L in ORB.NOISY for t in ("eps_p", "eps_q")])})
        return r

    def rec_pre(self, batch, buffer, indices):
        r = {"indices": np.array(indices, np.int64), "is_weight": np.array(batch.weight, np.float64)}
        b = orig_pre(self, batch, buffer, indices)
        r["returns"] = b.returns.numpy().copy()
        rec.append(r)
        return b

    def rec_upd(self, batch):
        r = rec[-1]
        r["old_training"] = bool(self.model_old.training) if self.model_old is not None else False
        stats = orig_upd(self, batch)
        r["prio"] = batch.weight.detach().numpy().copy()
        loss = stats.loss
        r["loss"] = np.array(loss.mean if hasattr(loss, "mean") and not isinstance(loss, float) else loss)
        return stats

    RainbowDQN._preprocess_batch, RainbowDQN._update_with_batch = rec_pre, rec_upd
    RainbowDQN._sample_noise = staticmethod(rec_sample)
    try:
        np.random.seed(seed + 7)
        for u in range(n_updates):
            d0 = len(drawn)
            with policy_within_training_step(algorithm.policy):
                algorithm.update(buffer=buf, sample_size=batch)
            r = rec[-1]
            assert len(drawn) - d0 == (2 if algorithm.use_target_network else 1)
            for k, v in drawn[d0].items():
                out[f"u{u}_noise_{k}"] = v
            if algorithm.use_target_network:
                for k, v in drawn[d0 + 1].items():
                    out[f"u{u}_noise_old_{k}"] = v
            for k in ("indices", "returns", "prio", "loss", "is_weight"):
                out[f"u{u}_{k}"] = r[k]
            out[f"u{u}_old_training"] = np.array(r["old_training"])
            sdm = net.state_dict()
            flat = torch.cat([sdm[k].reshape(-1) for k in ORB.TIANSHOU_KEYS]).numpy()
            out[f"u{u}_params_strided"] = flat[::97].copy()
            out[f"u{u}_conv1_w"] = sdm["net.0.weight"].numpy().copy()
            out[f"u{u}_V2_sigma_W"] = sdm["V.2.sigma_W"].numpy().copy()
            out[f"u{u}_Q2_mu_b"] = sdm["Q.2.mu_bias"].numpy().copy()
    finally:
        RainbowDQN._preprocess_batch, RainbowDQN._update_with_batch = orig_pre, orig_upd
        RainbowDQN._sample_noise = staticmethod(orig_sample)
    cfg = dict(gamma=algorithm.gamma, n_step=algorithm.n_step, target_update_freq=algorithm.target_update_freq, lr=lr,
               v_min=v_min, v_max=v_max)
    out["cfg_keys"], out["cfg_vals"] = np.array(list(cfg.keys())), np.array(list(cfg.values()), np.float64)
    np.savez_compressed(os.path.join(OUT, f"rainbow_{tag}.npz"), **out)


def gen_rainbow_all() -> None:
    gen_rainbow("lagged", E=3, slots=24, steps=30, c=2, h=44, w=36, n_act=3, n_atoms=11, batch=24, n_updates=3, seed=17,
                lr=3e-4, v_min=-2.0, v_max=3.0, gamma=0.95, n_step_return_horizon=3, target_update_freq=2)
    gen_rainbow("single", E=3, slots=24, steps=30, c=2, h=44, w=36, n_act=4, n_atoms=7, batch=24, n_updates=2, seed=19,
                lr=3e-4, v_min=-1.0, v_max=1.5, gamma=0.9, n_step_return_horizon=1, target_update_freq=0)


def gen_dqn_all() -> None:
    # atari layout (C3): frame stack through prev(), PER, n-step 3, double-Q with a lagged net, Huber
    gen_dqn("atari", E=2, slots=40, steps=50, c=4, h=84, w=84, n_act=6, batch=16, n_updates=3, seed=3,
            per=True, stack=True, gamma=0.99, n_step_return_horizon=3, target_update_freq=2, is_double=True,
            huber_loss_delta=1.0)
    # small images, plain buffer with full observations, vanilla max-Q target, MSE loss
    gen_dqn("small", E=3, slots=24, steps=20, c=2, h=44, w=36, n_act=3, batch=24, n_updates=2, seed=5,
            per=False, stack=False, lr=3e-4, gamma=0.9, n_step_return_horizon=1, target_update_freq=0,
            is_double=False, huber_loss_delta=None)


if __name__ == "__main__":
    main()
