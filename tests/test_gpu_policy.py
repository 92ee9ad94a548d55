"""SURVEY 8f N2 -- the collector's inference step (data/collector.py:735-744: `policy(batch)` then `policy.map_action(act)`)
on the engine, against what the REFERENCE itself computed: tests/golden/policy_forward.npz is written by
oracle/gen_golden.py::gen_policy_forward, which runs the unmodified `ProbabilisticActorPolicy.forward`
(reinforce.py:167-192) + `Algorithm.map_action` (algorithm_base.py:254-287), `SACPolicy.forward` (sac.py:108-131) and
`DiscreteQLearningPolicy.forward` (dqn.py:101-143) and records inputs, parameters, the N(0, 1) draws dist.sample() /
rsample() consumed, logits, actions and mapped actions.  The policies under test are the stand-ins of tests/standin.py turned
into engine-backed subclasses by `tianshou_amd.policy.attach` (the class HipPPO / HipSAC / HipDQN give their policy)."""
import copy
import os
import pickle

import numpy as np
import pytest
import torch
from torch import nn

from tests import standin as SI

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "policy_forward.npz")
G_TAGS = ["g_clip", "g_tanh", "g_none", "g_bounded", "g_bounded2", "g_net", "g_wide"]


def _gauss_policy(g, tag):
    """Stand-in ProbabilisticActorPolicy + ContinuousActorProbabilistic carrying the fixture's parameters and action space."""
    obs_dim, act_dim, bound, scaling, max_action, act_code, seed, training, det_eval, sampled = g[f"{tag}_cfg"]
    obs_dim, act_dim = int(obs_dim), int(act_dim)
    hidden = [int(h) for h in g[f"{tag}_hidden"]]
    act_cls = {0: nn.Tanh, 1: nn.ReLU, 2: None}[int(act_code)]
    actor = SI.ContinuousActorProbabilistic(SI.Net(obs_dim, hidden, act_cls), act_dim, unbounded=max_action == 0,
                                            max_action=float(max_action) or 1.0)
    sd = actor.state_dict()
    keys = [str(k) for k in g[f"{tag}_keys"]]
    assert sorted(keys) == sorted(sd.keys())
    actor.load_state_dict({k: torch.from_numpy(g[f"{tag}_p{i}"]) for i, k in enumerate(keys)})
    policy = SI.Policy(actor, action_space=SI.Box(g[f"{tag}_low"], g[f"{tag}_high"], (act_dim,)), action_scaling=bool(scaling),
                       action_bound_method={0: None, 1: "clip", 2: "tanh"}[int(bound)], deterministic_eval=bool(det_eval))
    policy.is_within_training_step = bool(training)
    return policy, obs_dim, act_dim, hidden, act_cls, (float(max_action) or None), bool(sampled)


def _attach_gauss(policy, obs_dim, act_dim, hidden, act_cls, max_action, **kw):
    """What integration._attach_gauss_policy does for the three engine kinds (here without an owner: the parameters are read
    from the torch modules)."""
    import ctypes as C

    from tianshou_amd import _lib
    from tianshou_amd import npg as NG
    from tianshou_amd import policy as HP
    from tianshou_amd.integration import _trunk_spec

    stems, _, act_name = _trunk_spec(policy.actor, "actor")
    keys = tuple(f"{st}.{x}" for st in stems for x in ("weight", "bias")) + ("mu.model.0.weight", "mu.model.0.bias", "sigma_param")
    base = dict(obs_dim=obs_dim, act_dim=act_dim, max_action=max_action, actor_keys=keys, **kw)
    if hidden == [64, 64] and act_cls is nn.Tanh and obs_dim <= 31 and act_dim <= 8:
        return HP.attach(policy, "gauss", None, **base), "gauss"
    if len(hidden) == 2 and hidden[0] == hidden[1] and hidden[0] % 32 == 0 and act_cls is nn.Tanh and max_action is None:
        return HP.attach(policy, "gauss_wide", None, hidden=hidden[0], n_actor=int(NG.layout(obs_dim, hidden[0], act_dim)["actor_count"]),
                         **base), "gauss_wide"
    out = (C.c_int64 * 3)()
    _lib.check(_lib.load().ts_net_layout(C.byref(_lib.NetDesc.make(obs_dim, hidden, act_name)), _lib.i64(act_dim), out))
    return HP.attach(policy, "gauss_net", None, hidden=tuple(hidden), activation=act_name, n_actor=int(out[1]), **base), "gauss_net"


@pytest.mark.gpu
@pytest.mark.parametrize("tag", G_TAGS)
def test_gaussian_policy_forward_and_map_action_replay_the_reference(tag, monkeypatch):
    from tianshou_amd import policy as HP

    g = np.load(GOLDEN)
    policy, obs_dim, act_dim, hidden, act_cls, max_action, sampled = _gauss_policy(g, tag)
    base_cls = type(policy)
    policy, fam = _attach_gauss(policy, obs_dim, act_dim, hidden, act_cls, max_action)
    assert fam == {"g_net": "gauss_net", "g_wide": "gauss_wide"}.get(tag, "gauss")
    assert isinstance(policy, base_cls) and type(policy).__name__ == "HipPolicy"
    noise = torch.from_numpy(g[f"{tag}_noise"])
    calls = []

    def fixed_noise(self, n, a, dev):               # the N(0, 1) draws the reference's dist.sample() consumed
        calls.append((n, a))
        return noise.to(dev)

    monkeypatch.setattr(HP._HipForward, "_hip_noise", fixed_noise)
    res = policy(SI.Batch(obs=g[f"{tag}_obs"], info={}), None)
    assert bool(calls) == sampled                                          # dist.mode draws nothing (reinforce.py:185-189)
    mu, sigma = res.logits
    np.testing.assert_allclose(mu.cpu().numpy(), g[f"{tag}_mu"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(sigma.cpu().numpy(), g[f"{tag}_sigma"], rtol=1e-6)
    np.testing.assert_allclose(res.act.numpy(), g[f"{tag}_act"], rtol=1e-5, atol=2e-6)
    assert res.state is None and tuple(res.dist.batch_shape) == (g[f"{tag}_obs"].shape[0],)
    np.testing.assert_allclose(res.dist.mean.cpu().numpy(), g[f"{tag}_mu"], rtol=1e-5, atol=2e-6)
    # collector.py:741-744: act_RA = to_numpy(act_batch.act); act_normalized_RA = policy.map_action(act_RA)
    act_np = res.act.detach().cpu().numpy()
    mapped = policy.map_action(act_np)
    np.testing.assert_allclose(mapped, g[f"{tag}_mapped"], rtol=1e-5, atol=2e-6)
    assert policy.__dict__["_hip_rt_last"][2] is not mapped                # a copy: the caller may edit it
    # any other array takes the reference's NumPy code (here: the stand-in's restatement of it) -- same numbers
    np.testing.assert_allclose(policy.map_action(act_np.copy()), g[f"{tag}_mapped"], rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
def test_torch_sampling_mode_consumes_torchs_generator_like_dist_sample():
    """sampling="torch": the action is mu + sigma * (the N(0, 1) draws `torch.empty(n, A).normal_()` takes from torch's CPU
    generator) -- the stream the fixture generator shows dist.sample() to consume (it asserts exactly this against the
    reference); sampling="device": torch's generator is left untouched and consecutive calls draw fresh noise."""
    g = np.load(GOLDEN)
    policy, obs_dim, act_dim, hidden, act_cls, max_action, _ = _gauss_policy(g, "g_clip")
    policy, _ = _attach_gauss(policy, obs_dim, act_dim, hidden, act_cls, max_action, sampling="torch")
    n = g["g_clip_obs"].shape[0]
    torch.manual_seed(123)
    ref_noise = torch.empty(n, act_dim).normal_().numpy()
    torch.manual_seed(123)
    res = policy(SI.Batch(obs=g["g_clip_obs"], info={}), None)
    np.testing.assert_allclose(res.act.numpy(), g["g_clip_mu"] + g["g_clip_sigma"] * ref_noise, rtol=1e-5, atol=2e-6)
    policy2, *_ = _gauss_policy(g, "g_clip")
    policy2, _ = _attach_gauss(policy2, obs_dim, act_dim, hidden, act_cls, max_action, sampling="device", noise_seed=5)
    state = torch.get_rng_state()
    a1 = policy2(SI.Batch(obs=g["g_clip_obs"], info={}), None).act.numpy().copy()
    a2 = policy2(SI.Batch(obs=g["g_clip_obs"], info={}), None).act.numpy().copy()
    assert torch.equal(state, torch.get_rng_state()) and not np.allclose(a1, a2)
    z = (a1 - g["g_clip_mu"]) / g["g_clip_sigma"]                          # standard normal draws
    assert abs(float(z.mean())) < 0.15 and 0.85 < float(z.std()) < 1.15


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["s_train", "s_eval", "s_h128"])
def test_sac_policy_forward_replays_the_reference(tag, monkeypatch):
    from tianshou_amd import policy as HP
    from tianshou_amd import sac as S

    g = np.load(GOLDEN)
    obs_dim, act_dim, hid, seed, training = (int(x) for x in g[f"{tag}_cfg"])
    actor = SI.ContinuousActorProbabilistic(SI.Net(obs_dim, [hid, hid], nn.ReLU), act_dim, unbounded=True, conditioned_sigma=True)
    assert list(actor.state_dict().keys()) == S.TIANSHOU_ACTOR_KEYS
    actor.load_state_dict({k: torch.from_numpy(g[f"{tag}_p{i}"]) for i, k in enumerate(S.TIANSHOU_ACTOR_KEYS)})
    policy = SI.Policy(actor, action_space=SI.Box(g[f"{tag}_low"], g[f"{tag}_high"], (act_dim,)), action_scaling=True,
                       deterministic_eval=True)
    policy.is_within_training_step = bool(training)
    HP.attach(policy, "sac", None, obs_dim=obs_dim, act_dim=act_dim, hidden=hid)
    noise = torch.from_numpy(g[f"{tag}_noise"])
    monkeypatch.setattr(HP._HipForward, "_hip_noise", lambda self, n, a, dev: noise.to(dev))
    res = policy(SI.Batch(obs=g[f"{tag}_obs"], info={}), None)
    mu, sigma = res.logits
    np.testing.assert_allclose(mu.cpu().numpy(), g[f"{tag}_mu"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(sigma.cpu().numpy(), g[f"{tag}_sigma"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(res.act.cpu().numpy(), g[f"{tag}_act"], rtol=1e-5, atol=2e-6)
    assert tuple(res.log_prob.shape) == (g[f"{tag}_obs"].shape[0], 1)
    np.testing.assert_allclose(res.log_prob.cpu().numpy(), g[f"{tag}_logp"], rtol=1e-5, atol=2e-5)
    # SACPolicy does not bound (tanh already did): map_action is the inherited scaling into the Box
    np.testing.assert_allclose(policy.map_action(res.act), g[f"{tag}_mapped"], rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["q_plain", "q_mask"])
def test_q_policy_forward_replays_the_reference(tag):
    from tianshou_amd import dqn as D
    from tianshou_amd import policy as HP

    g = np.load(GOLDEN)
    c, h, w, n_act, seed, masked = (int(x) for x in g[f"{tag}_cfg"])
    model = SI.DQNet(c, h, w, n_act)
    assert list(model.state_dict().keys()) == D.TIANSHOU_KEYS
    model.load_state_dict({k: torch.from_numpy(g[f"{tag}_p{i}"]) for i, k in enumerate(D.TIANSHOU_KEYS)})
    policy = HP.attach(SI.DiscreteQLearningPolicy(model), "q", None, n_act=n_act)
    obs = SI.Batch(obs=g[f"{tag}_obs"], mask=g[f"{tag}_mask"]) if masked else g[f"{tag}_obs"]
    res = policy(SI.Batch(obs=obs, info={}), None)
    ref = g[f"{tag}_logits"]
    np.testing.assert_allclose(res.logits.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * float(np.abs(ref).max()))
    assert isinstance(res.act, np.ndarray) and res.act.dtype == np.int64 and res.state is None
    # greedy actions: identical wherever the reference's top two Q-values are further apart than fp32 rounding
    q = ref + (1 - g[f"{tag}_mask"]) * (ref.min() - ref.max() - 1.0) if masked else ref
    top2 = np.sort(q, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-4 * np.abs(ref).max()
    assert clear.sum() >= 0.9 * len(clear) and np.array_equal(res.act[clear], g[f"{tag}_act"][clear])
    if masked:
        assert g[f"{tag}_mask"][np.arange(len(res.act)), res.act].all()
    # float observations (an un-stacked env) take the same kernels through the float32 NHWC path
    res_f = policy(SI.Batch(obs=g[f"{tag}_obs"].astype(np.float32) if not masked else
                            SI.Batch(obs=g[f"{tag}_obs"].astype(np.float32), mask=g[f"{tag}_mask"]), info={}), None)
    np.testing.assert_array_equal(res_f.logits.cpu().numpy(), res.logits.cpu().numpy())


@pytest.mark.gpu
def test_hip_ppo_gives_its_policy_the_engine_forward_and_it_reads_the_engine_parameters():
    """HipPPO(policy_forward="hip") (the default): after an update the collector-side forward reads `engine.params` itself --
    not a copy -- and equals the torch modules' forward on the written-back parameters; an outside write to the torch parameters
    (`policy.load_state_dict`) drops the engine (version counters) and the next forward reads the NEW torch parameters."""
    from oracle import oracle_ppo as OP
    from tianshou_amd.integration import make_hip_ppo
    from tianshou_amd.ppo import flat_from_modules

    torch.manual_seed(4)
    obs_dim, act_dim, E, T = 17, 6, 4, 40
    actor = SI.ContinuousActorProbabilistic(SI.Net(obs_dim, [64, 64], nn.Tanh), act_dim, unbounded=False, max_action=1.0)
    critic = SI.ContinuousCritic(SI.Net(obs_dim, [64, 64], nn.Tanh))
    policy = SI.Policy(actor, action_space=SI.Box(-2.0, 2.0, (act_dim,)), action_scaling=True, action_bound_method="clip")
    algo = make_hip_ppo("ppo", ref=SI)(policy=policy, critic=critic, device="cuda", permutations="host", lr=1e-3,
                                       max_grad_norm=0.5).to("cuda")
    assert type(algo.policy).__name__ == "HipPolicy" and algo.policy._hip_family == "gauss"
    rng = np.random.default_rng(0)
    obs = rng.normal(size=(50, obs_dim)).astype(np.float32)

    def torch_mu():
        p = OP.unflatten_params(flat_from_modules(algo.policy.actor, algo.critic, device="cpu"), obs_dim, act_dim)
        with torch.no_grad():
            return OP.actor_forward(p, torch.from_numpy(obs), 1.0)[0].numpy()

    algo.policy.deterministic_eval = True                 # outside a training step: dist.mode
    r0 = algo.policy(SI.Batch(obs=obs, info={}), None)
    assert algo._hip_engine is None                       # no engine yet: parameters came from the torch modules
    np.testing.assert_allclose(r0.act.numpy(), torch_mu(), rtol=1e-5, atol=2e-6)
    buf = SI.VectorReplayBuffer(E * T, E, obs_shape=(obs_dim,), act_shape=(act_dim,))
    o = rng.normal(size=(T + 1, E, obs_dim)).astype(np.float32)
    for t in range(T):
        buf.add(SI.Batch(obs=o[t], act=rng.normal(size=(E, act_dim)).astype(np.float32), rew=rng.normal(size=E).astype(np.float32),
                         terminated=rng.random(E) < 0.05, truncated=np.zeros(E, bool), obs_next=o[t + 1]))
    algo.policy.is_within_training_step = True
    algo.update(buf, 64, 2)
    algo.policy.is_within_training_step = False
    eng = algo._hip_engine
    assert eng is not None and algo.policy._hip_engine() is eng
    from tianshou_amd import policy as HP

    assert HP._gauss_params(algo.policy).data_ptr() == eng.params.data_ptr()
    r1 = algo.policy(SI.Batch(obs=obs, info={}), None)
    assert not np.allclose(r1.act.numpy(), r0.act.numpy())
    np.testing.assert_allclose(r1.act.numpy(), torch_mu(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(algo.policy.map_action(r1.act.numpy()), -2.0 + 4.0 * (np.clip(torch_mu(), -1, 1) + 1.0) / 2.0,
                               rtol=1e-5, atol=2e-6)
    # an outside write: the engine is dropped, the forward follows the torch parameters
    sd = {k: v * 0.5 for k, v in algo.policy.actor.state_dict().items()}
    algo.policy.actor.load_state_dict(sd)
    r2 = algo.policy(SI.Batch(obs=obs, info={}), None)
    assert algo._hip_engine is None
    np.testing.assert_allclose(r2.act.numpy(), torch_mu(), rtol=1e-5, atol=2e-6)


def test_attached_policies_stay_instances_of_their_class_pickle_and_refuse_the_cpu():
    """CPU: `attach` swaps the class for a subclass of the policy's OWN class (isinstance, state_dict keys and every inherited
    method stay), the result survives pickle / deepcopy (highlevel/persistence.py:106 pickles policies) and `detach` restores
    the reference's class; without a GPU the forward raises instead of computing anywhere else."""
    from tianshou_amd import policy as HP

    actor = SI.ContinuousActorProbabilistic(SI.Net(17, [64, 64], nn.Tanh), 6, unbounded=True)
    policy = SI.Policy(actor, action_space=SI.Box(-1.0, 1.0, (6,)), action_scaling=True, action_bound_method="clip")
    keys = list(policy.state_dict().keys())
    owner = nn.Linear(1, 1)                                                # any object can own; it is held weakly
    HP.attach(policy, "gauss", owner, obs_dim=17, act_dim=6, max_action=None, actor_keys=tuple(actor.state_dict().keys()), noise_seed=3)
    assert isinstance(policy, SI.Policy) and type(policy) is HP.hip_policy_class(SI.Policy, "gauss")
    assert list(policy.state_dict().keys()) == keys and policy._hip_owner() is owner
    for clone in (pickle.loads(pickle.dumps(policy)), copy.deepcopy(policy)):
        assert type(clone) is type(policy) and clone._hip_family == "gauss" and clone._hip_owner() is None
        assert clone.__dict__["_hip_noise_seed"] == 3 and clone._hip_spec == policy._hip_spec
        assert all(torch.equal(a, b) for a, b in zip(clone.state_dict().values(), policy.state_dict().values()))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            policy(SI.Batch(obs=np.zeros((2, 17), np.float32), info={}), None)
    del owner
    assert policy._hip_owner() is None
    # foreign arrays go through the class's own map_action
    np.testing.assert_allclose(policy.map_action(np.full((2, 6), 3.0, np.float32)), np.ones((2, 6)))
    HP.detach(policy)
    assert type(policy) is SI.Policy and not any(k.startswith("_hip_") for k in policy.__dict__)
