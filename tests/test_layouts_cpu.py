"""Host-side layout logic of every engine (no GPU needed: the layout entry points of libtsengine.so are pure host code):
reference tensor lists <-> the engines' flat vectors are exact, padding is zero, sizes agree with the C ABI."""
import numpy as np
import pytest
import torch

from oracle import oracle_distq as OQ
from oracle import oracle_dqn as OD
from oracle import oracle_dsac as ODS
from oracle import oracle_ppo as OP
from oracle import oracle_ppo_cnn as OC
from oracle import oracle_ppo_discrete as OPD
from oracle import oracle_rainbow as ORB
from oracle import oracle_redq as OR
from oracle import oracle_sac as OS


def _same(back, ref):
    assert len(back) == len(ref)
    for a, b in zip(back, ref):
        assert a.shape == b.shape and torch.equal(a, b)


def _perturbed(p: dict, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {k: v + torch.randn(v.shape, generator=g) for k, v in p.items()}        # biases etc. are not all zero


def test_dqn_and_distq_layouts():
    from tianshou_amd import distq as Q
    from tianshou_amd import dqn as D

    c, h, w, A = 4, 84, 84, 6
    p = _perturbed(OD.init_params(c, h, w, A, 1))
    t = [p[k] for k in OD.PARAM_ORDER]
    flat = D.flat_from_torch(t, c, h, w, A, device="cpu")
    assert flat.numel() == D.param_count(c, h, w, A) == OD.param_count(c, h, w, A)
    _same(D.flat_to_torch(flat, c, h, w, A), t)
    for N in (200, 51, 7):
        p = _perturbed(OQ.init_params(c, h, w, A, N, 2))
        t = [p[k] for k in OD.PARAM_ORDER]
        flat = Q.flat_from_torch(t, c, h, w, A, N, device="cpu")
        assert flat.numel() == Q.param_count(c, h, w, A, N)
        _same(Q.flat_to_torch(flat, c, h, w, A, N), t)
        head = flat[-513 * Q.head_width(A, N):].reshape(513, -1)
        assert Q.head_width(A, N) % 32 == 0 and torch.count_nonzero(head[:, A * N:]) == 0
    with pytest.raises(ValueError):
        Q.param_count(c, h, w, A, 300)


def test_rainbow_layout_and_noise_permutation():
    from tianshou_amd import rainbow as RB

    c, h, w, A, N = 2, 44, 36, 3, 11
    p, n = ORB.init_params(c, h, w, A, N, 3)
    p = _perturbed(p)
    t = [p[k] for k in ORB.PARAM_ORDER]
    lay = RB.layout(c, h, w, A, N)
    flat = RB.flat_from_torch(t, c, h, w, A, N, device="cpu")
    assert flat.numel() == lay["count"] and lay["F"] == ORB.feature_dim(h, w)
    _same(RB.flat_to_torch(flat, c, h, w, A, N), t)
    order = [f"{L}.{k}" for L in ORB.NOISY for k in ("eps_p", "eps_q")]
    nz = RB.noise_from_torch([n[k] for k in order], c, h, w, A, N, device="cpu")
    assert nz.numel() == lay["noise_count"]
    # the engine's F index is (h, w, c); torch flattens (c, h, w): effective weights must agree entry by entry
    eff_ref = p["Q0.mu_W"] + p["Q0.sigma_W"] * n["Q0.eps_q"].ger(n["Q0.eps_p"])          # [512, F] torch order
    F = lay["F"]
    mu = flat[lay["lin"][0]: lay["lin"][0] + (F + 1) * 512].reshape(F + 1, 512)
    sg = flat[lay["lin"][0] + (F + 1) * 512: lay["lin"][0] + 2 * (F + 1) * 512].reshape(F + 1, 512)
    eps_p, eps_q = nz[lay["noise"][0]: lay["noise"][0] + F], nz[lay["noise"][1]: lay["noise"][1] + 512]
    eff = mu[:F] + sg[:F] * (eps_q[None, :] * eps_p[:, None])                             # [F (h, w, c), 512]
    from tianshou_amd.rainbow import _hwc
    oh, ow = _hwc(c, h, w)
    eff_t = eff.t().reshape(512, oh, ow, 64).permute(0, 3, 1, 2).reshape(512, F)
    assert torch.allclose(eff_t, eff_ref, rtol=0, atol=1e-7)


def test_sac_family_layouts():
    from tianshou_amd import dsac as DS
    from tianshou_amd import redq as RQ
    from tianshou_amd import sac as S
    from tianshou_amd import td3 as T

    for obs_dim, act_dim in ((376, 17), (23, 5), (7, 1)):
        actor, c1, _ = OS.init_sac_params(obs_dim, act_dim, 0)
        lay = S.layout(obs_dim, act_dim)
        fa = S.actor_flat_from_torch([actor[k] for k in OS.ACTOR_ORDER], obs_dim, act_dim, "cpu")
        fc = S.critic_flat_from_torch([c1[k] for k in OS.CRITIC_ORDER], obs_dim, act_dim, "cpu")
        assert fa.numel() == lay["actor_count"] and fc.numel() == lay["critic_count"]
        _same(S.actor_flat_to_torch(fa, obs_dim, act_dim), [actor[k] for k in OS.ACTOR_ORDER])
        _same(S.critic_flat_to_torch(fc, obs_dim, act_dim), [c1[k] for k in OS.CRITIC_ORDER])
        da, _, _ = OS.init_td3_params(obs_dim, act_dim, 1)
        fd = T.actor_flat_from_torch([da[k] for k in OS.DET_ACTOR_ORDER], obs_dim, act_dim, "cpu")
        assert fd.numel() == T.layout(obs_dim, act_dim)["actor_count"]
        _same(T.actor_flat_to_torch(fd, obs_dim, act_dim), [da[k] for k in OS.DET_ACTOR_ORDER])
        _, ens = OR.init_params(obs_dim, act_dim, 5, 2)
        fe = RQ.ensemble_flat_from_torch([ens[k] for k in OR.CRITIC_ORDER], obs_dim, act_dim, "cpu")
        assert fe.numel() == 5 * lay["critic_count"]
        _same(RQ.ensemble_flat_to_torch(fe, 5, obs_dim, act_dim), [ens[k] for k in OR.CRITIC_ORDER])
    for obs_dim, n_act, hidden in ((11, 5, 64), (128, 18, 256), (40, 64, 96)):
        nets = ODS.init_params(obs_dim, n_act, hidden, 0)
        f = DS.net_flat_from_torch([nets[0][k] for k in ODS.NET_ORDER], obs_dim, n_act, hidden, "cpu")
        assert f.numel() == DS.layout(obs_dim, n_act, hidden)["count"]
        _same(DS.net_flat_to_torch(f, obs_dim, n_act, hidden), [nets[0][k] for k in ODS.NET_ORDER])
    for bad in ((11, 5, 48), (11, 65, 64), (11, 1, 64)):
        with pytest.raises(Exception):
            DS.layout(*bad)


def test_on_policy_layouts():
    from tianshou_amd import npg as NG
    from tianshou_amd import ppo_cnn as PC
    from tianshou_amd import ppo_discrete as PD

    p = _perturbed(OC.init_params(4, 84, 84, 6, 0))
    t = [p[k] for k in OC.PARAM_ORDER]
    _same(PC.flat_to_torch(PC.flat_from_torch(t, 4, 84, 84, 6, device="cpu"), 4, 84, 84, 6), t)
    for obs_dim, hidden, A in ((4, 64, 2), (33, 256, 31)):
        p = _perturbed(OPD.init_params(obs_dim, hidden, A, 0))
        t = [p[k] for k in OPD.PARAM_ORDER]
        f = PD.flat_from_torch(t, obs_dim, hidden, A, device="cpu")
        assert f.numel() == PD.layout(obs_dim, hidden, A)["count"]
        _same(PD.flat_to_torch(f, obs_dim, hidden, A), t)
    p = _perturbed(OP.init_params(17, 6))
    a_keys = ("a_w1", "a_b1", "a_w2", "a_b2", "a_wmu", "a_bmu", "a_sigma")
    c_keys = ("c_w1", "c_b1", "c_w2", "c_b2", "c_wv", "c_bv")
    lay = NG.layout(17, 64, 6)
    fa = NG.actor_flat_from_torch([p[k] for k in a_keys], 17, 64, 6, "cpu")
    fc = NG.critic_flat_from_torch([p[k] for k in c_keys], 17, 64, "cpu")
    assert (fa.numel(), fc.numel()) == (lay["actor_count"], lay["critic_count"])
    _same(NG.actor_flat_to_torch(fa, 17, 64, 6), [p[k] for k in a_keys])
    _same(NG.critic_flat_to_torch(fc, 17, 64), [p[k] for k in c_keys])
    assert torch.count_nonzero(fa[-32:][6:]) == 0                                  # log-sigma padding


def test_device_pointer_guard():
    from tianshou_amd import _lib

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.ptr(torch.zeros(4))
    assert _lib.ptr(None).value in (None, 0)
    assert np.dtype(np.int64).itemsize == 8


def test_recurrent_actor_critic_layout_converters_round_trip_and_reject_conditioned_sigma():
    """tianshou_amd.recurrent: state_dict() order <-> flat engine layout (exact inverses, zero padding), and the
    unsupported RecurrentActorProb(conditioned_sigma=True) is refused loudly (continuous.py:268-271)."""
    import types

    from tianshou_amd import recurrent as R

    obs_dim, act_dim, hidden, layers = 37, 5, 64, 2
    g = torch.Generator().manual_seed(0)
    shapes = {}
    for l in range(layers):
        shapes[f"nn.weight_ih_l{l}"] = (4 * hidden, obs_dim if l == 0 else hidden)
        shapes[f"nn.weight_hh_l{l}"] = (4 * hidden, hidden)
        shapes[f"nn.bias_ih_l{l}"] = shapes[f"nn.bias_hh_l{l}"] = (4 * hidden,)
    a_shapes = dict(shapes, **{"sigma_param": (act_dim, 1), "mu.weight": (act_dim, hidden), "mu.bias": (act_dim,)})
    c_shapes = dict(shapes, **{"fc2.weight": (1, hidden + act_dim), "fc2.bias": (1,)})
    ta = [torch.randn(a_shapes[k], generator=g) for k in R.actor_state_dict_keys(layers)]
    tc = [torch.randn(c_shapes[k], generator=g) for k in R.critic_state_dict_keys(layers)]
    flat, sigma = R.actor_flat_from_torch(ta, obs_dim, act_dim, hidden, layers, device="cpu")
    k0 = 64
    assert flat.numel() == (k0 + 1) * 4 * hidden + (hidden + 1) * 4 * hidden * 3 + (hidden + 1) * 32
    for a, b in zip(R.actor_flat_to_torch(flat, sigma, obs_dim, act_dim, hidden, layers), ta):
        assert torch.equal(a, b)
    w0 = flat[: (k0 + 1) * 4 * hidden].reshape(k0 + 1, 4 * hidden)
    assert torch.all(w0[obs_dim:k0] == 0) and torch.equal(w0[k0], ta[3])          # zero K padding, b_ih in the last row
    flat_c = R.critic_flat_from_torch(tc, obs_dim, act_dim, hidden, layers, device="cpu")
    head_in = 96                                                                 # 64 + 5 rounded up to 32
    assert flat_c.numel() == flat.numel() - (hidden + 1) * 32 + (head_in + 1) * 32
    for a, b in zip(R.critic_flat_to_torch(flat_c, obs_dim, act_dim, hidden, layers), tc):
        assert torch.equal(a, b)
    head = flat_c[-(head_in + 1) * 32:].reshape(head_in + 1, 32)
    assert torch.all(head[hidden + act_dim:head_in] == 0) and torch.all(head[:, 1:] == 0)
    fake = types.SimpleNamespace(_c_sigma=True)
    with pytest.raises(NotImplementedError, match="conditioned_sigma"):
        R.RecurrentActorProbEngine.from_module(fake)


def test_replay_stream_flattens_nested_batches_and_reinforce_statistics_round_trip():
    """Host logic that needs no GPU: dqn.ReplayStream._tensors walks the nested tuples a `prepare` callable returns (every
    tensor gets its record_stream call, None entries are skipped); ReinforceEngine.ret_rms is a plain list on the host until
    `preprocess` moves it to the device (assigning it replaces the device copy)."""
    from tianshou_amd import reinforce as RF
    from tianshou_amd.dqn import ReplayStream

    a, b, c, d = (torch.zeros(k + 1) for k in range(4))
    nested = (a, None, (b, (c, None)), [d], 3, "x")
    assert [t.numel() for t in ReplayStream._tensors(nested)] == [1, 2, 3, 4]
    assert list(ReplayStream._tensors(None)) == []
    eng = RF.ReinforceEngine.__new__(RF.ReinforceEngine)        # the statistics property alone (the constructor needs a GPU)
    eng._rms_host, eng._rms_dev = [0.0, 1.0, 0.0], None
    assert eng.ret_rms == [0.0, 1.0, 0.0]
    eng.ret_rms = (0.5, 2, 7)
    assert eng.ret_rms == [0.5, 2.0, 7.0] and eng._rms_dev is None
    eng._rms_dev = torch.tensor([1.5, 0.25, 9.0], dtype=torch.float64)          # what preprocess leaves behind
    assert eng.ret_rms == [1.5, 0.25, 9.0]
    got = eng.ret_rms
    got[0] = -1.0                                                # a copy: callers cannot edit the engine's state through it
    assert eng.ret_rms[0] == 1.5


def test_tensor_form_of_the_running_statistics_update_equals_the_host_form():
    """reinforce.rms_merge_tensors (what Reinforce's preprocess runs on the device) against ppo.rms_merge (Python floats, the
    form every other engine uses and the oracle pins): bit-identical over a chain of updates, including a zero-variance batch."""
    from tianshou_amd.ppo import rms_merge
    from tianshou_amd.reinforce import rms_merge_tensors

    rng = np.random.default_rng(0)
    host = [0.0, 1.0, 0.0]
    dev = torch.tensor(host, dtype=torch.float64)
    for it in range(40):
        n = int(rng.integers(1, 5000))
        x = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4) + rng.standard_normal() * 5
        if it == 7:
            x = np.full(n, 3.25)
        s1, s2 = float(x.sum()), float((x * x).sum())
        host = rms_merge(host, s1, s2, float(n))
        dev = rms_merge_tensors(dev, torch.tensor(s1, dtype=torch.float64), torch.tensor(s2, dtype=torch.float64), n)
        assert dev.tolist() == host, it
