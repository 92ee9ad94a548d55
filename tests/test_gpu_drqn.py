"""GPU parity of DQN on the recurrent Q network (SURVEY 8f N4's recurrent item; test/discrete/test_drqn.py's setup): LSTM
forward / backward through time on the GEMM kernels, stacked vector observations, n-step double-Q targets, DQN loss, Adam --
through the C ABI, against the oracle (oracle/oracle_drqn.py) and the reference fixtures tests/golden/drqn_*.npz."""
import numpy as np
import pytest
import torch

from oracle import oracle_dqn as OD
from oracle import oracle_drqn as ORQ
from tests import dqn_common as DC

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["layer_kernels", "per_step"], autouse=True)
def lstm_path(request, monkeypatch):
    """Every test runs on both recurrent paths: a whole LSTM layer per launch (lstm_layer_fwd / bwd_kernel, H = 32, 64, 128)
    and one GEMM + one cell launch per step (TS_RNN_PER_STEP, the path of the other widths)."""
    if request.param == "per_step":
        monkeypatch.setenv("TS_RNN_PER_STEP", "1")


def rand_params(obs_dim, hidden, layers, n_act, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = ORQ.param_shapes(obs_dim, hidden, layers, n_act)
    p = {}
    for k in ORQ.param_keys(layers):
        bound = 1.0 / np.sqrt(hidden if k.startswith("nn.") or k.startswith("fc2") else obs_dim)
        p[k] = (torch.rand(shapes[k], generator=g) * 2 - 1) * (2.0 * bound)
    return p


def make_engine(p, obs_dim, hidden, layers, n_act, ocfg):
    from tianshou_amd import dqn as D
    from tianshou_amd import drqn as R

    cfg = D.DQNConfig(gamma=ocfg.gamma, n_step=ocfg.n_step, target_update_freq=ocfg.target_update_freq, is_double=ocfg.is_double,
                      huber_delta=ocfg.huber_delta, lr=ocfg.lr, max_grad_norm=ocfg.max_grad_norm)
    flat = R.flat_from_torch([p[k] for k in ORQ.param_keys(layers)], obs_dim, hidden, layers, n_act)
    return R.RecurrentDQNEngine(obs_dim, hidden, layers, n_act, flat, cfg)


def torch_order(flat, obs_dim, hidden, layers, n_act):
    from tianshou_amd import drqn as R

    return torch.cat([t.reshape(-1) for t in R.flat_to_torch(flat, obs_dim, hidden, layers, n_act)]).cpu().numpy()


SHAPES = [(4, 128, 2, 2, 33, 4), (37, 64, 1, 5, 257, 3), (17, 32, 3, 32, 64, 1)]


@pytest.mark.parametrize("obs_dim,hidden,layers,n_act,B,T", SHAPES)
def test_forward_and_state_passing_vs_oracle(obs_dim, hidden, layers, n_act, B, T):
    from tianshou_amd import drqn as R

    p = rand_params(obs_dim, hidden, layers, n_act, 1)
    eng = make_engine(p, obs_dim, hidden, layers, n_act, OD.DQNConfig())
    flat = R.flat_from_torch([p[k] for k in ORQ.param_keys(layers)], obs_dim, hidden, layers, n_act)
    assert all(torch.equal(a.cpu(), p[k]) for a, k in zip(R.flat_to_torch(flat, obs_dim, hidden, layers, n_act), ORQ.param_keys(layers)))
    g = torch.Generator().manual_seed(B)
    obs = torch.randn(B, T, obs_dim, generator=g)
    q, act, (h, c) = eng.forward(obs, want_state=True)
    q_want, (h_want, c_want) = ORQ.forward(p, obs, want_state=True)
    np.testing.assert_allclose(q.cpu().numpy(), q_want.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(h.cpu().numpy(), h_want.transpose(0, 1).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c.cpu().numpy(), c_want.transpose(0, 1).numpy(), rtol=1e-5, atol=1e-5)
    qs = np.sort(q_want.numpy(), axis=1)
    clear = (qs[:, -1] - qs[:, -2] > 1e-4) if n_act > 1 else np.ones(B, bool)
    np.testing.assert_array_equal(act.cpu().numpy()[clear], q_want.argmax(dim=1).numpy()[clear])
    # one more evaluation-mode step from the carried state (obs [B, dim])
    o2 = torch.randn(B, obs_dim, generator=g)
    q2, _, (h2, c2) = eng.forward(o2, state=(h, c), want_state=True)
    q2_want, (h2_want, _) = ORQ.forward(p, o2, state=(h_want, c_want), want_state=True)
    np.testing.assert_allclose(q2.cpu().numpy(), q2_want.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(h2.cpu().numpy(), h2_want.transpose(0, 1).numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("obs_dim,hidden,layers,n_act,B,T", SHAPES)
@pytest.mark.parametrize("huber,weighted,clip", [(1.0, False, None), (None, True, 0.5)])
def test_gradient_and_step_vs_oracle(obs_dim, hidden, layers, n_act, B, T, huber, weighted, clip):
    p = rand_params(obs_dim, hidden, layers, n_act, 2)
    ocfg = OD.DQNConfig(huber_delta=huber, lr=3e-4, max_grad_norm=clip)
    eng = make_engine(p, obs_dim, hidden, layers, n_act, ocfg)
    st = OD.DQNState.create(p, ocfg)
    g = torch.Generator().manual_seed(B + 7)
    obs = torch.randn(B, T, obs_dim, generator=g)
    act = torch.randint(0, n_act, (B,), generator=g)
    ret = torch.randn(B, generator=g) * 2
    w = torch.rand(B, generator=g) + 0.5 if weighted else None
    col = {}
    loss_want, td_want = ORQ.update_with_batch(st, ocfg, obs, act, ret, w, collect=col)
    grad = torch.empty_like(eng.params)
    before = eng.params.clone()
    loss, td = eng.update_with_batch(obs, act, ret, w, grad_out=grad, apply=False)
    assert torch.equal(eng.params, before)
    assert abs(float(loss) - loss_want) <= 1e-5 * max(1.0, abs(loss_want))
    np.testing.assert_allclose(td.cpu().numpy(), td_want.numpy(), rtol=1e-5, atol=1e-5)
    g_want = torch.cat([col["grads"][k].reshape(-1) for k in ORQ.param_keys(layers)]).numpy()
    g_got = torch_order(grad, obs_dim, hidden, layers, n_act)
    np.testing.assert_allclose(g_got, g_want, rtol=1e-4, atol=1e-5 * np.abs(g_want).max())
    # the padding rows / columns receive exactly zero gradient (they must stay zero under Adam)
    lay = eng.lay
    if lay["k0"] > obs_dim:
        assert float(grad[lay["fc1"]:lay["fc1"] + (lay["k0"] + 1) * hidden].reshape(-1, hidden)[obs_dim:lay["k0"]].abs().max()) == 0.0
    if n_act < 32:
        assert float(grad[lay["fc2"]:].reshape(hidden + 1, 32)[:, n_act:].abs().max()) == 0.0
    eng.iter = 0
    loss2, _ = eng.update_with_batch(obs, act, ret, w)
    assert float(loss2) == float(loss)
    after = ORQ.flatten(st.params, layers).numpy()
    np.testing.assert_allclose(torch_order(eng.params, obs_dim, hidden, layers, n_act), after, rtol=1e-4, atol=0.02 * ocfg.lr)


@pytest.mark.parametrize("tag", ["cartpole", "per"])
def test_update_sequence_matches_reference_fixture(tag):
    """Replays the reference's DQN.update() sequence on the Recurrent network (sampled indices from the fixture)."""
    from tianshou_amd import drqn as R
    from tianshou_amd.buffer import DeviceReplayBuffer

    g, d, ocfg, bstate = DC.load_drqn(tag)
    dims = (d["obs_dim"], d["hidden"], d["layers"], d["n_act"])
    p0 = ORQ.unflatten(g["params0"], *dims)
    eng = make_engine(p0, *dims, ocfg)
    buf = DeviceReplayBuffer(offset=g["buf_offset"], last_index=g["buf_last_index"], lengths=g["buf_lengths"],
                             insertion=g["buf_insertion"], rew=g["rew"], terminated=g["terminated"], truncated=g["truncated"])
    rows = torch.as_tensor(g["obs_rows"]).cuda()
    act_all = torch.as_tensor(g["act"]).cuda()
    for u in range(d["n_updates"]):
        idx = torch.as_tensor(g[f"u{u}_indices"]).cuda()
        obs = R.gather_stacked_obs(rows, buf, idx, d["stack_num"])
        if u == 0:
            np.testing.assert_array_equal(obs.cpu().numpy(), g["u0_obs"])
        ret = eng.preprocess(buf, rows, idx, d["stack_num"])
        np.testing.assert_allclose(ret.cpu().numpy(), g[f"u{u}_returns"], rtol=1e-5, atol=1e-5)
        w = torch.as_tensor(g[f"u{u}_is_weight"]).cuda() if d["per"] else None
        loss, td = eng.update_with_batch(obs, act_all[idx], ret, w)
        np.testing.assert_allclose(td.cpu().numpy(), g[f"u{u}_td"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(float(loss), float(g[f"u{u}_loss"]), rtol=2e-5)
        tensors = dict(zip(ORQ.param_keys(d["layers"]), (t.cpu() for t in R.flat_to_torch(eng.params, *dims))))
        tol = dict(rtol=1e-4, atol=0.02 * ocfg.lr * (u + 1))
        np.testing.assert_allclose(ORQ.flatten(tensors, d["layers"]).numpy()[::17], g[f"u{u}_params_strided"], **tol)
        np.testing.assert_allclose(DC.drqn_small(tensors, d["layers"]), g[f"u{u}_small"], **tol)
    # evaluation-mode steps with the carried state, on the final parameters (as the fixture recorded them)
    q1, _, s1 = eng.forward(torch.as_tensor(g["eval_obs"][0]), want_state=True)
    q2, _, s2 = eng.forward(torch.as_tensor(g["eval_obs"][1]), state=s1, want_state=True)
    np.testing.assert_allclose(torch.stack([q1, q2]).cpu().numpy(), g["eval_q"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(s2[0].cpu().numpy(), g["eval_hidden"], rtol=1e-3, atol=1e-4)


def test_bad_arguments_fail_loudly():
    from tianshou_amd import _lib
    from tianshou_amd import dqn as D
    from tianshou_amd import drqn as R

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        R.RecurrentDQNEngine(4, 128, 2, 2, torch.zeros(10), D.DQNConfig())
    with pytest.raises((ValueError, _lib.EngineError)):
        R.layout(4, 48, 2, 2)                      # hidden not a multiple of 32
    with pytest.raises((ValueError, _lib.EngineError)):
        R.layout(4, 64, 2, 33)                     # more than 32 actions
    p = rand_params(4, 32, 1, 2, 0)
    eng = make_engine(p, 4, 32, 1, 2, OD.DQNConfig())
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(3, 2, 5))
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(3, 4), state=(torch.zeros(3, 2, 32), torch.zeros(3, 2, 32)))


def test_prefetched_forward_and_concurrent_target_pass_give_identical_updates():
    """RecurrentDQNEngine.preprocess_with_obs (batch.obs gathered first, its forward pass on the workspace's second side stream
    into a cache that ts_rnnq_update_cached consumes; the lagged network's pass of _target_q on the first side stream) against
    the plain order preprocess -> update_with_batch: four updates with a target sync in between, identical returns, losses, TD
    errors, parameters and Adam moments; a stale prefetch (parameters written since) is ignored."""
    from tianshou_amd import drqn as R
    from tianshou_amd.buffer import DeviceReplayBuffer

    obs_dim, hidden, layers, n_act, B, T, slots, E = 4, 128, 2, 2, 96, 4, 2048, 4
    p = rand_params(obs_dim, hidden, layers, n_act, 4)
    ocfg = OD.DQNConfig(gamma=0.95, n_step=3, target_update_freq=2, is_double=True, lr=1e-3)
    g = torch.Generator().manual_seed(9)
    Tn = slots // E
    off = np.arange(E + 1, dtype=np.int64) * Tn
    buf = DeviceReplayBuffer(offset=off, last_index=off[:-1] + Tn - 1, lengths=np.full(E, Tn, np.int64), insertion=np.zeros(E, np.int64),
                             rew=torch.randn(slots, generator=g).double().numpy(), terminated=(torch.rand(slots, generator=g) < 0.05).numpy(),
                             truncated=np.zeros(slots, bool), obs=torch.randn(slots, obs_dim, generator=g).numpy(),
                             act=torch.randint(0, n_act, (slots,), generator=g).numpy())
    plain, pre = make_engine(p, obs_dim, hidden, layers, n_act, ocfg), make_engine(p, obs_dim, hidden, layers, n_act, ocfg)
    for it in range(4):
        idx = torch.randint(0, slots, (B,), generator=g).cuda()
        ret0 = plain.preprocess(buf, buf.obs, idx, T)
        out0 = plain.update_with_batch(R.gather_stacked_obs(buf.obs, buf, idx, T), buf.act[idx], ret0)
        obs, ret1 = pre.preprocess_with_obs(buf, buf.obs, idx, T)
        assert pre._pre is not None and pre._pre[0] is obs
        out1 = pre.update_with_batch(obs, buf.act[idx], ret1)
        assert pre._pre is None
        assert torch.equal(ret0, ret1) and torch.equal(out0[0], out1[0]) and torch.equal(out0[1], out1[1]), it
        for name in ("params", "adam_m", "adam_v", "params_old"):
            assert torch.equal(getattr(plain, name), getattr(pre, name)), (it, name)
    # a prefetch that predates a parameter write is ignored (own forward pass)
    obs = pre.prefetch_forward(R.gather_stacked_obs(buf.obs, buf, idx, T))
    stale, pre._pre = pre._pre, None
    pre.update_with_batch(obs.clone(), buf.act[idx], ret1)
    plain.update_with_batch(obs.clone(), buf.act[idx], ret1)
    pre._pre = stale
    a, b = pre.update_with_batch(obs, buf.act[idx], ret1), plain.update_with_batch(obs, buf.act[idx], ret1)
    assert torch.equal(a[0], b[0]) and torch.equal(plain.params, pre.params)
