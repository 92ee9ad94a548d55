"""GPU parity of Reinforce (SURVEY 8a row a7's second call site, reinforce.py:266-310, and its minibatch loop :363-382):
discounted returns against the running-mean bootstrap, standardisation + ret_rms update, vanilla policy-gradient Adam steps
-- through the C ABI, against the oracle (oracle/oracle_reinforce.py) and the reference fixtures tests/golden/reinforce_*.npz."""
import numpy as np
import pytest
import torch

from oracle import oracle_ppo as OP
from oracle import oracle_reinforce as OR
from tests.test_oracle_golden import load_reinforce

pytestmark = pytest.mark.gpu


def make_engine(params, obs_dim, act_dim, cfg):
    from tianshou_amd import npg as NG
    from tianshou_amd import reinforce as RF

    ecfg = RF.ReinforceConfig(gamma=cfg.gamma, return_standardization=cfg.return_standardization, lr=cfg.lr, betas=cfg.betas,
                              adam_eps=cfg.adam_eps, max_grad_norm=cfg.max_grad_norm)
    return RF.ReinforceEngine(obs_dim, act_dim, 64,
                              NG.actor_flat_from_torch([params[k] for k in OR.ACTOR_KEYS], obs_dim, 64, act_dim), ecfg)


def torch_order(flat, obs_dim, act_dim):
    from tianshou_amd import npg as NG

    return torch.cat([x.reshape(-1) for x in NG.actor_flat_to_torch(flat, obs_dim, 64, act_dim)]).cpu().numpy()


@pytest.mark.parametrize("obs_dim,act_dim,B,clip", [(17, 6, 1000, None), (33, 1, 64, 0.05), (4, 32, 257, None)])
def test_gradient_and_step_vs_oracle(obs_dim, act_dim, B, clip):
    p = OP.init_params(obs_dim, act_dim, seed=3)
    g = torch.Generator().manual_seed(B)
    p["a_wmu"] = p["a_wmu"] * 30.0
    p["a_sigma"] = torch.randn(act_dim, generator=g) * 0.3 - 0.5
    params = {k: p[k] for k in OR.ACTOR_KEYS}
    cfg = OR.ReinforceConfig(lr=3e-4, max_grad_norm=clip)
    st = OP.PPOState(params={k: v.clone() for k, v in params.items()})
    eng = make_engine(params, obs_dim, act_dim, cfg)
    obs, act, ret = torch.randn(B, obs_dim, generator=g), torch.randn(B, act_dim, generator=g), torch.randn(B, generator=g) * 2 + 0.5
    col = {}
    want = OR.update(st, cfg, obs, act, ret, None, 1, [np.arange(B)], collect=col)
    grad = torch.empty_like(eng.actor)
    loss = eng.gradient(obs, act, ret, grad)
    assert abs(float(loss) - want[0]) <= 1e-5 * max(1.0, abs(want[0]))
    g_want = torch.cat([col["grads"][k].reshape(-1) for k in OR.ACTOR_KEYS]).numpy()
    g_got = torch_order(grad, obs_dim, act_dim)
    np.testing.assert_allclose(g_got, g_want, rtol=1e-4, atol=1e-5 * np.abs(g_want).max())
    from tianshou_amd import npg as NG
    k0 = NG.layout(obs_dim, 64, act_dim)["k0"]
    assert float(grad[: (k0 + 1) * 64].reshape(k0 + 1, 64)[obs_dim:k0].abs().max() if k0 > obs_dim else 0.0) == 0.0
    eng.apply_gradient(grad)
    after = torch.cat([st.params[k].reshape(-1) for k in OR.ACTOR_KEYS]).numpy()
    np.testing.assert_allclose(torch_order(eng.actor, obs_dim, act_dim), after, rtol=1e-4, atol=0.02 * cfg.lr)


@pytest.mark.parametrize("path", ["fused_step_kernel", "per_layer_gemms"])
@pytest.mark.parametrize("tag", ["std", "plain"])
def test_update_matches_reference_fixture(tag, path, monkeypatch):
    """Both routes of ReinforceEngine.update: the minibatch loop on the fused actor-critic step kernel of ts_ppo.hip (A2C's
    actor loss with adv := returns, a zero critic beside it) and the per-layer GEMM path (TS_REINFORCE_GEMM=1)."""
    if path == "per_layer_gemms":
        monkeypatch.setenv("TS_REINFORCE_GEMM", "1")
    g, d, cfg, params = load_reinforce(tag)
    eng = make_engine(params, d["obs_dim"], d["act_dim"], cfg)
    assert eng.fused_supported() == (path == "fused_step_kernel" and d["obs_dim"] <= 31 and d["act_dim"] <= 8)
    st = OP.PPOState(params={k: v.clone() for k, v in params.items()})
    for u in range(d["n_updates"]):
        idx, unfinished = g[f"u{u}_indices"], g[f"u{u}_unfinished"]
        take = lambda k: g[f"u{u}_{k}"][idx]  # noqa: E731
        cut = np.nonzero(np.isin(idx, unfinished))[0]
        ret = eng.preprocess(take("rew"), take("terminated"), take("truncated"), cut)
        want = OR.preprocess(st, cfg, g[f"u{u}_rew"], g[f"u{u}_terminated"], g[f"u{u}_truncated"], idx, unfinished)
        np.testing.assert_allclose(ret.cpu().numpy(), g[f"u{u}_returns"].astype(np.float32), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ret.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
        # the bootstrap constant ret_rms.mean crosses the C ABI as float32 v arrays (the reference keeps it float64)
        np.testing.assert_allclose(eng.ret_rms, g[f"u{u}_ret_rms"], rtol=1e-6)
        losses, steps = eng.update(take("obs"), take("act"), ret, d["batch_size"], d["repeat"], g[f"u{u}_perms"])
        OR.update(st, cfg, take("obs"), take("act"), want, d["batch_size"], d["repeat"], g[f"u{u}_perms"])
        assert steps == len(g[f"u{u}_losses"])
        np.testing.assert_allclose(losses.cpu().numpy().reshape(-1), g[f"u{u}_losses"], rtol=1e-4, atol=1e-5)
        got = torch_order(eng.actor, d["obs_dim"], d["act_dim"])
        steps_so_far = eng.adam_step
        np.testing.assert_allclose(got, g[f"u{u}_actor"], rtol=1e-4, atol=0.02 * cfg.lr * steps_so_far)
        oracle_flat = torch.cat([st.params[k].reshape(-1) for k in OR.ACTOR_KEYS]).numpy()
        np.testing.assert_allclose(got, oracle_flat, rtol=1e-4, atol=0.02 * cfg.lr * steps_so_far)
    if eng.fused_supported():           # the critic half of the fused engine never moves
        fe = eng._fused[0]
        n_actor = int(eng._fused[2].numel())
        assert float(fe.params[n_actor:].abs().max()) == 0.0 and float(fe.adam_v[n_actor:].abs().max()) == 0.0


def test_host_tensor_is_refused():
    from tianshou_amd import reinforce as RF

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        RF.ReinforceEngine(17, 6, 64, torch.zeros(10), RF.ReinforceConfig())
