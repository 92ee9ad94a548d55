"""GPU parity of the REDQ row: ensemble critics (EnsembleLinear), random-subset min / mean target, one
ensemble loss + Adam step, delayed actor / alpha steps, Polyak -- through the C ABI, against the oracle
(oracle/oracle_redq.py, pinned to the reference by tests/golden/redq_*.npz).  Tolerance 1e-5 relative on each tensor's
scale."""
import numpy as np
import pytest
import torch

from oracle import oracle_redq as OR
from oracle import oracle_sac as OS
from tests.test_oracle_golden import load_redq

pytestmark = pytest.mark.gpu
CFG_KEYS = ("gamma", "tau", "n_step", "alpha", "auto_alpha", "target_entropy", "log_alpha0", "actor_lr", "critic_lr",
            "alpha_lr", "ensemble_size", "subset_size", "actor_delay", "target_mode")


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def make_engine(obs_dim, act_dim, seed, cfg, hidden=256):
    from tianshou_amd import redq as RQ
    from tianshou_amd import sac as S

    from tianshou_amd import widths as W

    actor, critic = OR.init_params(obs_dim, act_dim, cfg.ensemble_size, seed, hidden)     # hidden: int, four widths or a nested pair
    H = W.round32(max(max(x) for x in OS.layer_sizes(hidden)))
    eng = RQ.REDQEngine(obs_dim, act_dim, S.actor_flat_from_torch(list(actor.values()), obs_dim, act_dim, hidden=H),
                        RQ.ensemble_flat_from_torch(list(critic.values()), obs_dim, act_dim, hidden=H),
                        RQ.REDQConfig(**{k: getattr(cfg, k) for k in CFG_KEYS}), hidden=H, depth=OS.depth_of(actor))
    return eng, actor, critic


@pytest.mark.parametrize("mode,E,S", [("min", 10, 2), ("mean", 5, 5), ("min", 3, 1)])
def test_target_q_vs_oracle(mode, E, S):
    from tianshou_amd import redq as RQ

    obs_dim, act_dim, B = 23, 5, 300
    cfg = OR.REDQConfig(auto_alpha=True, log_alpha0=-0.4, ensemble_size=E, subset_size=S, target_mode=mode)
    eng, actor, critic = make_engine(obs_dim, act_dim, 3, cfg)
    for a, k in zip(RQ.ensemble_flat_to_torch(eng.critics, E, obs_dim, act_dim), OR.CRITIC_ORDER):
        assert torch.equal(a.cpu(), critic[k]), k
    st = OR.REDQState.create(actor, critic, cfg)
    g = torch.Generator().manual_seed(1)
    st.critic_old = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in critic.items()}
    eng.critics_old = RQ.ensemble_flat_from_torch([st.critic_old[k] for k in OR.CRITIC_ORDER], obs_dim, act_dim)
    obs, noise = torch.randn(B, obs_dim, generator=g), torch.randn(B, act_dim, generator=g)
    subset = np.random.default_rng(0).choice(E, S, replace=False)
    ref = OR.target_q(st, cfg, obs, noise, subset).flatten()
    out = eng.target_q(obs, noise, subset)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * act_dim)


@pytest.mark.parametrize("obs_dim,act_dim,B,E,auto,weighted", [(376, 17, 1024, 4, True, False), (23, 5, 200, 3, False, True)])
def test_update_gradients_vs_oracle(obs_dim, act_dim, B, E, auto, weighted):
    """Ensemble and actor gradients of one update with actor_delay = 1 (learning rates 0: the actor phase sees the same
    critics).  The actor's batch sum cancels heavily: float64 yardstick as in the SAC test."""
    from tianshou_amd import redq as RQ
    from tianshou_amd import sac as S

    cfg = OR.REDQConfig(auto_alpha=auto, log_alpha0=-0.3, alpha=0.15, target_entropy=-float(act_dim), actor_lr=0.0,
                        critic_lr=0.0, alpha_lr=0.0, tau=0.0, ensemble_size=E, subset_size=2, actor_delay=1)
    eng, actor, critic = make_engine(obs_dim, act_dim, 5, cfg)
    g = torch.Generator().manual_seed(7)
    obs = torch.randn(B, obs_dim, generator=g)
    act = torch.rand(B, act_dim, generator=g) * 2 - 1
    ret, noise = torch.randn(B, generator=g) * 2, torch.randn(B, act_dim, generator=g)
    weight = torch.rand(B, generator=g) if weighted else None
    st = OR.REDQState.create(actor, critic, cfg)
    col: dict = {}
    ref = OR.update_with_batch(st, cfg, obs, act, ret, noise, weight, collect=col)
    pc, pa = eng.lay["critic_count"], eng.lay["actor_count"]
    grads = torch.empty(E * pc + pa, dtype=torch.float32, device="cuda")
    stats, w_out = eng.update_with_batch(obs, act, ret, noise, weight, grads_out=grads)
    s = stats.cpu().numpy()
    np.testing.assert_allclose(s[:2], [ref["actor_loss"], ref["critic_loss"]], rtol=1e-5)
    np.testing.assert_allclose(w_out.cpu().numpy(), ref["weight"].numpy(), rtol=1e-5, atol=1e-5)
    for t, key in zip(RQ.ensemble_flat_to_torch(grads[:E * pc], E, obs_dim, act_dim), OR.CRITIC_ORDER):
        assert rel_err(t.cpu(), col["critic_grads"][key]) < 1e-5, key
    # actor: float64 reference of the same loss
    p64 = {k: v.double().requires_grad_(True) for k, v in actor.items()}
    c64 = {k: v.double() for k, v in critic.items()}
    a64, logp64, _, _ = OS.policy_forward(p64, obs.double(), noise.double())
    loss64 = (OS.alpha_value(st, cfg) * logp64.flatten() - OR.critic_forward(c64, obs.double(), a64).mean(dim=0).flatten()).mean()
    g64 = dict(zip(p64.keys(), torch.autograd.grad(loss64, list(p64.values()))))
    for t, key in zip(S.actor_flat_to_torch(grads[E * pc:], obs_dim, act_dim), OS.ACTOR_ORDER):
        e_gpu, e_ref = rel_err(t.cpu(), g64[key]), rel_err(col["actor_grads"][key], g64[key])
        assert e_gpu < max(1e-5, 2 * e_ref), (key, e_gpu, e_ref)


@pytest.mark.parametrize("hidden,obs_dim,act_dim,B,E", [(128, 23, 5, 96, 3), (64, 11, 3, 40, 4)])
def test_other_hidden_widths_vs_oracle(hidden, obs_dim, act_dim, B, E):
    """Net / EnsembleLinear widths other than test_redq.py's 256 (utils/net/common.py:246-369 takes any hidden_sizes): two
    updates with the actor step (actor_delay = 1) against the oracle -- losses, PER weights, parameters -- on the per-layer
    GEMM kernels; the width travels with every call (ts_mlp_set_hidden)."""
    from tianshou_amd import redq as RQ
    from tianshou_amd import sac as S

    cfg = OR.REDQConfig(auto_alpha=True, log_alpha0=-0.3, target_entropy=-float(act_dim), actor_lr=3e-4, critic_lr=1e-3,
                        alpha_lr=1e-3, tau=0.02, ensemble_size=E, subset_size=2, actor_delay=1)
    eng, actor, critic = make_engine(obs_dim, act_dim, 8, cfg, hidden)
    assert eng.lay == S.layout(obs_dim, act_dim, hidden)
    for a, k in zip(RQ.ensemble_flat_to_torch(eng.critics, E, obs_dim, act_dim, hidden), OR.CRITIC_ORDER):
        assert torch.equal(a.cpu(), critic[k]), k
    st = OR.REDQState.create(actor, critic, cfg)
    g = torch.Generator().manual_seed(B)
    for _ in range(2):
        obs = torch.randn(B, obs_dim, generator=g)
        act = torch.rand(B, act_dim, generator=g) * 2 - 1
        ret, noise = torch.randn(B, generator=g), torch.randn(B, act_dim, generator=g)
        ref = OR.update_with_batch(st, cfg, obs, act, ret, noise, None)
        stats, w = eng.update_with_batch(obs, act, ret, noise)
        s = stats.cpu().numpy()
        np.testing.assert_allclose(s[:2], [ref["actor_loss"], ref["critic_loss"]], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(w.cpu().numpy(), ref["weight"].numpy(), rtol=1e-5, atol=1e-5)
    for t, k in zip(RQ.ensemble_flat_to_torch(eng.critics, E, obs_dim, act_dim, hidden), OR.CRITIC_ORDER):
        np.testing.assert_allclose(t.cpu().numpy(), st.critic[k].numpy(), rtol=1e-4, atol=0.1 * cfg.critic_lr, err_msg=k)
    for t, k in zip(S.actor_flat_to_torch(eng.actor, obs_dim, act_dim, hidden), OS.ACTOR_ORDER):
        np.testing.assert_allclose(t.cpu().numpy(), st.actor[k].numpy(), rtol=1e-4, atol=0.1 * cfg.actor_lr, err_msg=k)


@pytest.mark.parametrize("tag", ["min", "mean", "widths", "depth1"])
def test_redq_update_matches_reference_golden(tag):
    """(`widths`: actor Net[48, 80], EnsembleLinear critics [72, 40] in the reference, embedded in Net[96, 96].  `depth1`: ONE hidden
    layer, actor [64], ensemble [48], mean target -- gen_golden.py::gen_depth.)"""
    from tianshou_amd import redq as RQ
    from tianshou_amd import sac as S
    from tianshou_amd.buffer import DeviceReplayBuffer

    g, d, cfg, bstate = load_redq(tag)
    eng, _, _ = make_engine(d["obs_dim"], d["act_dim"], d["seed"], cfg, d["hidden"])
    sa, sc = OS.layer_sizes(d["hidden"])
    od, ad, H = d["obs_dim"], d["act_dim"], eng.hidden
    buf = DeviceReplayBuffer(offset=g["buf_offset"], last_index=g["buf_last_index"], lengths=g["buf_lengths"],
                             insertion=g["buf_insertion"], rew=g["rew"], terminated=g["terminated"],
                             truncated=g["truncated"], obs=g["obs"], act=g["act"], obs_next=g["obs_next"])
    for u in range(d["n_updates"]):
        idx = torch.as_tensor(g[f"u{u}_indices"]).cuda()
        ret = eng.preprocess(buf, idx, g[f"u{u}_noise_target"], g[f"u{u}_subset"])
        np.testing.assert_allclose(ret.cpu().numpy(), g[f"u{u}_returns"], rtol=1e-5, atol=2e-5)
        assert eng.will_update_actor() == (f"u{u}_noise_actor" in g)
        noise = g[f"u{u}_noise_actor"] if eng.will_update_actor() else None
        stats, _ = eng.update_with_batch(buf.obs[idx], buf.act[idx], ret, noise)
        s, ref = stats.cpu().numpy(), g[f"u{u}_stats"]
        np.testing.assert_allclose(s[:3], ref[:3], rtol=2e-5, atol=1e-7)
        assert np.isnan(s[3]) == np.isnan(ref[3])
        if not np.isnan(ref[3]):
            np.testing.assert_allclose(s[3], ref[3], rtol=1e-5, atol=1e-6)
        E = cfg.ensemble_size
        cat = lambda ts: torch.cat([t.reshape(-1) for t in ts])  # noqa: E731
        for name, flat, lr in (("actor", cat(S.actor_flat_to_torch(eng.actor, od, ad, H, sizes=sa)), cfg.actor_lr),
                               ("critic", cat(RQ.ensemble_flat_to_torch(eng.critics, E, od, ad, H, sizes=sc)), cfg.critic_lr),
                               ("critic_old", cat(RQ.ensemble_flat_to_torch(eng.critics_old, E, od, ad, H, sizes=sc)), cfg.critic_lr)):
            np.testing.assert_allclose(flat.cpu().numpy()[::61], g[f"u{u}_{name}"], rtol=1e-5, atol=0.02 * lr, err_msg=name)


def test_bad_arguments_fail_loudly():
    from tianshou_amd import redq as RQ

    cfg = OR.REDQConfig(ensemble_size=3, subset_size=2, actor_delay=1)
    eng, _, _ = make_engine(7, 2, 0, cfg)
    z = torch.zeros
    with pytest.raises(ValueError):                         # the actor step needs noise
        eng.update_with_batch(z(4, 7), z(4, 2), z(4))
    with pytest.raises(ValueError):
        eng.target_q(z(4, 7), z(4, 2), [0])                 # subset of the wrong size
    with pytest.raises(Exception):
        eng.target_q(z(4, 7), z(4, 2), [0, 3])              # member index out of range
    with pytest.raises(RuntimeError):
        RQ.REDQEngine(7, 2, eng.actor.cpu(), eng.critics.cpu(), eng.cfg)
