"""GPU parity of the DiscreteSAC row: Categorical policy, twin critics over all actions, entropy-
regularised target, three Adam steps, auto alpha, Polyak -- through the C ABI, against the oracle
(oracle/oracle_dsac.py, pinned to the reference by tests/golden/dsac_*.npz).  Tolerance 1e-5 relative on each
tensor's scale."""
import numpy as np
import pytest
import torch

from oracle import oracle_dsac as ODS
from oracle import oracle_sac as OS
from tests.test_oracle_golden import load_dsac

pytestmark = pytest.mark.gpu
CFG_KEYS = ("gamma", "tau", "n_step", "alpha", "auto_alpha", "target_entropy", "log_alpha0", "actor_lr", "critic_lr",
            "alpha_lr")


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def make_engine(obs_dim, n_act, hidden, seed, cfg):
    from tianshou_amd import dsac as DS
    from tianshou_amd.sac import SACConfig

    from tianshou_amd import widths as W

    nets = ODS.init_params(obs_dim, n_act, hidden, seed)            # hidden: int, (actor h1, actor h2, critic h1, critic h2) or a nested pair
    H = W.engine_hidden([W.layer_widths(list(p.values()), 1) for p in nets])
    flats = [DS.net_flat_from_torch(list(p.values()), obs_dim, n_act, H) for p in nets]
    eng = DS.DiscreteSACEngine(obs_dim, n_act, H, *flats, SACConfig(**{k: getattr(cfg, k) for k in CFG_KEYS}),
                               depth=(len(nets[0]) - 2) // 2)
    return eng, nets


@pytest.mark.parametrize("obs_dim,n_act,hidden,B", [(11, 5, 64, 300), (128, 18, 256, 4096), (33, 2, 32, 1), (4, 64, 96, 77)])
def test_logits_and_target_q_vs_oracle(obs_dim, n_act, hidden, B):
    from tianshou_amd import dsac as DS

    cfg = OS.SACConfig(auto_alpha=True, log_alpha0=-0.7)
    eng, (actor, c1, c2) = make_engine(obs_dim, n_act, hidden, 3, cfg)
    for p, flat in ((actor, eng.actor), (c1, eng.critic1)):                 # layout round trip
        back = DS.net_flat_to_torch(flat, obs_dim, n_act, hidden)
        assert all(torch.equal(a.cpu(), p[k]) for a, k in zip(back, ODS.NET_ORDER))
    obs = torch.randn(B, obs_dim, generator=torch.Generator().manual_seed(B))
    logits = eng.policy_forward(obs)
    assert rel_err(logits.cpu(), ODS.net_forward(actor, obs)) < 1e-5
    st = OS.SACState.create(actor, c1, c2, cfg)
    g = torch.Generator().manual_seed(1)                                  # lagged critics that differ from the live ones
    st.critic1_old = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in c1.items()}
    st.critic2_old = {k: v + 0.05 * torch.randn(v.shape, generator=g) for k, v in c2.items()}
    eng.critic1_old = DS.net_flat_from_torch([st.critic1_old[k] for k in ODS.NET_ORDER], obs_dim, n_act, hidden)
    eng.critic2_old = DS.net_flat_from_torch([st.critic2_old[k] for k in ODS.NET_ORDER], obs_dim, n_act, hidden)
    tq_ref = ODS.target_q(st, cfg, obs)
    np.testing.assert_allclose(eng.target_q(obs).cpu().numpy(), tq_ref.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("obs_dim,n_act,hidden,B,auto,weighted", [(128, 18, 256, 4096, True, False),
                                                                  (11, 5, 64, 200, False, True)])
def test_update_gradients_vs_oracle(obs_dim, n_act, hidden, B, auto, weighted):
    """All three gradients of one update (learning rates 0, so the actor phase sees the same critics)."""
    from tianshou_amd import dsac as DS

    cfg = OS.SACConfig(auto_alpha=auto, log_alpha0=-0.3, alpha=0.15, target_entropy=0.98 * float(np.log(n_act)),
                       actor_lr=0.0, critic_lr=0.0, alpha_lr=0.0, tau=0.0)
    eng, (actor, c1, c2) = make_engine(obs_dim, n_act, hidden, 5, cfg)
    g = torch.Generator().manual_seed(7)
    obs = torch.randn(B, obs_dim, generator=g)
    act = torch.randint(0, n_act, (B,), generator=g)
    ret = torch.randn(B, generator=g) * 2
    weight = torch.rand(B, generator=g) if weighted else None
    st = OS.SACState.create(actor, c1, c2, cfg)
    col: dict = {}
    ref = ODS.update_with_batch(st, cfg, obs, act, ret, weight, collect=col)
    P = eng.lay["count"]
    grads = torch.empty(3 * P, dtype=torch.float32, device="cuda")
    stats, w_out = eng.update_with_batch(obs, act, ret, weight, grads_out=grads)
    s = stats.cpu().numpy()
    np.testing.assert_allclose(s[:3], [ref["actor_loss"], ref["critic1_loss"], ref["critic2_loss"]], rtol=1e-5)
    np.testing.assert_allclose(w_out.cpu().numpy(), ref["weight"].numpy(), rtol=1e-5, atol=1e-5)
    for i, name in enumerate(("critic1", "critic2", "actor")):
        got = DS.net_flat_to_torch(grads[i * P:(i + 1) * P], obs_dim, n_act, hidden)
        for t, key in zip(got, ODS.NET_ORDER):
            assert rel_err(t.cpu(), col[name + "_grads"][key]) < 1e-5, (name, key)
    # zero padding of the internal layout must receive exactly zero gradient
    hw = eng.lay["hw"]
    head = grads[:P][-(hidden + 1) * hw:].reshape(hidden + 1, hw)
    assert torch.count_nonzero(head[:, n_act:]) == 0


@pytest.mark.parametrize("tag", ["auto", "fixed", "widths", "depth3"])
def test_dsac_update_matches_reference_golden(tag):
    """(`widths`: actor Net[40, 72], critics Net[56, 24] in the reference, embedded in Net[96, 96]: tianshou_amd.widths.  `depth3`:
    THREE hidden layers, actor [40, 72, 24], critics [56, 24, 48], embedded in Net[96] * 3 -- gen_golden.py::gen_depth.)"""
    from tianshou_amd import dsac as DS
    from tianshou_amd import widths as W
    from tianshou_amd.buffer import DeviceReplayBuffer

    g, d, cfg, bstate = load_dsac(tag)
    sa, sc = OS.layer_sizes(d["hidden"])
    eng, _ = make_engine(d["obs_dim"], d["n_act"], d["hidden"], d["seed"], cfg)
    assert eng.depth == len(sa)
    buf = DeviceReplayBuffer(offset=g["buf_offset"], last_index=g["buf_last_index"], lengths=g["buf_lengths"],
                             insertion=g["buf_insertion"], rew=g["rew"], terminated=g["terminated"],
                             truncated=g["truncated"], obs=g["obs"], act=g["act"], obs_next=g["obs_next"])
    for u in range(d["n_updates"]):
        idx = torch.as_tensor(g[f"u{u}_indices"]).cuda()
        ret = eng.preprocess(buf, idx)
        np.testing.assert_allclose(ret.cpu().numpy(), g[f"u{u}_returns"], rtol=1e-5, atol=2e-5)
        stats, w = eng.update_with_batch(buf.obs[idx], buf.act[idx], ret, g[f"u{u}_is_weight"])
        s, ref = stats.cpu().numpy(), g[f"u{u}_stats"]
        np.testing.assert_allclose(s[:3], ref[:3], rtol=2e-5)
        np.testing.assert_allclose(s[3], ref[3], rtol=1e-5)
        if cfg.auto_alpha:
            np.testing.assert_allclose(s[4], ref[4], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(w.cpu().numpy(), g[f"u{u}_new_weight"], rtol=1e-5, atol=2e-5)
        for name in ("actor", "critic1", "critic2", "critic1_old", "critic2_old"):
            sz = sa if name == "actor" else sc
            full = DS.net_flat_to_torch(getattr(eng, name), d["obs_dim"], d["n_act"], eng.hidden, depth=eng.depth)
            assert W.padding_is_zero_layers(full, sz), name
            flat = torch.cat([t.reshape(-1) for t in W.unpad_layers(full, sz)])
            lr = cfg.actor_lr if name == "actor" else cfg.critic_lr
            np.testing.assert_allclose(flat.cpu().numpy()[::5], g[f"u{u}_{name}"], rtol=1e-5, atol=0.02 * lr, err_msg=name)


def test_two_updates_vs_oracle_with_live_learning_rates():
    obs_dim, n_act, hidden, B = 40, 6, 128, 512
    cfg = OS.SACConfig(auto_alpha=True, log_alpha0=-0.2, target_entropy=0.98 * float(np.log(n_act)), actor_lr=3e-4,
                       critic_lr=1e-3, alpha_lr=1e-3, tau=0.02)
    eng, (actor, c1, c2) = make_engine(obs_dim, n_act, hidden, 9, cfg)
    st = OS.SACState.create(actor, c1, c2, cfg)
    g = torch.Generator().manual_seed(B)
    for _ in range(2):
        obs = torch.randn(B, obs_dim, generator=g)
        act = torch.randint(0, n_act, (B,), generator=g)
        ret = torch.randn(B, generator=g)
        ref = ODS.update_with_batch(st, cfg, obs, act, ret)
        stats, w = eng.update_with_batch(obs, act, ret)
        s = stats.cpu().numpy()
        np.testing.assert_allclose(s[:3], [ref["actor_loss"], ref["critic1_loss"], ref["critic2_loss"]], rtol=2e-5,
                                   atol=1e-6)
        np.testing.assert_allclose(s[3:], [ref["alpha"], ref["alpha_loss"]], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(w.cpu().numpy(), ref["weight"].numpy(), rtol=1e-5, atol=1e-5)


def test_bad_arguments_fail_loudly():
    from tianshou_amd import dsac as DS
    from tianshou_amd.sac import SACConfig

    with pytest.raises(Exception):
        DS.layout(11, 5, 48)                               # hidden must be a multiple of 32
    with pytest.raises(Exception):
        DS.layout(11, 65, 64)
    n = DS.layout(11, 5, 64)["count"]
    with pytest.raises(RuntimeError):
        DS.DiscreteSACEngine(11, 5, 64, torch.zeros(n), torch.zeros(n), torch.zeros(n), SACConfig())
    eng, _ = make_engine(11, 5, 64, 0, OS.SACConfig())
    with pytest.raises(ValueError):
        eng.update_with_batch(torch.zeros(4, 12), torch.zeros(4, dtype=torch.int64), torch.zeros(4))
