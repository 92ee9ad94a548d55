"""GPU parity of TD3 / DDPG through the C ABI against oracle/oracle_sac.py's TD3 restatement
(pinned to the reference by tests/golden/td3_*.npz)."""
import numpy as np
import pytest
import torch

from oracle import oracle_sac as OS
from tests.test_oracle_golden import load_td3

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def make_engine(obs_dim, act_dim, seed, cfg, hidden=256, activation="relu"):
    """hidden: int, (h1, h2) or (actor h1, actor h2, critic h1, critic h2) -- embedded by zero padding (tianshou_amd.widths)."""
    from tianshou_amd import td3 as T
    from tianshou_amd import widths as W

    actor, c1, c2 = OS.init_td3_params(obs_dim, act_dim, seed, cfg.twin, hidden)
    lists = [list(actor.values()), list(c1.values())] + ([list(c2.values())] if cfg.twin else [])          # (dicts are in layer order)
    H = W.engine_hidden([W.layer_widths(t, 1) for t in lists])
    eng = T.TD3Engine(
        obs_dim, act_dim, T.actor_flat_from_torch(lists[0], obs_dim, act_dim, hidden=H),
        T.critic_flat_from_torch(lists[1], obs_dim, act_dim, hidden=H),
        T.critic_flat_from_torch(lists[2], obs_dim, act_dim, hidden=H) if cfg.twin else None,
        T.TD3Config(**{k: getattr(cfg, k) for k in ("gamma", "tau", "n_step", "twin", "policy_noise", "noise_clip",
                                                     "update_actor_freq", "max_action", "actor_lr", "critic_lr")}),
        hidden=H, depth=OS.depth_of(actor), activation=activation)
    return eng, (actor, c1, c2)


@pytest.mark.parametrize("twin,hidden", [(True, 128), (False, 64)])
def test_other_hidden_widths_vs_oracle(twin, hidden):
    """TD3 / DDPG with Net[h, h], h other than 256 (mujoco_td3.py's width): two updates (the second one steps the actor)
    against the oracle on the per-layer GEMM path."""
    obs_dim, act_dim, B = 17, 6, 80
    cfg = OS.TD3Config(twin=twin, max_action=1.0, actor_lr=3e-4, critic_lr=1e-3, tau=0.01, update_actor_freq=2 if twin else 1)
    eng, (actor, c1, c2) = make_engine(obs_dim, act_dim, 4, cfg, hidden)
    st = OS.TD3State.create(actor, c1, c2, cfg)
    g = torch.Generator().manual_seed(hidden)
    for _ in range(2):
        obs = torch.randn(B, obs_dim, generator=g)
        act = torch.rand(B, act_dim, generator=g) * 2 - 1
        ret = torch.randn(B, generator=g)
        ref = OS.td3_update_with_batch(st, cfg, obs, act, ret)
        stats, w = eng.update_with_batch(obs, act, ret)
        s = stats.cpu().numpy()
        np.testing.assert_allclose(s[:2], [ref["actor_loss"], ref["critic1_loss"]], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(w.cpu().numpy(), ref["weight"].numpy(), rtol=1e-5, atol=1e-5)
    with torch.no_grad():
        assert rel_err(eng.policy_forward(obs).cpu(), OS.det_actor_forward(st.actor, obs, cfg.max_action)) < 1e-5


@pytest.mark.parametrize("twin", [True, False])
def test_policy_target_and_gradients_vs_oracle(twin):
    from tianshou_amd import td3 as T

    obs_dim, act_dim, B = 376, 17, 512
    cfg = OS.TD3Config(twin=twin, max_action=1.5, actor_lr=0.0, critic_lr=0.0, tau=0.0, update_actor_freq=1)
    eng, (actor, c1, c2) = make_engine(obs_dim, act_dim, 21, cfg)
    for a, k in zip(T.actor_flat_to_torch(eng.actor, obs_dim, act_dim), OS.DET_ACTOR_ORDER):
        assert torch.equal(a.cpu(), actor[k]), k
    g = torch.Generator().manual_seed(3)
    obs = torch.randn(B, obs_dim, generator=g)
    act = torch.rand(B, act_dim, generator=g) * 3 - 1.5
    ret, noise = torch.randn(B, generator=g), torch.randn(B, act_dim, generator=g)
    st = OS.TD3State.create(actor, c1, c2, cfg)
    with torch.no_grad():
        assert rel_err(eng.policy_forward(obs).cpu(), OS.det_actor_forward(actor, obs, cfg.max_action)) < 1e-5
    np.testing.assert_allclose(eng.target_q(obs, noise if twin else None).cpu().numpy(),
                               OS.td3_target_q(st, cfg, obs, noise).flatten().numpy(), rtol=1e-5, atol=1e-5)
    col: dict = {}
    ref = OS.td3_update_with_batch(st, cfg, obs, act, ret, collect=col)
    lay = eng.lay
    grads = torch.zeros(2 * lay["critic_count"] + lay["actor_count"], dtype=torch.float32, device="cuda")
    stats, w = eng.update_with_batch(obs, act, ret, grads_out=grads)
    s = stats.cpu().numpy()
    np.testing.assert_allclose(s[:2], [ref["actor_loss"], ref["critic1_loss"]], rtol=1e-5)
    np.testing.assert_allclose(w.cpu().numpy(), ref["weight"].numpy(), rtol=1e-5, atol=1e-5)
    pc = lay["critic_count"]
    got = {"critic1": T.critic_flat_to_torch(grads[:pc], obs_dim, act_dim),
           "actor": T.actor_flat_to_torch(grads[2 * pc:], obs_dim, act_dim)}
    if twin:
        got["critic2"] = T.critic_flat_to_torch(grads[pc:2 * pc], obs_dim, act_dim)
    for name, tensors in got.items():
        order = OS.DET_ACTOR_ORDER if name == "actor" else OS.CRITIC_ORDER
        for t, key in zip(tensors, order):
            assert rel_err(t.cpu(), col[name + "_grads"][key]) < 2e-5, (name, key)


@pytest.mark.parametrize("tag", ["twin", "ddpg", "widths", "ddpg_widths", "depth4", "ddpg_depth1", "tanh3"])
def test_update_matches_reference_golden(tag):
    """(`widths`: the TD3 paper's Net[400, 300] embedded in Net[416, 416]; `ddpg_widths`: actor [24, 56], critic [40, 24] in 64;
    `depth4`: FOUR hidden layers, actor [64, 64, 32, 32] and critics [48, 64, 64, 40] in Net[64] * 4, max_action 1.5;
    `ddpg_depth1`: ONE hidden layer, actor [128], critic [64] -- fixtures the unmodified reference wrote, gen_golden.py::gen_depth.)"""
    from tianshou_amd import td3 as T
    from tianshou_amd.buffer import DeviceReplayBuffer

    from tianshou_amd import widths as W

    g, d, cfg, bstate = load_td3(tag)
    eng, _ = make_engine(d["obs_dim"], d["act_dim"], d["seed"], cfg, d["hidden"], d["activation"])      # (`tanh3`: three nn.Tanh layers)
    sa, sc = OS.layer_sizes(d["hidden"])
    assert eng.depth == len(sa) == len(sc)
    buf = DeviceReplayBuffer(offset=g["buf_offset"], last_index=g["buf_last_index"], lengths=g["buf_lengths"],
                             insertion=g["buf_insertion"], rew=g["rew"], terminated=g["terminated"],
                             truncated=g["truncated"], obs=g["obs"], act=g["act"], obs_next=g["obs_next"])
    for u in range(d["n_updates"]):
        idx = torch.as_tensor(g[f"u{u}_indices"]).cuda()
        ret = eng.preprocess(buf, idx, g[f"u{u}_noise"] if d["twin"] else None)
        np.testing.assert_allclose(ret.cpu().numpy(), g[f"u{u}_returns"], rtol=1e-5, atol=2e-5)
        stats, _ = eng.update_with_batch(buf.obs[idx], buf.act[idx], ret)
        s, ref = stats.cpu().numpy(), g[f"u{u}_stats"]
        np.testing.assert_allclose(s[:len(ref)], ref, rtol=2e-5, atol=1e-7)
        names = ["actor", "critic1", "actor_old", "critic1_old"] + (["critic2", "critic2_old"] if d["twin"] else [])
        for name in names:
            conv = T.actor_flat_to_torch if name.startswith("actor") else T.critic_flat_to_torch
            sz = sa if name.startswith("actor") else sc
            full = conv(getattr(eng, name), d["obs_dim"], d["act_dim"], eng.hidden, depth=eng.depth)
            assert W.padding_is_zero_layers(full, sz), name       # the embedding of Net[h1, ...] in Net[h] * d stays an embedding
            flat = torch.cat([t.reshape(-1) for t in W.unpad_layers(full, sz)])
            lr = cfg.actor_lr if name.startswith("actor") else cfg.critic_lr
            np.testing.assert_allclose(flat.cpu().numpy()[::61], g[f"u{u}_{name}"], rtol=1e-5, atol=0.02 * lr,
                                       err_msg=name)
