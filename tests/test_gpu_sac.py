"""GPU parity of the SAC row: tanh-Gaussian policy, twin lagged critics, n-step target,
three Adam steps, auto alpha, Polyak -- through the C ABI, against the oracle (oracle/oracle_sac.py, pinned to
the reference by tests/golden/sac_*.npz).  Tolerance 1e-5 relative on each tensor's scale."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle import oracle_sac as OS
from tests.test_oracle_golden import load_sac

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def make_engine(obs_dim, act_dim, seed, cfg, hidden=256, activation="relu"):
    """hidden: int, or (actor h1, actor h2, critic h1, critic h2) -- embedded by zero padding (tianshou_amd.widths)."""
    from tianshou_amd import sac as S
    from tianshou_amd import widths as W

    actor, c1, c2 = OS.init_sac_params(obs_dim, act_dim, seed, hidden)
    lists = [list(actor.values()), list(c1.values()), list(c2.values())]            # (dicts are in layer order)
    H = W.engine_hidden([W.layer_widths(t, 2 if i == 0 else 1) for i, t in enumerate(lists)])
    eng = S.SACEngine(
        obs_dim, act_dim,
        S.actor_flat_from_torch(lists[0], obs_dim, act_dim, hidden=H),
        S.critic_flat_from_torch(lists[1], obs_dim, act_dim, hidden=H),
        S.critic_flat_from_torch(lists[2], obs_dim, act_dim, hidden=H),
        S.SACConfig(**{k: getattr(cfg, k) for k in ("gamma", "tau", "n_step", "alpha", "auto_alpha", "target_entropy",
                                                     "log_alpha0", "actor_lr", "critic_lr", "alpha_lr")}), hidden=H,
        depth=OS.depth_of(actor), max_action=getattr(cfg, "max_action", 0.0), activation=activation)
    return eng, (actor, c1, c2)


@pytest.mark.parametrize("obs_dim,act_dim", [(376, 17), (23, 5), (32, 32), (7, 1)])
def test_layout_round_trip(obs_dim, act_dim):
    from tianshou_amd import sac as S

    actor, c1, _ = OS.init_sac_params(obs_dim, act_dim, 1)
    fa = S.actor_flat_from_torch([actor[k] for k in OS.ACTOR_ORDER], obs_dim, act_dim)
    fc = S.critic_flat_from_torch([c1[k] for k in OS.CRITIC_ORDER], obs_dim, act_dim)
    for a, k in zip(S.actor_flat_to_torch(fa, obs_dim, act_dim), OS.ACTOR_ORDER):
        assert torch.equal(a.cpu(), actor[k]), k
    for a, k in zip(S.critic_flat_to_torch(fc, obs_dim, act_dim), OS.CRITIC_ORDER):
        assert torch.equal(a.cpu(), c1[k]), k


@pytest.mark.parametrize("obs_dim,act_dim,B", [(376, 17, 300), (23, 5, 64), (7, 1, 33), (1000, 6, 50)])
def test_policy_and_target_q_vs_oracle(obs_dim, act_dim, B):
    cfg = OS.SACConfig(auto_alpha=True, log_alpha0=-0.7, target_entropy=-float(act_dim))
    eng, (actor, c1, c2) = make_engine(obs_dim, act_dim, 3, cfg)
    g = torch.Generator().manual_seed(B)
    obs = torch.randn(B, obs_dim, generator=g)
    noise = torch.randn(B, act_dim, generator=g)
    act_ref, logp_ref, _, _ = OS.policy_forward(actor, obs, noise)
    act, logp = eng.policy_forward(obs.cuda(), noise.cuda())
    assert rel_err(act.cpu(), act_ref) < 1e-5
    np.testing.assert_allclose(logp.cpu().numpy(), logp_ref.numpy(), rtol=1e-5, atol=1e-5 * act_dim)
    act_det, _ = eng.policy_forward(obs.cuda(), None)                  # dist.mode: tanh(mu)
    mu, _ = OS.actor_forward(actor, obs)
    assert rel_err(act_det.cpu(), torch.tanh(mu)) < 1e-5
    st = OS.SACState.create(actor, c1, c2, cfg)
    tq_ref = OS.target_q(st, cfg, obs, noise).flatten()
    tq = eng.target_q(obs.cuda(), noise.cuda())
    np.testing.assert_allclose(tq.cpu().numpy(), tq_ref.numpy(), rtol=1e-5, atol=1e-5 * act_dim)


@pytest.mark.parametrize("obs_dim,act_dim,B,auto,weighted", [(376, 17, 4096, True, False), (23, 5, 200, False, True)])
def test_update_gradients_vs_oracle(obs_dim, act_dim, B, auto, weighted):
    """All three gradients of one SAC update (learning rates 0, so the actor phase sees the same critics)."""
    from tianshou_amd import sac as S

    cfg = OS.SACConfig(auto_alpha=auto, log_alpha0=-0.3, alpha=0.15, target_entropy=-float(act_dim),
                       actor_lr=0.0, critic_lr=0.0, alpha_lr=0.0, tau=0.0)
    eng, (actor, c1, c2) = make_engine(obs_dim, act_dim, 5, cfg)
    g = torch.Generator().manual_seed(7)
    obs = torch.randn(B, obs_dim, generator=g)
    act = torch.rand(B, act_dim, generator=g) * 2 - 1
    ret = torch.randn(B, generator=g) * 2
    noise = torch.randn(B, act_dim, generator=g)
    weight = torch.rand(B, generator=g) if weighted else None
    st = OS.SACState.create(actor, c1, c2, cfg)
    col: dict = {}
    ref = OS.update_with_batch(st, cfg, obs, act, ret, noise, weight, collect=col)
    lay = eng.lay
    grads = torch.empty(2 * lay["critic_count"] + lay["actor_count"], dtype=torch.float32, device="cuda")
    stats, w_out = eng.update_with_batch(obs, act, ret, noise, weight, grads_out=grads)
    s = stats.cpu().numpy()
    np.testing.assert_allclose(s[:3], [ref["actor_loss"], ref["critic1_loss"], ref["critic2_loss"]], rtol=1e-5)
    np.testing.assert_allclose(w_out.cpu().numpy(), ref["weight"].numpy(), rtol=1e-5, atol=1e-5)
    # yardstick: the same gradients in float64.  The engine must be within 1e-5 of them, or -- where the sum
    # over the batch cancels heavily (policy-gradient terms) and float32 itself is noisier than that -- at least
    # as close as the reference's own float32 evaluation.
    g64 = OS.gradients(actor, c1, c2, OS.alpha_value(st, cfg), obs, act, ret, noise, weight, dtype=torch.float64)
    pc = lay["critic_count"]
    got = {"critic1": S.critic_flat_to_torch(grads[:pc], obs_dim, act_dim),
           "critic2": S.critic_flat_to_torch(grads[pc:2 * pc], obs_dim, act_dim),
           "actor": S.actor_flat_to_torch(grads[2 * pc:], obs_dim, act_dim)}
    for name, order in (("critic1", OS.CRITIC_ORDER), ("critic2", OS.CRITIC_ORDER), ("actor", OS.ACTOR_ORDER)):
        for t, key in zip(got[name], order):
            exact = g64[name + "_grads"][key]
            e_gpu = rel_err(t.cpu(), exact)
            e_ref = rel_err(col[name + "_grads"][key], exact)
            assert e_gpu < max(1e-5, 2 * e_ref), (name, key, e_gpu, e_ref)
    # zero padding of the internal layout must receive exactly zero gradient
    l1 = grads[:lay["critic_l2"]].reshape(lay["kc"] + 1, 256)
    assert torch.count_nonzero(l1[obs_dim + act_dim:lay["kc"]]) == 0


@pytest.mark.parametrize("tag", ["auto", "fixed", "widths", "depth3", "depth1", "bounded", "bounded_depth3", "tanh"])
def test_sac_update_matches_reference_golden(tag):
    """(`widths`: actor Net[48, 80], critics Net[72, 40] in the reference; the engine runs them embedded in Net[96, 96] and every
    padding entry of parameters, lagged parameters and Adam moments stays exactly zero.  `depth3`: THREE hidden layers, actor
    [64, 48, 32] and critics [40, 56, 24] in Net[64] * 3; `depth1`: ONE hidden layer [96], fixed alpha, 2-step returns --
    fixtures the unmodified reference wrote, gen_golden.py::gen_depth; the engine runs them layer by layer, ts_mlp_set_trunk.
    `bounded` / `bounded_depth3`: the class-default actor `unbounded=False` -- mu = max_action * tanh(mu), max_action 1.5 / 0.8,
    ts_sac_set_actor_bound -- on Net[256, 256] (the fused kernels) and on actor [48, 64, 40] / critics [64, 32, 32].
    `tanh`: Net(activation=nn.Tanh) trunks, actor [64, 48] / critics [40, 72] -- ts_mlp_set_activation.)"""
    from tianshou_amd import sac as S
    from tianshou_amd import widths as W
    from tianshou_amd.buffer import DeviceReplayBuffer

    g, d, cfg, bstate = load_sac(tag)
    eng, _ = make_engine(d["obs_dim"], d["act_dim"], d["seed"], cfg, d["hidden"], d["activation"])
    sa, sc = OS.layer_sizes(d["hidden"])
    assert eng.depth == len(sa) == len(sc)
    sizes = {"actor": sa, "critic1": sc, "critic2": sc, "critic1_old": sc, "critic2_old": sc}
    buf = DeviceReplayBuffer(offset=g["buf_offset"], last_index=g["buf_last_index"], lengths=g["buf_lengths"],
                             insertion=g["buf_insertion"], rew=g["rew"], terminated=g["terminated"],
                             truncated=g["truncated"], obs=g["obs"], act=g["act"], obs_next=g["obs_next"])
    for u in range(d["n_updates"]):
        idx = torch.as_tensor(g[f"u{u}_indices"]).cuda()
        ret = eng.preprocess(buf, idx, g[f"u{u}_noise_target"])
        np.testing.assert_allclose(ret.cpu().numpy(), g[f"u{u}_returns"], rtol=1e-5, atol=2e-5)
        stats, _ = eng.update_with_batch(buf.obs[idx], buf.act[idx], ret, g[f"u{u}_noise_actor"])
        s, ref = stats.cpu().numpy(), g[f"u{u}_stats"]
        np.testing.assert_allclose(s[:3], ref[:3], rtol=2e-5)
        np.testing.assert_allclose(s[3], ref[3], rtol=1e-5)
        if cfg.auto_alpha:
            np.testing.assert_allclose(s[4], ref[4], rtol=1e-5, atol=1e-6)
        for name, to_torch in (("actor", S.actor_flat_to_torch), ("critic1", S.critic_flat_to_torch),
                               ("critic2", S.critic_flat_to_torch), ("critic1_old", S.critic_flat_to_torch),
                               ("critic2_old", S.critic_flat_to_torch)):
            full = to_torch(getattr(eng, name), d["obs_dim"], d["act_dim"], eng.hidden, depth=eng.depth)
            assert W.padding_is_zero_layers(full, sizes[name]), name
            for sfx in ("_m", "_v"):
                if hasattr(eng, name + sfx):
                    assert W.padding_is_zero_layers(to_torch(getattr(eng, name + sfx), d["obs_dim"], d["act_dim"], eng.hidden,
                                                             depth=eng.depth), sizes[name])
            flat = torch.cat([t.reshape(-1) for t in W.unpad_layers(full, sizes[name])])
            lr = cfg.actor_lr if name == "actor" else cfg.critic_lr
            # Adam's first steps move every weight by ~lr whatever its gradient: compare on lr's scale
            np.testing.assert_allclose(flat.cpu().numpy()[::61], g[f"u{u}_{name}"], rtol=1e-5, atol=0.02 * lr,
                                       err_msg=name)


@pytest.mark.parametrize("obs_dim,act_dim,B", [(7, 1, 33), (64, 32, 40), (33, 3, 1), (1000, 6, 50)])
def test_update_other_shapes_vs_oracle(obs_dim, act_dim, B):
    """Edge shapes: one action, 32 actions (the head's limit), obs widths around the 32-column padding, B = 1."""
    cfg = OS.SACConfig(auto_alpha=True, log_alpha0=-0.2, target_entropy=-float(act_dim), actor_lr=3e-4,
                       critic_lr=1e-3, alpha_lr=1e-3, tau=0.02)
    eng, (actor, c1, c2) = make_engine(obs_dim, act_dim, 9, cfg)
    st = OS.SACState.create(actor, c1, c2, cfg)
    g = torch.Generator().manual_seed(B)
    for _ in range(2):
        obs = torch.randn(B, obs_dim, generator=g)
        act = torch.rand(B, act_dim, generator=g) * 2 - 1
        ret, noise = torch.randn(B, generator=g), torch.randn(B, act_dim, generator=g)
        ref = OS.update_with_batch(st, cfg, obs, act, ret, noise)
        stats, w = eng.update_with_batch(obs, act, ret, noise)
        s = stats.cpu().numpy()
        np.testing.assert_allclose(s[:3], [ref["actor_loss"], ref["critic1_loss"], ref["critic2_loss"]], rtol=2e-5,
                                   atol=1e-6)
        np.testing.assert_allclose(s[3], ref["alpha"], rtol=1e-5)
        np.testing.assert_allclose(w.cpu().numpy(), ref["weight"].numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("hidden,obs_dim,act_dim,B", [(128, 23, 5, 96), (96, 376, 17, 64), (512, 11, 3, 40)])
def test_other_hidden_widths_vs_oracle(hidden, obs_dim, act_dim, B):
    """Net(hidden_sizes=[h, h]) with h other than the example's 256 (utils/net/common.py:246-369 takes any): the same update
    on the per-layer GEMM kernels (ts_mlp_set_hidden), two updates against the oracle; a 256-wide engine sharing the
    device's workspace is interleaved to show that the width travels with each call."""
    from tianshou_amd import sac as S

    cfg = OS.SACConfig(auto_alpha=True, log_alpha0=-0.2, target_entropy=-float(act_dim), actor_lr=3e-4,
                       critic_lr=1e-3, alpha_lr=1e-3, tau=0.02)
    eng, (actor, c1, c2) = make_engine(obs_dim, act_dim, 9, cfg, hidden)
    other, _ = make_engine(7, 2, 1, cfg)                      # hidden 256, same default workspace
    lay = S.layout(obs_dim, act_dim, hidden)
    assert eng.actor.numel() == lay["actor_count"] == (lay["ka"] + 1) * hidden + (hidden + 1) * hidden + (hidden + 1) * 64
    back = S.actor_flat_to_torch(eng.actor, obs_dim, act_dim, hidden)
    for t, k in zip(back, OS.ACTOR_ORDER):
        assert torch.equal(t.cpu(), actor[k])
    st = OS.SACState.create(actor, c1, c2, cfg)
    g = torch.Generator().manual_seed(B)
    for _ in range(2):
        obs = torch.randn(B, obs_dim, generator=g)
        act = torch.rand(B, act_dim, generator=g) * 2 - 1
        ret, noise = torch.randn(B, generator=g), torch.randn(B, act_dim, generator=g)
        other.update_with_batch(torch.randn(8, 7), torch.rand(8, 2), torch.randn(8), torch.randn(8, 2))
        ref = OS.update_with_batch(st, cfg, obs, act, ret, noise)
        stats, w = eng.update_with_batch(obs, act, ret, noise)
        s = stats.cpu().numpy()
        np.testing.assert_allclose(s[:3], [ref["actor_loss"], ref["critic1_loss"], ref["critic2_loss"]], rtol=2e-5,
                                   atol=1e-6)
        np.testing.assert_allclose(s[3], ref["alpha"], rtol=1e-5)
        np.testing.assert_allclose(w.cpu().numpy(), ref["weight"].numpy(), rtol=1e-5, atol=1e-5)
    a_act, a_logp = eng.policy_forward(obs, noise)
    r_act, r_logp = OS.policy_forward(st.actor, obs, noise)[:2]
    assert rel_err(a_act.cpu(), r_act) < 1e-5 and rel_err(a_logp.cpu().flatten(), r_logp.flatten()) < 1e-5
    assert make_engine(5, 2, 0, cfg, hidden=100)[0].hidden == 128       # no multiple of 32: embedded by zero padding (round 6)
    with pytest.raises(NotImplementedError):
        make_engine(5, 2, 0, cfg, hidden=1100)                # beyond the kernels' 1024


@pytest.mark.parametrize("hidden,obs_dim,act_dim,B", [(((256, 256, 256), (256, 256, 256)), 376, 17, 512), (((64,), (96,)), 23, 5, 200),
                                                      (((40, 72, 56, 24, 88), (32, 32, 64, 64, 32)), 11, 3, 65),
                                                      (((128,) * 6, (128,) * 6), 17, 6, 96)])
def test_other_depths_vs_oracle(hidden, obs_dim, act_dim, B):
    _other_depths(hidden, obs_dim, act_dim, B, "relu")


@pytest.mark.parametrize("hidden,obs_dim,act_dim,B", [(((256, 256), (256, 256)), 376, 17, 512), (((40, 72, 56), (32, 64, 32)), 11, 3, 65)])
def test_tanh_trunks_vs_oracle(hidden, obs_dim, act_dim, B):
    """Net(activation=nn.Tanh) trunks (ts_mlp_set_activation): the same checks as test_other_depths_vs_oracle -- at [256, 256] too,
    where ReLU trunks take the fused kernels and tanh trunks the per-layer path."""
    with OS.activation("tanh"):
        _other_depths(hidden, obs_dim, act_dim, B, "tanh")


def _other_depths(hidden, obs_dim, act_dim, B, activation):
    """Net(hidden_sizes=[...]) of 1, 3, 5 and 6 hidden layers (utils/net/common.py:246-369 takes any list; round 6): the update
    runs layer by layer on the GEMM kernels (ts_mlp_set_trunk).  Gradients of the first update against the float64 yardstick
    as in test_update_gradients_vs_oracle, two more updates against the oracle, the policy / target entry points, and a
    two-layer engine on the same workspace in between (the depth travels with each call)."""
    from tianshou_amd import sac as S
    from tianshou_amd import widths as W

    cfg = OS.SACConfig(auto_alpha=True, log_alpha0=-0.2, target_entropy=-float(act_dim), actor_lr=3e-4,
                       critic_lr=1e-3, alpha_lr=1e-3, tau=0.02)
    eng, (actor, c1, c2) = make_engine(obs_dim, act_dim, 9, cfg, hidden, activation)
    other, _ = make_engine(7, 2, 1, cfg)                      # Net[256, 256] ReLU, same default workspace
    sa, sc = OS.layer_sizes(hidden)
    assert eng.depth == len(sa)
    for t, k in zip(S.actor_flat_to_torch(eng.actor, obs_dim, act_dim, eng.hidden, sizes=sa), actor):
        assert torch.equal(t.cpu(), actor[k]), k
    for t, k in zip(S.critic_flat_to_torch(eng.critic2, obs_dim, act_dim, eng.hidden, sizes=sc), c2):
        assert torch.equal(t.cpu(), c2[k]), k
    st = OS.SACState.create(actor, c1, c2, cfg)
    g = torch.Generator().manual_seed(B)
    for u in range(3):
        obs = torch.randn(B, obs_dim, generator=g)
        act = torch.rand(B, act_dim, generator=g) * 2 - 1
        ret, noise = torch.randn(B, generator=g), torch.randn(B, act_dim, generator=g)
        weight = torch.rand(B, generator=g) + 0.5 if u == 1 else None
        other.update_with_batch(torch.randn(8, 7), torch.rand(8, 2), torch.randn(8), torch.randn(8, 2))
        pc = eng.critic1.numel()
        grads = torch.empty(2 * pc + eng.actor.numel(), dtype=torch.float32, device="cuda") if u == 0 else None
        col = {}
        before = (dict(st.actor), dict(st.critic1), dict(st.critic2), OS.alpha_value(st, cfg))
        ref = OS.update_with_batch(st, cfg, obs, act, ret, noise, weight, collect=col)
        stats, w = eng.update_with_batch(obs, act, ret, noise, weight, grads_out=grads)
        s = stats.cpu().numpy()
        np.testing.assert_allclose(s[:3], [ref["actor_loss"], ref["critic1_loss"], ref["critic2_loss"]], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(s[3], ref["alpha"], rtol=1e-5)
        # TD errors: float32-exact on the first update (measured 1e-7 .. 4e-7); afterwards the two implementations' parameters
        # differ where Adam's sign-like first steps (lr * g / (|g| + eps)) amplify the rounding noise of near-zero gradients --
        # the reason the fixture replays compare parameters at 0.02 * lr -- and the TD errors follow (3e-5 after two updates)
        np.testing.assert_allclose(w.cpu().numpy(), ref["weight"].numpy(), rtol=1e-5, atol=1e-5 if u == 0 else 2e-4)
        if u == 0:
            # (the actor's gradient is taken against the UPDATED critics: evaluate the float64 yardstick with them)
            g64c = OS.gradients(before[0], before[1], before[2], before[3], obs, act, ret, noise, weight, dtype=torch.float64)
            g64a = OS.gradients(before[0], st.critic1, st.critic2, before[3], obs, act, ret, noise, weight, dtype=torch.float64)
            got = {"critic1": S.critic_flat_to_torch(grads[:pc], obs_dim, act_dim, eng.hidden, sizes=sc),
                   "critic2": S.critic_flat_to_torch(grads[pc:2 * pc], obs_dim, act_dim, eng.hidden, sizes=sc),
                   "actor": S.actor_flat_to_torch(grads[2 * pc:], obs_dim, act_dim, eng.hidden, sizes=sa)}
            for name in ("critic1", "critic2", "actor"):
                exact_all = (g64a if name == "actor" else g64c)[name + "_grads"]
                for t, key in zip(got[name], exact_all):
                    e_gpu, e_ref = rel_err(t.cpu(), exact_all[key]), rel_err(col[name + "_grads"][key], exact_all[key])
                    assert e_gpu < max(1e-5, 2 * e_ref), (name, key, e_gpu, e_ref)
            # padding entries of the embedded network receive exactly zero gradient
            assert W.padding_is_zero_layers(S.actor_flat_to_torch(grads[2 * pc:], obs_dim, act_dim, eng.hidden, depth=eng.depth), sa)
            assert W.padding_is_zero_layers(S.critic_flat_to_torch(grads[:pc], obs_dim, act_dim, eng.hidden, depth=eng.depth), sc)
    a_act, a_logp = eng.policy_forward(obs, noise)
    r_act, r_logp = OS.policy_forward(st.actor, obs, noise)[:2]
    assert rel_err(a_act.cpu(), r_act) < 1e-5 and rel_err(a_logp.cpu().flatten(), r_logp.flatten()) < 1e-5
    tq = eng.target_q(obs, noise)
    assert rel_err(tq.cpu(), OS.target_q(st, cfg, obs, noise).flatten()) < 2e-5
    for name, sizes, conv in (("actor", sa, S.actor_flat_to_torch), ("critic1", sc, S.critic_flat_to_torch), ("critic2_old", sc, S.critic_flat_to_torch)):
        for t, k in zip(conv(getattr(eng, name), obs_dim, act_dim, eng.hidden, sizes=sizes), getattr(st, name)):
            lr = cfg.actor_lr if name == "actor" else cfg.critic_lr
            # (every entry of every tensor after three Adam steps: the same sign-like amplification, up to 0.04 * lr on one entry in 10^5)
            np.testing.assert_allclose(t.cpu().numpy(), getattr(st, name)[k].numpy(), rtol=1e-5, atol=0.1 * lr, err_msg=f"{name}.{k}")
    with pytest.raises(NotImplementedError):
        make_engine(5, 2, 0, cfg, hidden=((32,) * 7, (32,) * 7))          # beyond TS_MLP_MAX_HIDDEN_LAYERS


@pytest.mark.parametrize("hidden,obs_dim,act_dim,B", [(256, 376, 17, 1024), (((64, 32, 48), (32, 64, 32)), 23, 5, 130)])
def test_bounded_actor_vs_oracle(hidden, obs_dim, act_dim, B):
    """ContinuousActorProbabilistic(unbounded=False) -- the class default: mu = max_action * tanh(mu), continuous.py:230-231 --
    under SAC (ts_sac_set_actor_bound): the policy / target entry points, the three gradients against the float64 yardstick
    (the actor's goes back through max_action * (1 - tanh^2)), and an unbounded engine on the same workspace in between."""
    from tianshou_amd import sac as S

    cfg = OS.SACConfig(auto_alpha=True, log_alpha0=-0.3, target_entropy=-float(act_dim), actor_lr=0.0, critic_lr=0.0, alpha_lr=0.0,
                       tau=0.0, max_action=1.3)
    eng, (actor, c1, c2) = make_engine(obs_dim, act_dim, 5, cfg, hidden)
    other, _ = make_engine(7, 2, 1, OS.SACConfig())                # unbounded, same default workspace
    assert eng.max_action == 1.3 and other.max_action == 0.0
    sa, sc = OS.layer_sizes(hidden)
    st = OS.SACState.create(actor, c1, c2, cfg)
    g = torch.Generator().manual_seed(B)
    obs = torch.randn(B, obs_dim, generator=g) * 2
    act = torch.rand(B, act_dim, generator=g) * 2 - 1
    ret, noise = torch.randn(B, generator=g), torch.randn(B, act_dim, generator=g)
    a_act, a_logp = eng.policy_forward(obs, noise)
    other.policy_forward(torch.randn(8, 7), torch.randn(8, 2))
    r_act, r_logp, r_mu, _ = OS.policy_forward(st.actor, obs, noise, cfg.max_action)
    assert float(r_mu.abs().max()) <= 1.3 and float(r_mu.abs().max()) > 0.05           # the bound is live
    assert rel_err(a_act.cpu(), r_act) < 1e-5 and rel_err(a_logp.cpu().flatten(), r_logp.flatten()) < 1e-5
    u_act, _, _, _ = OS.policy_forward(st.actor, obs, noise, 0.0)
    assert rel_err(a_act.cpu(), u_act) > 1e-3                                           # and differs from the unbounded actor
    assert rel_err(eng.target_q(obs, noise).cpu(), OS.target_q(st, cfg, obs, noise).flatten()) < 2e-5
    pc = eng.critic1.numel()
    grads = torch.empty(2 * pc + eng.actor.numel(), dtype=torch.float32, device="cuda")
    col = {}
    ref = OS.update_with_batch(st, cfg, obs, act, ret, noise, None, collect=col)
    stats, _ = eng.update_with_batch(obs, act, ret, noise, None, grads_out=grads)
    np.testing.assert_allclose(stats.cpu().numpy()[:3], [ref["actor_loss"], ref["critic1_loss"], ref["critic2_loss"]], rtol=1e-5)
    g64 = OS.gradients(actor, c1, c2, OS.alpha_value(st, cfg), obs, act, ret, noise, None, dtype=torch.float64, max_action=cfg.max_action)
    got = S.actor_flat_to_torch(grads[2 * pc:], obs_dim, act_dim, eng.hidden, sizes=sa)
    for t, key in zip(got, g64["actor_grads"]):
        e_gpu, e_ref = rel_err(t.cpu(), g64["actor_grads"][key]), rel_err(col["actor_grads"][key], g64["actor_grads"][key])
        assert e_gpu < max(1e-5, 2 * e_ref), (key, e_gpu, e_ref)


def test_twin_critics_on_two_streams_with_generation_2():
    """hidden = 512 is not on the fused-MLP path, so the twin critics run their backward chains concurrently on the caller's
    stream and the workspace's side stream; with the second-generation kernels forced on, each chain transposes its
    weights into workspace scratch (ts_conv2.hip conv2_dgrad).  The scratch and the class tables are per launch stream:
    three updates must agree with the first-generation kernels (different summation order, amplified by three Adam steps
    whose first updates are lr * sign-like: 1e-4 of the largest parameter; a clobbered scratch buffer would show up as
    errors of order one)."""
    from tianshou_amd import _lib

    lib = _lib.load()
    obs_dim, act_dim, B, hidden = 40, 4, 128, 512
    cfg = OS.SACConfig(auto_alpha=True, log_alpha0=-0.2, target_entropy=-float(act_dim), actor_lr=3e-4,
                       critic_lr=1e-3, alpha_lr=1e-3, tau=0.02)
    out = {}
    prev = lib.ts_conv_set_generation(0)
    try:
        for gen in (-1, 1):
            lib.ts_conv_set_generation(gen)
            eng, _ = make_engine(obs_dim, act_dim, 4, cfg, hidden)
            g = torch.Generator().manual_seed(77)
            stats = None
            for _ in range(3):
                obs = torch.randn(B, obs_dim, generator=g)
                act = torch.rand(B, act_dim, generator=g) * 2 - 1
                ret, noise = torch.randn(B, generator=g), torch.randn(B, act_dim, generator=g)
                stats, _w = eng.update_with_batch(obs, act, ret, noise)
            torch.cuda.synchronize()
            out[gen] = (stats.cpu(), eng.actor.cpu().clone(), eng.critic1.cpu().clone(), eng.critic2.cpu().clone())
    finally:
        lib.ts_conv_set_generation(prev)
    np.testing.assert_allclose(out[1][0].numpy()[:4], out[-1][0].numpy()[:4], rtol=2e-5, atol=1e-6)
    for a, b in zip(out[1][1:], out[-1][1:]):
        assert rel_err(a, b) < 1e-4


def test_bad_arguments_fail_loudly():
    from tianshou_amd import _lib
    from tianshou_amd import sac as S

    with pytest.raises(_lib.EngineError):
        S.layout(17, 33)                                  # act_dim > 32 is outside the head layout
    cfg = OS.SACConfig()
    eng, _ = make_engine(11, 3, 1, cfg)
    with pytest.raises(ValueError):
        eng.update_with_batch(torch.zeros(4, 12), torch.zeros(4, 3), torch.zeros(4), torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        S.SACEngine(11, 3, eng.actor.cpu(), eng.critic1.cpu(), eng.critic2.cpu(), S.SACConfig())


@pytest.mark.parametrize("auto,weighted", [(True, True), (False, False)])
def test_phased_update_is_bit_identical(auto, weighted):
    """ts_sac_update_phase 1, 2, 4, 8 (DataParallelSAC at world 1: no exchange) == ts_sac_update, bit for bit:
    parameters, Adam moments, lagged critics, log_alpha, stats and the new PER weights, over three updates."""
    from tianshou_amd.distributed import DataParallelSAC

    obs_dim, act_dim, B = 23, 5, 300
    cfg = OS.SACConfig(auto_alpha=auto, log_alpha0=-0.3, alpha=0.15, target_entropy=-float(act_dim), tau=0.01)
    eng_a, _ = make_engine(obs_dim, act_dim, 5, cfg)
    eng_b, _ = make_engine(obs_dim, act_dim, 5, cfg)
    dp = DataParallelSAC(eng_b)
    g = torch.Generator().manual_seed(3)
    for _ in range(3):
        obs = torch.randn(B, obs_dim, generator=g).cuda()
        act = (torch.rand(B, act_dim, generator=g) * 2 - 1).cuda()
        ret = (torch.randn(B, generator=g) * 2).cuda()
        noise = torch.randn(B, act_dim, generator=g).cuda()
        weight = torch.rand(B, generator=g).cuda() if weighted else None
        s_a, w_a = eng_a.update_with_batch(obs, act, ret, noise, weight)
        s_b, w_b = dp.update_with_batch(obs, act, ret, noise, weight)
        assert torch.equal(s_a, s_b) and torch.equal(w_a, w_b)
    for name in ("actor", "critic1", "critic2", "critic1_old", "critic2_old", "actor_m", "actor_v", "critic1_m",
                 "critic1_v", "critic2_m", "critic2_v", "log_alpha", "log_alpha_m", "log_alpha_v"):
        assert torch.equal(getattr(eng_a, name), getattr(eng_b, name)), name
    assert eng_a.adam_step == eng_b.adam_step == 3


@pytest.mark.parametrize("obs_dim,act_dim,B", [(376, 17, 4096), (23, 5, 300)])
def test_one_launch_slab_sums_and_adam_are_bit_identical(obs_dim, act_dim, B, monkeypatch):
    """slab_adam_kernel (weight-gradient slab sums + Adam + Polyak of both critics in one launch; the actor's + the alpha step
    in another) == the separate slab_sum_multi / adam / alpha launches (TS_SAC_SPLIT_ADAM=1), bit for bit, over three
    updates: parameters, moments, lagged critics, log_alpha, stats and PER weights."""
    cfg = OS.SACConfig(auto_alpha=True, log_alpha0=-0.3, target_entropy=-float(act_dim), tau=0.01)
    out = {}
    for split in ("", "1"):
        if split:
            monkeypatch.setenv("TS_SAC_SPLIT_ADAM", "1")
        else:
            monkeypatch.delenv("TS_SAC_SPLIT_ADAM", raising=False)
        eng, _ = make_engine(obs_dim, act_dim, 5, cfg)
        g = torch.Generator().manual_seed(3)
        stats = []
        for _ in range(3):
            obs = torch.randn(B, obs_dim, generator=g).cuda()
            act = (torch.rand(B, act_dim, generator=g) * 2 - 1).cuda()
            ret = (torch.randn(B, generator=g) * 2).cuda()
            noise = torch.randn(B, act_dim, generator=g).cuda()
            weight = torch.rand(B, generator=g).cuda()
            stats.append(eng.update_with_batch(obs, act, ret, noise, weight))
        out[split] = (eng, stats)
    (ea, sa), (eb, sb) = out[""], out["1"]
    for (s_a, w_a), (s_b, w_b) in zip(sa, sb):
        assert torch.equal(s_a, s_b) and torch.equal(w_a, w_b)
    for name in ("actor", "critic1", "critic2", "critic1_old", "critic2_old", "actor_m", "actor_v", "critic1_m",
                 "critic1_v", "critic2_m", "critic2_v", "log_alpha", "log_alpha_m", "log_alpha_v"):
        assert torch.equal(getattr(ea, name), getattr(eb, name)), name


@pytest.mark.parametrize("auto,weighted", [(True, True), (False, False)])
def test_row_indexed_entry_points_are_bit_identical(auto, weighted):
    """ts_sac_returns_rows (gather of obs_next inside the input packing + _target_q + the 1-step return in one launch
    sequence) == gather_rows + ts_sac_target_q + ts_nstep_return_fused, and ts_sac_update_rows == gather_rows + ts_sac_update:
    returns, statistics, PER weights, parameters and Adam moments bit for bit over three updates on a buffer with
    terminations and repeated indices."""
    from tianshou_amd.buffer import DeviceReplayBuffer

    obs_dim, act_dim, B, slots, E = 23, 5, 200, 4096, 4
    cfg = OS.SACConfig(auto_alpha=auto, log_alpha0=-0.2, alpha=0.15, target_entropy=-float(act_dim), actor_lr=3e-4,
                       critic_lr=1e-3, alpha_lr=1e-3, tau=0.02, n_step=1)
    g = torch.Generator().manual_seed(3)
    T = slots // E
    off = np.arange(E + 1, dtype=np.int64) * T
    buf = DeviceReplayBuffer(offset=off, last_index=off[:-1] + T - 1, lengths=np.full(E, T, np.int64),
                             insertion=np.zeros(E, np.int64), rew=torch.randn(slots, generator=g).double().numpy(),
                             terminated=(torch.rand(slots, generator=g) < 0.1).numpy(), truncated=np.zeros(slots, bool),
                             obs=torch.randn(slots, obs_dim, generator=g).numpy(), act=(torch.rand(slots, act_dim, generator=g) * 2 - 1).numpy(),
                             obs_next=torch.randn(slots, obs_dim, generator=g).numpy())
    out = {}
    import os
    for mode in ("rows", "gather"):
        if mode == "gather":
            os.environ["TS_SAC_NO_ROWS"] = "1"
        try:
            eng, _ = make_engine(obs_dim, act_dim, 11, cfg)
            gg = torch.Generator().manual_seed(5)
            rets = []
            for _ in range(3):
                idx = torch.randint(0, slots, (B,), generator=gg)
                n0, n1 = torch.randn(B, act_dim, generator=gg), torch.randn(B, act_dim, generator=gg)
                w = torch.rand(B, generator=gg) if weighted else None
                ret = eng.preprocess(buf, idx, n0)
                stats, w_out = eng.update_with_rows(buf, idx, ret, n1, w)
                rets.append((ret.cpu(), stats.cpu(), w_out.cpu()))
            torch.cuda.synchronize()
            out[mode] = (rets, [getattr(eng, k).cpu().clone() for k in ("actor", "critic1", "critic2", "critic1_old", "actor_m",
                                                                          "critic1_v", "log_alpha")])
        finally:
            os.environ.pop("TS_SAC_NO_ROWS", None)
    for (r0, s0, w0), (r1, s1, w1) in zip(out["rows"][0], out["gather"][0]):
        assert torch.equal(r0, r1) and torch.equal(s0, s1) and torch.equal(w0, w1)
    for a, b in zip(out["rows"][1], out["gather"][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("auto,weighted,hidden,activation,fill", [(True, True, 256, "relu", False), (False, False, 256, "relu", True),
                                                                  (True, False, ((64, 48, 32), (40, 56, 24)), "relu", True), (True, True, 256, "tanh", False)])
def test_one_call_update_is_bit_identical(auto, weighted, hidden, activation, fill):
    """ts_sac_learn_rows (one call: both packing passes + the noise draw in one launch, the target pass, the 1-step return
    formed inside the critic-loss launch, the update) == normal_noise + ts_sac_returns_rows + ts_sac_update_rows: noise, returns,
    statistics, PER weights, every parameter vector, lagged critic and Adam moment bit for bit over four updates on a buffer with
    terminations and repeated indices.  Net[256, 256] ReLU rides on the fused sequence; three hidden layers / a tanh trunk run the
    two sequences back to back inside the call."""
    from tianshou_amd.buffer import DeviceReplayBuffer, normal_noise

    obs_dim, act_dim, B, slots, E = 23, 5, 300, 4096, 4
    cfg = OS.SACConfig(auto_alpha=auto, log_alpha0=-0.2, alpha=0.15, target_entropy=-float(act_dim), actor_lr=3e-4,
                       critic_lr=1e-3, alpha_lr=1e-3, tau=0.02, n_step=1)
    g = torch.Generator().manual_seed(3)
    T = slots // E
    off = np.arange(E + 1, dtype=np.int64) * T
    buf = DeviceReplayBuffer(offset=off, last_index=off[:-1] + T - 1, lengths=np.full(E, T, np.int64),
                             insertion=np.zeros(E, np.int64), rew=torch.randn(slots, generator=g).double().numpy(),
                             terminated=(torch.rand(slots, generator=g) < 0.1).numpy(), truncated=np.zeros(slots, bool),
                             obs=torch.randn(slots, obs_dim, generator=g).numpy(), act=(torch.rand(slots, act_dim, generator=g) * 2 - 1).numpy(),
                             obs_next=torch.randn(slots, obs_dim, generator=g).numpy())
    names = ("actor", "critic1", "critic2", "critic1_old", "critic2_old", "actor_m", "actor_v", "critic1_m", "critic1_v", "critic2_m",
             "critic2_v", "log_alpha")
    out = {}
    for mode in ("one", "two"):
        eng, _ = make_engine(obs_dim, act_dim, 11, cfg, hidden=hidden, activation=activation)
        gg = torch.Generator().manual_seed(5)
        rec = []
        for u in range(4):
            idx = torch.randint(0, slots, (B,), generator=gg)
            noise = torch.randn(2, B, act_dim, generator=gg)
            w = torch.rand(B, generator=gg) if weighted else None
            if mode == "one":
                stats, w_out, ret, nz = eng.learn_rows(buf, idx, None if fill else noise, noise_key=(0x5AC, u) if fill else None, weight=w)
            else:
                nz = normal_noise((2, B, act_dim), 0x5AC, u) if fill else noise.cuda()
                ret = eng.preprocess(buf, idx, nz[0])
                stats, w_out = eng.update_with_rows(buf, idx, ret, nz[1], w)
            rec.append((nz.cpu(), ret.cpu(), stats.cpu(), w_out.cpu()))
        torch.cuda.synchronize()
        assert eng.adam_step == 4
        out[mode] = (rec, [getattr(eng, k).cpu().clone() for k in names])
    for u, (a, b) in enumerate(zip(out["one"][0], out["two"][0])):
        for what, x, y in zip(("noise", "returns", "stats", "weight"), a, b):
            assert torch.equal(x, y), (u, what)
    for k, a, b in zip(names, out["one"][1], out["two"][1]):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("n_step", [1, 3])
def test_returns_without_a_stored_obs_next_read_obs_of_the_next_slot(n_step):
    """save_obs_next=False (buffer_base.py:622-626): batch.obs_next = obs[next(index)].  A buffer without the obs_next column
    gives the same n-step returns as one whose obs_next column was filled with obs[next(.)] for every slot (next() from the
    oracle's restatement of _next_index, manager.py:340-363: an episode's last / newest transition maps to itself)."""
    from oracle import oracle as O
    from tianshou_amd.buffer import DeviceReplayBuffer

    obs_dim, act_dim, B, slots, E = 23, 5, 300, 2048, 4
    cfg = OS.SACConfig(auto_alpha=True, log_alpha0=-0.2, target_entropy=-float(act_dim), actor_lr=3e-4, critic_lr=1e-3,
                       alpha_lr=1e-3, tau=0.02, n_step=n_step)
    g = torch.Generator().manual_seed(7)
    T = slots // E
    off = np.arange(E + 1, dtype=np.int64) * T
    term = (torch.rand(slots, generator=g) < 0.05).numpy()
    trunc = (torch.rand(slots, generator=g) < 0.03).numpy() & ~term
    obs = torch.randn(slots, obs_dim, generator=g).numpy()
    last, lengths = off[:-1] + T - 1, np.full(E, T, np.int64)
    nxt = O._next_index(np.arange(slots), off, term | trunc, last, lengths)
    assert (nxt != np.arange(slots) + 1).any() and (nxt == np.arange(slots)).any()
    common = dict(offset=off, last_index=last, lengths=lengths, insertion=np.zeros(E, np.int64),
                  rew=torch.randn(slots, generator=g).double().numpy(), terminated=term, truncated=trunc, obs=obs,
                  act=(torch.rand(slots, act_dim, generator=g) * 2 - 1).numpy())
    with_col = DeviceReplayBuffer(obs_next=obs[nxt], **common)
    without = DeviceReplayBuffer(obs_next=None, **common)
    idx = torch.randint(0, slots, (B,), generator=g)
    assert torch.equal(without.obs_next_rows(idx).cpu(), torch.from_numpy(obs[nxt[idx.numpy()]]))
    noise = torch.randn(B, act_dim, generator=g)
    eng, _ = make_engine(obs_dim, act_dim, 11, cfg)
    ret_a = eng.preprocess(with_col, idx, noise)
    ret_b = eng.preprocess(without, idx, noise)
    torch.cuda.synchronize()
    # n_step = 1 with the stored column runs the row-indexed launch sequence, the other three the general path: same values
    np.testing.assert_allclose(ret_b.cpu().numpy(), ret_a.cpu().numpy(), rtol=1e-6, atol=1e-6)
