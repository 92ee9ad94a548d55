"""CPU: pins oracle/ (C restatement + torch-fp32 PPO restatement) to the reference.

Sources of truth, both committed under tests/golden/ by oracle/gen_golden.py:
  * literal known-answer vectors of the reference's own tests (test/base/test_returns.py,
    test/base/test_buffer.py), and
  * outputs of the unmodified reference executed in the authoring container.
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle import oracle_ppo as OP

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


# ------------------------------------------------------------------------------------------ GAE
def test_gae_known_answers():
    g = load("returns_kat.npz")
    for c in range(int(g["n_gae"])):
        gamma, lam = g[f"gae{c}_gamma_lambda"]
        v_next = g[f"gae{c}_v_next"] if bool(g[f"gae{c}_has_v"]) else None
        ret, adv = O.compute_episodic_return(
            g[f"gae{c}_rew"], g[f"gae{c}_terminated"], g[f"gae{c}_truncated"],
            g[f"gae{c}_indices"], g[f"gae{c}_unfinished"], v_next, None, gamma, lam)
        # literal vectors are given to 4-5 significant digits in the reference test (np.allclose)
        assert np.allclose(ret, g[f"gae{c}_literal"]), c
        # the reference itself, run here: same f64 arithmetic -> essentially exact
        np.testing.assert_allclose(ret, g[f"gae{c}_ref_returns"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(adv, g[f"gae{c}_ref_adv"], rtol=1e-13, atol=1e-13)


def test_mc_return_to_go_known_answers():
    # test/base/test_policy.py:26-30
    assert np.all(O.episode_mc_return_to_go([1, 1, 1], 0.9) == np.array([0.9**2 + 0.9 + 1, 0.9 + 1, 1]))
    assert O.episode_mc_return_to_go([1, 2, 3], 0.5)[0] == 1 + 0.5 * (2 + 0.5 * 3)


# --------------------------------------------------------------------------------------- n-step
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 10])
def test_nstep_known_answers(variant, n):
    g = load("returns_kat.npz")
    pre = f"nstep{variant}_"
    st = O.BufferState(g[pre + "offset"], g[pre + "last_index"], g[pre + "lengths"],
                       g[pre + "insertion"], g[pre + "rew"], g[pre + "terminated"],
                       g[pre + "truncated"], g[pre + "done"])
    indices = g[pre + "indices"]
    assert np.array_equal(st.sample_indices_all(), indices)

    def target_q(after):  # test_returns.py:162-165: -rew[next(idx)]
        return -st.rew[st.next(after)].astype(np.float32)

    ret, _ = O.compute_nstep_return(st, indices, target_q, gamma=0.1, n_step=n)
    assert np.allclose(ret.reshape(-1), g[pre + f"n{n}_literal"])
    # reference output is float32 (to_torch_as, algorithm_base.py:811): bit-exact after the cast
    assert np.array_equal(ret.astype(np.float32).reshape(-1), g[pre + f"n{n}_ref"].reshape(-1))

    def target_q_multi(after):  # :167-168
        return np.repeat(target_q(after)[:, None], 51, axis=1)

    retm, _ = O.compute_nstep_return(st, indices, target_q_multi, gamma=0.1, n_step=n)
    assert np.array_equal(retm.astype(np.float32), g[pre + f"n{n}_ref_multidim"])


# ---------------------------------------------------------------------------------- index math
def test_buffer_index_math_bit_exact():
    g = load("buffer_index.npz")
    for s in range(int(g["n_scen"])):
        t = f"s{s}_"
        args = (g[t + "offset"], g[t + "done"], g[t + "last_index"], g[t + "lengths"])
        q = g[t + "query"]
        assert np.array_equal(O._next_index(q, *args), g[t + "next"]), s
        assert np.array_equal(O._prev_index(q, *args), g[t + "prev"]), s
        assert np.array_equal(O.unfinished_index(*args), g[t + "unfinished"]), s
        assert np.array_equal(
            O.sample_indices_all(g[t + "offset"], g[t + "lengths"], g[t + "insertion"]),
            g[t + "sample0"]), s
    # literal vectors of test/base/test_buffer.py:822-961
    for lit in ("litA", "litB"):
        t = f"s{int(g[lit + '_scen'])}_"
        args = (g[t + "offset"], g[t + "done"], g[t + "last_index"], g[t + "lengths"])
        idx = np.arange(20)
        assert np.array_equal(O._next_index(idx, *args), g[lit + "_next"])
        assert np.array_equal(O._prev_index(idx, *args), g[lit + "_prev"])
        assert np.array_equal(O.unfinished_index(*args), g[lit + "_unfinished"])
    assert np.array_equal(g[f"s{int(g['litA_scen'])}_done"], g["litA_done"])


# ------------------------------------------------------------------------------------ sum tree
def test_segtree_against_reference():
    g = load("segtree_per.npz")
    for c in range(int(g["n_tree"])):
        size, bound = g[f"t{c}_size_bound"]
        tree = None
        for r in range(3):
            tree = g[f"t{c}_r{r}_tree_before"].copy()
            O._setitem(tree, g[f"t{c}_r{r}_idx"] + bound, g[f"t{c}_r{r}_val"])
            assert np.array_equal(tree, g[f"t{c}_r{r}_tree_after"]), (c, r)
        q = g[f"t{c}_query"].copy()
        assert np.array_equal(O._get_prefix_sum_idx(q, int(bound), tree), g[f"t{c}_prefix_idx"])
        for (lo, hi), ref in zip(g[f"t{c}_range"].T, g[f"t{c}_range_sum"]):
            got = tree[1] if (lo == 0 and hi == size and False) else O._reduce(
                tree, int(lo) + int(bound) - 1, int(hi) + int(bound))
            assert got == ref


def test_segtree_naive_properties():
    # mirrors test/base/test_buffer.py:553-633: random updates vs naive sums
    rng = np.random.default_rng(0)
    size, bound = 100, 128
    tree = np.zeros(2 * bound)
    naive = np.zeros(size)
    for _ in range(200):
        k = rng.integers(1, 10)
        idx = rng.integers(0, size, k)
        val = rng.random(k)
        O._setitem(tree, idx + bound, val)
        naive[idx] = val
        lo = int(rng.integers(0, size))
        hi = int(rng.integers(lo + 1, size + 1))
        assert np.isclose(O._reduce(tree, lo + bound - 1, hi + bound), naive[lo:hi].sum())
    q = rng.random(64) * tree[1]
    idx = O._get_prefix_sum_idx(q.copy(), bound, tree)
    cs = np.cumsum(naive)
    for v, i in zip(q, idx):
        assert cs[i] >= v - 1e-9 and (i == 0 or cs[i - 1] < v + 1e-9)


def test_per_weights_against_reference():
    g = load("segtree_per.npz")
    bound = int(g["per_bound"])
    alpha, beta = g["per_alpha_beta"]
    tree = g["per_tree0"].copy()
    mx, mn = g["per_prio_before"]
    mx, mn = O.per_update_weight(tree, bound, g["per_upd_idx"], g["per_upd_td"], alpha, mx, mn)
    # float32 pow: libm powf vs NumPy's SIMD float32 power may differ by an ulp
    np.testing.assert_allclose(tree, g["per_tree1"], rtol=3e-7)
    np.testing.assert_allclose([mx, mn], g["per_prio_after"], rtol=1e-7)
    tree = g["per_tree1"].copy()
    scalar = g["per_uniform"] * tree[1]               # prio.py:65
    sidx = O._get_prefix_sum_idx(scalar.copy(), bound, tree)
    assert np.array_equal(sidx, g["per_sample_idx"])
    w = O.per_get_weight(tree, bound, sidx, g["per_prio_after"][1], beta, True)
    np.testing.assert_allclose(w, g["per_is_weight"], rtol=1e-12)


def test_random_sample_indices_restatement_matches_reference():
    """manager.py:216-234 with batch_size > 0: the restatement replays the reference's own draws (recovered from copies
    of its RandomStates by gen_golden.gen_sample_random) on uneven / wrapped buffers, two consecutive calls each."""
    g = load("sample_random.npz")
    for c in range(int(g["n_cases"][0])):
        for r in range(int(g["n_cases"][1])):
            k = f"c{c}_r{r}_"
            out = O.sample_indices_random(g[k + "offset"], g[k + "lengths"], g[k + "u"], g[k + "within"])
            assert out.dtype == np.int64 and np.array_equal(out, g[k + "result"]), k


def test_stacked_sample_indices_restatement_matches_reference():
    """manager.py:205-216 / buffer_base.py:532-545 (`stack_num > 1 and sample_avail`): available indices (batch_size 0) and
    the reference's `RandomState.choice(all_indices, bs)` result through the positions it drew."""
    g = load("sample_stack.npz")
    for c in range(int(g["n_cases"][0])):
        k = f"c{c}_"
        B = int(g[k + "offset"][-1])
        st = O.BufferState(g[k + "offset"], g[k + "last_index"], g[k + "lengths"], g[k + "insertion"], np.zeros(B),
                           g[k + "done"], np.zeros(B, np.uint8), done=g[k + "done"])
        stack = int(g[k + "stack"][0])
        assert np.array_equal(O.sample_indices_stack(st, g[k + "insertion"], stack), g[k + "all"]), k
        assert np.array_equal(O.sample_indices_stack(st, g[k + "insertion"], stack, g[k + "positions"]), g[k + "result"]), k


# ------------------------------------------------------------------------------------ PPO path
def _cfg_from(g):
    c = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    return OP.PPOConfig(
        gamma=c["gamma"], gae_lambda=c["gae_lambda"], eps_clip=c["eps_clip"],
        dual_clip=(c["dual_clip"] or None), value_clip=bool(c["value_clip"]),
        advantage_normalization=bool(c["advantage_normalization"]),
        recompute_advantage=bool(c["recompute_advantage"]), vf_coef=c["vf_coef"],
        ent_coef=c["ent_coef"], max_grad_norm=(c["max_grad_norm"] or None),
        return_scaling=bool(c["return_scaling"]), lr=c["lr"],
        max_batchsize=int(c["max_batchsize"]), algo="a2c" if c.get("is_a2c") else "ppo",
        # round 6 (absent from the older fixtures: Adam without weight decay on an unbounded actor)
        optimizer="rmsprop" if c.get("opt_rmsprop") else "adam", weight_decay=float(c.get("weight_decay", 0.0)),
        adam_eps=float(c.get("opt_eps", 1e-8)), rms_alpha=float(c.get("rms_alpha", 0.99)),
        rms_momentum=float(c.get("rms_momentum", 0.0)), rms_centered=bool(c.get("rms_centered", 0.0)),
        max_action=(float(c["max_action"]) if c.get("max_action") else None))


# round 6: the reference's default (bounded) Gaussian actor, RMSprop (mujoco_a2c.py:117), Adam with weight decay
R6_TAGS = ["bounded", "a2c_rmsprop", "adam_wd", "rms_momentum", "rms_centered"]


@pytest.mark.parametrize("tag", ["mujoco", "defaults", "a2c", "sched"] + R6_TAGS)
def test_ppo_restatement_matches_reference(tag):
    torch.set_num_threads(4)
    g = load(f"ppo_{tag}.npz")
    E, T, obs_dim, act_dim, batch_size, repeat, n_updates = [int(x) for x in g["dims"]]
    cfg = _cfg_from(g)
    state = OP.PPOState(params=OP.unflatten_params(torch.from_numpy(g["flat_params0"]), obs_dim, act_dim))
    for u in range(n_updates):
        pre_ = "" if u == 0 else f"u{u}_"
        if f"u{u}_lr" in g.files:        # "sched": LRSchedulerFactoryLinear stepped after every update()
            cfg.lr = float(g[f"u{u}_lr"])
            if tag == "sched":
                assert cfg.lr == pytest.approx(3e-4 * (1 - u / 6), rel=1e-12)
        obs = torch.from_numpy(g[pre_ + "obs"])
        obs_next = torch.from_numpy(g[pre_ + "obs_next"])
        act = torch.from_numpy(g[pre_ + "act"])
        rew, term, trunc = g[pre_ + "rew"], g[pre_ + "terminated"], g[pre_ + "truncated"]
        bs = O.BufferState(g["buf_offset"], g["buf_last_index"], g["buf_lengths"],
                           g["buf_insertion"], rew, term, trunc)
        indices = bs.sample_indices_all()
        unfinished = bs.unfinished_index()
        if u == 0:
            assert np.array_equal(indices, g["pre_indices"])
            assert np.array_equal(unfinished, g["pre_unfinished"])
        args = (obs[indices], obs_next[indices], act[indices], rew[indices], term[indices],
                trunc[indices], indices, unfinished)
        pre = OP.preprocess(state, cfg, *args)
        if u == 0:
            np.testing.assert_allclose(pre["v_s"].numpy(), g["pre_v_s"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(pre["returns"].numpy(), g["pre_returns"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(pre["adv"].numpy(), g["pre_adv"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(pre["logp_old"].numpy(), g["pre_logp_old"], rtol=1e-5, atol=1e-6)

        def recompute():
            return OP.add_returns_and_advantages(state, cfg, args[0], args[1], *args[3:])

        losses = OP.update(state, cfg, {"obs": args[0], "act": args[2]}, pre, batch_size, repeat,
                           list(g[f"u{u}_perms"]), recompute=recompute)
        ref = g[f"u{u}_losses"]
        assert losses.shape == ref.shape and losses.shape[0] == int(g[f"u{u}_gradient_steps"])
        np.testing.assert_allclose(losses, ref, rtol=1e-5, atol=2e-6)
        flat = OP.flatten_params(state.params).numpy()
        np.testing.assert_allclose(flat, g[f"u{u}_flat_params"], rtol=1e-4, atol=2e-6)
        m = torch.cat([state.adam_m[k].reshape(-1) for k in OP.PARAM_ORDER]).numpy()
        v = torch.cat([state.adam_v[k].reshape(-1) for k in OP.PARAM_ORDER]).numpy()
        # (RMSprop's momentum buffer holds g / sqrt(E[g^2]) sums of O(1..10): the floor follows the vector's scale)
        np.testing.assert_allclose(m, g[f"u{u}_adam_m"], rtol=1e-3, atol=max(1e-7, 2e-7 * float(np.abs(g[f"u{u}_adam_m"]).max())))
        np.testing.assert_allclose(v, g[f"u{u}_adam_v"], rtol=1e-3, atol=1e-10)
        np.testing.assert_allclose(
            [state.ret_rms.mean, state.ret_rms.var, state.ret_rms.count], g[f"u{u}_ret_rms"],
            rtol=1e-6)


# ------------------------------------------------------------------------------------ DQN path
@pytest.mark.parametrize("tag", ["atari", "small"])
def test_dqn_restatement_matches_reference(tag):
    """oracle_dqn (frame stack, n-step double-Q target, Huber / MSE, Adam, hard sync, PER weights)
    against the unmodified reference DQN.update() (gen_golden.gen_dqn)."""
    from oracle import oracle_dqn as OD
    from tests import dqn_common as DC

    g, d, cfg, bstate = DC.load(tag)
    stack_num = d["c"] if d["stack"] else 1
    frames = g["frames"]
    st = OD.DQNState.create(OD.init_params(d["c"], d["h"], d["w"], d["n_act"], d["seed"]), cfg)
    if d["per"]:
        tree = g["tree0"].copy()
        bound = 1
        while bound < d["E"] * d["slots"]:
            bound *= 2
        np.random.seed(d["seed"] + 7)
        mx, mn = 1.0, 1.0                                         # prio.py:42-43
    for u in range(d["n_updates"]):
        idx = g[f"u{u}_indices"]
        if d["per"]:
            scalar = np.random.rand(d["batch"]) * tree[1]       # prio.py:65
            assert np.array_equal(O._get_prefix_sum_idx(scalar, bound, tree), idx)
            w = O.per_get_weight(tree, bound, idx, mn, 0.4, True)
            np.testing.assert_allclose(w, g[f"u{u}_is_weight"], rtol=1e-4)      # tree carries the td rounding
        obs = OD.stacked_frames(bstate, frames, idx, stack_num)
        if u == 0:
            assert np.array_equal(obs[:2], g["u0_obs_sample"])
        ret = OD.preprocess(st, cfg, bstate, frames, idx, stack_num)
        np.testing.assert_allclose(ret, g[f"u{u}_returns"], rtol=1e-6, atol=1e-6)
        loss, td = OD.update_with_batch(st, cfg, obs, g["act"][idx], ret)
        np.testing.assert_allclose(td.numpy(), g[f"u{u}_td"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(loss, float(g[f"u{u}_loss"]), rtol=1e-6)
        flat = DC.torch_order_flat(st.params)
        np.testing.assert_allclose(flat[::61], g[f"u{u}_params_strided"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(st.params["conv1.w"].numpy(), g[f"u{u}_conv1_w"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(st.params["fc2.w"].numpy(), g[f"u{u}_fc2_w"], rtol=1e-6, atol=1e-7)
        if d["per"]:
            mx, mn = O.per_update_weight(tree, bound, idx, td.numpy(), 0.6, mx, mn)
            # priorities are (|td| + eps)^alpha of a float32 TD error that itself carries ~1e-6 rounding
            # (a small |td| has a large relative rounding error; torch's CPU conv also changes its
            # summation order with the thread count)
            np.testing.assert_allclose(tree, g[f"u{u}_tree"], rtol=1e-4)
            np.testing.assert_allclose([mn, mx], g[f"u{u}_prio_minmax"], rtol=1e-4)


# ------------------------------------------------------------------------------------ QRDQN / C51 paths
@pytest.mark.parametrize("kind", ["qr", "c51"])
def test_distq_restatement_matches_reference(kind):
    """oracle_distq (quantile Huber / categorical projection + cross entropy, n-step returns of whole
    distributions, PER weights and new priorities, Adam, hard sync) against the unmodified reference
    QRDQN.update() / C51.update() (gen_golden.gen_distq)."""
    from oracle import oracle_distq as OQ
    from oracle import oracle_dqn as OD
    from tests import dqn_common as DC

    g, d, cfg, bstate = DC.load_distq(kind)
    st = OD.DQNState.create(OQ.init_params(d["c"], d["h"], d["w"], d["n_act"], d["n_atoms"], d["seed"]), cfg.dqn())
    tree = g["tree0"].copy()
    bound = 1
    while bound < d["E"] * d["slots"]:
        bound *= 2
    np.random.seed(d["seed"] + 7)
    mx, mn = 1.0, 1.0
    for u in range(d["n_updates"]):
        idx = g[f"u{u}_indices"]
        scalar = np.random.rand(d["batch"]) * tree[1]
        assert np.array_equal(O._get_prefix_sum_idx(scalar, bound, tree), idx)
        w = O.per_get_weight(tree, bound, idx, mn, 0.4, True)
        np.testing.assert_allclose(w, g[f"u{u}_is_weight"], rtol=1e-4)
        ret = OQ.preprocess(st, cfg, bstate, g["frames"], idx, d["n_act"], 1, g["frames_next"])
        assert ret.shape == (d["batch"], d["n_atoms"])
        np.testing.assert_allclose(ret, g[f"u{u}_returns"], rtol=1e-6, atol=1e-6)
        loss, prio = OQ.update_with_batch(st, cfg, g["frames"][idx], g["act"][idx], ret, d["n_act"], weight=w,
                                          obs_next=g["frames_next"][idx])
        np.testing.assert_allclose(prio.numpy(), g[f"u{u}_prio"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(loss, float(g[f"u{u}_loss"]), rtol=1e-5)
        flat = DC.torch_order_flat(st.params)
        np.testing.assert_allclose(flat[::61], g[f"u{u}_params_strided"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(st.params["conv1.w"].numpy(), g[f"u{u}_conv1_w"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(st.params["fc2.w"].numpy().reshape(-1)[::7], g[f"u{u}_fc2_w_strided"], rtol=1e-6,
                                   atol=1e-7)
        mx, mn = O.per_update_weight(tree, bound, idx, prio.numpy(), 0.6, mx, mn)
        np.testing.assert_allclose(tree, g[f"u{u}_tree"], rtol=1e-4)


# ------------------------------------------------------------------------------------ Rainbow path
@pytest.mark.parametrize("tag", ["lagged", "single"])
def test_rainbow_restatement_matches_reference(tag):
    """oracle_rainbow (NoisyLinear layers with the recorded noise of both networks, dueling heads, C51 projection and
    cross entropy, hard sync that carries the noise along) against the unmodified reference RainbowDQN.update()."""
    from oracle import oracle_rainbow as ORB
    from tests import dqn_common as DC

    g, d, cfg, bstate = DC.load_rainbow(tag)
    p0, n0 = ORB.init_params(d["c"], d["h"], d["w"], d["n_act"], d["n_atoms"], d["seed"])
    st = ORB.RainbowState(p0, n0, cfg)
    for u in range(d["n_updates"]):
        idx = g[f"u{u}_indices"]
        ret = ORB.preprocess(cfg, bstate, idx)
        np.testing.assert_allclose(ret, g[f"u{u}_returns"], rtol=1e-6, atol=1e-6)
        loss, prio = ORB.update_with_batch(st, cfg, g["frames"][idx], g["act"][idx], ret, g["frames_next"][idx], d["n_act"],
                                           DC.rainbow_noise(g, u), DC.rainbow_noise(g, u, old=True),
                                           weight=g[f"u{u}_is_weight"], old_training=bool(g[f"u{u}_old_training"]))
        np.testing.assert_allclose(prio.numpy(), g[f"u{u}_prio"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(loss, float(g[f"u{u}_loss"]), rtol=1e-5)
        flat = ORB.flatten_params(st.dqn.params).numpy()
        np.testing.assert_allclose(flat[::97], g[f"u{u}_params_strided"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(st.dqn.params["conv1.w"].numpy(), g[f"u{u}_conv1_w"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(st.dqn.params["V2.sigma_W"].numpy(), g[f"u{u}_V2_sigma_W"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(st.dqn.params["Q2.mu_b"].numpy(), g[f"u{u}_Q2_mu_b"], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------ SAC path
def _fixture_hidden(g):
    """`hidden` of a SAC / TD3 fixture: 256 (the early fixtures), four widths (actor h1, h2, critic h1, h2), or -- other depths,
    round 6 -- the nested pair (actor sizes, critic sizes) that oracle_sac.layer_sizes takes."""
    if "hidden_actor" in g:
        return (tuple(int(x) for x in g["hidden_actor"]), tuple(int(x) for x in g["hidden_critic"]))
    return tuple(int(x) for x in g["hidden"]) if "hidden" in g else 256


def load_sac(tag):
    from oracle import oracle_sac as OS

    g = load(f"sac_{tag}.npz")
    E, slots, steps, obs_dim, act_dim, batch, n_updates, seed, auto, n_step = (int(x) for x in g["dims"])
    c = dict(zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist()))
    cfg = OS.SACConfig(gamma=c["gamma"], tau=c["tau"], n_step=int(c["n_step"]), alpha=c["alpha"],
                       auto_alpha=bool(c["auto_alpha"]), target_entropy=c["target_entropy"],
                       log_alpha0=c["log_alpha0"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"],
                       alpha_lr=c["alpha_lr"], max_action=c.get("max_action", 0.0))      # (> 0: the bounded class-default actor)
    d = dict(E=E, slots=slots, obs_dim=obs_dim, act_dim=act_dim, batch=batch, n_updates=n_updates, seed=seed,
             hidden=_fixture_hidden(g), activation="tanh" if c.get("tanh_trunks") else "relu")
    bstate = O.BufferState(g["buf_offset"], g["buf_last_index"], g["buf_lengths"], g["buf_insertion"],
                           g["rew"], g["terminated"], g["truncated"])
    return g, d, cfg, bstate


@pytest.mark.parametrize("tag", ["auto", "fixed", "widths", "depth3", "depth1", "bounded", "bounded_depth3", "tanh"])
def test_sac_restatement_matches_reference(tag):
    """oracle_sac (tanh-Gaussian policy, twin lagged critics, n-step target, three Adam steps, auto alpha,
    Polyak) against the unmodified reference SAC.update() with its rsample() noise replayed."""
    from oracle import oracle_sac as OS

    g, d, cfg, bstate = load_sac(tag)
    actor, c1, c2 = OS.init_sac_params(d["obs_dim"], d["act_dim"], d["seed"], d["hidden"])
    st = OS.SACState.create(actor, c1, c2, cfg)
    obs_all, obs_next_all = torch.as_tensor(g["obs"]), torch.as_tensor(g["obs_next"])
    with OS.activation(d["activation"]):          # (`tanh`: Net(activation=nn.Tanh) trunks)
        _sac_restatement_updates(g, d, cfg, bstate, st, obs_all, obs_next_all)


def _sac_restatement_updates(g, d, cfg, bstate, st, obs_all, obs_next_all):
    from oracle import oracle_sac as OS

    for u in range(d["n_updates"]):
        idx = g[f"u{u}_indices"]

        def tq_fn(after):
            return OS.target_q(st, cfg, obs_next_all[after], torch.as_tensor(g[f"u{u}_noise_target"])).numpy()

        ret, _ = O.compute_nstep_return(bstate, idx, tq_fn, cfg.gamma, cfg.n_step)
        ret = ret.astype(np.float32).reshape(-1)
        np.testing.assert_allclose(ret, g[f"u{u}_returns"], rtol=1e-5, atol=1e-5)
        out = OS.update_with_batch(st, cfg, obs_all[idx], g["act"][idx], ret, g[f"u{u}_noise_actor"])
        ref = g[f"u{u}_stats"]
        np.testing.assert_allclose([out["actor_loss"], out["critic1_loss"], out["critic2_loss"]], ref[:3], rtol=1e-5)
        np.testing.assert_allclose(out["alpha"], ref[3], rtol=1e-6)
        if cfg.auto_alpha:
            np.testing.assert_allclose(out["alpha_loss"], ref[4], rtol=1e-5, atol=1e-6)
        for name in ("actor", "critic1", "critic2", "critic1_old", "critic2_old"):
            flat = OS.flatten(getattr(st, name), OS.order_of(getattr(st, name))).numpy()
            np.testing.assert_allclose(flat[::61], g[f"u{u}_{name}"], rtol=1e-5, atol=1e-6, err_msg=name)


# ------------------------------------------------------------------------------------ PPO on the Atari actor-critic
def load_ppo_cnn():
    from oracle import oracle_ppo as OPm

    g = load("ppo_cnn.npz")
    E, T, c, h, w, n_act, batch_size, repeat, seed = (int(x) for x in g["dims"])
    cv = dict(zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist()))
    cfg = OPm.PPOConfig(gamma=cv["gamma"], gae_lambda=cv["gae_lambda"], eps_clip=cv["eps_clip"],
                        dual_clip=cv["dual_clip"] or None, value_clip=bool(cv["value_clip"]),
                        advantage_normalization=bool(cv["advantage_normalization"]), vf_coef=cv["vf_coef"],
                        ent_coef=cv["ent_coef"], max_grad_norm=cv["max_grad_norm"] or None,
                        return_scaling=bool(cv["return_scaling"]), lr=cv["lr"], adam_eps=cv["adam_eps"],
                        max_batchsize=int(cv["max_batchsize"]))
    return g, dict(E=E, T=T, c=c, h=h, w=w, n_act=n_act, batch_size=batch_size, repeat=repeat, seed=seed), cfg


def test_ppo_cnn_restatement_matches_reference():
    """oracle_ppo_cnn (shared DQNet trunk, Categorical policy, GAE, clipped surrogate + clipped value + entropy,
    clip_grad_norm_ + Adam) against the unmodified reference PPO.update()."""
    from oracle import oracle_ppo_cnn as OC

    g, d, cfg = load_ppo_cnn()
    st = OP.PPOState(params=OC.init_params(d["c"], d["h"], d["w"], d["n_act"], d["seed"]))
    idx, unf = g["pre_indices"], g["pre_unfinished"]
    assert np.array_equal(idx, np.arange(d["E"] * d["T"]))
    pre = OC.preprocess(st, cfg, g["obs"], g["obs_next"], g["act"], g["rew"], g["terminated"], g["truncated"], idx, unf)
    for k in ("v_s", "returns", "adv", "logp_old"):
        np.testing.assert_allclose(pre[k].numpy(), g["pre_" + k], rtol=1e-5, atol=1e-5, err_msg=k)
    losses = OC.update(st, cfg, g["obs"], g["act"], pre, d["batch_size"], d["repeat"], g["perms"])
    assert len(losses) == int(g["gradient_steps"])
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(OC.flatten_params(st.params).numpy()[::17], g["params_strided"], rtol=1e-5,
                               atol=0.02 * cfg.lr)


def load_ppo_discrete(tag):
    from oracle import oracle_ppo as OPm

    g = load(f"ppo_discrete_{tag}.npz")
    E, T, obs_dim, hidden, n_act, batch_size, repeat, seed, softmax = (int(x) for x in g["dims"])
    cv = dict(zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist()))
    cfg = OPm.PPOConfig(gamma=cv["gamma"], gae_lambda=cv["gae_lambda"], eps_clip=cv["eps_clip"],
                        dual_clip=cv["dual_clip"] or None, value_clip=bool(cv["value_clip"]),
                        advantage_normalization=bool(cv["advantage_normalization"]), vf_coef=cv["vf_coef"],
                        ent_coef=cv["ent_coef"], max_grad_norm=cv["max_grad_norm"] or None,
                        return_scaling=bool(cv["return_scaling"]), lr=cv["lr"], adam_eps=cv["adam_eps"],
                        max_batchsize=int(cv["max_batchsize"]), algo="a2c" if cv.get("a2c") else "ppo")
    d = dict(E=E, T=T, obs_dim=obs_dim, hidden=hidden, n_act=n_act, batch_size=batch_size, repeat=repeat, seed=seed,
             softmax=bool(softmax))
    return g, d, cfg


@pytest.mark.parametrize("tag", ["c1", "opts", "a2c"])
def test_ppo_discrete_restatement_matches_reference(tag):
    """BASELINE.json configs[0] (CartPole shape: obs 4, MLP[64, 64] shared by a softmax actor and a critic, batch 64)
    and an every-option variant: oracle_ppo_discrete against the unmodified reference PPO.update()."""
    from oracle import oracle_ppo_cnn as OC
    from oracle import oracle_ppo_discrete as OD

    g, d, cfg = load_ppo_discrete(tag)
    net = OD.MlpNet(softmax_output=d["softmax"])
    st = OP.PPOState(params=OD.init_params(d["obs_dim"], d["hidden"], d["n_act"], d["seed"]))
    idx, unf = g["pre_indices"], g["pre_unfinished"]
    assert np.array_equal(idx, np.arange(d["E"] * d["T"]))
    pre = OC.preprocess(st, cfg, g["obs"], g["obs_next"], g["act"], g["rew"], g["terminated"], g["truncated"], idx, unf,
                        net=net)
    for k in ("v_s", "returns", "adv") + (("logp_old",) if cfg.algo == "ppo" else ()):
        np.testing.assert_allclose(pre[k].numpy(), g["pre_" + k], rtol=1e-5, atol=1e-5, err_msg=k)
    losses = OC.update(st, cfg, g["obs"], g["act"], pre, d["batch_size"], d["repeat"], g["perms"], net=net)
    assert len(losses) == int(g["gradient_steps"])
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(OD.flatten_params(st.params).numpy(), g["params"], rtol=1e-5, atol=0.02 * cfg.lr)
    np.testing.assert_allclose([st.ret_rms.mean, st.ret_rms.var, st.ret_rms.count], g["ret_rms"], rtol=1e-6)


# ------------------------------------------------------------------------------------ write side
def replay_buffer_add(g, s, make_writer, add):
    """Replays scenario s of buffer_add.npz through `add(writer, rows, ids)`; returns the writer."""
    total, E, steps, obs_dim = (int(x) for x in g[f"s{s}_dims"])
    T = total // E
    wr = make_writer(np.arange(E + 1) * T, obs_dim)
    pos = 0
    for t, k in enumerate(g[f"s{s}_counts"]):
        k = int(k)
        sl = slice(pos, pos + k)
        ret = add(wr, {key: g[f"s{s}_in_{key}"][sl] for key in ("rew", "term", "trunc", "obs", "act", "obs_next")},
                  g[f"s{s}_ids"][sl])
        got = np.stack([np.asarray(x, np.float64) for x in ret], axis=1)
        assert np.array_equal(got, g[f"s{s}_returned"][sl]), (s, t)          # bit-exact, incl. the f64 returns
        pos += k
    return wr


def test_buffer_add_restatement_matches_reference():
    g = load("buffer_add.npz")
    for s in range(int(g["n_scen"])):
        wr = replay_buffer_add(g, s, lambda off, d: O.BufferWriter(off),
                               lambda w, rows, ids: w.add(rows["rew"], rows["term"], rows["trunc"], ids))
        assert np.array_equal(wr.last_index, g[f"s{s}_final_last_index"])
        assert np.array_equal(wr.lengths, g[f"s{s}_final_lengths"])
        assert np.array_equal(wr.insertion, g[f"s{s}_final_insertion"])
        assert np.array_equal(wr.rew, g[f"s{s}_final_rew"])
        assert np.array_equal(wr.done.astype(bool), g[f"s{s}_final_done"])
        assert np.array_equal(wr.terminated.astype(bool), g[f"s{s}_final_terminated"])


# ------------------------------------------------------------------------------------ TD3 / DDPG
def load_td3(tag):
    from oracle import oracle_sac as OS

    g = load(f"td3_{tag}.npz")
    E, slots, steps, obs_dim, act_dim, batch, n_updates, seed, twin, n_step = (int(x) for x in g["dims"])
    c = dict(zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist()))
    cfg = OS.TD3Config(gamma=c["gamma"], tau=c["tau"], n_step=int(c["n_step"]), twin=bool(c["twin"]),
                       policy_noise=c["policy_noise"], noise_clip=c["noise_clip"],
                       update_actor_freq=int(c["update_actor_freq"]), max_action=c["max_action"],
                       actor_lr=c["actor_lr"], critic_lr=c["critic_lr"])
    d = dict(obs_dim=obs_dim, act_dim=act_dim, batch=batch, n_updates=n_updates, seed=seed, twin=bool(twin),
             hidden=_fixture_hidden(g), activation="tanh" if c.get("tanh_trunks") else "relu")
    bstate = O.BufferState(g["buf_offset"], g["buf_last_index"], g["buf_lengths"], g["buf_insertion"],
                           g["rew"], g["terminated"], g["truncated"])
    return g, d, cfg, bstate


@pytest.mark.parametrize("tag", ["twin", "ddpg", "widths", "ddpg_widths", "depth4", "ddpg_depth1", "tanh3"])
def test_td3_ddpg_restatement_matches_reference(tag):
    from oracle import oracle_sac as OS

    g, d, cfg, bstate = load_td3(tag)
    st = OS.TD3State.create(*OS.init_td3_params(d["obs_dim"], d["act_dim"], d["seed"], d["twin"], d["hidden"]), cfg)
    obs_all, obs_next_all = torch.as_tensor(g["obs"]), torch.as_tensor(g["obs_next"])
    with OS.activation(d["activation"]):
        _td3_restatement_updates(g, d, cfg, bstate, st, obs_all, obs_next_all)


def _td3_restatement_updates(g, d, cfg, bstate, st, obs_all, obs_next_all):
    from oracle import oracle_sac as OS

    for u in range(d["n_updates"]):
        idx = g[f"u{u}_indices"]
        noise = g[f"u{u}_noise"] if d["twin"] else None
        ret, _ = O.compute_nstep_return(bstate, idx, lambda after: OS.td3_target_q(st, cfg, obs_next_all[after], noise).numpy(),
                                        cfg.gamma, cfg.n_step)
        ret = ret.astype(np.float32).reshape(-1)
        np.testing.assert_allclose(ret, g[f"u{u}_returns"], rtol=1e-5, atol=1e-5)
        out = OS.td3_update_with_batch(st, cfg, obs_all[idx], g["act"][idx], ret)
        ref = g[f"u{u}_stats"]
        got = [out["actor_loss"], out["critic1_loss"]] + ([out["critic2_loss"]] if d["twin"] else [])
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-7)
        names = ["actor", "critic1", "actor_old", "critic1_old"] + (["critic2", "critic2_old"] if d["twin"] else [])
        for name in names:
            order = OS.order_of(getattr(st, name))
            np.testing.assert_allclose(OS.flatten(getattr(st, name), order).numpy()[::61], g[f"u{u}_{name}"], rtol=1e-5,
                                       atol=1e-6, err_msg=name)


# ------------------------------------------------------------------------------------ DiscreteSAC path
def load_dsac(tag):
    from oracle import oracle_sac as OS

    g = load(f"dsac_{tag}.npz")
    E, slots, steps, obs_dim, n_act, hidden, batch, n_updates, seed, auto, n_step = (int(x) for x in g["dims"])
    c = dict(zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist()))
    cfg = OS.SACConfig(gamma=c["gamma"], tau=c["tau"], n_step=int(c["n_step"]), alpha=c["alpha"],
                       auto_alpha=bool(c["auto_alpha"]), target_entropy=c["target_entropy"],
                       log_alpha0=c["log_alpha0"], actor_lr=c["actor_lr"], critic_lr=c["critic_lr"],
                       alpha_lr=c["alpha_lr"])
    if "hidden" in g or "hidden_actor" in g:   # (actor h1, actor h2, critic h1, critic h2): unequal widths; other depths: a nested pair
        hidden = _fixture_hidden(g)
    d = dict(E=E, slots=slots, obs_dim=obs_dim, n_act=n_act, hidden=hidden, batch=batch, n_updates=n_updates, seed=seed)
    bstate = O.BufferState(g["buf_offset"], g["buf_last_index"], g["buf_lengths"], g["buf_insertion"],
                           g["rew"], g["terminated"], g["truncated"])
    return g, d, cfg, bstate


@pytest.mark.parametrize("tag", ["auto", "fixed", "widths", "depth3"])
def test_dsac_restatement_matches_reference(tag):
    """oracle_dsac (categorical target with entropy bonus, gathered-Q critic losses with PER weights, actor step
    against the updated critics, alpha step, Polyak) against the unmodified reference DiscreteSAC.update()."""
    from oracle import oracle_dsac as ODS
    from oracle import oracle_sac as OS

    g, d, cfg, bstate = load_dsac(tag)
    actor, c1, c2 = ODS.init_params(d["obs_dim"], d["n_act"], d["hidden"], d["seed"])
    st = OS.SACState.create(actor, c1, c2, cfg)
    for u in range(d["n_updates"]):
        idx = g[f"u{u}_indices"]

        def tq_fn(after):
            return ODS.target_q(st, cfg, g["obs_next"][after]).numpy().reshape(-1, 1)

        ret, _ = O.compute_nstep_return(bstate, idx, tq_fn, cfg.gamma, cfg.n_step)
        ret = ret.astype(np.float32).reshape(-1)
        np.testing.assert_allclose(ret, g[f"u{u}_returns"], rtol=1e-5, atol=1e-6)
        out = ODS.update_with_batch(st, cfg, g["obs"][idx], g["act"][idx], ret, weight=g[f"u{u}_is_weight"])
        stats = g[f"u{u}_stats"]
        np.testing.assert_allclose([out["actor_loss"], out["critic1_loss"], out["critic2_loss"], out["alpha"]],
                                   stats[:4], rtol=1e-5)
        if cfg.auto_alpha:
            np.testing.assert_allclose(out["alpha_loss"], stats[4], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(out["weight"].numpy(), g[f"u{u}_new_weight"], rtol=1e-5, atol=1e-6)
        for name in ("actor", "critic1", "critic2", "critic1_old", "critic2_old"):
            flat = torch.cat([t.reshape(-1) for t in getattr(st, name).values()]).numpy()[::5]          # (dicts are in layer order)
            np.testing.assert_allclose(flat, g[f"u{u}_{name}"], rtol=1e-5, atol=1e-6, err_msg=name)


# ------------------------------------------------------------------------------------ REDQ path
def load_redq(tag):
    from oracle import oracle_redq as OR

    g = load(f"redq_{tag}.npz")
    (E, slots, steps, obs_dim, act_dim, batch, n_updates, seed, auto, n_step, ens, sub, delay, mean) = (int(x) for x in g["dims"])
    c = dict(zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist()))
    cfg = OR.REDQConfig(gamma=c["gamma"], tau=c["tau"], n_step=int(c["n_step"]), alpha=c["alpha"],
                        auto_alpha=bool(c["auto_alpha"]), target_entropy=c["target_entropy"], log_alpha0=c["log_alpha0"],
                        actor_lr=c["actor_lr"], critic_lr=c["critic_lr"], alpha_lr=c["alpha_lr"], ensemble_size=ens,
                        subset_size=sub, actor_delay=delay, target_mode="mean" if mean else "min")
    d = dict(E=E, slots=slots, obs_dim=obs_dim, act_dim=act_dim, batch=batch, n_updates=n_updates, seed=seed,
             hidden=_fixture_hidden(g))
    bstate = O.BufferState(g["buf_offset"], g["buf_last_index"], g["buf_lengths"], g["buf_insertion"],
                           g["rew"], g["terminated"], g["truncated"])
    return g, d, cfg, bstate


@pytest.mark.parametrize("tag", ["min", "mean", "widths", "depth1"])
def test_redq_restatement_matches_reference(tag):
    """oracle_redq (EnsembleLinear critics, random-subset min / mean target, one ensemble loss, delayed actor and alpha
    steps, Polyak) against the unmodified reference REDQ.update()."""
    from oracle import oracle_redq as OR
    from oracle import oracle_sac as OS

    g, d, cfg, bstate = load_redq(tag)
    actor, critic = OR.init_params(d["obs_dim"], d["act_dim"], cfg.ensemble_size, d["seed"], d["hidden"])
    st = OR.REDQState.create(actor, critic, cfg)
    for u in range(d["n_updates"]):
        idx = g[f"u{u}_indices"]

        def tq_fn(after):
            return OR.target_q(st, cfg, g["obs_next"][after], g[f"u{u}_noise_target"], g[f"u{u}_subset"]).numpy().reshape(-1, 1)

        ret, _ = O.compute_nstep_return(bstate, idx, tq_fn, cfg.gamma, cfg.n_step)
        ret = ret.astype(np.float32).reshape(-1)
        np.testing.assert_allclose(ret, g[f"u{u}_returns"], rtol=1e-5, atol=1e-6)
        noise = g[f"u{u}_noise_actor"] if f"u{u}_noise_actor" in g else None
        out = OR.update_with_batch(st, cfg, g["obs"][idx], g["act"][idx], ret, noise)
        stats = g[f"u{u}_stats"]
        np.testing.assert_allclose([out["actor_loss"], out["critic_loss"], out["alpha"]], stats[:3], rtol=2e-5, atol=1e-7)
        if out["alpha_loss"] is not None:
            np.testing.assert_allclose(out["alpha_loss"], stats[3], rtol=1e-5, atol=1e-7)
        else:
            assert np.isnan(stats[3])
        for name, p, order in (("actor", st.actor, list(st.actor)), ("critic", st.critic, list(st.critic)),
                               ("critic_old", st.critic_old, list(st.critic_old))):
            flat = torch.cat([p[k].reshape(-1) for k in order]).numpy()[::61]
            np.testing.assert_allclose(flat, g[f"u{u}_{name}"], rtol=1e-5, atol=1e-6, err_msg=name)


# ------------------------------------------------------------------------------------ NPG / TRPO paths
def load_npg(tag):
    from oracle import oracle_npg as ON

    g = load(f"npg_{tag}.npz")
    E, T, obs_dim, act_dim, batch_size, repeat, is_trpo = (int(x) for x in g["dims"])
    c = dict(zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist()))
    cfg = ON.NPGConfig(algo="trpo" if is_trpo else "npg", gamma=c["gamma"], gae_lambda=c["gae_lambda"],
                       optim_critic_iters=int(c["optim_critic_iters"]), trust_region_size=c["trust_region_size"],
                       advantage_normalization=bool(c["advantage_normalization"]), return_scaling=bool(c["return_scaling"]),
                       max_batchsize=int(c["max_batchsize"]), damping=c["damping"], max_kl=c["max_kl"],
                       backtrack_coeff=c["backtrack_coeff"], max_backtracks=int(c["max_backtracks"]), lr=c["lr"])
    return g, dict(E=E, T=T, obs_dim=obs_dim, act_dim=act_dim, batch_size=batch_size, repeat=repeat), cfg


@pytest.mark.parametrize("tag", ["npg", "trpo"])
def test_npg_trpo_restatement_matches_reference(tag):
    """oracle_npg (natural gradient by conjugate gradients on double-backward Fisher-vector products, NPG's fixed step /
    TRPO's step size + backtracking line search, critic iterations) against the unmodified reference update()."""
    from oracle import oracle_npg as ON

    g, d, cfg = load_npg(tag)
    st = OP.PPOState(params=OP.unflatten_params(torch.as_tensor(g["flat_params0"]), d["obs_dim"], d["act_dim"]))
    obs, obs_next, act = (torch.as_tensor(g[k]) for k in ("obs", "obs_next", "act"))
    pre = ON.preprocess(st, cfg, obs, obs_next, act, g["rew"], g["terminated"], g["truncated"], g["pre_indices"],
                        g["pre_unfinished"])
    for k in ("v_s", "returns", "adv", "logp_old"):
        np.testing.assert_allclose(pre[k].numpy(), g["pre_" + k], rtol=1e-5, atol=1e-5, err_msg=k)
    stats = ON.update(st, cfg, g["obs"], g["act"], pre, d["batch_size"], d["repeat"], g["perms"])
    assert stats.shape == g["stats"].shape
    np.testing.assert_allclose(stats, g["stats"], rtol=2e-4, atol=1e-6)         # conjugate gradients amplify fp32 noise
    np.testing.assert_allclose(OP.flatten_params(st.params).numpy(), g["flat_params"], rtol=1e-4, atol=2e-5)


def load_reinforce(tag):
    from oracle import oracle_reinforce as OR

    g = load(f"reinforce_{tag}.npz")
    E, T, obs_dim, act_dim, batch_size, repeat, n_updates = (int(x) for x in g["dims"])
    c = dict(zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist()))
    cfg = OR.ReinforceConfig(gamma=c["gamma"], return_standardization=bool(c["return_standardization"]), lr=c["lr"])
    shapes = OP.param_shapes(obs_dim, act_dim)
    params, off = {}, 0
    for k in OR.ACTOR_KEYS:
        n = int(np.prod(shapes[k]))
        params[k] = torch.as_tensor(g["actor0"][off:off + n]).reshape(shapes[k]).clone()
        off += n
    assert off == len(g["actor0"])
    return g, dict(E=E, T=T, obs_dim=obs_dim, act_dim=act_dim, batch_size=batch_size or None, repeat=repeat,
                   n_updates=n_updates), cfg, params


@pytest.mark.parametrize("tag", ["std", "plain"])
def test_reinforce_restatement_matches_reference(tag):
    """oracle_reinforce (discounted returns against the running-mean bootstrap, optional standardisation + ret_rms update,
    the vanilla policy-gradient minibatch steps) against the unmodified reference Reinforce.update()."""
    from oracle import oracle_reinforce as OR

    g, d, cfg, params = load_reinforce(tag)
    st = OP.PPOState(params=params)
    for u in range(d["n_updates"]):
        idx = g[f"u{u}_indices"]
        ret = OR.preprocess(st, cfg, g[f"u{u}_rew"], g[f"u{u}_terminated"], g[f"u{u}_truncated"], idx, g[f"u{u}_unfinished"])
        np.testing.assert_allclose(ret.numpy(), g[f"u{u}_returns"].astype(np.float32), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose([st.ret_rms.mean, st.ret_rms.var, st.ret_rms.count], g[f"u{u}_ret_rms"], rtol=1e-12)
        losses = OR.update(st, cfg, g[f"u{u}_obs"][idx], g[f"u{u}_act"][idx], ret, d["batch_size"], d["repeat"], g[f"u{u}_perms"])
        np.testing.assert_allclose(losses, g[f"u{u}_losses"], rtol=2e-5, atol=1e-6)
        flat = torch.cat([st.params[k].reshape(-1) for k in OR.ACTOR_KEYS]).numpy()
        np.testing.assert_allclose(flat, g[f"u{u}_actor"], rtol=1e-4, atol=0.02 * cfg.lr)


@pytest.mark.parametrize("tag", ["cartpole", "per"])
def test_drqn_restatement_matches_reference(tag):
    """oracle_drqn (written-out LSTM cell, stacked observations through prev(), n-step double-Q targets, DQN loss, Adam)
    against the unmodified reference DQN.update() on the Recurrent network, and its evaluation-mode state passing."""
    from oracle import oracle_dqn as OD
    from oracle import oracle_drqn as ORQ
    from tests.dqn_common import drqn_small, load_drqn

    g, d, cfg, bstate = load_drqn(tag)
    L = d["layers"]
    st = OD.DQNState.create(ORQ.unflatten(g["params0"], d["obs_dim"], d["hidden"], L, d["n_act"]), cfg)
    for u in range(d["n_updates"]):
        idx = g[f"u{u}_indices"]
        obs = OD.stacked_frames(bstate, g["obs_rows"], idx, d["stack_num"])
        if u == 0:
            np.testing.assert_array_equal(obs, g["u0_obs"])
        ret = ORQ.preprocess(st, cfg, bstate, g["obs_rows"], idx, d["stack_num"])
        np.testing.assert_allclose(ret, g[f"u{u}_returns"], rtol=1e-5, atol=1e-6)
        w = g[f"u{u}_is_weight"] if d["per"] else None
        loss, td = ORQ.update_with_batch(st, cfg, obs, g["act"][idx], ret, w)
        np.testing.assert_allclose(loss, g[f"u{u}_loss"], rtol=1e-5)
        np.testing.assert_allclose(td.numpy(), g[f"u{u}_td"], rtol=1e-5, atol=1e-6)
        tol = dict(rtol=1e-4, atol=0.02 * cfg.lr * (u + 1))
        np.testing.assert_allclose(ORQ.flatten(st.params, L).numpy()[::17], g[f"u{u}_params_strided"], **tol)
        np.testing.assert_allclose(drqn_small(st.params, L), g[f"u{u}_small"], **tol)
    q1, s1 = ORQ.forward(st.params, g["eval_obs"][0], want_state=True)
    q2, s2 = ORQ.forward(st.params, g["eval_obs"][1], state=s1, want_state=True)
    np.testing.assert_allclose(torch.stack([q1, q2]).numpy(), g["eval_q"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(s2[0].transpose(0, 1).numpy(), g["eval_hidden"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(s2[1].transpose(0, 1).numpy(), g["eval_cell"], rtol=1e-4, atol=1e-5)
