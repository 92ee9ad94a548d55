"""GPU parity of the NPG / TRPO rows: natural gradient by conjugate gradients on Fisher-vector products,
NPG's fixed step and TRPO's step size + line search, critic iterations -- through the C ABI, against the oracle
(oracle/oracle_npg.py: the reference's autograd double backward, pinned by tests/golden/npg_*.npz).

The engine forms F v as the Gauss-Newton product J^T diag(1 / sigma^2) J v / B, the reference by differentiating the mean KL
twice; both are the same matrix, evaluated with different float32 rounding, and conjugate gradients amplify that rounding.
Gradients and single products are held to 1e-5; the conjugate-gradient solution and what follows from it to "within 1e-5
of the float64 solution or at least as close to it as the reference's own float32 evaluation" (x2), as for SAC's actor."""
import numpy as np
import pytest
import torch

from oracle import oracle_npg as ON
from oracle import oracle_ppo as OP
from tests.test_oracle_golden import load_npg

pytestmark = pytest.mark.gpu
A_KEYS = ("a_w1", "a_b1", "a_w2", "a_b2", "a_wmu", "a_bmu", "a_sigma")
C_KEYS = ("c_w1", "c_b1", "c_w2", "c_b2", "c_wv", "c_bv")


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def make_engine(p, obs_dim, act_dim, cfg):
    from tianshou_amd import npg as NG

    ecfg = NG.NPGConfig(**{k: getattr(cfg, k) for k in ("algo", "gamma", "gae_lambda", "optim_critic_iters",
                                                       "trust_region_size", "advantage_normalization", "return_scaling",
                                                       "damping", "max_kl", "backtrack_coeff", "max_backtracks", "lr", "betas",
                                                       "adam_eps", "max_grad_norm")})
    return NG.NPGEngine(obs_dim, act_dim, 64, NG.actor_flat_from_torch([p[k] for k in A_KEYS], obs_dim, 64, act_dim),
                        NG.critic_flat_from_torch([p[k] for k in C_KEYS], obs_dim, 64), ecfg)


def oracle_order(flat, obs_dim, act_dim):
    """engine actor vector -> the reference's flat order (sigma_param first)."""
    from tianshou_amd import npg as NG

    t = NG.actor_flat_to_torch(flat, obs_dim, 64, act_dim)
    return torch.cat([t[6].reshape(-1)] + [x.reshape(-1) for x in t[:6]]).cpu()


def rand_params(obs_dim, act_dim, seed):
    p = OP.init_params(obs_dim, act_dim, seed=seed)
    g = torch.Generator().manual_seed(seed)
    p["a_wmu"] = p["a_wmu"] * 30.0                            # a policy whose mean actually depends on the observation
    p["a_sigma"] = torch.randn(act_dim, generator=g) * 0.3 - 0.5
    return p


@pytest.mark.parametrize("obs_dim,act_dim,B", [(17, 6, 1000), (33, 1, 64), (4, 32, 257), (3, 1, 33), (27, 8, 5000), (8, 3, 70001),
                                               (32, 2, 31)])
def test_infer_vs_oracle(obs_dim, act_dim, B):
    p = rand_params(obs_dim, act_dim, 1)
    eng = make_engine(p, obs_dim, act_dim, ON.NPGConfig())
    g = torch.Generator().manual_seed(B)
    obs, act = torch.randn(B, obs_dim, generator=g), torch.randn(B, act_dim, generator=g)
    v, logp, mu = eng.infer(obs, act, want_mu=True)
    with torch.no_grad():
        mu_ref, sigma = OP.actor_forward(p, obs)
        assert rel_err(mu.cpu(), mu_ref) < 1e-5 and rel_err(v.cpu(), OP.critic_forward(p, obs).flatten()) < 1e-5
        np.testing.assert_allclose(logp.cpu().numpy(), OP.dist_of(mu_ref, sigma).log_prob(act).numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("fvp_path", ["one_launch_kernel", "per_layer_gemms"])
@pytest.mark.parametrize("algo,B", [("npg", 4096), ("trpo", 700)])
def test_actor_step_vs_oracle_and_float64(algo, B, fvp_path, monkeypatch):
    """(both routes of the Fisher-vector product: ts_npg_q.h's one-launch kernel and the per-layer GEMM passes)"""
    monkeypatch.setenv("TS_NPG_FVP", "1" if fvp_path == "one_launch_kernel" else "0")
    obs_dim, act_dim = 17, 6
    p = rand_params(obs_dim, act_dim, 3)
    cfg = ON.NPGConfig(algo=algo, trust_region_size=0.1, optim_critic_iters=1)
    g = torch.Generator().manual_seed(5)
    obs, act, adv = torch.randn(B, obs_dim, generator=g), torch.randn(B, act_dim, generator=g) * 0.8, torch.randn(B, generator=g)
    with torch.no_grad():
        logp_old = OP.dist_of(*OP.actor_forward(p, obs)).log_prob(act)
    ret = torch.zeros(B)
    runs = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        st = OP.PPOState(params={k: v.to(dt) for k, v in p.items()})
        col: dict = {}
        ON.minibatch_step(st, cfg, obs.to(dt), act.to(dt), adv.to(dt), ret.to(dt), logp_old.to(dt), collect=col)
        runs[name] = (col, torch.cat([st.params[k].reshape(-1) for k in ON.ACTOR_KEYS]))
    eng = make_engine(p, obs_dim, act_dim, cfg)
    stats, dbg = eng.actor_step(obs, act, adv, logp_old, want_debug=True)
    (c32, new32), (c64, new64) = runs["f32"], runs["f64"]
    grad = oracle_order(dbg[0], obs_dim, act_dim)
    assert rel_err(grad, c64["flat_grads"]) < max(1e-5, 2 * rel_err(c32["flat_grads"], c64["flat_grads"]))
    fg = oracle_order(dbg[2], obs_dim, act_dim)
    assert rel_err(fg, c64["mvp_of_grad"]) < max(1e-5, 2 * rel_err(c32["mvp_of_grad"], c64["mvp_of_grad"]))
    sd = -oracle_order(dbg[1], obs_dim, act_dim)
    e_gpu, e_ref = rel_err(sd, c64["search_direction"]), rel_err(c32["search_direction"], c64["search_direction"])
    assert e_gpu < max(1e-5, 2 * e_ref), (e_gpu, e_ref)
    new = oracle_order(eng.actor, obs_dim, act_dim)
    step64 = (new64 - torch.cat([p[k].reshape(-1) for k in ON.ACTOR_KEYS]).double()).abs().max().item()
    e_gpu = (new.double() - new64).abs().max().item() / step64
    e_ref = (new32.double() - new64).abs().max().item() / step64
    assert e_gpu < max(1e-5, 2 * e_ref), (e_gpu, e_ref)      # the parameter step, on the scale of the step itself


@pytest.mark.parametrize("algo", ["npg", "trpo"])
@pytest.mark.parametrize("obs_dim,act_dim,B", [(17, 6, 1000), (3, 1, 33), (11, 3, 4101), (27, 8, 20000), (32, 2, 64),
                                               (8, 4, 65536)])
def test_one_launch_actor_passes_match_the_per_layer_passes(obs_dim, act_dim, B, algo, monkeypatch):
    """The surrogate's gradient, F g + damping g, the conjugate-gradient solution, the step's statistics and the new
    parameters from ts_npg_q.h's kernels (every instantiated layer-1 depth, ragged last tiles, one tile .. several tiles per
    workgroup, NPG's single candidate and TRPO's ten in one launch) against the GEMM passes they replace, and against the
    float64 evaluation of the reference's formulation."""
    p = rand_params(obs_dim, act_dim, 11 + obs_dim)
    cfg = ON.NPGConfig(algo=algo, trust_region_size=0.1, optim_critic_iters=1)
    g = torch.Generator().manual_seed(B)
    obs, act, adv = torch.randn(B, obs_dim, generator=g), torch.randn(B, act_dim, generator=g) * 0.8, torch.randn(B, generator=g)
    with torch.no_grad():
        logp_old = OP.dist_of(*OP.actor_forward(p, obs)).log_prob(act)
    out = {}
    for path in ("1", "0"):
        monkeypatch.setenv("TS_NPG_FVP", path)
        eng = make_engine(p, obs_dim, act_dim, cfg)
        stats, dbg = eng.actor_step(obs, act, adv, logp_old, want_debug=True)
        out[path] = (dbg.cpu(), eng.actor.cpu(), stats.cpu())
    assert rel_err(out["1"][0][0], out["0"][0][0]) < 1e-5                                   # gradient
    assert rel_err(out["1"][0][2], out["0"][0][2]) < 1e-5                                   # F g + damping g
    st = OP.PPOState(params={k: v.double() for k, v in p.items()})
    col: dict = {}
    ON.minibatch_step(st, cfg, obs.double(), act.double(), adv.double(), torch.zeros(B).double(), logp_old.double(), collect=col)
    new64 = torch.cat([st.params[k].reshape(-1) for k in ON.ACTOR_KEYS])
    for key, row in (("flat_grads", 0), ("mvp_of_grad", 2)):
        e1 = rel_err(oracle_order(out["1"][0][row].cuda(), obs_dim, act_dim), col[key])
        e0 = rel_err(oracle_order(out["0"][0][row].cuda(), obs_dim, act_dim), col[key])
        assert e1 < max(1e-5, 2 * e0), (key, e1, e0)
    sd1 = -oracle_order(out["1"][0][1].cuda(), obs_dim, act_dim)
    sd0 = -oracle_order(out["0"][0][1].cuda(), obs_dim, act_dim)
    e1, e0 = rel_err(sd1, col["search_direction"]), rel_err(sd0, col["search_direction"])
    assert e1 < max(1e-4, 2 * e0), (e1, e0)
    # the surrogate, the chosen candidate's kl and the step size; the parameter step on the scale of the step itself
    s1, s0 = out["1"][2].double().numpy(), out["0"][2].double().numpy()
    np.testing.assert_allclose(s1[0], s0[0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(s1[1:], s0[1:], rtol=2e-3, atol=1e-7)
    old = torch.cat([p[k].reshape(-1) for k in ON.ACTOR_KEYS]).double()
    step64 = (new64 - old).abs().max().item()
    if step64 > 0:
        e1 = (oracle_order(out["1"][1].cuda(), obs_dim, act_dim).double() - new64).abs().max().item() / step64
        e0 = (oracle_order(out["0"][1].cuda(), obs_dim, act_dim).double() - new64).abs().max().item() / step64
        assert e1 < max(1e-4, 2 * e0), (e1, e0)


@pytest.mark.parametrize("path", ["one_launch_kernel", "per_layer_gemms"])
@pytest.mark.parametrize("obs_dim,B", [(17, 3000), (3, 33), (30, 40000)])
def test_critic_step_vs_oracle(obs_dim, B, path, monkeypatch):
    monkeypatch.setenv("TS_NPG_FVP", "0" if path == "per_layer_gemms" else "1")
    act_dim = 6
    p = rand_params(obs_dim, act_dim, 7)
    cfg = ON.NPGConfig(lr=1e-3, max_grad_norm=0.5)
    eng = make_engine(p, obs_dim, act_dim, cfg)
    g = torch.Generator().manual_seed(2)
    obs, ret = torch.randn(B, obs_dim, generator=g), torch.randn(B, generator=g) * 2
    st = OP.PPOState(params={k: v.clone() for k, v in p.items()})
    from tianshou_amd import npg as NG
    for _ in range(3):
        pc = {k: st.params[k].clone().requires_grad_(True) for k in C_KEYS}
        value = OP.critic_forward({**st.params, **pc}, obs).flatten()
        vf = torch.nn.functional.mse_loss(ret, value)
        gs = dict(zip(C_KEYS, torch.autograd.grad(vf, [pc[k] for k in C_KEYS])))
        grad = torch.empty(eng.lay["critic_count"], device="cuda")
        loss = eng.critic_step(obs, ret, grad_out=grad, apply=False)
        assert abs(float(loss) - float(vf)) <= 1e-5 * abs(float(vf))
        # float64 evaluation of the same gradient: the bias gradients are sums of 3000 terms that cancel to ~1e-3 of
        # their magnitude, so the float32 autograd result itself (whose summation order depends on the host's thread
        # count) is only good to a few 1e-6..1e-5 there; the bar is 1e-5 or "as close to float64 as the reference"
        p64 = {k: v.double() for k, v in st.params.items()}
        pc64 = {k: p64[k].clone().requires_grad_(True) for k in C_KEYS}
        vf64 = torch.nn.functional.mse_loss(ret.double(), OP.critic_forward({**p64, **pc64}, obs.double()).flatten())
        gs64 = dict(zip(C_KEYS, torch.autograd.grad(vf64, [pc64[k] for k in C_KEYS])))
        # (40,000 samples: the head bias gradient is a mean of 40,000 terms of magnitude ~4 that cancels to ~1e-3: 4e-5 of it
        # is 0.1 ulp of a mean term.  The head bias gradient is also accepted within eps32 * mean |term|: the forward error
        # bound of a float32 tree summation is log2(n) times that)
        with torch.no_grad():
            mean_abs_term = float((2 * (OP.critic_forward(p64, obs.double()).flatten() - ret.double())).abs().mean())
        for t, k in zip(NG.critic_flat_to_torch(grad, obs_dim, 64), C_KEYS):
            ok = rel_err(t.cpu(), gs64[k]) < max(1e-5, 2 * rel_err(gs[k], gs64[k]))
            if k == "c_bv":
                ok = ok or abs(float(t.cpu().reshape(-1)[0]) - float(gs64[k].reshape(-1)[0])) < 1.19e-7 * mean_abs_term
            assert ok, k
        ON._critic_adam(st, cfg, gs)
        eng.critic_step(obs, ret)
        for t, k in zip(NG.critic_flat_to_torch(eng.critic, obs_dim, 64), C_KEYS):
            np.testing.assert_allclose(t.cpu().numpy(), st.params[k].numpy(), rtol=1e-5, atol=0.02 * cfg.lr, err_msg=k)


@pytest.mark.parametrize("path", ["one_launch_kernel", "per_layer_gemms"])
def test_critic_steps_is_the_loop_of_critic_step(path, monkeypatch):
    """ts_npg_critic_steps (all optim_critic_iters iterations of a minibatch in one call) == the same number of
    ts_npg_critic_step calls: bit-identical parameters, moments and last loss."""
    monkeypatch.setenv("TS_NPG_FVP", "0" if path == "per_layer_gemms" else "1")
    obs_dim, act_dim, B = 17, 6, 5000
    p = rand_params(obs_dim, act_dim, 9)
    cfg = ON.NPGConfig(lr=1e-3, max_grad_norm=0.5)
    g = torch.Generator().manual_seed(4)
    obs, ret = torch.randn(B, obs_dim, generator=g), torch.randn(B, generator=g) * 2
    a, b = make_engine(p, obs_dim, act_dim, cfg), make_engine(p, obs_dim, act_dim, cfg)
    la = a.critic_steps(obs, ret, 5)
    for _ in range(5):
        lb = b.critic_step(obs, ret)
    assert a.adam_step == b.adam_step == 5
    for x, y in ((a.critic, b.critic), (a.critic_m, b.critic_m), (a.critic_v, b.critic_v), (la, lb)):
        np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())


@pytest.mark.parametrize("critic_path", ["one_launch_passes", "per_layer_gemms"])
@pytest.mark.parametrize("tag", ["npg", "trpo"])
def test_update_matches_reference_golden(tag, critic_path, monkeypatch):
    """(both routes of every network pass -- preprocessing, gradient, Fisher-vector products, candidate evaluations, critic
    iterations: the one-launch kernels of csrc/ts_npg_q.h and the per-layer GEMM passes, TS_NPG_FVP=0)"""
    from tianshou_amd import npg as NG

    monkeypatch.setenv("TS_NPG_FVP", "0" if critic_path == "per_layer_gemms" else "1")
    g, d, cfg = load_npg(tag)
    p0 = OP.unflatten_params(torch.as_tensor(g["flat_params0"]), d["obs_dim"], d["act_dim"])
    eng = make_engine(p0, d["obs_dim"], d["act_dim"], cfg)
    idx = g["pre_indices"]
    cut = np.nonzero(np.isin(idx, g["pre_unfinished"]))[0]
    pre = eng.preprocess(g["obs"], g["obs_next"], g["act"], g["rew"], g["terminated"], g["truncated"], cut)
    for k in ("v_s", "returns", "adv", "logp_old"):
        np.testing.assert_allclose(pre[k].cpu().numpy(), g["pre_" + k], rtol=1e-5, atol=2e-5, err_msg=k)
    stats, steps = eng.update(pre, d["batch_size"], d["repeat"], list(g["perms"]))
    s, ref = stats.cpu().numpy(), g["stats"]
    assert steps == ref.shape[0]
    # reference (float32 double backward + CG) vs engine (float32 Gauss-Newton + CG): both a few 1e-4 from exact arithmetic
    np.testing.assert_allclose(s[:, :ref.shape[1]], ref, rtol=2e-3, atol=2e-5)
    a = NG.actor_flat_to_torch(eng.actor, d["obs_dim"], 64, d["act_dim"])
    c = NG.critic_flat_to_torch(eng.critic, d["obs_dim"], 64)
    flat = torch.cat([t.reshape(-1) for t in a + c]).cpu().numpy()            # = oracle_ppo.PARAM_ORDER
    step = np.abs(g["flat_params"] - g["flat_params0"]).max()
    assert np.abs(flat - g["flat_params"]).max() < 5e-3 * step


def _split_flat(flat, obs_dim, act_dim, ha, hc):
    """A fixture's flat vector (gen_golden._flat_from_modules: actor trunk (w, b)*, mu (w, b), sigma_param | critic trunk
    (w, b)*, last (w, b)) -> (actor tensors, critic tensors) in nn.Linear layout."""
    flat = torch.as_tensor(flat)
    shapes_a, k = [], obs_dim
    for h in ha:
        shapes_a += [(h, k), (h,)]
        k = h
    shapes_a += [(act_dim, k), (act_dim,), (act_dim,)]
    shapes_c, k = [], obs_dim
    for h in hc:
        shapes_c += [(h, k), (h,)]
        k = h
    shapes_c += [(1, k), (1,)]
    out, off = [], 0
    for shp in shapes_a + shapes_c:
        n = int(np.prod(shp))
        out.append(flat[off:off + n].reshape(shp).clone())
        off += n
    assert off == flat.numel()
    return out[:len(shapes_a)], out[len(shapes_a):]


@pytest.mark.parametrize("tag", ["npg_relu3", "trpo_tanh1"])
def test_generic_trunks_match_reference_golden(tag):
    """Round 6: NPG / TRPO on `Net` trunks outside [h1, h2] tanh -- NPG on THREE ReLU layers (actor [64, 48, 32]) beside a critic
    of another depth ([40, 56]); TRPO on ONE tanh layer (actor [96], critic [80]) -- NetNPGEngine, layer by layer on the GEMM
    kernels (ts_npg_net_actor_step / ts_npg_net_critic_steps), against what the unmodified REFERENCE's update() produced
    (tests/golden/npg_{npg_relu3,trpo_tanh1}.npz, oracle/gen_golden.py::gen_depth): preprocessing, per-step statistics, parameters."""
    from tianshou_amd import npg as NG
    from tianshou_amd.ppo_wide import net_flat_from_tensors as nf

    g, d, cfg = load_npg(tag)
    obs_dim, act_dim = d["obs_dim"], d["act_dim"]
    ha, hc = [int(x) for x in g["hidden_a"]], [int(x) for x in g["hidden_c"]]
    act_name = {0: "tanh", 1: "relu", 2: "none"}[int(g["activation"])]
    ta, tc = _split_flat(g["flat_params0"], obs_dim, act_dim, ha, hc)
    ecfg = NG.NPGConfig(**{k: getattr(cfg, k) for k in ("algo", "gamma", "gae_lambda", "optim_critic_iters", "trust_region_size",
                                                       "advantage_normalization", "return_scaling", "damping", "max_kl",
                                                       "backtrack_coeff", "max_backtracks", "lr")})
    eng = NG.NetNPGEngine(obs_dim, act_dim, ha, hc, act_name, nf(ta, obs_dim, ha, act_dim), nf(tc, obs_dim, hc, None), ecfg)
    for got, want in zip(eng.actor_to_tensors(eng.actor) + eng.critic_to_tensors(eng.critic), ta + tc):        # layout round trip
        assert torch.equal(got.cpu().reshape(want.shape), want)
    idx = g["pre_indices"]
    cut = np.nonzero(np.isin(idx, g["pre_unfinished"]))[0]
    pre = eng.preprocess(g["obs"], g["obs_next"], g["act"], g["rew"], g["terminated"], g["truncated"], cut)
    for k in ("v_s", "returns", "adv", "logp_old"):
        np.testing.assert_allclose(pre[k].cpu().numpy(), g["pre_" + k], rtol=1e-5, atol=2e-5, err_msg=k)
    stats, steps = eng.update(pre, d["batch_size"], d["repeat"], list(g["perms"]))
    s, ref = stats.cpu().numpy(), g["stats"]
    assert steps == ref.shape[0]
    np.testing.assert_allclose(s[:, :ref.shape[1]], ref, rtol=2e-3, atol=2e-5)       # (float32 CG solves on both sides, as above)
    flat = torch.cat([t.reshape(-1) for t in eng.actor_to_tensors(eng.actor) + eng.critic_to_tensors(eng.critic)]).cpu().numpy()
    step = np.abs(g["flat_params"] - g["flat_params0"]).max()
    assert step > 0 and np.abs(flat - g["flat_params"]).max() < 5e-3 * step


@pytest.mark.parametrize("act_name,ha", [("relu", [64, 48, 32]), ("tanh", [96]), ("none", [32, 32]), ("tanh", [64, 64])])
def test_generic_trunk_fisher_vector_product_vs_float64(act_name, ha):
    """The three vectors an actor step exposes -- gradient, conjugate-gradient solution, F g + damping g -- of the per-layer path
    (one forward-mode + one reverse pass per product) against float64 autograd of the same network: the gradient of the
    surrogate and the DOUBLE-BACKWARD Fisher-vector product of the mean KL the reference computes (npg.py:195-200) at the
    expansion point, where it equals the Gauss-Newton form the engine evaluates."""
    from torch.distributions import Independent, Normal, kl_divergence

    from tianshou_amd import npg as NG
    from tianshou_amd.ppo_wide import net_flat_from_tensors as nf

    obs_dim, act_dim, B = 13, 4, 600
    gen = torch.Generator().manual_seed(3)
    ta, k = [], obs_dim
    for h in ha:
        ta += [torch.randn(h, k, generator=gen) / np.sqrt(k), torch.randn(h, generator=gen) * 0.1]
        k = h
    ta += [torch.randn(act_dim, k, generator=gen) * 0.3 / np.sqrt(k), torch.randn(act_dim, generator=gen) * 0.1,
           torch.full((act_dim,), -0.5) + 0.1 * torch.randn(act_dim, generator=gen)]
    tc = [torch.randn(32, obs_dim, generator=gen) * 0.1, torch.zeros(32), torch.randn(1, 32, generator=gen) * 0.1, torch.zeros(1)]
    cfg = NG.NPGConfig(algo="npg", damping=0.1, trust_region_size=0.05)
    eng = NG.NetNPGEngine(obs_dim, act_dim, ha, [32], act_name, nf(ta, obs_dim, ha, act_dim), nf(tc, obs_dim, [32], None), cfg)
    obs, act = torch.randn(B, obs_dim, generator=gen), torch.randn(B, act_dim, generator=gen) * 0.7
    adv = torch.randn(B, generator=gen)
    _, dbg = eng.actor_step(obs, act, adv, want_debug=True)
    fn = {"relu": torch.relu, "tanh": torch.tanh, "none": lambda x: x}[act_name]
    p64 = [t.double().requires_grad_(True) for t in ta]

    def dist_of(params):
        h = obs.double()
        for i in range(len(ha)):
            h = fn(torch.nn.functional.linear(h, params[2 * i], params[2 * i + 1]))
        mu = torch.nn.functional.linear(h, params[2 * len(ha)], params[2 * len(ha) + 1])
        return Independent(Normal(mu, (params[-1].view(1, -1) + torch.zeros_like(mu)).exp()), 1)

    dist = dist_of(p64)
    loss = -(dist.log_prob(act.double()) * adv.double()).mean()
    g64 = torch.cat([x.reshape(-1) for x in torch.autograd.grad(loss, p64, retain_graph=True)])
    kl = kl_divergence(dist_of([t.detach() for t in p64]), dist).mean()
    gk = torch.cat([x.reshape(-1) for x in torch.autograd.grad(kl, p64, create_graph=True)])
    fg64 = torch.cat([x.reshape(-1) for x in torch.autograd.grad((gk * g64).sum(), p64)]) + 0.1 * g64
    got_g = torch.cat([t.reshape(-1) for t in eng.actor_to_tensors(dbg[0])]).cpu()
    got_fg = torch.cat([t.reshape(-1) for t in eng.actor_to_tensors(dbg[2])]).cpu()
    assert rel_err(got_g, g64) < 2e-5, rel_err(got_g, g64)
    assert rel_err(got_fg, fg64) < 5e-5, rel_err(got_fg, fg64)


def test_bad_arguments_fail_loudly():
    from tianshou_amd import npg as NG

    with pytest.raises(Exception):
        NG.layout(17, 48, 6)
    with pytest.raises(Exception):
        NG.layout(17, 64, 33)
    p = rand_params(17, 6, 0)
    eng = make_engine(p, 17, 6, ON.NPGConfig(algo="trpo"))
    z = torch.zeros
    with pytest.raises(ValueError):
        eng.actor_step(z(8, 17), z(8, 6), z(8))                # TRPO without logp_old
    with pytest.raises(RuntimeError):
        NG.NPGEngine(17, 6, 64, eng.actor.cpu(), eng.critic.cpu(), eng.cfg)
