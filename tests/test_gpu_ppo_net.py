"""PPO / A2C on actor-critics with trunks of any depth / width / activation (ts_ppo_net_step, ts_ppo_net_infer;
tianshou_amd.ppo_wide.NetPPOEngine) against what the REFERENCE itself produced: tests/golden/ppo_net_*.npz are written by
oracle/gen_golden.py::gen_ppo_net, which runs the unmodified tianshou PPO / A2C update() on Net(hidden_sizes=...,
activation=...) actor-critics (utils/net/common.py:90-178, 246-369) and records inputs, Batch.split's permutations, the
preprocessing outputs, per-step losses, final parameters and Adam moments."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
# round 6: "bounded_*" = the reference's default actor (unbounded=False: mu = max_action * tanh(.), continuous.py:230-231)
# on the per-layer engine, with RMSprop (optim.py:113-140) / Adam + weight decay (optim.py:95-109)
TAGS = ["relu3", "tanh1_a2c", "linear4", "csigma", "bounded_relu3", "bounded_cs", "ln_relu3", "ln_tanh1_a2c"]


def _load(tag):
    g = dict(np.load(os.path.join(GOLDEN, f"ppo_net_{tag}.npz"), allow_pickle=False))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], [float(v) for v in g["cfg_vals"]]))
    ha, hc = [int(x) for x in g["hidden_a"]], [int(x) for x in g["hidden_c"]]
    cs = bool(int(g["conditioned_sigma"])) if "conditioned_sigma" in g else False
    g["_cs"] = cs
    g["_max_action"] = float(g["max_action"]) if "max_action" in g and float(g["max_action"]) > 0 else None
    g["_ln"] = bool(int(g["layer_norm"])) if "layer_norm" in g else False       # MLP(norm_layer=nn.LayerNorm): (w, b, gamma, beta)*
    g["_ln_eps"] = float(g["ln_eps"]) if g["_ln"] else 1e-5
    per = 4 if g["_ln"] else 2
    na, nc = per * len(ha) + 2 + (2 if cs else 1), per * len(hc) + 2
    return g, cfg, ha, hc, na, nc


@pytest.mark.parametrize("tag", TAGS)
def test_flat_layout_roundtrip(tag):
    """CPU: nn.Linear tensors -> ts_net_layout vector -> tensors; padding entries are zero; sizes match ts_net_layout."""
    import ctypes as C

    from tianshou_amd import _lib
    from tianshou_amd.ppo_wide import net_flat_from_tensors, net_flat_to_tensors

    g, cfg, ha, hc, na, nc = _load(tag)
    obs_dim, act_dim = int(g["dims"][2]), int(g["dims"][3])
    act_name = {0: "tanh", 1: "relu", 2: "none"}[int(g["activation"])]
    a = [torch.from_numpy(g[f"a{i}_0"]) for i in range(na)]
    c = [torch.from_numpy(g[f"c{i}_0"]) for i in range(nc)]
    ln = g["_ln"]
    fa = net_flat_from_tensors(a, obs_dim, ha, act_dim, "cpu", conditioned_sigma=g["_cs"], layer_norm=ln)
    fc = net_flat_from_tensors(c, obs_dim, hc, None, "cpu", layer_norm=ln)
    out = (C.c_int64 * 3)()
    flags = int(g["_cs"]) | (_lib.NetDesc.LAYERNORM if ln else 0)
    _lib.check(_lib.load().ts_net_layout(C.byref(_lib.NetDesc.make(obs_dim, ha, act_name, flags)), _lib.i64(act_dim), out))
    assert fa.numel() == out[1]
    _lib.check(_lib.load().ts_net_layout(C.byref(_lib.NetDesc.make(obs_dim, hc, act_name, flags & _lib.NetDesc.LAYERNORM)), _lib.i64(act_dim), out))
    assert fc.numel() == out[2]
    ta, tc = net_flat_to_tensors(fa, obs_dim, ha, act_dim, True, g["_cs"], ln), net_flat_to_tensors(fc, obs_dim, hc, 1, False, layer_norm=ln)
    assert len(ta) == len(a) and len(tc) == len(c)
    for t0, t1 in zip(a, ta):
        assert torch.equal(t0.reshape(t1.shape), t1)
    for t0, t1 in zip(c, tc):
        assert torch.equal(t0.reshape(t1.shape), t1)
    assert int((fa != 0).sum()) <= sum(t.numel() for t in a)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_update_matches_the_reference(tag):
    from tianshou_amd import ppo as P
    from tianshou_amd.ppo_wide import NetPPOEngine, net_flat_from_tensors

    g, cfg, ha, hc, na, nc = _load(tag)
    E, T, obs_dim, act_dim, batch_size, repeat = (int(x) for x in g["dims"])
    act_name = {0: "tanh", 1: "relu", 2: "none"}[int(g["activation"])]
    a2c = cfg["is_a2c"] > 0
    pcfg = P.PPOConfig(algo="a2c" if a2c else "ppo", gamma=cfg["gamma"], gae_lambda=cfg["gae_lambda"], eps_clip=cfg["eps_clip"],
                       dual_clip=cfg["dual_clip"] or None, value_clip=bool(cfg["value_clip"]),
                       advantage_normalization=bool(cfg["advantage_normalization"]), vf_coef=cfg["vf_coef"], ent_coef=cfg["ent_coef"],
                       max_grad_norm=cfg["max_grad_norm"] or None, return_scaling=bool(cfg["return_scaling"]), lr=cfg["lr"],
                       optimizer="rmsprop" if cfg.get("opt_rmsprop") else "adam", weight_decay=cfg.get("weight_decay", 0.0),
                       adam_eps=cfg.get("opt_eps", 1e-8), rms_alpha=cfg.get("rms_alpha", 0.99),
                       rms_momentum=cfg.get("rms_momentum", 0.0), rms_centered=bool(cfg.get("rms_centered", 0.0)),
                       max_action=g["_max_action"])
    a0 = [torch.from_numpy(g[f"a{i}_0"]) for i in range(na)]
    c0 = [torch.from_numpy(g[f"c{i}_0"]) for i in range(nc)]
    ln = g["_ln"]
    flat = torch.cat([net_flat_from_tensors(a0, obs_dim, ha, act_dim, conditioned_sigma=g["_cs"], layer_norm=ln),
                      net_flat_from_tensors(c0, obs_dim, hc, None, layer_norm=ln)])
    eng = NetPPOEngine(obs_dim, act_dim, ha, hc, act_name, flat, pcfg, conditioned_sigma=g["_cs"], layer_norm=ln, ln_eps=g["_ln_eps"])
    idx = g["pre_indices"]
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")          # noqa: E731
    cut = np.searchsorted(idx, g["pre_unfinished"])
    assert np.array_equal(idx[cut], g["pre_unfinished"])
    b = eng.preprocess(dev(g["obs"][idx]), dev(g["obs_next"][idx]), dev(g["act"][idx]), dev(g["rew"][idx]),
                       dev(g["terminated"][idx]), dev(g["truncated"][idx]), dev(cut))
    for k in ("v_s", "returns", "adv") + (() if a2c else ("logp_old",)):
        np.testing.assert_allclose(b[k].cpu().numpy(), g["pre_" + k], rtol=2e-5, atol=2e-5, err_msg=k)
    losses, steps = eng.update(b, batch_size, repeat, [p for p in g["perms"]])
    torch.cuda.synchronize()
    assert steps == int(g["gradient_steps"])
    np.testing.assert_allclose(losses.cpu().numpy(), g["losses"], rtol=2e-5, atol=2e-6)
    fa, fc = eng.flat_to_tensors(eng.params.cpu())
    ma, mc = eng.flat_to_tensors(eng.adam_m.cpu())
    va, vc = eng.flat_to_tensors(eng.adam_v.cpu())
    lr = cfg["lr"]
    for i in range(na):
        np.testing.assert_allclose(fa[i].numpy().reshape(g[f"a{i}_1"].shape), g[f"a{i}_1"], rtol=1e-4, atol=0.02 * lr, err_msg=f"a{i}")
        np.testing.assert_allclose(ma[i].numpy().reshape(g[f"a{i}_m"].shape), g[f"a{i}_m"], rtol=1e-3, atol=1e-6, err_msg=f"a{i} m")
        np.testing.assert_allclose(va[i].numpy().reshape(g[f"a{i}_v"].shape), g[f"a{i}_v"], rtol=2e-3, atol=1e-9, err_msg=f"a{i} v")
    for i in range(nc):
        np.testing.assert_allclose(fc[i].numpy().reshape(g[f"c{i}_1"].shape), g[f"c{i}_1"], rtol=1e-4, atol=0.02 * lr, err_msg=f"c{i}")
        np.testing.assert_allclose(mc[i].numpy().reshape(g[f"c{i}_m"].shape), g[f"c{i}_m"], rtol=1e-3, atol=1e-6, err_msg=f"c{i} m")
        np.testing.assert_allclose(vc[i].numpy().reshape(g[f"c{i}_v"].shape), g[f"c{i}_v"], rtol=2e-3, atol=1e-9, err_msg=f"c{i} v")
    # padding entries of the flat vector (widths rounded up to 32) stay exactly zero through the update
    pad = torch.ones(eng.P, dtype=torch.bool)
    mask_src = [torch.ones_like(t) for t in a0], [torch.ones_like(t) for t in c0]
    pad &= (torch.cat([net_flat_from_tensors(mask_src[0], obs_dim, ha, act_dim, "cpu", conditioned_sigma=g["_cs"], layer_norm=ln),
                       net_flat_from_tensors(mask_src[1], obs_dim, hc, None, "cpu", layer_norm=ln)]) == 0)
    assert torch.all(eng.params.cpu()[pad] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("act_name,hidden_a,hidden_c,eps,B", [("relu", [96, 72, 40], [64, 48], 1e-5, 1000),
                                                              ("tanh", [33], [130, 7], 1e-3, 257),
                                                              ("none", [64, 64], [32], 1e-5, 64)])
def test_layer_norm_gradients_vs_float64(act_name, hidden_a, hidden_c, eps, B):
    """MLP(norm_layer=nn.LayerNorm) trunks (utils/net/common.py:25-39): the A2C loss gradient of ts_ppo_net_step (lr < 0:
    gradient only) against torch autograd in float64 on the same parameters -- every block (weights, biases, gamma, beta,
    log_sigma) within 1e-5 of the block's largest entry; values / log-probabilities of ts_ppo_net_infer within 1e-5."""
    import torch.nn.functional as F

    from tianshou_amd import ppo as P
    from tianshou_amd.ppo_wide import NetPPOEngine, net_flat_from_tensors

    obs_dim, act_dim = 13, 5
    gen = torch.Generator().manual_seed(7)
    rnd = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)            # noqa: E731

    def make(hidden, n_head):
        t, k = [], obs_dim
        for h in hidden:
            t += [rnd(h, k) / np.sqrt(k), 0.1 * rnd(h), 1.0 + 0.3 * rnd(h), 0.2 * rnd(h)]
            k = h
        return t + [rnd(n_head, k) / np.sqrt(k), 0.1 * rnd(n_head)]

    a = [t.requires_grad_() for t in make(hidden_a, act_dim) + [-0.5 + 0.1 * rnd(act_dim)]]
    c = [t.requires_grad_() for t in make(hidden_c, 1)]
    obs, act, adv, ret = rnd(B, obs_dim), rnd(B, act_dim), rnd(B), rnd(B)
    fn = {"relu": torch.relu, "tanh": torch.tanh, "none": lambda x: x}[act_name]

    def trunk(t, n_hidden):
        h = obs
        for i in range(n_hidden):
            w, b, g_, be = t[4 * i: 4 * i + 4]
            h = fn(F.layer_norm(h @ w.t() + b, (w.shape[0],), g_, be, eps))
        return h @ t[4 * n_hidden].t() + t[4 * n_hidden + 1]

    vf_coef, ent_coef = 0.5, 0.01
    mu, v = trunk(a, len(hidden_a)), trunk(c, len(hidden_c)).reshape(-1)
    dist = torch.distributions.Independent(torch.distributions.Normal(mu, a[-1].exp().expand_as(mu)), 1)
    logp = dist.log_prob(act)
    loss = -(logp * adv).mean() + vf_coef * F.mse_loss(v, ret) - ent_coef * dist.entropy().mean()      # a2c.py:262-272
    loss.backward()

    cfg = P.PPOConfig(algo="a2c", vf_coef=vf_coef, ent_coef=ent_coef, max_grad_norm=None, lr=1e-3)
    f32 = lambda ts: [t.detach().float() for t in ts]                                # noqa: E731
    flat = torch.cat([net_flat_from_tensors(f32(a), obs_dim, hidden_a, act_dim, layer_norm=True),
                      net_flat_from_tensors(f32(c), obs_dim, hidden_c, None, layer_norm=True)])
    eng = NetPPOEngine(obs_dim, act_dim, hidden_a, hidden_c, act_name, flat, cfg, layer_norm=True, ln_eps=eps)
    dev = lambda x: x.detach().float().cuda().contiguous()                           # noqa: E731
    v_e, lp_e = eng.infer(dev(obs), dev(act))
    np.testing.assert_allclose(v_e.cpu().numpy(), v.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lp_e.cpu().numpy(), logp.detach().numpy(), rtol=1e-5, atol=2e-5)
    b = dict(obs=dev(obs), act=dev(act), adv=dev(adv), returns=dev(ret), logp_old=dev(logp), v_s=dev(v))
    losses = torch.zeros(4, device="cuda")
    grad = torch.zeros(eng.P, device="cuda")
    eng.step(b, None, losses, grad_out=grad, apply=False)
    torch.cuda.synchronize()
    assert abs(float(losses[0]) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    ga, gc = eng.flat_to_tensors(grad.cpu())
    for name, got, want in [("actor", ga, a), ("critic", gc, c)]:
        assert len(got) == len(want)
        for i, (g_, w_) in enumerate(zip(got, want)):
            ref = w_.grad.numpy()
            np.testing.assert_allclose(g_.numpy().reshape(ref.shape), ref, rtol=0, atol=1e-5 * max(np.abs(ref).max(), 1e-6),
                                       err_msg=f"{name} tensor {i}")
    # the padding entries of the gradient (widths rounded up to 32, incl. gamma / beta) are exactly zero
    ones = lambda ts: [torch.ones_like(t.detach().float()) for t in ts]              # noqa: E731
    pad = torch.cat([net_flat_from_tensors(ones(a), obs_dim, hidden_a, act_dim, "cpu", layer_norm=True),
                     net_flat_from_tensors(ones(c), obs_dim, hidden_c, None, "cpu", layer_norm=True)]) == 0
    assert torch.all(grad.cpu()[pad] == 0)
