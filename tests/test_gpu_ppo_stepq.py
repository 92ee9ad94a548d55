"""The two fused PPO step kernels against the oracle on the same minibatches (gradient per parameter block, the four loss
figures, parameters after the Adam steps): ppo_step2_kernel (TS_PPO_STEPQ=0: 128-sample workgroups, LDS weight image) and
the feature-split kernel of ts_ppo_q.h in its 128- and 168-register builds (TS_PPO_STEPQ=1 / 2).  Reference lines:
PPO._update_with_batch ppo.py:164-224, A2C a2c.py:249-290, Optimizer.step algorithm_base.py:484-500."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import oracle_ppo as OP

pytestmark = pytest.mark.gpu

PPO_KW = dict(eps_clip=0.2, vf_coef=0.25, ent_coef=0.0, max_grad_norm=0.5, value_clip=True, advantage_normalization=False, lr=3e-4)
CASES = {
    "c2_like": (4096, 17, 6, 1024, 2, PPO_KW),
    "ragged_advnorm_dualclip_entropy": (3001, 17, 6, 1000, 1, dict(eps_clip=0.2, dual_clip=3.0, vf_coef=0.5, ent_coef=0.01,
                                                                  max_grad_norm=None, value_clip=False,
                                                                  advantage_normalization=True, lr=1e-3)),
    "a2c_obs11_act3": (2048, 11, 3, 512, 1, dict(algo="a2c", vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5, lr=7e-4,
                                                 advantage_normalization=False)),
    "obs3_act1": (777, 3, 1, 256, 1, dict(PPO_KW, advantage_normalization=True)),
    "obs27_act8": (1024, 27, 8, 1024, 1, PPO_KW),
    "obs21_act1_last_kstep_leaves_record": (1000, 21, 1, 500, 1, dict(PPO_KW, ent_coef=0.01)),
    "one_partial_tile": (20, 17, 6, 20, 1, PPO_KW),
    "many_tiles_per_workgroup": (65536 + 40, 17, 6, 65536 + 40, 1, PPO_KW),
}


def _make(n, obs_dim, act_dim, seed):
    rng = np.random.default_rng(seed)
    params = OP.init_params(obs_dim, act_dim, seed=seed)
    params["a_wmu"] = params["a_wmu"] * 30.0       # wide heads: every branch of the clipped loss is taken by some sample
    params["a_bmu"] = torch.from_numpy(rng.normal(size=act_dim).astype(np.float32) * 0.1)
    params["c_bv"] = torch.from_numpy(rng.normal(size=1).astype(np.float32) * 0.1)
    params["a_b2"] = torch.from_numpy(rng.normal(size=64).astype(np.float32) * 0.1)
    params["c_b1"] = torch.from_numpy(rng.normal(size=64).astype(np.float32) * 0.1)
    b = dict(obs=rng.normal(size=(n, obs_dim)).astype(np.float32), act=rng.normal(size=(n, act_dim)).astype(np.float32),
             adv=rng.normal(size=n).astype(np.float32), returns=rng.normal(size=n).astype(np.float32),
             logp_old=(rng.normal(size=n) * 0.3 - 1.2 * act_dim).astype(np.float32), v_s=rng.normal(size=n).astype(np.float32))
    return params, b


def _engine_run(variant, monkeypatch, params, b, obs_dim, act_dim, kw, batch, repeat, perms):
    from tianshou_amd import ppo as P

    monkeypatch.setenv("TS_PPO_STEPQ", str(variant))
    eng = P.PPOEngine(obs_dim, act_dim, OP.flatten_params(params).cuda(), P.PPOConfig(**kw))
    db = {k: torch.as_tensor(v, device="cuda") for k, v in b.items()}
    losses, _, grads = eng.update(db, batch, repeat, perms, want_grad=True)
    torch.cuda.synchronize()
    return losses.cpu().numpy().astype(np.float64), grads.cpu().numpy(), eng.params.cpu().numpy()


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("case", list(CASES))
def test_step_kernel_matches_oracle(case, variant, monkeypatch):
    from tianshou_amd import ppo as P

    n, obs_dim, act_dim, batch, repeat, kw = CASES[case]
    params, b = _make(n, obs_dim, act_dim, seed=n)
    perms = [np.random.default_rng(1).permutation(n) for _ in range(repeat)]
    st = OP.PPOState(params={k: v.clone() for k, v in params.items()})
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    lo, go = OP.update(st, OP.PPOConfig(**kw), {"obs": tb["obs"], "act": tb["act"]},
                       {k: tb[k] for k in ("adv", "returns", "logp_old", "v_s")}, batch, repeat, perms, collect_grads=True)
    go = go.numpy()
    l, g, p = _engine_run(variant, monkeypatch, params, b, obs_dim, act_dim, kw, batch, repeat, perms)
    np.testing.assert_allclose(l, lo, rtol=1e-5, atol=2e-6)                # north_star: losses within 1e-5
    off = 0
    for k, shp in P.param_shapes(obs_dim, act_dim).items():
        m = int(np.prod(shp))
        ref = go[off:off + m]
        # gradients: 1e-4 of the block's largest entry (sums of 1e3..6e4 fp32 products in a different order)
        assert np.abs(g[off:off + m] - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-12), (case, variant, k)
        off += m
    np.testing.assert_allclose(p, OP.flatten_params(st.params).numpy(), rtol=1e-4, atol=0.02 * kw["lr"])


@pytest.mark.parametrize("nets", [1, 2])
def test_one_network_steps_agree_between_the_kernels(nets, monkeypatch):
    """ts_ppo_hparams.nets = 1 / 2 (Reinforce's actor steps, NPG / TRPO's critic iterations): the live network's gradient and
    parameters from the feature-split kernel equal the 128-sample kernel's; the other network takes a zero gradient."""
    from tianshou_amd import ppo as P

    n, obs_dim, act_dim = 2048, 17, 6
    kw = dict(algo="a2c", nets=nets, vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5, lr=7e-4, advantage_normalization=False)
    params, b = _make(n, obs_dim, act_dim, seed=5)
    perms = [np.random.default_rng(2).permutation(n)]
    res = [_engine_run(v, monkeypatch, params, b, obs_dim, act_dim, kw, 512, 1, perms) for v in (0, 1, 2)]
    n_actor = sum(int(np.prod(s)) for k, s in P.param_shapes(obs_dim, act_dim).items() if k.startswith("a_"))
    live = slice(0, n_actor) if nets == 1 else slice(n_actor, None)
    dead = slice(n_actor, None) if nets == 1 else slice(0, n_actor)
    for l, g, p in res[1:]:
        assert np.all(g[dead] == 0.0)
        scale = np.abs(res[0][1][live]).max()
        assert np.abs(g[live] - res[0][1][live]).max() <= 2e-6 * scale
        np.testing.assert_allclose(p, res[0][2], rtol=0, atol=2e-6)
        np.testing.assert_allclose(l[:, nets], res[0][0][:, nets], rtol=1e-6)


def test_default_dispatch_by_row_count(monkeypatch):
    """ts_ppo_step_plan: the feature-split kernel up to 3 tiles per workgroup of a two-per-CU grid, the 128-sample kernel above."""
    from tianshou_amd import _lib

    monkeypatch.delenv("TS_PPO_STEPQ", raising=False)
    monkeypatch.delenv("TS_PPO_STEPQ_PAIRS", raising=False)
    lib = _lib.load()
    cus = torch.cuda.get_device_properties(0).multi_processor_count

    def plan(rows):
        v, g, s = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        _lib.check(lib.ts_ppo_step_plan(_lib.i64(17), _lib.i64(6), _lib.i64(rows), C.c_int32(0), C.byref(v), C.byref(g), C.byref(s)))
        return v.value, g.value, s.value

    assert plan(32 * 3 * cus) == (2, 2 * cus, cus)
    assert plan(32 * cus + 32) == (2, 2 * cus, cus)
    assert plan(32 * cus) == (2, cus, cus // 2)            # up to one tile per CU and network: one workgroup per CU
    assert plan(64) == (2, 4, 2)
    v, g, s = plan(65536)
    assert v == 0 and g == s == min(512, 2 * cus)


def test_update_whose_minibatches_run_both_kernels(monkeypatch):
    """Default dispatch inside ONE update: Batch.split(merge_last=True) makes the last minibatch the largest -- here 20,000-row
    minibatches on the feature-split kernel (256 slabs of 11,608 floats, 182 reduction workgroups) and a 28,000-row last one
    on the 128-sample kernel (219 slabs of 11,088, 174 workgroups).  The slab area must hold the largest NEED (not the
    largest minibatch's), and every step's clip_grad_norm_ factor must come from the partial sums of ITS reduction.
    Oracle: losses of every step, gradient of the last step, parameters."""
    from tianshou_amd import ppo as P

    monkeypatch.delenv("TS_PPO_STEPQ", raising=False)
    n, obs_dim, act_dim, batch, repeat = 68000, 17, 6, 20000, 2
    kw = dict(PPO_KW, max_grad_norm=0.05)                  # small enough that every step is clipped
    params, b = _make(n, obs_dim, act_dim, seed=3)
    perms = [np.random.default_rng(4 + r).permutation(n) for r in range(repeat)]
    st = OP.PPOState(params={k: v.clone() for k, v in params.items()})
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    lo, go = OP.update(st, OP.PPOConfig(**kw), {"obs": tb["obs"], "act": tb["act"]},
                       {k: tb[k] for k in ("adv", "returns", "logp_old", "v_s")}, batch, repeat, perms, collect_grads=True)
    eng = P.PPOEngine(obs_dim, act_dim, OP.flatten_params(params).cuda(), P.PPOConfig(**kw))
    db = {k: torch.as_tensor(v, device="cuda") for k, v in b.items()}
    losses, steps, grads = eng.update(db, batch, repeat, perms, want_grad=True)
    assert steps == 6
    np.testing.assert_allclose(losses.cpu().numpy().astype(np.float64), lo, rtol=1e-5, atol=2e-6)
    g, ref = grads.cpu().numpy(), go.numpy()
    assert np.abs(g - ref).max() <= 1e-4 * np.abs(ref).max()
    np.testing.assert_allclose(eng.params.cpu().numpy(), OP.flatten_params(st.params).numpy(), rtol=1e-4, atol=0.02 * kw["lr"])
    # Adam's first steps move every parameter by ~lr whatever the clip factor is; a wrong factor shows up in the moments
    mm = torch.cat([st.adam_m[k].reshape(-1) for k in P.PARAM_ORDER]).numpy()
    assert np.abs(eng.adam_m.cpu().numpy() - mm).max() <= 1e-3 * np.abs(mm).max()
