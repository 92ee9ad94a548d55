"""GPU parity: PPO / A2C on the GEMM path (tianshou_amd/ppo_wide.py, ts_ppo_wide_step) for actor-critic MLPs outside the
fused kernels' envelope - Net[h, h] with h in {32 .. 1024}, any obs_dim, act_dim <= 32 (utils/net/common.py:90-178) -
against the torch-fp32 CPU oracle (oracle/oracle_ppo.py with `hidden`), same bars as tests/test_gpu_ppo.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle import oracle_ppo as OP

pytestmark = pytest.mark.gpu


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    return t if dtype is None else t.to(dtype)


def problem(n, obs_dim, act_dim, hidden, seed):
    rng = np.random.default_rng(seed)
    params = OP.init_params(obs_dim, act_dim, hidden=hidden, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in params:
        params[k] = params[k] + 0.03 * torch.randn(params[k].shape, generator=g)
    data = dict(obs=rng.normal(size=(n, obs_dim)).astype(np.float32), obs_next=rng.normal(size=(n, obs_dim)).astype(np.float32),
                act=rng.normal(size=(n, act_dim)).astype(np.float32), rew=rng.normal(size=n).astype(np.float32).astype(np.float64),
                terminated=rng.random(n) < 0.02, truncated=np.zeros(n, bool))
    return params, data


def engine_flat(params, obs_dim, act_dim, hidden):
    from tianshou_amd.ppo_wide import flat_from_tensors

    a = [params[k] for k in OP.PARAM_ORDER[:7]]
    c = [params[k] for k in OP.PARAM_ORDER[7:]]
    return flat_from_tensors(a, c, obs_dim, hidden, act_dim)


CFGS = {
    "mujoco": dict(eps_clip=0.2, vf_coef=0.25, ent_coef=0.0, max_grad_norm=0.5, value_clip=True, advantage_normalization=False,
                   return_scaling=True, lr=3e-4),
    "defaults": dict(dual_clip=3.0, recompute_advantage=True, lr=1e-3),
    "a2c": dict(algo="a2c", vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5, return_scaling=True, lr=7e-4, advantage_normalization=False),
}


@pytest.mark.parametrize("obs_dim,act_dim,hidden,cfg_name", [(376, 17, 256, "mujoco"), (376, 17, 256, "defaults"), (40, 10, 128, "defaults"),
                                                             (17, 6, 64, "mujoco"), (5, 32, 32, "a2c"), (111, 8, 512, "a2c")])
def test_wide_update_matches_oracle(obs_dim, act_dim, hidden, cfg_name):
    from tianshou_amd import ppo as P
    from tianshou_amd.ppo_wide import WidePPOEngine, flat_to_tensors

    n, n_env, batch_size, repeat = 600, 4, 160, 2
    params, data = problem(n, obs_dim, act_dim, hidden, seed=obs_dim + hidden)
    kw = CFGS[cfg_name]
    ocfg, cfg = OP.PPOConfig(max_batchsize=4096, **kw), P.PPOConfig(**kw)
    rng = np.random.default_rng(3)
    perms = [rng.permutation(n) for _ in range(repeat)]
    torch.set_num_threads(8)
    st = OP.PPOState(params={k: v.clone() for k, v in params.items()})
    bs = O.BufferState.from_vector_fill(data["rew"], data["terminated"], data["truncated"], n_env)
    idx, unf = bs.sample_indices_all(), bs.unfinished_index()
    args = (torch.from_numpy(data["obs"]), torch.from_numpy(data["obs_next"]), torch.from_numpy(data["act"]), data["rew"],
            data["terminated"], data["truncated"], idx, unf)
    pre_o = OP.preprocess(st, ocfg, *args)

    def recompute():
        return OP.add_returns_and_advantages(st, ocfg, args[0], args[1], *args[3:])

    losses_o, grads_o = OP.update(st, ocfg, {"obs": args[0], "act": args[2]}, pre_o, batch_size, repeat, perms,
                                  recompute=recompute, collect_grads=True)
    eng = WidePPOEngine(obs_dim, act_dim, hidden, engine_flat(params, obs_dim, act_dim, hidden), cfg)
    b = eng.preprocess(dev(data["obs"]), dev(data["obs_next"]), dev(data["act"]), dev(data["rew"]), dev(data["terminated"]),
                       dev(data["truncated"]), dev(unf))
    np.testing.assert_allclose(b["v_s"].cpu().numpy(), pre_o["v_s"].numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(b["adv"].cpu().numpy(), pre_o["adv"].numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(b["returns"].cpu().numpy(), pre_o["returns"].numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(b["logp_old"].cpu().numpy(), pre_o["logp_old"].numpy(), rtol=1e-5, atol=1e-4 if act_dim > 16 else 1e-5)
    losses, steps, grads = eng.update(b, batch_size, repeat, perms, want_grad=True)
    assert steps == losses_o.shape[0] and eng.adam_step == st.adam_step
    np.testing.assert_allclose(losses.cpu().numpy(), losses_o, rtol=1e-5, atol=3e-6)
    ga, gc = flat_to_tensors(grads, obs_dim, hidden, act_dim)
    shapes = OP.param_shapes(obs_dim, act_dim, hidden)
    off = 0
    gscale = float(grads_o.abs().max())
    for k, t in zip(OP.PARAM_ORDER, ga + gc):
        cnt = int(np.prod(shapes[k]))
        np.testing.assert_allclose(t.cpu().numpy().reshape(-1), grads_o[off:off + cnt].numpy(), rtol=1e-4, atol=5e-6 * max(gscale, 1.0),
                                   err_msg="grad " + k)
        off += cnt
    pa, pc = flat_to_tensors(eng.params, obs_dim, hidden, act_dim)
    for k, t in zip(OP.PARAM_ORDER, pa + pc):
        # post-Adam parameters on the scale of one step: with no gradient clipping and 65,536-element layers a handful of
        # elements whose gradient is rounding noise (|g| ~ 1e-9) get m / sqrt(v) of either sign in the first steps
        got, want = t.cpu().numpy().reshape(-1), st.params[k].numpy().reshape(-1)
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=0.1 * cfg.lr, err_msg=k)
        assert np.mean(np.abs(got - want) > 0.02 * cfg.lr + 1e-4 * np.abs(want)) < 1e-4, k
    np.testing.assert_allclose(eng.ret_rms, [st.ret_rms.mean, st.ret_rms.var, st.ret_rms.count], rtol=1e-5)


def test_wide_engine_rejects_unsupported_shapes():
    from tianshou_amd import ppo as P
    from tianshou_amd.ppo_wide import WidePPOEngine

    with pytest.raises(NotImplementedError):
        WidePPOEngine(17, 6, 100, torch.zeros(10, device="cuda"), P.PPOConfig())
    with pytest.raises(NotImplementedError):
        WidePPOEngine(17, 33, 64, torch.zeros(10, device="cuda"), P.PPOConfig())


@pytest.mark.parametrize("obs_dim,act_dim,hidden,cfg_name", [(376, 17, 256, "mujoco"), (40, 10, 128, "defaults")])
def test_wide_data_parallel_wrapper_at_world_one_is_bit_identical(obs_dim, act_dim, hidden, cfg_name):
    """DataParallelWidePPO (ts_ppo_wide_step in gradient-only mode -> exchange -> ts_adam_step) with one rank and no
    exchange == WidePPOEngine.update (ts_ppo_wide_step with its own clip + Adam): the same kernels on the same buffers,
    bit for bit -- losses, parameters, Adam moments, ret_rms -- incl. recompute_advantage and advantage normalisation with
    the wrapper's (global) minibatch statistics.  The multi-rank behaviour of the wrapper's base class is covered on CPU
    (tests/test_dp_gloo.py)."""
    from tianshou_amd import ppo as P
    from tianshou_amd.distributed import DataParallelWidePPO
    from tianshou_amd.ppo_wide import WidePPOEngine

    n, n_env, batch_size, repeat = 600, 4, 160, 2
    params, data = problem(n, obs_dim, act_dim, hidden, seed=7)
    cfg = P.PPOConfig(**CFGS[cfg_name])
    rng = np.random.default_rng(5)
    perms = [rng.permutation(n) for _ in range(repeat)]
    unf = (np.arange(n_env) + 1) * (n // n_env) - 1
    out = []
    for wrap in (False, True):
        eng = WidePPOEngine(obs_dim, act_dim, hidden, engine_flat(params, obs_dim, act_dim, hidden), cfg)
        runner = DataParallelWidePPO(eng) if wrap else eng
        b = runner.preprocess(dev(data["obs"]), dev(data["obs_next"]), dev(data["act"]), dev(data["rew"]), dev(data["terminated"]),
                              dev(data["truncated"]), dev(unf))
        losses, steps = runner.update(b, batch_size, repeat, perms)
        torch.cuda.synchronize()
        out.append((losses.cpu(), eng.params.cpu().clone(), eng.adam_m.cpu().clone(), eng.adam_v.cpu().clone(), steps, eng.adam_step,
                    list(eng.ret_rms)))
    a, w = out
    assert a[4] == w[4] and a[5] == w[5] and a[6] == w[6]
    for x, y in zip(a[1:4], w[1:4]):
        if cfg.advantage_normalization:      # the wrapper's statistics come from sums (global_adv_stats), the engine's from
            np.testing.assert_allclose(y.numpy(), x.numpy(), rtol=1e-5, atol=1e-7)      # torch.std(): last-bit differences
        else:
            assert torch.equal(x, y)
    # the wrapper recomposes loss = clip + vf_coef * vf - ent_coef * ent on the host side of the exchange buffer
    np.testing.assert_allclose(w[0].numpy(), a[0].numpy(), rtol=1e-5, atol=1e-6)
