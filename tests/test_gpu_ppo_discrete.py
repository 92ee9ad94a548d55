"""GPU parity of BASELINE.json configs[0] -- PPO on the CartPole-shape networks (obs 4, MLP[64, 64] trunk shared by a
discrete actor and a critic, minibatch 64) -- through the C ABI, against oracle/oracle_ppo_discrete.py (pinned to the
reference by tests/golden/ppo_discrete_*.npz) and against the golden files themselves."""
import numpy as np
import pytest
import torch

from oracle import oracle_ppo as OP
from oracle import oracle_ppo_cnn as OC
from oracle import oracle_ppo_discrete as OD
from tests.test_gpu_ppo_cnn import engine_cfg, rel_err
from tests.test_oracle_golden import load_ppo_discrete

pytestmark = pytest.mark.gpu


def make_engine(obs_dim, hidden, n_act, seed, cfg):
    from tianshou_amd import ppo_discrete as PD

    p = OD.init_params(obs_dim, hidden, n_act, seed)
    flat = PD.flat_from_torch([p[k] for k in OD.PARAM_ORDER], obs_dim, hidden, n_act)
    return p, PD.DiscretePPOEngine(obs_dim, hidden, n_act, flat, engine_cfg(cfg))


@pytest.mark.parametrize("obs_dim,hidden,A,B", [(4, 64, 2, 64), (4, 64, 2, 1), (33, 256, 31, 1000), (128, 32, 7, 65536)])
def test_layout_round_trip_and_inference(obs_dim, hidden, A, B):
    from tianshou_amd import ppo_discrete as PD

    p, eng = make_engine(obs_dim, hidden, A, 3, OP.PPOConfig())
    for a, k in zip(PD.flat_to_torch(eng.params, obs_dim, hidden, A), OD.PARAM_ORDER):
        assert torch.equal(a.cpu(), p[k]), k
    g = torch.Generator().manual_seed(B)
    obs = torch.randn(B, obs_dim, generator=g)
    act = torch.randint(0, A, (B,), generator=g)
    v, logp, logits = eng.infer(obs, act, True)
    net = OD.MlpNet(softmax_output=True)
    with torch.no_grad():
        lg_ref, v_ref = net.logits(p, obs), net.critic_forward(p, obs).flatten()
        lp_ref = net.dist(p, obs).log_prob(act)                 # Categorical(probs = softmax), as configs[0]
    assert rel_err(logits.cpu(), lg_ref) < 1e-5 and rel_err(v.cpu(), v_ref) < 1e-5
    np.testing.assert_allclose(logp.cpu().numpy(), lp_ref.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", ["c1", "opts", "a2c"])
def test_update_matches_reference_golden(tag):
    from tianshou_amd import ppo_discrete as PD
    from tianshou_amd.buffer import DeviceReplayBuffer

    g, d, cfg = load_ppo_discrete(tag)
    dims = (d["obs_dim"], d["hidden"], d["n_act"])
    _, eng = make_engine(*dims, d["seed"], cfg)
    buf = DeviceReplayBuffer(offset=g["buf_offset"], last_index=g["buf_last_index"], lengths=g["buf_lengths"],
                             insertion=g["buf_insertion"], rew=g["rew"], terminated=g["terminated"],
                             truncated=g["truncated"], obs=g["obs"], act=g["act"], obs_next=g["obs_next"])
    pre = eng.preprocess(buf)
    assert np.array_equal(pre["indices"].cpu().numpy(), g["pre_indices"])
    for k in ("v_s", "returns", "adv") + (("logp_old",) if cfg.algo == "ppo" else ()):
        np.testing.assert_allclose(pre[k].cpu().numpy(), g["pre_" + k], rtol=1e-5, atol=2e-5, err_msg=k)
    losses, steps = eng.update(buf, pre, d["batch_size"], d["repeat"], list(g["perms"]))
    assert steps == int(g["gradient_steps"])
    np.testing.assert_allclose(losses.cpu().numpy(), g["losses"], rtol=5e-5, atol=2e-6)
    flat = torch.cat([t.reshape(-1) for t in PD.flat_to_torch(eng.params, *dims)]).cpu().numpy()
    np.testing.assert_allclose(flat, g["params"], rtol=1e-5, atol=0.05 * cfg.lr)
    np.testing.assert_allclose(eng.ret_rms, g["ret_rms"], rtol=1e-6)


@pytest.mark.parametrize("obs_dim,hidden,A,B,adv_norm,dual,vclip", [(4, 64, 2, 64, False, None, False),
                                                                   (4, 64, 2, 65536, True, None, True),
                                                                   (17, 128, 6, 777, False, 3.0, False)])
def test_minibatch_gradient_vs_oracle(obs_dim, hidden, A, B, adv_norm, dual, vclip):
    """losses and every layer's gradient of one minibatch, from the configs[0] minibatch (64) to PPO's 65,536."""
    from tianshou_amd import ppo_discrete as PD

    g = torch.Generator().manual_seed(B + A)
    obs = torch.randn(B, obs_dim, generator=g)
    act = torch.randint(0, A, (B,), generator=g)
    adv, ret = torch.randn(B, generator=g), torch.randn(B, generator=g) * 2
    cfg = OP.PPOConfig(eps_clip=0.2, dual_clip=dual, value_clip=vclip, advantage_normalization=adv_norm, vf_coef=0.5,
                       ent_coef=0.01, max_grad_norm=0.5, lr=3e-4)
    p, eng = make_engine(obs_dim, hidden, A, 6, cfg)
    net = OD.MlpNet(softmax_output=False)
    with torch.no_grad():
        logp_old = net.dist(p, obs).log_prob(act) + torch.randn(B, generator=g) * 0.2
        v_old = net.critic_forward(p, obs).flatten() + torch.randn(B, generator=g) * 0.3
    pg = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss, clip, vf, ent = OC.minibatch_loss(pg, cfg, obs, act, adv, ret, logp_old, v_old, net=net)
    loss.backward()
    grad = torch.empty(eng.P, dtype=torch.float32, device="cuda")
    losses = eng.step(obs, act, adv, ret, logp_old, v_old, grad_out=grad, apply=False)
    np.testing.assert_allclose(losses.cpu().numpy(), [loss.item(), clip.item(), vf.item(), ent.item()], rtol=2e-5,
                               atol=1e-6)
    got = PD.flat_to_torch(grad, obs_dim, hidden, A)
    for t, k in zip(got, OD.PARAM_ORDER):
        assert rel_err(t.cpu(), pg[k].grad) < 2e-5, k
    lay = PD.layout(obs_dim, hidden, A)
    l1 = grad[:(lay["k0"] + 1) * hidden].reshape(lay["k0"] + 1, hidden)
    assert torch.count_nonzero(l1[obs_dim:lay["k0"]]) == 0               # padding rows get exactly zero gradient
    # the optimizer step on the same minibatch (clip_grad_norm_ + Adam)
    st = OP.PPOState(params={k: v.clone() for k, v in p.items()})
    OC._clip_adam(st, cfg, {k: pg[k].grad for k in OD.PARAM_ORDER})
    eng.step(obs, act, adv, ret, logp_old, v_old)
    new = torch.cat([t.reshape(-1) for t in PD.flat_to_torch(eng.params, obs_dim, hidden, A)]).cpu().numpy()
    np.testing.assert_allclose(new, OD.flatten_params(st.params).numpy(), rtol=1e-5, atol=0.05 * cfg.lr)


def test_bad_arguments_fail_loudly():
    from tianshou_amd import ppo_discrete as PD
    from tianshou_amd.ppo import PPOConfig

    with pytest.raises(Exception):
        PD.layout(4, 48, 2)
    with pytest.raises(Exception):
        PD.layout(4, 64, 32)
    n = PD.layout(4, 64, 2)["count"]
    with pytest.raises(RuntimeError):
        PD.DiscretePPOEngine(4, 64, 2, torch.zeros(n), PPOConfig())
    _, eng = make_engine(4, 64, 2, 0, OP.PPOConfig())
    z = torch.zeros(8)
    with pytest.raises(ValueError):
        eng.step(torch.zeros(8, 4), torch.zeros(7, dtype=torch.int64), z, z, z, z)


@pytest.mark.parametrize("obs_dim,A,n,batch,adv_norm,dual,vclip,algo", [
    (4, 2, 2000, 64, False, None, False, "ppo"),       # configs[0]: the merged last minibatch has 80 rows = two chunks
    (17, 6, 1500, 150, True, 3.0, True, "ppo"),        # three-chunk minibatches, per-minibatch advantage statistics
    (32, 31, 500, 64, False, None, False, "a2c"),      # widest observation / head the one-launch kernel takes
    (3, 1, 70, 64, True, None, True, "ppo")])          # a single merged minibatch of 70 rows, one action
def test_one_launch_update_equals_the_per_step_path_and_the_oracle(obs_dim, A, n, batch, adv_norm, dual, vclip, algo, monkeypatch):
    """ts_mlp_ppo_update (ts_mlp_small.hip: every minibatch of every repeat + clip + Adam in one persistent workgroup)
    against the per-step entry point (ts_mlp_ppo_step per minibatch, the path for networks outside the small envelope) and
    against the CPU oracle's update(): losses of every gradient step, final parameters and Adam moments."""
    from tianshou_amd import ppo_discrete as PD
    from tianshou_amd.buffer import DeviceReplayBuffer

    hidden, repeat = 64, 3
    cfg = OP.PPOConfig(eps_clip=0.2, dual_clip=dual, value_clip=vclip, advantage_normalization=adv_norm, vf_coef=0.5,
                       ent_coef=0.01, max_grad_norm=0.5, lr=3e-4, algo=algo)
    rng = np.random.default_rng(n + obs_dim)
    obs = rng.normal(size=(n, obs_dim)).astype(np.float32)
    act = rng.integers(0, A, size=n)
    pre_np = {"adv": rng.normal(size=n).astype(np.float32), "returns": (rng.normal(size=n) * 2).astype(np.float32),
              "logp_old": (-np.log(A) + 0.1 * rng.normal(size=n)).astype(np.float32) if A > 1 else (0.05 * rng.normal(size=n)).astype(np.float32),
              "v_s": rng.normal(size=n).astype(np.float32)}
    perms = [rng.permutation(n) for _ in range(repeat)]
    # oracle
    p0 = OD.init_params(obs_dim, hidden, A, 5)
    st = OP.PPOState(params={k: v.clone() for k, v in p0.items()})
    pre_t = {k: torch.as_tensor(v) for k, v in pre_np.items()}
    losses_o = OC.update(st, cfg, obs, act, pre_t, batch, repeat, perms, net=OD.MlpNet(softmax_output=True))
    # the two GPU paths
    buf = DeviceReplayBuffer.from_vector_fill(1, rew=np.zeros(n), terminated=np.zeros(n, bool), truncated=np.zeros(n, bool),
                                              obs=obs, act=act, obs_next=obs)
    out = {}
    for mode in ("one_launch", "per_step"):
        if mode == "per_step":
            monkeypatch.setenv("TS_MLP_PPO_PER_STEP", "1")
        _, eng = make_engine(obs_dim, hidden, A, 5, cfg)
        pre = {k: torch.as_tensor(v).cuda() for k, v in pre_np.items()}
        pre["indices"] = torch.arange(n, device="cuda")
        pre["act"] = torch.as_tensor(act).cuda()
        losses, steps = eng.update(buf, pre, batch, repeat, perms)
        torch.cuda.synchronize()
        out[mode] = (losses.cpu().numpy(), eng.params.cpu().numpy(), eng.adam_m.cpu().numpy(), eng.adam_v.cpu().numpy(), steps,
                     eng.adam_step)
    monkeypatch.delenv("TS_MLP_PPO_PER_STEP")
    a, b = out["one_launch"], out["per_step"]
    assert a[4] == b[4] == losses_o.shape[0] and a[5] == b[5] == a[4]
    np.testing.assert_allclose(a[0], np.asarray(losses_o), rtol=5e-5, atol=2e-6)
    np.testing.assert_allclose(a[0], b[0], rtol=5e-5, atol=2e-6)
    flat_o = PD.flat_from_torch([st.params[k] for k in OD.PARAM_ORDER], obs_dim, hidden, A, device="cpu").numpy()
    np.testing.assert_allclose(a[1], flat_o, rtol=1e-5, atol=0.05 * cfg.lr)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-5, atol=0.05 * cfg.lr)
    assert rel_err(a[2], b[2]) < 1e-4 and rel_err(a[3], b[3]) < 1e-4


@pytest.mark.parametrize("per_step", [False, True])
def test_recompute_advantage_matches_oracle(per_step, monkeypatch):
    """recompute_advantage=True (ppo.py:174-178; the default of examples/mujoco/mujoco_ppo.py:56): V(s), V(s'), GAE and the
    return scaling -- with one more RunningMeanStd update per repeat -- are redone with the current parameters before
    every repeat after the first.  One-launch path (one launch per repeat) and per-step path against the oracle."""
    from tianshou_amd import ppo_discrete as PD
    from tianshou_amd.buffer import DeviceReplayBuffer

    if per_step:
        monkeypatch.setenv("TS_MLP_PPO_PER_STEP", "1")
    obs_dim, hidden, A, n_env, T, batch, repeat = 6, 64, 3, 4, 60, 64, 3
    n = n_env * T
    cfg = OP.PPOConfig(eps_clip=0.2, value_clip=True, advantage_normalization=True, recompute_advantage=True, vf_coef=0.5,
                       ent_coef=0.01, max_grad_norm=0.5, return_scaling=True, lr=1e-3, max_batchsize=4096)
    rng = np.random.default_rng(12)
    obs = rng.normal(size=(n, obs_dim)).astype(np.float32)
    obs_next = rng.normal(size=(n, obs_dim)).astype(np.float32)
    act = rng.integers(0, A, size=n)
    rew = rng.normal(size=n)
    term = rng.random(n) < 0.05
    trunc = np.zeros(n, bool)
    perms = [rng.permutation(n) for _ in range(repeat)]
    p0 = OD.init_params(obs_dim, hidden, A, 9)
    net = OD.MlpNet(softmax_output=True)
    st = OP.PPOState(params={k: v.clone() for k, v in p0.items()})
    idx, unf = np.arange(n), np.arange(n_env) * T + T - 1
    o_args = (obs, obs_next, act, rew, term, trunc, idx, unf)
    pre_o = OC.preprocess(st, cfg, *o_args, net=net)
    losses_o = OC.update(st, cfg, obs, act, pre_o, batch, repeat, perms, net=net,
                         recompute=lambda: OC.preprocess(st, cfg, *o_args, net=net))
    _, eng = make_engine(obs_dim, hidden, A, 9, cfg)
    buf = DeviceReplayBuffer.from_vector_fill(n_env, rew=rew, terminated=term, truncated=trunc, obs=obs, act=act, obs_next=obs_next)
    pre = eng.preprocess(buf)
    losses, steps = eng.update(buf, pre, batch, repeat, perms)
    assert steps == losses_o.shape[0] == eng.adam_step
    np.testing.assert_allclose(losses.cpu().numpy(), np.asarray(losses_o), rtol=5e-5, atol=3e-6)
    flat_o = PD.flat_from_torch([st.params[k] for k in OD.PARAM_ORDER], obs_dim, hidden, A, device="cpu").numpy()
    np.testing.assert_allclose(eng.params.cpu().numpy(), flat_o, rtol=1e-5, atol=0.05 * cfg.lr)
    np.testing.assert_allclose(eng.ret_rms, [st.ret_rms.mean, st.ret_rms.var, st.ret_rms.count], rtol=1e-5)
