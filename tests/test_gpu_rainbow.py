"""GPU parity of the Rainbow row: NoisyLinear layers, dueling heads, C51 projection / cross entropy --
through the C ABI, against the oracle (oracle/oracle_rainbow.py, pinned to the reference by tests/golden/rainbow_*.npz)
and against the golden files themselves.  Tolerance 1e-5 relative on each tensor's scale."""
import numpy as np
import pytest
import torch

from oracle import oracle_distq as OQ
from oracle import oracle_rainbow as ORB
from tests import dqn_common as DC

pytestmark = pytest.mark.gpu
NOISE_ORDER = [f"{L}.{t}" for L in ORB.NOISY for t in ("eps_p", "eps_q")]


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def dev_noise(noise: dict, dims):
    from tianshou_amd import rainbow as RB

    return RB.noise_from_torch([noise[k] for k in NOISE_ORDER], *dims)


def make_engine(c, h, w, A, N, seed, **kw):
    from tianshou_amd import distq as Q
    from tianshou_amd import rainbow as RB

    p, n = ORB.init_params(c, h, w, A, N, seed)
    cfg = Q.DistQConfig(kind="c51", n_atoms=N, **kw)
    eng = RB.RainbowEngine(c, h, w, A, RB.flat_from_torch([p[k] for k in ORB.PARAM_ORDER], c, h, w, A, N),
                           dev_noise(n, (c, h, w, A, N)), cfg)
    return p, n, eng


@pytest.mark.parametrize("c,h,w,A,N", [(4, 84, 84, 6, 51), (2, 44, 36, 3, 7)])
def test_layout_round_trip_and_forward_vs_oracle(c, h, w, A, N):
    from tianshou_amd import rainbow as RB

    p, n, eng = make_engine(c, h, w, A, N, seed=3, v_min=-4.0, v_max=9.0)
    for a, k in zip(RB.flat_to_torch(eng.params, c, h, w, A, N), ORB.PARAM_ORDER):
        assert torch.equal(a.cpu(), p[k]), k
    ocfg = OQ.DistQConfig(kind="c51", n_atoms=N, v_min=-4.0, v_max=9.0)
    obs = np.random.default_rng(1).integers(0, 256, size=(29, c, h, w), dtype=np.uint8)
    x = torch.as_tensor(obs).permute(0, 2, 3, 1).contiguous().cuda()
    for training in (True, False):
        d_ref = ORB.dist(p, n if training else None, obs, A, N)
        q_ref = (d_ref * OQ.support(ocfg)).sum(2)
        dist, q, act = eng.forward(x, training=training)
        assert rel_err(dist.cpu(), d_ref) < 1e-5 and rel_err(q.cpu(), q_ref) < 1e-5
        assert torch.equal(act.cpu(), q_ref.argmax(dim=1))
    assert rel_err(ORB.dist(p, n, obs, A, N), ORB.dist(p, None, obs, A, N)) > 1e-3          # the noise matters


@pytest.mark.parametrize("weighted", [True, False])
def test_batch_gradient_vs_oracle(weighted):
    """loss, priorities, projected target and the gradient of every tensor (conv, mu and sigma of the four noisy layers) of
    one minibatch with a lagged, separately-noised network, then the Adam step."""
    from tianshou_amd import rainbow as RB

    c, h, w, A, N, B = 4, 84, 84, 6, 51, 48
    kw = dict(v_min=-3.0, v_max=5.0, lr=1e-4, target_update_freq=7)
    p, n, eng = make_engine(c, h, w, A, N, seed=4, **kw)
    ocfg = OQ.DistQConfig(kind="c51", n_atoms=N, **kw)
    st = ORB.RainbowState(p, n, ocfg)
    g = torch.Generator().manual_seed(0)
    st.dqn.params_old = {k: v + 0.01 * torch.randn(v.shape, generator=g) for k, v in p.items()}
    st.dqn.iter = eng.iter = 1                                   # no sync at this update
    torch.manual_seed(5)
    noise, noise_old = ORB.sample_noise(h, w, A, N), ORB.sample_noise(h, w, A, N)
    eng.params_old = RB.flat_from_torch([st.dqn.params_old[k] for k in ORB.PARAM_ORDER], c, h, w, A, N)
    eng.set_noise(dev_noise(noise, (c, h, w, A, N)), dev_noise(noise_old, (c, h, w, A, N)))
    rng = np.random.default_rng(9)
    obs = rng.integers(0, 256, size=(B, c, h, w), dtype=np.uint8)
    obs_next = rng.integers(0, 256, size=(B, c, h, w), dtype=np.uint8)
    act = rng.integers(0, A, size=B)
    ret = (rng.normal(size=(B, N)) * 2.5).astype(np.float32)
    weight = rng.random(B).astype(np.float32) if weighted else None
    col: dict = {}
    loss_ref, prio_ref = ORB.update_with_batch(st, ocfg, obs, act, ret, obs_next, A, noise, noise_old, weight=weight, collect=col)
    to_dev = lambda a: torch.as_tensor(a).permute(0, 2, 3, 1).contiguous().cuda()
    grad = torch.empty(eng.P, dtype=torch.float32, device="cuda")
    loss, prio, tgt = eng.update_with_batch(to_dev(obs), act, ret, to_dev(obs_next), weight, grad_out=grad, apply=False,
                                            want_target=True)
    assert abs(float(loss) - loss_ref) <= 1e-5 * abs(loss_ref)
    assert rel_err(prio.cpu(), prio_ref) < 1e-5 and rel_err(tgt.cpu(), col["target_dist"]) < 1e-5
    for t, k in zip(RB.flat_to_torch(grad, c, h, w, A, N), ORB.PARAM_ORDER):
        assert rel_err(t.cpu(), col["grads"][k]) < 2e-5, k
    eng.update_with_batch(to_dev(obs), act, ret, to_dev(obs_next), weight)
    new = torch.cat([t.reshape(-1) for t in RB.flat_to_torch(eng.params, c, h, w, A, N)]).cpu().numpy()
    ref = ORB.flatten_params(st.dqn.params).numpy()
    bad = np.abs(new - ref) > 1e-5 * np.abs(ref) + 0.02 * ocfg.lr
    assert bad.mean() < 1e-4 and np.abs(new - ref).max() <= 2 * ocfg.lr


@pytest.mark.parametrize("tag", ["lagged", "single"])
def test_update_sequence_matches_reference_golden(tag):
    from tianshou_amd import dqn as D
    from tianshou_amd import rainbow as RB
    from tianshou_amd.buffer import DeviceReplayBuffer

    g, d, ocfg, bstate = DC.load_rainbow(tag)
    c, h, w, A, N = d["c"], d["h"], d["w"], d["n_act"], d["n_atoms"]
    dims = (c, h, w, A, N)
    p, n, eng = make_engine(*dims, seed=d["seed"], v_min=ocfg.v_min, v_max=ocfg.v_max, gamma=ocfg.gamma, n_step=ocfg.n_step,
                            target_update_freq=ocfg.target_update_freq, lr=ocfg.lr)
    buf = DeviceReplayBuffer(offset=g["buf_offset"], last_index=g["buf_last_index"], lengths=g["buf_lengths"],
                             insertion=g["buf_insertion"], rew=g["rew"], terminated=g["terminated"], truncated=g["truncated"])
    frames, frames_next = torch.as_tensor(g["frames"]).cuda(), torch.as_tensor(g["frames_next"]).cuda()
    act_all = torch.as_tensor(g["act"]).cuda()
    for u in range(d["n_updates"]):
        idx = torch.as_tensor(g[f"u{u}_indices"]).cuda()
        ret = eng.preprocess(buf, idx)
        np.testing.assert_allclose(ret.cpu().numpy(), g[f"u{u}_returns"], rtol=1e-5, atol=1e-5)
        old = DC.rainbow_noise(g, u, old=True)
        eng.set_noise(dev_noise(DC.rainbow_noise(g, u), dims), None if old is None else dev_noise(old, dims))
        obs = D.gather_obs_nhwc(frames, buf, idx, 1, as_u8=True)
        obs_next = D.gather_obs_nhwc(frames_next, buf, idx, 1, as_u8=True)
        loss, prio = eng.update_with_batch(obs, act_all[idx], ret, obs_next, g[f"u{u}_is_weight"])
        np.testing.assert_allclose(prio.cpu().numpy(), g[f"u{u}_prio"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(float(loss), float(g[f"u{u}_loss"]), rtol=1e-5)
        tensors = RB.flat_to_torch(eng.params, *dims)
        flat = torch.cat([t.reshape(-1) for t in tensors]).cpu().numpy()
        np.testing.assert_allclose(flat[::97], g[f"u{u}_params_strided"], rtol=1e-5, atol=0.02 * ocfg.lr)
        by = dict(zip(ORB.PARAM_ORDER, tensors))
        np.testing.assert_allclose(by["conv1.w"].cpu().numpy(), g[f"u{u}_conv1_w"], rtol=1e-5, atol=0.02 * ocfg.lr)
        np.testing.assert_allclose(by["V2.sigma_W"].cpu().numpy(), g[f"u{u}_V2_sigma_W"], rtol=1e-5, atol=0.02 * ocfg.lr)
        np.testing.assert_allclose(by["Q2.mu_b"].cpu().numpy(), g[f"u{u}_Q2_mu_b"], rtol=1e-5, atol=0.02 * ocfg.lr)


def test_argument_errors():
    from tianshou_amd import rainbow as RB

    p, n, eng = make_engine(2, 44, 36, 3, 7, seed=0)
    x = torch.zeros((5, 44, 36, 2), dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):                       # returns must be [B, n_atoms]
        eng.update_with_batch(x, np.zeros(5, np.int64), np.zeros(5, np.float32), x)
    with pytest.raises(ValueError):
        eng.forward(torch.zeros((5, 36, 44, 2), dtype=torch.uint8, device="cuda"))
    with pytest.raises(RuntimeError):
        RB.RainbowEngine(2, 44, 36, 3, eng.params.cpu(), eng.noise.cpu(), eng.cfg)
    with pytest.raises(ValueError):
        RB.RainbowEngine(2, 44, 36, 3, eng.params[:-1], eng.noise, eng.cfg)


def test_replay_stream_cycle_equals_the_sequential_cycle():
    """Rainbow's update with its backward pass spread over the workspace's streams, in a dqn.ReplayStream cycle (priority
    update, next batch, support returns and both noise draws on the replay stream) against the sequential order on one
    stream: five updates, identical indices, returns, losses, priorities, parameters and sum tree."""
    import bench_dqn as BD
    from tianshou_amd import distq as Q
    from tianshou_amd import dqn as D

    def cycle(use_stream: bool):
        frames, act, buf, per = BD.build(4096, 4, seed=3)
        _, _, eng = make_engine(4, 84, 84, 6, 51, seed=2, gamma=0.99, n_step=3, target_update_freq=2, lr=1e-4)
        gen = torch.Generator(device="cuda").manual_seed(11)
        nn = eng.lay["noise_count"]

        def noise():
            x = torch.randn(nn, generator=gen, device="cuda")
            return x.sign() * x.abs().sqrt()

        draw = lambda: torch.rand(64, generator=gen, device="cuda", dtype=torch.float64)  # noqa: E731
        base = Q.replay_prepare(eng, buf, frames, 4)
        rs = (D.ReplayStream(eng, buf, frames, per, 4, draw, lambda i: act[i], prepare=lambda i: base(i) + (noise(), noise()))
              if use_stream else None)
        log = []
        for _ in range(5):
            if rs is None:
                idx, wt = per.sample(draw())
                ret = eng.preprocess(buf, idx)
                eng.set_noise(noise(), noise())
                obs = D.gather_obs_nhwc(frames, buf, idx, 4, as_u8=True)
                obs_next = D.gather_obs_nhwc(frames, buf, buf.next(idx), 4, as_u8=True)
                a = act[idx]
            else:
                idx, wt, a, obs, obs_next, ret, n1, n2 = rs.take()
                eng.set_noise(n1, n2)
            loss, prio = eng.update_with_batch(obs, a, ret, obs_next, wt)
            if rs is None:
                per.update_weight(idx, prio)
            else:
                rs.give(idx, prio)
            log.append((idx.clone(), wt.float(), ret.clone(), loss.clone(), prio.clone()))
        torch.cuda.synchronize()
        return log, eng.params.clone(), per.weight._value.clone(), per.prio_minmax.clone()

    a, b = cycle(False), cycle(True)
    for it, (x, y) in enumerate(zip(a[0], b[0])):
        for u, v in zip(x, y):
            assert torch.equal(u, v), it
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
