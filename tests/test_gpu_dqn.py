"""GPU parity of the DQN row: conv / linear layers on fp32 MFMA, frame-stack gather,
double-Q n-step target, Huber / MSE TD loss, Adam -- through the C ABI, against the oracle
(oracle/oracle_dqn.py, pinned to the reference by tests/golden/dqn_*.npz).
Tolerance: 1e-5 relative (north_star), on the scale of each tensor."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O
from oracle import oracle_dqn as OD
from tests import dqn_common as DC

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def to_wb(w, b):
    """torch conv weight [oc, ic, kh, kw] + bias -> engine matrix [(kh, kw, ic) + 1, oc]."""
    return torch.cat([w.permute(2, 3, 1, 0).reshape(-1, w.shape[0]), b[None, :]]).contiguous()


LAYERS = [  # (B, IH, IW, IC, K, S, OC, relu-masked input)
    (5, 84, 84, 4, 8, 4, 32, False),       # conv1 (no dx needed, input is the observation)
    (3, 44, 36, 2, 8, 4, 32, False),       # conv1 of the small golden net
    (5, 20, 20, 32, 4, 2, 64, True),       # conv2
    (7, 10, 8, 32, 4, 2, 64, True),        # conv2, rectangular
    (5, 9, 9, 64, 3, 1, 64, True),         # conv3
    (37, 1, 1, 3136, 1, 1, 512, True),     # fc1
    (512, 1, 1, 128, 1, 1, 512, True),     # fc1 of the small net, full batch
]


@pytest.mark.parametrize("shape", LAYERS)
def test_layer_forward_backward_vs_torch(shape):
    from tianshou_amd import dqn as D

    B, IH, IW, IC, K, S, OC, masked = shape
    g = torch.Generator().manual_seed(B * 1000 + IH)
    x = torch.randn(B, IC, IH, IW, generator=g)
    if masked:
        x = F.relu(x)                                    # a ReLU output: zeros carry the mask
    w = (torch.randn(OC, IC, K, K, generator=g) / np.sqrt(IC * K * K)).requires_grad_(True)
    b = torch.randn(OC, generator=g).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y = F.relu(F.conv2d(xr, w, b, stride=S))
    dy = torch.randn(y.shape, generator=g) * (y > 0)     # gradient w.r.t. the pre-activation
    y.backward(dy)

    dev = "cuda"
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wb = to_wb(w.detach(), b.detach()).to(dev)
    y_gpu = D.conv_forward(x_nhwc, wb, K, K, S, relu=True)
    assert rel_err(y_gpu.permute(0, 3, 1, 2).cpu(), y.detach()) < 1e-5
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    need_dx = IC % 32 == 0
    d_wb, dx = D.conv_backward(x_nhwc, wb, dy_nhwc, K, K, S, mask=x_nhwc if masked else None, need_dx=need_dx)
    assert rel_err(d_wb.cpu(), to_wb(w.grad, b.grad)) < 1e-5
    if need_dx:
        ref_dx = xr.grad * (x > 0) if masked else xr.grad
        assert rel_err(dx.permute(0, 3, 1, 2).cpu(), ref_dx) < 1e-5


def test_layout_round_trip_and_forward_matches_oracle():
    from tianshou_amd import dqn as D

    c, h, w, A = 4, 84, 84, 6
    p = OD.init_params(c, h, w, A, seed=11)
    tensors = [p[k] for k in OD.PARAM_ORDER]
    flat = D.flat_from_torch(tensors, c, h, w, A)
    assert flat.numel() == OD.param_count(c, h, w, A) == D.param_count(c, h, w, A) == 1_687_206
    back = D.flat_to_torch(flat, c, h, w, A)
    for a, b in zip(back, tensors):
        assert torch.equal(a.cpu(), b)
    rng = np.random.default_rng(0)
    obs = rng.integers(0, 256, size=(9, c, h, w), dtype=np.uint8)
    q_ref = OD.forward(p, obs)
    eng = D.DQNEngine(c, h, w, A, flat, D.DQNConfig())
    obs_nhwc = torch.as_tensor(obs).permute(0, 2, 3, 1).float().contiguous().cuda()
    q, act = eng.forward(obs_nhwc)
    assert rel_err(q.cpu(), q_ref) < 1e-5
    assert torch.equal(act.cpu(), q_ref.argmax(dim=1))


@pytest.mark.parametrize("tag", ["atari", "small"])
def test_frame_stack_gather_bit_exact(tag):
    from tianshou_amd import dqn as D
    from tianshou_amd.buffer import DeviceReplayBuffer

    g, d, cfg, bstate = DC.load(tag)
    stack_num = d["c"] if d["stack"] else 1
    buf = DeviceReplayBuffer(offset=g["buf_offset"], last_index=g["buf_last_index"], lengths=g["buf_lengths"],
                             insertion=g["buf_insertion"], rew=g["rew"], terminated=g["terminated"],
                             truncated=g["truncated"])
    frames = torch.as_tensor(g["frames"]).cuda()
    idx = np.arange(d["E"] * d["slots"])
    ref = OD.stacked_frames(bstate, g["frames"], idx, stack_num)          # [I, C, H, W] uint8
    out = D.gather_obs_nhwc(frames, buf, idx, stack_num)
    assert torch.equal(out.cpu(), torch.as_tensor(ref).permute(0, 2, 3, 1).float())
    out8 = D.gather_obs_nhwc(frames, buf, idx, stack_num, as_u8=True)
    assert out8.dtype == torch.uint8 and torch.equal(out8.cpu(), torch.as_tensor(ref).permute(0, 2, 3, 1))
    if stack_num > 1:
        st = D.stack_indices(buf, idx, stack_num).cpu().numpy()
        cur = idx.copy()
        for j in range(stack_num):
            assert np.array_equal(st[:, stack_num - 1 - j], cur)
            cur = bstate.prev(cur)


@pytest.mark.parametrize("tag", ["atari", "small"])
def test_dqn_update_matches_reference_golden(tag):
    """Replays the reference's DQN.update() sequence (sampled indices from the fixture) on the engine."""
    from tianshou_amd import dqn as D
    from tianshou_amd.buffer import DeviceReplayBuffer

    g, d, ocfg, bstate = DC.load(tag)
    c, h, w, A = d["c"], d["h"], d["w"], d["n_act"]
    stack_num = c if d["stack"] else 1
    cfg = D.DQNConfig(gamma=ocfg.gamma, n_step=ocfg.n_step, target_update_freq=ocfg.target_update_freq,
                      is_double=ocfg.is_double, huber_delta=ocfg.huber_delta, lr=ocfg.lr)
    p0 = OD.init_params(c, h, w, A, d["seed"])
    eng = D.DQNEngine(c, h, w, A, D.flat_from_torch([p0[k] for k in OD.PARAM_ORDER], c, h, w, A), cfg)
    buf = DeviceReplayBuffer(offset=g["buf_offset"], last_index=g["buf_last_index"], lengths=g["buf_lengths"],
                             insertion=g["buf_insertion"], rew=g["rew"], terminated=g["terminated"],
                             truncated=g["truncated"])
    frames = torch.as_tensor(g["frames"]).cuda()
    act_all = torch.as_tensor(g["act"]).cuda()
    for u in range(d["n_updates"]):
        idx = torch.as_tensor(g[f"u{u}_indices"]).cuda()
        ret = eng.preprocess(buf, frames, idx, stack_num)
        np.testing.assert_allclose(ret.cpu().numpy(), g[f"u{u}_returns"], rtol=1e-5, atol=1e-5)
        obs = D.gather_obs_nhwc(frames, buf, idx, stack_num)
        loss, td = eng.update_with_batch(obs, act_all[idx], ret)
        np.testing.assert_allclose(td.cpu().numpy(), g[f"u{u}_td"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(float(loss), float(g[f"u{u}_loss"]), rtol=1e-5)
        tensors = D.flat_to_torch(eng.params, c, h, w, A)
        flat = torch.cat([t.reshape(-1) for t in tensors]).cpu().numpy()
        # Adam's first steps move every weight by ~lr whatever the gradient scale: compare on lr's scale
        np.testing.assert_allclose(flat[::61], g[f"u{u}_params_strided"], rtol=1e-5, atol=0.02 * cfg.lr)
        np.testing.assert_allclose(tensors[0].cpu().numpy(), g[f"u{u}_conv1_w"], rtol=1e-5, atol=0.02 * cfg.lr)
        np.testing.assert_allclose(tensors[8].cpu().numpy(), g[f"u{u}_fc2_w"], rtol=1e-5, atol=0.02 * cfg.lr)


@pytest.mark.parametrize("huber,weighted", [(1.0, False), (None, True), (None, False)])
def test_c3_batch_gradient_vs_oracle(huber, weighted):
    """Full C3 minibatch (512 x u8[4,84,84], 6 actions): loss, TD errors and the whole gradient."""
    from tianshou_amd import dqn as D

    c, h, w, A, B = 4, 84, 84, 6, 512
    rng = np.random.default_rng(7)
    obs = rng.integers(0, 256, size=(B, c, h, w), dtype=np.uint8)
    act = rng.integers(0, A, size=B)
    ret = rng.normal(size=B).astype(np.float32) * 3
    weight = rng.random(B).astype(np.float32) if weighted else None
    p = OD.init_params(c, h, w, A, seed=2)
    ocfg = OD.DQNConfig(huber_delta=huber, lr=1e-4)
    st = OD.DQNState.create(p, ocfg)
    col: dict = {}
    loss_ref, td_ref = OD.update_with_batch(st, ocfg, obs, act, ret, weight, collect=col)

    cfg = D.DQNConfig(huber_delta=huber, lr=1e-4)
    eng = D.DQNEngine(c, h, w, A, D.flat_from_torch([p[k] for k in OD.PARAM_ORDER], c, h, w, A), cfg)
    obs_nhwc = torch.as_tensor(obs).permute(0, 2, 3, 1).float().contiguous().cuda()
    grad = torch.empty(eng.P, dtype=torch.float32, device="cuda")
    loss, td = eng.update_with_batch(obs_nhwc, act, ret, weight, grad_out=grad, apply=False)
    assert rel_err(td.cpu(), td_ref) < 1e-5
    assert abs(float(loss) - loss_ref) <= 1e-5 * abs(loss_ref)
    g_ref = D.flat_from_torch([col["grads"][k] for k in OD.PARAM_ORDER], c, h, w, A, device="cpu")
    off, _ = D.layer_layout(c, h, w, A)
    for i in range(5):                                   # per layer, on the layer's own scale
        assert rel_err(grad[off[i]:off[i + 1]].cpu(), g_ref[off[i]:off[i + 1]]) < 1e-5, f"layer {i}"
    # the optimizer step on top of the same gradient
    loss2, _ = eng.update_with_batch(obs_nhwc, act, ret, weight)
    new = torch.cat([t.reshape(-1) for t in D.flat_to_torch(eng.params, c, h, w, A)]).cpu().numpy()
    ref = OD.flatten_params(st.params).numpy()
    # Adam's first step is lr * g / (|g| + eps): for the handful of weights whose gradient is ~eps = 1e-8
    # (1e-7 of the layer's scale) the rounding of g shows up at full size, bounded by lr
    bad = np.abs(new - ref) > 1e-5 * np.abs(ref) + 0.02 * cfg.lr
    assert bad.mean() < 1e-4 and np.abs(new - ref).max() <= 2 * cfg.lr


def test_adam_step_vs_torch():
    """ts_adam_step (clip_grad_norm_ + Adam) on a given gradient: same arithmetic as torch.optim.Adam."""
    import ctypes as C

    from tianshou_amd import _lib

    n = 1_687_206
    gen = torch.Generator().manual_seed(0)
    p = torch.randn(n, generator=gen)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], lr=1e-4)
    dp, dm, dv = p.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    ws = _lib.default_workspace(0)
    for step in range(1, 4):
        g = torch.randn(n, generator=gen) * 10 ** float(torch.randint(-6, 2, (1,), generator=gen))
        # clip_grad_norm_ (algorithm_base.py:496-499) with the norm accumulated in float64: torch's own
        # float32 CPU accumulation over 1.7 M elements is only good to ~2e-5, the kernel's tree to ~1e-7
        coef = min(0.5 / (float(g.double().norm()) + 1e-6), 1.0)
        ref.grad = g * torch.tensor(coef, dtype=torch.float32)
        opt.step()
        g_dev = g.cuda()
        _lib.check(_lib.load().ts_adam_step(ws.handle, _lib.ptr(dp), _lib.ptr(dm), _lib.ptr(dv), _lib.ptr(g_dev),
                                            _lib.i64(n), _lib.i64(step), _lib.f64(1e-4), _lib.f64(0.9), _lib.f64(0.999),
                                            _lib.f64(1e-8), _lib.f64(0.5), _lib.current_stream()))
        np.testing.assert_allclose(dp.cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-9)
        st = opt.state[ref]
        m_ref, v_ref = st["exp_avg"].numpy(), st["exp_avg_sq"].numpy()
        np.testing.assert_allclose(dm.cpu().numpy(), m_ref, rtol=2e-6, atol=1e-6 * np.abs(m_ref).max())
        np.testing.assert_allclose(dv.cpu().numpy(), v_ref, rtol=2e-6, atol=1e-6 * np.abs(v_ref).max())


def test_gradient_plus_apply_equals_update():
    """The data-parallel halves (DQNEngine.gradient / apply_gradient) reproduce update_with_batch bit for bit."""
    from tianshou_amd import dqn as D

    c, h, w, A, B = 2, 44, 36, 3, 40
    rng = np.random.default_rng(1)
    p = OD.init_params(c, h, w, A, seed=4)
    flat = D.flat_from_torch([p[k] for k in OD.PARAM_ORDER], c, h, w, A)
    cfg = D.DQNConfig(huber_delta=1.0, lr=1e-3, target_update_freq=2, max_grad_norm=0.5)
    e1, e2 = D.DQNEngine(c, h, w, A, flat, cfg), D.DQNEngine(c, h, w, A, flat, cfg)
    grad = torch.empty(e2.P, dtype=torch.float32, device="cuda")
    for _ in range(3):
        obs = torch.as_tensor(rng.integers(0, 256, size=(B, h, w, c)).astype(np.float32)).cuda()
        act, ret = rng.integers(0, A, size=B), rng.normal(size=B).astype(np.float32)
        l1, td1 = e1.update_with_batch(obs, act, ret)
        l2, td2 = e2.gradient(obs, act, ret, None, grad)
        e2.apply_gradient(grad)
        assert torch.equal(td1, td2) and torch.equal(l1, l2)
        assert torch.equal(e1.params, e2.params) and torch.equal(e1.params_old, e2.params_old)
        assert (e1.iter, e1.adam_step) == (e2.iter, e2.adam_step)


@pytest.mark.parametrize("n_act,B", [(18, 3), (1, 5), (64, 2)])
def test_other_action_counts_and_tiny_batches(n_act, B):
    """Atari's largest action set (18), a single action, the head limit (64); batches smaller than any tile."""
    from tianshou_amd import dqn as D

    c, h, w = 4, 84, 84
    rng = np.random.default_rng(n_act)
    p = OD.init_params(c, h, w, n_act, seed=n_act)
    cfg = OD.DQNConfig(huber_delta=None, lr=1e-4, max_grad_norm=10.0)
    st = OD.DQNState.create(p, cfg)
    obs = rng.integers(0, 256, size=(B, c, h, w), dtype=np.uint8)
    act, ret = rng.integers(0, n_act, size=B), rng.normal(size=B).astype(np.float32)
    wgt = rng.random(B).astype(np.float32)
    loss_ref, td_ref = OD.update_with_batch(st, cfg, obs, act, ret, wgt)
    eng = D.DQNEngine(c, h, w, n_act, D.flat_from_torch([p[k] for k in OD.PARAM_ORDER], c, h, w, n_act),
                      D.DQNConfig(huber_delta=None, lr=1e-4, max_grad_norm=10.0))
    loss, td = eng.update_with_batch(torch.as_tensor(obs).permute(0, 2, 3, 1).float().contiguous().cuda(), act, ret, wgt)
    assert rel_err(td.cpu(), td_ref) < 1e-5
    np.testing.assert_allclose(float(loss), loss_ref, rtol=1e-5)
    new = torch.cat([t.reshape(-1) for t in D.flat_to_torch(eng.params, c, h, w, n_act)]).cpu().numpy()
    ref = OD.flatten_params(st.params).numpy()
    assert (np.abs(new - ref) > 1e-5 * np.abs(ref) + 0.02 * cfg.lr).mean() < 1e-4


def test_bad_arguments_fail_loudly():
    from tianshou_amd import _lib
    from tianshou_amd import dqn as D

    with pytest.raises(ValueError):
        D.param_count(4, 20, 20, 6)                       # observation too small for the three convolutions
    with pytest.raises(_lib.EngineError):
        D.layer_layout(4, 84, 84, 65)                     # more actions than the head supports
    p = OD.init_params(4, 84, 84, 6, seed=0)
    flat = D.flat_from_torch([p[k] for k in OD.PARAM_ORDER], 4, 84, 84, 6)
    eng = D.DQNEngine(4, 84, 84, 6, flat, D.DQNConfig())
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(2, 84, 84, 3, device="cuda"))
    with pytest.raises(ValueError):
        eng.update_with_batch(torch.zeros(2, 84, 84, 4, device="cuda"), [0, 1, 2], [0.0, 0.0])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        D.DQNEngine(4, 84, 84, 6, flat.cpu(), D.DQNConfig())
    with pytest.raises(_lib.EngineError):                 # conv shape outside the kernel's limits: K % 32 != 0
        D.conv_forward(torch.zeros(1, 9, 9, 3, device="cuda"), torch.zeros(3 * 3 * 3 + 1, 32, device="cuda"), 3, 3, 1, True)


def test_uint8_observations_give_bitwise_identical_results():
    """obs_u8 mode (frames converted on load inside the conv kernels) vs the float32 copy of the same frames:
    the conversion is exact, so Q-values, TD errors, loss and the post-Adam parameters are bit-identical."""
    from tianshou_amd import dqn as D

    c, h, w, A, B = 4, 84, 84, 6, 37
    rng = np.random.default_rng(5)
    obs8 = torch.as_tensor(rng.integers(0, 256, size=(B, h, w, c), dtype=np.uint8)).cuda()
    obs32 = obs8.float()
    act, ret = rng.integers(0, A, size=B), rng.normal(size=B).astype(np.float32)
    p = OD.init_params(c, h, w, A, seed=8)
    flat = D.flat_from_torch([p[k] for k in OD.PARAM_ORDER], c, h, w, A)
    cfg = D.DQNConfig(huber_delta=1.0, lr=1e-4, target_update_freq=1)
    e8, e32 = D.DQNEngine(c, h, w, A, flat, cfg), D.DQNEngine(c, h, w, A, flat, cfg)
    assert torch.equal(e8.forward(obs8)[0], e32.forward(obs32)[0])
    assert torch.equal(e8.target_q(obs8), e32.target_q(obs32))
    l8, td8 = e8.update_with_batch(obs8, act, ret)
    l32, td32 = e32.update_with_batch(obs32, act, ret)
    assert torch.equal(td8, td32) and torch.equal(l8, l32) and torch.equal(e8.params, e32.params)
    # single layer, backward included
    wb = torch.randn(8 * 8 * 4 + 1, 32, device="cuda") * 0.05
    y8, y32 = D.conv_forward(obs8, wb, 8, 8, 4, True), D.conv_forward(obs32, wb, 8, 8, 4, True)
    assert torch.equal(y8, y32)
    dy = torch.randn_like(y8)
    assert torch.equal(D.conv_backward(obs8, wb, dy, 8, 8, 4, need_dx=False)[0],
                       D.conv_backward(obs32, wb, dy, 8, 8, 4, need_dx=False)[0])


@pytest.mark.parametrize("h,w,B", [(84, 84, 300), (16, 16, 7), (12, 20, 65)])
def test_u8_plane_gather_vector_path_equals_indexing(h, w, B):
    """ts_gather_planes_nhwc_u8, C = 4 and plane size a multiple of 16 bytes (the 16-pixel-per-thread kernel with the
    in-register 4 x 4 byte transposes): bit-identical to torch indexing for arbitrary (repeated, unordered) plane indices."""
    from tianshou_amd import _lib

    g = torch.Generator().manual_seed(h * w + B)
    n_planes = 500
    src = torch.randint(0, 256, (n_planes, h, w), generator=g, dtype=torch.uint8).cuda()
    planes = torch.randint(0, n_planes, (B, 4), generator=g).cuda()
    out = torch.empty((B, h, w, 4), dtype=torch.uint8, device="cuda")
    _lib.check(_lib.load().ts_gather_planes_nhwc_u8(_lib.ptr(src), _lib.i64(n_planes), _lib.i64(h * w), _lib.ptr(planes), _lib.i64(B),
                                                     _lib.i64(4), _lib.ptr(out), _lib.current_stream(src.device)))
    assert torch.equal(out, src[planes].permute(0, 2, 3, 1))


def test_prefetched_forward_gives_bit_identical_updates():
    """DQNEngine.prefetch_forward (Q_online(batch.obs) on a side stream into a cache, consumed by ts_dqn_update_cached) against
    the plain update: three updates with interleaved target passes, identical bits in loss, td errors, parameters and Adam
    moments; a prefetch that does not match the update's tensor, or that predates a parameter write, is ignored."""
    from tianshou_amd import dqn as D

    c, h, w, A, B = 4, 84, 84, 6, 64
    p = OD.init_params(c, h, w, A, 3)
    cfg = D.DQNConfig(gamma=0.99, n_step=1, target_update_freq=2, is_double=True, huber_delta=1.0, lr=1e-4, max_grad_norm=10.0)
    flat = D.flat_from_torch([p[k] for k in OD.PARAM_ORDER], c, h, w, A)
    plain, pre = D.DQNEngine(c, h, w, A, flat, cfg), D.DQNEngine(c, h, w, A, flat, cfg)
    g = torch.Generator().manual_seed(0)
    for it in range(3):
        obs = torch.randint(0, 256, (B, h, w, c), generator=g, dtype=torch.uint8).cuda()
        obs_next = torch.randint(0, 256, (B, h, w, c), generator=g, dtype=torch.uint8).cuda()
        act = torch.randint(0, A, (B,), generator=g).cuda()
        rew = torch.randn(B, generator=g).cuda()
        wt = torch.rand(B, generator=g).cuda()
        pre.prefetch_forward(obs)
        outs = []
        for eng in (plain, pre):
            ret = rew + 0.99 * eng.target_q(obs_next)
            outs.append(eng.update_with_batch(obs, act, ret, wt))
        assert pre._pre is None                                           # consumed
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), it
        for name in ("params", "adam_m", "adam_v", "params_old"):
            assert torch.equal(getattr(plain, name), getattr(pre, name)), (it, name)
    # a prefetch for another tensor is ignored (own forward pass) ...
    obs2 = obs.clone()
    pre.prefetch_forward(obs)
    l1, l0 = pre.update_with_batch(obs2, act, ret, wt), plain.update_with_batch(obs2, act, ret, wt)
    assert torch.equal(l0[0], l1[0]) and torch.equal(plain.params, pre.params)
    # ... and so is one that predates a parameter write
    pre.prefetch_forward(obs)
    stale, pre._pre = pre._pre, None
    pre.update_with_batch(obs2, act, ret, wt)
    plain.update_with_batch(obs2, act, ret, wt)
    pre._pre = stale
    a, b = pre.update_with_batch(obs, act, ret, wt), plain.update_with_batch(obs, act, ret, wt)
    assert torch.equal(a[0], b[0]) and torch.equal(plain.params, pre.params)


@pytest.mark.parametrize("n_step", [1, 3, 7])
def test_pair_gather_equals_the_six_launch_path(n_step):
    """ts_dqn_gather_pair (both stacked gathers of a DQN update in one launch) against nstep_indices -> next() ->
    2 x (stack_indices + gather): identical bytes over EVERY slot of the reference-generated Atari fixture buffer (two
    sub-buffers with episode ends and a wrapped write pointer) and of a ragged synthetic one, and the same returns through
    DQNEngine.preprocess_with_obs as through DQNEngine.preprocess."""
    from tianshou_amd import dqn as D
    from tianshou_amd.buffer import DeviceReplayBuffer
    from tianshou_amd.returns import nstep_indices

    g, d, ocfg, bstate = DC.load("atari")
    buf = DeviceReplayBuffer(offset=g["buf_offset"], last_index=g["buf_last_index"], lengths=g["buf_lengths"],
                             insertion=g["buf_insertion"], rew=g["rew"], terminated=g["terminated"],
                             truncated=g["truncated"])
    frames = torch.as_tensor(g["frames"]).cuda()
    idx = torch.arange(d["E"] * d["slots"]).cuda()
    cases = [(buf, frames, idx)]
    # ragged: 5 sub-buffers of different fill, random episode ends, 16 x 16 frames, repeated / unordered indices
    rng = np.random.default_rng(n_step)
    sizes, fill = np.array([9, 1, 30, 17, 4]), np.array([9, 1, 12, 17, 0])
    off = np.concatenate([[0], np.cumsum(sizes)])
    total = int(off[-1])
    term = rng.random(total) < 0.15
    trunc = rng.random(total) < 0.1
    last = off[:-1] + np.array([3, 0, 11, 5, 0])
    rb = DeviceReplayBuffer(offset=off, last_index=last, lengths=fill, insertion=(last + 1 - off[:-1]) % sizes,
                            rew=rng.standard_normal(total), terminated=term, truncated=trunc)
    fr = torch.as_tensor(rng.integers(0, 256, (total, 16, 16), dtype=np.uint8)).cuda()
    valid = np.concatenate([np.arange(off[e], off[e] + fill[e]) for e in range(5)])
    cases.append((rb, fr, torch.as_tensor(rng.choice(valid, 300)).cuda()))
    for b, f, ix in cases:
        pair = D.gather_obs_pair(f, b, ix, n_step, 4)
        assert pair is not None
        after = nstep_indices(b, ix, n_step)
        assert torch.equal(pair[0], D.gather_obs_nhwc(f, b, ix, 4, as_u8=True))
        assert torch.equal(pair[1], D.gather_obs_nhwc(f, b, b.next(after), 4, as_u8=True))
    # the engine's two preprocess routes
    c, h, w, A = d["c"], d["h"], d["w"], d["n_act"]
    cfg = D.DQNConfig(gamma=0.99, n_step=n_step, target_update_freq=5, is_double=True, huber_delta=1.0, lr=1e-4)
    p0 = OD.init_params(c, h, w, A, 1)
    eng = D.DQNEngine(c, h, w, A, D.flat_from_torch([p0[k] for k in OD.PARAM_ORDER], c, h, w, A), cfg)
    ix = idx[torch.randperm(idx.numel(), device="cuda")[:32]].contiguous()
    obs, ret = eng.preprocess_with_obs(buf, frames, ix, 4)
    assert eng._pre is not None and eng._pre[0] is obs
    assert torch.equal(ret, eng.preprocess(buf, frames, ix, 4))
    assert torch.equal(obs, D.gather_obs_nhwc(frames, buf, ix, 4, as_u8=True))
    # layouts outside the kernel's fall back (None), with the same results from preprocess_with_obs
    odd = torch.zeros((10, 5, 5), dtype=torch.uint8, device="cuda")
    assert D.gather_obs_pair(odd, rb, torch.zeros(2, dtype=torch.int64).cuda(), n_step, 4) is None
    assert D.gather_obs_pair(fr, rb, torch.zeros(2, dtype=torch.int64).cuda(), n_step, 2) is None


def test_replay_stream_cycle_equals_the_sequential_cycle():
    """dqn.ReplayStream (priority update + next batch's draws, sum-tree descent and frame gathers on a second stream behind
    ts_dqn_wait_td, beside the update's backward pass) against the reference order sample -> preprocess -> update ->
    update_weight on one stream: six updates on a 4096-slot prioritized frame buffer, identical indices, weights, losses, TD
    errors, parameters and sum tree."""
    import bench_dqn as BD
    from tianshou_amd import dqn as D

    def cycle(use_stream: bool):
        frames, act, buf, per = BD.build(4096, 4, seed=3)
        p0 = OD.init_params(4, 84, 84, 6, 2)
        cfg = D.DQNConfig(gamma=0.99, n_step=3, target_update_freq=2, is_double=True, huber_delta=1.0, lr=1e-4)
        eng = D.DQNEngine(4, 84, 84, 6, D.flat_from_torch([p0[k] for k in OD.PARAM_ORDER], 4, 84, 84, 6), cfg)
        gen = torch.Generator(device="cuda").manual_seed(11)
        draw = lambda: torch.rand(64, generator=gen, device="cuda", dtype=torch.float64)  # noqa: E731
        rs = D.ReplayStream(eng, buf, frames, per, 4, draw, lambda i: act[i]) if use_stream else None
        log = []
        for _ in range(6):
            if rs is None:
                idx, wt = per.sample(draw())
                a, pair, coef = act[idx], None, None
            else:
                idx, wt, a, pair, coef = rs.take()
            obs, ret = eng.preprocess_with_obs(buf, frames, idx, 4, pair=pair, coef=coef)
            loss, td = eng.update_with_batch(obs, a, ret, wt)
            if rs is None:
                per.update_weight(idx, td)
            else:
                rs.give(idx, td)
            log.append((idx.clone(), wt.float(), loss.clone(), td.clone()))
        torch.cuda.synchronize()
        return log, eng.params.clone(), per.weight._value.clone(), per.prio_minmax.clone()

    a, b = cycle(False), cycle(True)
    # reset() drops the prepared batch (transitions were added): the next take() samples afresh on the current tree
    frames, act, buf, per = BD.build(4096, 4, seed=3)
    p0 = OD.init_params(4, 84, 84, 6, 2)
    eng = D.DQNEngine(4, 84, 84, 6, D.flat_from_torch([p0[k] for k in OD.PARAM_ORDER], 4, 84, 84, 6),
                      D.DQNConfig(gamma=0.99, n_step=3, target_update_freq=2, is_double=True, huber_delta=1.0, lr=1e-4))
    gen = torch.Generator(device="cuda").manual_seed(5)
    rs = D.ReplayStream(eng, buf, frames, per, 4, lambda: torch.rand(64, generator=gen, device="cuda", dtype=torch.float64),
                        lambda i: act[i])
    first = rs.take()
    obs, ret = eng.preprocess_with_obs(buf, frames, first[0], 4, pair=first[3], coef=first[4])
    _, td = eng.update_with_batch(obs, first[2], ret, first[1])
    rs.give(first[0], td)
    assert rs._next is not None
    rs.reset()
    assert rs._next is None
    again = rs.take()
    assert again[0].shape == first[0].shape and bool((again[0] >= 0).all())
    for it, (x, y) in enumerate(zip(a[0], b[0])):
        for u, v in zip(x, y):
            assert torch.equal(u, v), it
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])


@pytest.mark.parametrize("prioritized,huber,graph", [(True, 1.0, False), (True, 0.0, False), (False, 1.0, False), (True, 1.0, True)])
def test_learn_step_equals_sample_preprocess_update_postprocess(prioritized, huber, graph, monkeypatch):
    """DQNEngine.learn_step (ts_dqn_learn_step: the draws, the sum-tree descent and importance weights, the pair gather, n-step
    coefficients, target passes, the periodic sync, the cached update and the priority update of one OffPolicyAlgorithm.update in
    one library call, the next batch prepared on the replay stream) against the separate calls in the reference's order on one
    stream with the same draws: fourteen updates on a 4096-slot frame buffer -- identical losses, TD errors, parameters, lagged
    parameters, optimizer state, sum tree and running max / min priority; with learn_reset() and a changed seed key on the way.  graph: the opt-in replay of the
    steady state from captured HIP graphs (TS_DQN_GRAPH=1), same values."""
    import bench_dqn as BD
    from tianshou_amd import dqn as D
    B = 64
    monkeypatch.setenv("TS_DQN_GRAPH", "1" if graph else "0")

    def make():
        frames, act, buf, per = BD.build(4096, 4, seed=3)
        p0 = OD.init_params(4, 84, 84, 6, 2)
        cfg = D.DQNConfig(gamma=0.99, n_step=3, target_update_freq=3, is_double=True, huber_delta=huber, lr=1e-4)
        eng = D.DQNEngine(4, 84, 84, 6, D.flat_from_torch([p0[k] for k in OD.PARAM_ORDER], 4, 84, 84, 6), cfg)
        return frames, act, buf, (per if prioritized else None), eng

    def seeds():
        for it in range(14):
            yield (77, it) if it < 11 else (78, it)     # a new key: nothing prepared for it

    def separate():
        frames, act, buf, per, eng = make()
        log = []
        for sd in seeds():
            if per is not None:
                idx, wt = per.sample(D.uniform_draws(B, sd))
            else:
                idx, wt = buf.sample_indices(B, seed=sd), None
            obs, ret = eng.preprocess_with_obs(buf, frames, idx, 4)
            loss, td = eng.update_with_batch(obs, act[idx], ret, wt)
            if per is not None:
                per.update_weight(idx, td)
            log.append((loss.clone(), td.clone()))
        torch.cuda.synchronize()
        return log, eng, per

    def one_call():
        frames, act, buf, per, eng = make()
        log, before = [], eng.learn_graph_launches()
        for k, sd in enumerate(seeds()):
            if k == 6:
                eng.learn_reset()                        # the prepared batch is dropped and drawn again: same values
            loss, td = eng.learn_step(buf, frames, act, per, B, sd, want_td=True)
            log.append((loss.clone(), td.clone()))
        torch.cuda.synchronize()
        graphs.append(eng.learn_graph_launches() - before if before >= 0 else -1)
        return log, eng, per

    graphs = []
    a, b = separate(), one_call()
    for it, (x, y) in enumerate(zip(a[0], b[0])):
        assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]), it
    for name in ("params", "params_old", "adam_m", "adam_v"):
        assert torch.equal(getattr(a[1], name), getattr(b[1], name)), name
    assert a[1].adam_step == b[1].adam_step == 14 and a[1].iter == b[1].iter
    # TS_DQN_GRAPH=1: the steady-state updates (from the third call with unchanged arguments on) are replayed from captured graphs
    assert graphs[0] == (9 if graph else 0), graphs
    if prioritized:
        assert torch.equal(a[2].weight._value, b[2].weight._value) and torch.equal(a[2].prio_minmax, b[2].prio_minmax)
    # layouts outside the call's are refused, not converted
    frames, act, buf, per, eng = make()
    with pytest.raises(ValueError):
        eng.learn_step(buf, frames.float(), act, per, B, (1, 0))
    with pytest.raises(ValueError):
        eng.learn_step(buf, frames, act.int(), per, B, (1, 0))


def test_uniform_draws_are_a_counter_based_stream():
    """ts_uniform_fill_f64: doubles in [0, 1), a pure function of (key, counter, position) -- the prefix of a longer fill equals
    the shorter one, different counters / keys give different streams --, uniform to sampling error."""
    from tianshou_amd import dqn as D
    u = D.uniform_draws(1 << 16, (5, 9))
    assert u.dtype == torch.float64 and float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert torch.equal(u[:100], D.uniform_draws(100, (5, 9)))
    assert not torch.equal(u[:100], D.uniform_draws(100, (5, 10))) and not torch.equal(u[:100], D.uniform_draws(100, (6, 9)))
    assert abs(float(u.mean()) - 0.5) < 0.01 and abs(float(u.var()) - 1 / 12) < 0.005
    hist = torch.histc(u.float(), bins=16, min=0.0, max=1.0)
    assert float((hist - 4096).abs().max()) < 400
