"""GPU parity: HIP GAE / n-step kernels (through the C ABI) vs the CPU oracle and the committed
golden vectors of the reference.  Tolerances: float64 scan 1e-12 relative to the value scale
(the parallel scan re-associates), float32 outputs 1e-6; n-step and all index math bit-exact."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    return t if dtype is None else t.to(dtype)


def make_buffer(g, pre, **extra):
    from tianshou_amd.buffer import DeviceReplayBuffer

    return DeviceReplayBuffer(offset=g[pre + "offset"], last_index=g[pre + "last_index"],
                              lengths=g[pre + "lengths"], insertion=g[pre + "insertion"], **extra)


def test_gae_known_answers_from_reference_tests():
    from tianshou_amd import returns as R
    from tianshou_amd.buffer import DeviceReplayBuffer

    g = load("returns_kat.npz")
    for c in range(int(g["n_gae"])):
        gamma, lam = g[f"gae{c}_gamma_lambda"]
        n = len(g[f"gae{c}_rew"])
        # ReplayBuffer(20) holding n transitions written from slot 0
        buf = DeviceReplayBuffer(offset=[0, 20], last_index=[n - 1], lengths=[n], insertion=[n % 20],
                                 rew=np.pad(g[f"gae{c}_rew"], (0, 20 - n)),
                                 terminated=np.pad(g[f"gae{c}_terminated"], (0, 20 - n)),
                                 truncated=np.pad(g[f"gae{c}_truncated"], (0, 20 - n)))
        idx = buf.sample_indices(0)
        assert np.array_equal(idx.cpu().numpy(), g[f"gae{c}_indices"])
        assert np.array_equal(buf.unfinished_index().cpu().numpy(), g[f"gae{c}_unfinished"])
        batch = SimpleNamespace(rew=dev(g[f"gae{c}_rew"]), terminated=dev(g[f"gae{c}_terminated"]),
                                truncated=dev(g[f"gae{c}_truncated"]))
        v = dev(g[f"gae{c}_v_next"], torch.float32) if bool(g[f"gae{c}_has_v"]) else None
        ret, adv = R.compute_episodic_return(batch, buf, idx, v, None, gamma=gamma, gae_lambda=lam)
        assert np.allclose(ret.cpu().numpy(), g[f"gae{c}_literal"])
        np.testing.assert_allclose(ret.cpu().numpy(), g[f"gae{c}_ref_returns"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(adv.cpu().numpy(), g[f"gae{c}_ref_adv"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n,n_env,rew64", [(1, 1, True), (7, 1, False), (2048, 4, True),
                                           (2049, 1, True), (5000, 5, False), (65536, 64, True),
                                           (300001, 7, True), (300001, 1500, True), (70000, 9000, False)])
def test_gae_matches_oracle_ragged_sizes(n, n_env, rew64):
    from tianshou_amd import returns as R

    rng = np.random.default_rng(n)
    v_s = rng.normal(size=n).astype(np.float32)
    v_n = rng.normal(size=n).astype(np.float32)
    rew = rng.normal(size=n).astype(np.float32)
    term = rng.random(n) < 0.01
    trunc = (rng.random(n) < 0.005) & ~term
    cuts = np.unique(rng.integers(0, n, size=n_env))
    idx = np.arange(n)
    ret_o, adv_o = O.compute_episodic_return(rew, term, trunc, idx, cuts, v_n, v_s, 0.99, 0.95)
    out = R.gae_scan(dev(v_s), dev(v_n), dev(rew.astype(np.float64) if rew64 else rew), dev(term),
                     dev(trunc), dev(cuts), gamma=0.99, gae_lambda=0.95, want_f64=True,
                     want_ret_stats=True)
    scale = max(1.0, float(np.abs(adv_o).max()))
    np.testing.assert_allclose(out["adv64"].cpu().numpy(), adv_o, rtol=0, atol=1e-12 * scale)
    np.testing.assert_allclose(out["ret64"].cpu().numpy(), ret_o, rtol=0, atol=1e-12 * scale)
    np.testing.assert_allclose(out["adv"].cpu().numpy(), adv_o.astype(np.float32), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out["returns"].cpu().numpy(), ret_o.astype(np.float32), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(float(out["ret_sum"]), ret_o.sum(), rtol=1e-10, atol=1e-8)
    np.testing.assert_allclose(float(out["ret_sumsq"]), (ret_o**2).sum(), rtol=1e-10)


def _long_cut_case():
    from tianshou_amd import returns as R

    n, valid = 131_071, 3000
    rng = np.random.default_rng(5)
    v_s, v_n = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
    rew = rng.normal(size=n)
    term = rng.random(n) < 0.003
    trunc = np.zeros(n, bool)
    cuts = rng.permutation(n)[:valid]
    cuts[:10] = n - 1 - np.arange(10)                                   # cuts in the partial last word
    padded = np.concatenate([cuts, cuts[:50], rng.integers(0, n, size=1046)])   # duplicates, then ignored garbage
    d_n = torch.tensor([valid + 50], dtype=torch.int64, device="cuda")
    ret_o, adv_o = O.compute_episodic_return(rew, term, trunc, np.arange(n), np.unique(cuts), v_n, v_s, 0.99, 0.95)
    out = R.gae_scan(dev(v_s), dev(v_n), dev(rew), dev(term), dev(trunc), dev(padded), gamma=0.99, gae_lambda=0.95,
                     want_f64=True, d_n_cut=d_n)
    scale = max(1.0, float(np.abs(adv_o).max()))
    np.testing.assert_allclose(out["adv64"].cpu().numpy(), adv_o, rtol=0, atol=1e-12 * scale)
    np.testing.assert_allclose(out["ret64"].cpu().numpy(), ret_o, rtol=0, atol=1e-12 * scale)


def test_gae_long_cut_list_with_device_side_count():
    """More than 1024 cuts go through the global cut bitmask; only the first *d_n_cut entries of the (unordered,
    padded) cut list count, duplicates and the tail of the last word are harmless."""
    _long_cut_case()


def test_gae_no_episode_end_carries_across_all_tiles():
    """gamma*lambda = 1, no end flag: the carry crosses every tile boundary (Monte-Carlo
    return of one 100k-step episode) - exercises the multi-tile look-back without early exit."""
    from tianshou_amd import returns as R

    n = 100_000
    rng = np.random.default_rng(3)
    rew = rng.normal(size=n)
    z = np.zeros(n, np.float32)
    f = np.zeros(n, bool)
    ret_o, adv_o = O.compute_episodic_return(rew, f, f, np.arange(n), np.array([n - 1]), z, z, 1.0, 1.0)
    out = R.gae_scan(dev(z), dev(z), dev(rew), dev(f), dev(f), dev(np.array([n - 1])), gamma=1.0,
                     gae_lambda=1.0, want_f64=True)
    np.testing.assert_allclose(out["adv64"].cpu().numpy(), adv_o, rtol=0, atol=1e-9)
    # leaf-level drop-in with the njit signature
    a = R._gae(dev(z), dev(z), dev(rew), dev(f), 1.0, 1.0)
    np.testing.assert_allclose(a.cpu().numpy(), O._gae(z, z, rew, f, 1.0, 1.0), rtol=0, atol=1e-9)


def test_gae_return_scaling_and_unaligned_views():
    from tianshou_amd import returns as R

    n = 4099
    rng = np.random.default_rng(11)
    v_s = rng.normal(size=n + 3).astype(np.float32)
    v_n = rng.normal(size=n + 3).astype(np.float32)
    rew = rng.normal(size=n + 3)
    term = rng.random(n + 3) < 0.02
    trunc = np.zeros(n + 3, bool)
    scale = float(np.sqrt(3.7 + 1e-8))
    sl = slice(3, None)  # 12-byte offset: forces the scalar-load path
    ret_o, adv_o = O.compute_episodic_return(rew[sl], term[sl], trunc[sl], np.arange(n), np.array([n - 1]),
                                             v_n[sl].astype(np.float64) * scale,
                                             v_s[sl].astype(np.float64) * scale, 0.99, 0.95)
    out = R.gae_scan(dev(v_s)[sl], dev(v_n)[sl], dev(rew)[sl], dev(term)[sl], dev(trunc)[sl],
                     dev(np.array([n - 1])), v_scale=scale, ret_div=scale, want_f64=True)
    np.testing.assert_allclose(out["adv64"].cpu().numpy(), adv_o, rtol=0, atol=1e-11)
    np.testing.assert_allclose(out["returns"].cpu().numpy(), (ret_o / scale).astype(np.float32), rtol=1e-6, atol=1e-6)


def test_gae_general_indices_via_isin():
    """Wrapped sub-buffers: sample_indices(0) is not arange; cuts come from isin(indices, unfinished)."""
    from tianshou_amd import returns as R
    from tianshou_amd.buffer import DeviceReplayBuffer

    rng = np.random.default_rng(5)
    E, T = 6, 50
    B = E * T
    rew = rng.normal(size=B)
    term = rng.random(B) < 0.05
    trunc = np.zeros(B, bool)
    insertion = rng.integers(0, T, size=E)
    offset = np.arange(E + 1) * T
    last = offset[:-1] + (insertion - 1) % T
    st = O.BufferState(offset, last, np.full(E, T), insertion, rew, term, trunc)
    buf = DeviceReplayBuffer(offset=offset, last_index=last, lengths=np.full(E, T), insertion=insertion,
                             rew=rew, terminated=term, truncated=trunc)
    idx_o = st.sample_indices_all()
    idx = buf.sample_indices(0)
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    v_s = rng.normal(size=B).astype(np.float32)
    v_n = rng.normal(size=B).astype(np.float32)
    ret_o, adv_o = O.compute_episodic_return(rew[idx_o], term[idx_o], trunc[idx_o], idx_o,
                                             st.unfinished_index(), v_n, v_s, 0.99, 0.95)
    batch = SimpleNamespace(rew=buf.gather("rew", idx), terminated=buf.gather("terminated", idx),
                            truncated=buf.gather("truncated", idx))
    ret, adv = R.compute_episodic_return(batch, buf, idx, dev(v_n), dev(v_s), 0.99, 0.95)
    np.testing.assert_allclose(adv.cpu().numpy(), adv_o, rtol=0, atol=1e-12)
    np.testing.assert_allclose(ret.cpu().numpy(), ret_o, rtol=0, atol=1e-12)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 10])
def test_nstep_known_answers_bit_exact(variant, n):
    from tianshou_amd import returns as R

    g = load("returns_kat.npz")
    pre = f"nstep{variant}_"
    buf = make_buffer(g, pre, rew=g[pre + "rew"], terminated=g[pre + "terminated"],
                      truncated=g[pre + "truncated"])
    st = O.BufferState(g[pre + "offset"], g[pre + "last_index"], g[pre + "lengths"],
                       g[pre + "insertion"], g[pre + "rew"], g[pre + "terminated"], g[pre + "truncated"])
    indices = buf.sample_indices(0)
    assert np.array_equal(indices.cpu().numpy(), g[pre + "indices"])

    def target_q_fn(b, after):  # test_returns.py:162-165
        return (-b.rew[b.next(after)]).to(torch.float32)

    batch = SimpleNamespace()
    R.compute_nstep_return(batch, buf, indices, target_q_fn, gamma=0.1, n_step=n, want_f64=True)
    got = batch.returns.cpu().numpy()
    assert np.allclose(got, g[pre + f"n{n}_literal"])
    assert np.array_equal(got.reshape(-1), g[pre + f"n{n}_ref"].reshape(-1))        # float32 bit-exact
    ora, _ = O.compute_nstep_return(st, g[pre + "indices"],
                                    lambda a: -st.rew[st.next(a)].astype(np.float32), 0.1, n)
    assert np.array_equal(batch.returns64.cpu().numpy().reshape(-1), ora.reshape(-1))  # float64 bit-exact

    def target_q_multi(b, after):  # :167-168
        return target_q_fn(b, after).unsqueeze(1).repeat(1, 51)

    batch2 = SimpleNamespace()
    R.compute_nstep_return(batch2, buf, indices, target_q_multi, gamma=0.1, n_step=n)
    assert np.array_equal(batch2.returns.cpu().numpy(), g[pre + f"n{n}_ref_multidim"])
    # leaf with the njit signature: stacked indices + explicit end flags
    after, stack = R.nstep_indices(buf, indices, n, want_stack=True)
    end = st.done.copy()
    end[st.unfinished_index()] = True
    tq = target_q_fn(buf, after) * (buf.terminated[after] == 0)
    leaf, leaf64 = R._nstep_return(buf.rew, dev(end), tq.reshape(-1, 1), stack, 0.1, n, want_f64=True)
    assert np.array_equal(leaf64.cpu().numpy().reshape(-1), ora.reshape(-1))


def test_nstep_random_buffers_bit_exact_vs_oracle():
    from tianshou_amd import returns as R
    from tianshou_amd.buffer import DeviceReplayBuffer

    rng = np.random.default_rng(21)
    E, T = 16, 257
    B = E * T
    for n_step, A in [(1, 1), (3, 1), (5, 6), (32, 2)]:
        rew = rng.normal(size=B)
        term = rng.random(B) < 0.03
        trunc = (rng.random(B) < 0.01) & ~term
        lengths = rng.integers(1, T + 1, size=E)
        lengths[0] = T
        insertion = np.where(lengths == T, rng.integers(0, T, size=E), lengths)
        offset = np.arange(E + 1) * T
        last = offset[:-1] + (insertion - 1) % np.maximum(lengths, 1)
        st = O.BufferState(offset, last, lengths, insertion % T, rew, term, trunc)
        buf = DeviceReplayBuffer(offset=offset, last_index=last, lengths=lengths, insertion=insertion % T,
                                 rew=rew, terminated=term, truncated=trunc)
        valid = st.sample_indices_all()
        indices = rng.choice(valid, size=777)
        q = rng.normal(size=(B, A)).astype(np.float32)
        ora, after_o = O.compute_nstep_return(st, indices, lambda a: q[a], 0.97, n_step)
        batch = SimpleNamespace()
        qd = dev(q)
        R.compute_nstep_return(batch, buf, dev(indices), lambda b, a: qd[a], gamma=0.97, n_step=n_step,
                               want_f64=True)
        assert np.array_equal(R.nstep_indices(buf, dev(indices), n_step).cpu().numpy(), after_o)
        assert np.array_equal(batch.returns64.cpu().numpy(), ora.reshape(777, A))
        assert np.array_equal(batch.returns.cpu().numpy(), ora.reshape(777, A).astype(np.float32))


def test_nstep_shape_mismatch_raises_value_error():
    from tianshou_amd import returns as R
    from tianshou_amd.buffer import DeviceReplayBuffer

    buf = DeviceReplayBuffer.from_vector_fill(2, rew=np.zeros(8), terminated=np.zeros(8, bool),
                                              truncated=np.zeros(8, bool))

    class B3:
        def __len__(self):
            return 3

    with pytest.raises(ValueError, match="mismatch"):
        R.compute_nstep_return(B3(), buf, dev(np.arange(4)), lambda b, a: torch.zeros(4, device="cuda"))


def test_gae_single_pass_handoff_under_load():
    """Many tiles (more than can be resident), no episode end at all and gamma*lambda = 1: every
    workgroup has to fold the maps of ALL following tiles, i.e. the cross-workgroup hand-off is
    exercised at full depth; repeated launches reuse the persistent hand-off area (epoch tags)."""
    from tianshou_amd import _lib
    from tianshou_amd import returns as R

    n = (1 << 23) + 777
    rng = np.random.default_rng(17)
    rew = rng.normal(size=n)
    z = np.zeros(n, np.float32)
    f = np.zeros(n, bool)
    ref = O._gae(z, z, rew, f, 1.0, 1.0)
    dz, drew, df = dev(z), dev(rew), dev(f)
    for rep in range(3):
        out = R.gae_scan(dz, dz, drew, df, df, None, gamma=1.0, gae_lambda=1.0, want_f64=True)
        np.testing.assert_allclose(out["adv64"].cpu().numpy(), ref, rtol=0, atol=1e-7)
    # a different size right after (ticket base / epoch bookkeeping)
    m = 123457
    out = R.gae_scan(dz[:m], dz[:m], drew[:m], df[:m], df[:m], None, gamma=1.0, gae_lambda=1.0, want_f64=True)
    np.testing.assert_allclose(out["adv64"].cpu().numpy(), O._gae(z[:m], z[:m], rew[:m], f[:m], 1.0, 1.0), rtol=0, atol=1e-8)
    assert _lib.default_workspace(0).gae_check() == 0
