"""Checkpoint paths of the device replay mirror: HDF5 round trip in the reference's converter layout
(tianshou/data/utils/converter.py:93-165), write-back into a reference-shaped buffer, and -- when the reference is
mounted -- its own from_hdf5 / VectorReplayBuffer reading what we wrote."""
import sys

import numpy as np
import pytest
import torch

from tests import fake_h5py
from tests.standin import Batch, VectorReplayBuffer
from tianshou_amd import persist
from tianshou_amd.buffer import DeviceReplayBuffer


def _filled(n_env=3, size=5, steps=8, seed=0):
    rng = np.random.default_rng(seed)
    vb = VectorReplayBuffer(n_env * size, n_env, obs_shape=(4,), act_shape=(2,))
    for t in range(steps):
        ids = np.arange(n_env) if t % 3 else np.array([0, 2])
        k = len(ids)
        vb.add(Batch(obs=rng.normal(size=(k, 4)).astype(np.float32), act=rng.normal(size=(k, 2)).astype(np.float32),
                     rew=rng.normal(size=k), terminated=rng.random(k) < 0.2, truncated=rng.random(k) < 0.1,
                     obs_next=rng.normal(size=(k, 4)).astype(np.float32)), buffer_ids=ids)
    return vb


def _same(a: DeviceReplayBuffer, b: DeviceReplayBuffer):
    for k in ("obs", "act", "obs_next", "rew", "terminated", "truncated", "done"):
        assert torch.equal(getattr(a, k).cpu(), getattr(b, k).cpu()), k
        assert getattr(a, k).dtype == getattr(b, k).dtype, k
    for k in ("h_offset", "h_last_index", "h_lengths", "h_insertion"):
        np.testing.assert_array_equal(getattr(a, k), getattr(b, k))


def test_hdf5_round_trip_and_layout():
    vb = _filled()
    m = DeviceReplayBuffer.from_tianshou(vb, device="cpu")
    persist.save_hdf5(m, "mem://a.h5", compression="gzip", h5py=fake_h5py)
    back = persist.load_hdf5("mem://a.h5", device="cpu", h5py=fake_h5py)
    _same(m, back)
    t = fake_h5py.tree(fake_h5py._FILES["mem://a.h5"])
    # converter.py conventions: ints -> group attributes, tensors -> "Tensor" datasets, arrays -> "ndarray" datasets,
    # strings (no int / float / array) -> pickled byte datasets tagged with the class name
    assert t["/"][1]["buffer_num"] == 3 and t["/"][1]["maxsize"] == 15 and t["/"][1]["version"] == 1
    assert t["/format"][1] == "int8" and t["/format"][3]["__data_type__"] == "str"
    assert t["/columns/obs"] == ("dataset", "float32", (15, 4), {"__data_type__": "Tensor"})
    assert t["/columns/rew"][1] == "float64" and t["/columns/done"][1] == "uint8"
    assert t["/manager/offset"] == ("dataset", "int64", (4,), {"__data_type__": "ndarray"})
    assert "/episode" not in t                                     # no device-side add() happened


def test_hdf5_without_h5py_fails_loudly(monkeypatch):
    monkeypatch.setitem(sys.modules, "h5py", None)                 # "import h5py" raises ImportError
    m = DeviceReplayBuffer.from_tianshou(_filled(), device="cpu")
    with pytest.raises(ImportError, match="h5py"):
        m.save_hdf5("/tmp/never_written.h5")


def test_to_tianshou_restores_a_fresh_buffer():
    vb = _filled(steps=11)                                           # sub-buffers 0 and 2 have wrapped
    m = DeviceReplayBuffer.from_tianshou(vb, device="cpu")
    fresh = VectorReplayBuffer(15, 3, obs_shape=(4,), act_shape=(2,))
    m.to_tianshou(fresh)
    for k in ("obs", "act", "obs_next", "rew", "terminated", "truncated", "done"):
        np.testing.assert_array_equal(getattr(fresh, k), getattr(vb, k))
    np.testing.assert_array_equal(fresh._lengths, vb._lengths)
    np.testing.assert_array_equal(fresh.last_index, vb.last_index)
    assert [b._insertion_idx for b in fresh.buffers] == [b._insertion_idx for b in vb.buffers]
    assert [len(b) for b in fresh.buffers] == [len(b) for b in vb.buffers]
    np.testing.assert_array_equal(fresh.sample_indices(0), vb.sample_indices(0))
    np.testing.assert_array_equal(fresh.unfinished_index(), vb.unfinished_index())
    # and the mirror of the restored buffer is the mirror we started from
    _same(m, DeviceReplayBuffer.from_tianshou(fresh, device="cpu"))


def test_to_tianshou_rejects_a_different_layout():
    m = DeviceReplayBuffer.from_tianshou(_filled(), device="cpu")
    with pytest.raises(ValueError, match="sub-buffers"):
        m.to_tianshou(VectorReplayBuffer(15, 5, obs_shape=(4,), act_shape=(2,)))
    with pytest.raises(ValueError, match="size"):
        m.to_tianshou(VectorReplayBuffer(30, 3, obs_shape=(4,), act_shape=(2,)))


# ---- against the reference itself (this container only) ---------------------------------------------------------------
from oracle import ref_shim  # noqa: E402

needs_ref = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference not mounted")


@needs_ref
def test_reference_from_hdf5_reads_our_file_and_vector_buffer_accepts_the_write_back(monkeypatch):
    ref_shim.install()
    monkeypatch.setitem(sys.modules, "h5py", fake_h5py)              # the converter's isinstance checks use h5py.Dataset
    import importlib

    import tianshou.data.utils.converter as conv

    conv = importlib.reload(conv)
    from tianshou.data import Batch as RBatch
    from tianshou.data import VectorReplayBuffer as RVB

    rng = np.random.default_rng(1)
    rvb = RVB(15, 3)
    for t in range(9):
        rvb.add(RBatch(obs=rng.normal(size=(3, 4)).astype(np.float32), act=rng.normal(size=(3, 2)).astype(np.float32),
                       rew=rng.normal(size=3), terminated=rng.random(3) < 0.2, truncated=rng.random(3) < 0.1,
                       obs_next=rng.normal(size=(3, 4)).astype(np.float32), info={}))
    m = DeviceReplayBuffer.from_tianshou(rvb, device="cpu")
    persist.save_hdf5(m, "mem://ref.h5", h5py=fake_h5py)
    got = conv.from_hdf5(fake_h5py._FILES["mem://ref.h5"])           # the REFERENCE's reader on OUR file
    assert got["buffer_num"] == 3 and got["format"] == "tianshou_amd.DeviceReplayBuffer"
    np.testing.assert_array_equal(got["columns"]["obs"].numpy(), np.asarray(rvb.obs))
    np.testing.assert_array_equal(got["manager"]["last_index"], np.asarray(rvb.last_index))
    # the reference's writer and ours agree on the layout of the same dict
    g = fake_h5py.File("mem://ref_writer.h5", "w")
    conv.to_hdf5(persist.mirror_state(m), g)
    assert fake_h5py.tree(g) == fake_h5py.tree(fake_h5py._FILES["mem://ref.h5"])
    # write-back into a fresh reference buffer: same indices, same data, and it keeps working
    fresh = RVB(15, 3)
    fresh.add(RBatch(obs=np.zeros((3, 4), np.float32), act=np.zeros((3, 2), np.float32), rew=np.zeros(3),
                     terminated=np.zeros(3, bool), truncated=np.zeros(3, bool), obs_next=np.zeros((3, 4), np.float32), info={}))
    m.to_tianshou(fresh)
    np.testing.assert_array_equal(fresh.sample_indices(0), rvb.sample_indices(0))
    np.testing.assert_array_equal(fresh.unfinished_index(), rvb.unfinished_index())
    idx = rvb.sample_indices(0)
    np.testing.assert_array_equal(fresh.next(idx), rvb.next(idx))
    np.testing.assert_array_equal(fresh.prev(idx), rvb.prev(idx))
    np.testing.assert_array_equal(np.asarray(fresh.obs), np.asarray(rvb.obs))
    assert len(fresh) == len(rvb)
