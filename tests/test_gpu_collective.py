"""ts_allreduce_* (csrc/ts_collective.hip) on one MI355X: a world-1 RCCL communicator through the C ABI.  The multi-rank
behaviour of the exchange (scaling, ordering, identical replicas) is covered on CPU by the world-2 gloo tests
(tests/test_dp_gloo.py, test_dp_dqn_gloo.py, test_dp_sac_gloo.py), which accept the same `allreduce=` hook."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_native_allreduce_world_one_through_the_c_abi():
    from tianshou_amd import _lib

    lib = _lib.load()
    uid = (C.c_uint8 * 128)()
    _lib.check(lib.ts_allreduce_unique_id(uid))
    assert any(uid)                                            # RCCL filled the id
    comm = C.c_void_p()
    _lib.check(lib.ts_allreduce_init(uid, _lib.i64(0), _lib.i64(1), C.c_int(0), C.byref(comm)))
    x = torch.randn(1_000_003, device="cuda")
    ref = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                              # issued on the caller's stream, whichever it is
        _lib.check(lib.ts_allreduce(comm, _lib.ptr(x), _lib.i64(x.numel()), _lib.current_stream(x.device)))
        y = x * 2
    side.synchronize()
    assert torch.equal(x, ref) and torch.equal(y, ref * 2)     # sum over one rank
    _lib.check(lib.ts_allreduce(comm, None, _lib.i64(0), _lib.current_stream(x.device)))      # empty exchange
    with pytest.raises(_lib.EngineError):
        _lib.check(lib.ts_allreduce(None, _lib.ptr(x), _lib.i64(4), _lib.current_stream(x.device)))
    with pytest.raises(_lib.EngineError):
        _lib.check(lib.ts_allreduce_init(uid, _lib.i64(2), _lib.i64(2), C.c_int(0), C.byref(C.c_void_p())))
    _lib.check(lib.ts_allreduce_destroy(comm))


def test_native_allreduce_as_the_exchange_of_the_dp_wrappers():
    """NativeAllReduce without a process group = world 1; DataParallelSAC takes it as `allreduce=`."""
    from tianshou_amd.collective import NativeAllReduce
    from tianshou_amd.distributed import DataParallelSAC

    ar = NativeAllReduce(torch.device("cuda", 0))
    assert (ar.rank, ar.world) == (0, 1)
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    assert torch.equal(ar(x), torch.arange(1000, dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError):
        ar(torch.zeros(4, dtype=torch.float64, device="cuda"))
    dp = DataParallelSAC(engine=None, allreduce=ar)
    dp.world = 2                                               # force the exchange path on one rank: sum, then * 1/2
    buf = torch.full((10,), 3.0, device="cuda")
    dp._reduce(buf)
    assert torch.equal(buf, torch.full((10,), 1.5, device="cuda"))
    ar.close()
