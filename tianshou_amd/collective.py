"""The data-parallel exchange through the C ABI (ts_allreduce_*, csrc/ts_collective.hip), issued on the current HIP stream:

* RCCL over xGMI, one communicator per process (ts_allreduce_init / ts_allreduce), and
* the one-shot path for payloads up to 64 KB (ts_allreduce_small_*): every rank's device buffer is mapped by the others
  through HIP IPC and a call is one single-workgroup kernel (publish, poll the peers' flags, sum in rank order) -- the
  44 KB [gradient | loss parts] vector of a PPO minibatch step sits between two ~55 us kernels, where RCCL's ring
  latency (tens of microseconds) would be a third of the step.

`tianshou_amd.distributed` falls back to `torch.distributed.all_reduce` (backend "nccl" is the same RCCL) when no
`NativeAllReduce` is given; this class is the path a non-PyTorch host takes and what `ts_ppo_dp_step` needs to put one
data-parallel minibatch behind one call.  The 128-byte RCCL id and the 64-byte IPC handles travel through whatever
process group already exists (any backend, gloo included)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


class NativeAllReduce:
    def __init__(self, device: torch.device | int, group=None, small_floats: int = 16384, rccl: bool = True):
        """`small_floats` > 0: also build the one-shot path for payloads up to that many floats (<= 16384) and let
        ts_allreduce choose by size; it is dropped silently (RCCL only) when HIP IPC is not available between the ranks.
        `rccl=False`: one-shot path only (ranks that share one GPU cannot form an RCCL communicator)."""
        device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("NativeAllReduce runs on MI355X GPUs; there is no CPU path")
        lib = _lib.load()
        lib.ts_allreduce_small_capacity.restype = C.c_int64
        multi = dist.is_initialized() and dist.get_world_size(group) > 1
        self.rank = dist.get_rank(group) if multi else 0
        self.world = dist.get_world_size(group) if multi else 1
        self.device = device
        self.group = group
        self._comm = C.c_void_p()
        self._small = C.c_void_p()
        on = device if multi and dist.get_backend(group) == "nccl" else torch.device("cpu")
        src = dist.get_global_rank(group, 0) if (multi and group is not None) else 0
        if rccl:
            uid = (C.c_uint8 * 128)()
            if self.rank == 0:
                _lib.check(lib.ts_allreduce_unique_id(uid))
            if multi:
                t = torch.tensor(list(uid), dtype=torch.uint8, device=on)
                dist.broadcast(t, src=src, group=group)
                uid = (C.c_uint8 * 128)(*t.cpu().tolist())
            _lib.check(lib.ts_allreduce_init(uid, _lib.i64(self.rank), _lib.i64(self.world), C.c_int(device.index or 0),
                                             C.byref(self._comm)))
        if small_floats > 0 and (multi or not rccl):
            self._init_small(lib, int(small_floats), on, strict=not rccl)
        if not rccl:
            if not self._small:
                raise RuntimeError("NativeAllReduce(rccl=False): the one-shot path could not be set up")
            _lib.check(lib.ts_allreduce_from_small(self._small, C.byref(self._comm)))
        elif self._small:
            _lib.check(lib.ts_allreduce_attach_small(self._comm, self._small))

    def _init_small(self, lib, small_floats: int, on, strict: bool) -> None:
        handle = (C.c_uint8 * 64)()
        small = C.c_void_p()
        rc = lib.ts_allreduce_small_create(C.c_int(self.device.index or 0), _lib.i64(small_floats), C.byref(small), handle)
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=on)
        if self.world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:                       # some rank could not export its buffer: every rank backs out
            if rc == 0:
                lib.ts_allreduce_small_destroy(small)
            if strict:
                _lib.check(rc if rc != 0 else _lib.TS_ERR_HIP)
            return
        mine = torch.tensor(list(handle), dtype=torch.uint8, device=on)
        allh = [torch.empty_like(mine) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(allh, mine, group=self.group)
        else:
            allh = [mine]
        flat = (C.c_uint8 * (64 * self.world))(*[int(b) for t in allh for b in t.cpu().tolist()])
        rc = lib.ts_allreduce_small_connect(small, flat, _lib.i64(self.rank), _lib.i64(self.world))
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=on)
        if self.world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            lib.ts_allreduce_small_destroy(small)
            if strict:
                _lib.check(rc if rc != 0 else _lib.TS_ERR_HIP)
            return
        if not self._probe_small(lib, small, on):     # mapped, but a trial call did not deliver the sum: not a path to trust
            lib.ts_allreduce_small_destroy(small)
            if strict:
                raise RuntimeError("NativeAllReduce: the one-shot path failed its trial all-reduce")
            return
        self._small = small

    def _probe_small(self, lib, small, on) -> bool:
        """One trial call through the freshly connected one-shot path (rank r contributes r + 1 in every slot, so the sum is
        world (world + 1) / 2 exactly); a peer that never arrives shows up in the status word.  All ranks take the same
        decision."""
        n = int(min(lib.ts_allreduce_small_capacity(small), 256))
        ok = 1
        try:
            buf = torch.full((n,), float(self.rank + 1), dtype=torch.float32, device=self.device)
            stream = _lib.current_stream(self.device)
            _lib.check(lib.ts_allreduce_small(small, _lib.ptr(buf), _lib.i64(n), stream))
            _lib.check(lib.ts_allreduce_small_status(small, stream))
            ok = int(bool((buf == self.world * (self.world + 1) / 2).all().item()))
        except (_lib.EngineError, RuntimeError, ValueError):
            ok = 0
        t = torch.tensor([ok], dtype=torch.int32, device=on)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item()) == 1

    @property
    def small_capacity(self) -> int:
        """Floats per call the one-shot path takes (0: RCCL only)."""
        return int(_lib.load().ts_allreduce_small_capacity(self._small)) if self._small else 0

    def ranks(self) -> tuple[int, int]:
        """(ranks the communicator was built for, ranks RCCL reports for it -- 0 without an RCCL side)."""
        w, r = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.load().ts_allreduce_ranks(self._comm, C.byref(w), C.byref(r)))
        return int(w.value), int(r.value)

    def __call__(self, buf: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks, ordered on the current stream of `buf`'s device."""
        if not self._comm:
            raise RuntimeError("NativeAllReduce: the communicator has been closed")
        if buf.dtype != torch.float32 or not buf.is_contiguous() or buf.device != self.device:
            raise ValueError("NativeAllReduce: a contiguous float32 tensor on the communicator's device")
        _lib.check(_lib.load().ts_allreduce(self._comm, _lib.ptr(buf), _lib.i64(buf.numel()), _lib.current_stream(self.device)))
        return buf

    def check(self) -> None:
        """Synchronises the stream and raises if a peer never arrived in a one-shot call (bounded spin)."""
        if self._small:
            _lib.check(_lib.load().ts_allreduce_small_status(self._small, _lib.current_stream(self.device)))

    def close(self) -> None:
        lib = _lib.load()
        if self._comm:
            lib.ts_allreduce_destroy(self._comm)
            self._comm = C.c_void_p()
        if self._small:
            lib.ts_allreduce_small_destroy(self._small)
            self._small = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
