"""The data-parallel exchange through the C ABI (ts_allreduce_*, csrc/ts_collective.hip): RCCL over xGMI, one
communicator per process, issued on the current HIP stream.

`tianshou_amd.distributed` uses `torch.distributed.all_reduce` by default (backend "nccl" is the same RCCL); this
class is the path a non-PyTorch host takes and can be handed to DataParallelPPO / DQN / SAC as `allreduce=`.  The
128-byte RCCL id travels through whatever process group already exists (any backend, gloo included)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


class NativeAllReduce:
    def __init__(self, device: torch.device | int, group=None):
        device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("NativeAllReduce runs RCCL on an MI355X; there is no CPU path")
        lib = _lib.load()
        multi = dist.is_initialized() and dist.get_world_size(group) > 1
        self.rank = dist.get_rank(group) if multi else 0
        self.world = dist.get_world_size(group) if multi else 1
        self.device = device
        uid = (C.c_uint8 * 128)()
        if self.rank == 0:
            _lib.check(lib.ts_allreduce_unique_id(uid))
        if multi:
            on = device if dist.get_backend(group) == "nccl" else torch.device("cpu")
            t = torch.tensor(list(uid), dtype=torch.uint8, device=on)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = (C.c_uint8 * 128)(*t.cpu().tolist())
        self._comm = C.c_void_p()
        _lib.check(lib.ts_allreduce_init(uid, _lib.i64(self.rank), _lib.i64(self.world), C.c_int(device.index or 0),
                                         C.byref(self._comm)))

    def __call__(self, buf: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks, ordered on the current stream of `buf`'s device."""
        if buf.dtype != torch.float32 or not buf.is_contiguous() or buf.device != self.device:
            raise ValueError("NativeAllReduce: a contiguous float32 tensor on the communicator's device")
        _lib.check(_lib.load().ts_allreduce(self._comm, _lib.ptr(buf), _lib.i64(buf.numel()), _lib.current_stream(self.device)))
        return buf

    def close(self) -> None:
        if self._comm:
            _lib.load().ts_allreduce_destroy(self._comm)
            self._comm = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
