"""Sum tree and prioritized-replay weights on the MI355X.

Host-side mirror of ``SegmentTree`` (tianshou/data/utils/segtree.py:5-134) and of the PER
arithmetic of ``PrioritizedReplayBuffer`` (tianshou/data/buffer/prio.py:12-113): same method
names and semantics, float64 tree in HBM, kernels in tianshou_amd/csrc/ts_segtree.hip.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .buffer import _dev_index, _i64_dev


def _setitem(tree: torch.Tensor, index, value) -> None:
    """segtree.py:95-101; `index` already includes +bound; later duplicates win."""
    if tree.dtype != torch.float64 or not tree.is_contiguous():
        raise ValueError("tree must be a contiguous float64 tensor")
    dev = tree.device
    index = _i64_dev(index, dev).reshape(-1)
    if not isinstance(value, torch.Tensor):
        value = torch.as_tensor(np.asarray(value), device=dev)
    value = value.to(dev).reshape(-1)
    if value.dtype not in (torch.float32, torch.float64):
        value = value.to(torch.float64)
    value = value.contiguous()
    if value.numel() != index.numel():
        raise ValueError("index / value size mismatch")
    bound = tree.numel() // 2
    ws = _lib.default_workspace(_dev_index(tree))
    _lib.check(_lib.load().ts_segtree_setitem(
        ws.handle, _lib.ptr(tree), _lib.i64(bound), _lib.ptr(index), _lib.ptr(value),
        1 if value.dtype == torch.float64 else 0, _lib.i64(index.numel()), _lib.current_stream(dev)))


def _reduce(tree: torch.Tensor, start: int, end: int) -> torch.Tensor:
    """segtree.py:104-116 -> 0-d float64 device tensor."""
    out = torch.empty((), dtype=torch.float64, device=tree.device)
    _lib.check(_lib.load().ts_segtree_reduce(_lib.ptr(tree), _lib.i64(start), _lib.i64(end),
                                             _lib.ptr(out), _lib.current_stream(tree.device)))
    return out


def _get_prefix_sum_idx(value: torch.Tensor, bound: int, sums: torch.Tensor) -> torch.Tensor:
    """segtree.py:119-134; `value` (float64) is mutated in place like the reference."""
    if value.dtype != torch.float64 or not value.is_contiguous():
        raise ValueError("value must be a contiguous float64 tensor")
    out = torch.empty(value.shape, dtype=torch.int64, device=value.device)
    _lib.check(_lib.load().ts_segtree_prefix_sum_idx(
        _lib.ptr(value), _lib.i64(value.numel()), _lib.i64(bound), _lib.ptr(sums), _lib.ptr(out),
        _lib.current_stream(value.device)))
    return out


class SegmentTree:
    """segtree.py:5-92 with the tree in HBM."""

    def __init__(self, size: int, device="cuda") -> None:
        bound = 1
        while bound < size:
            bound *= 2
        self._size, self._bound = size, bound
        self._value = torch.zeros(bound * 2, dtype=torch.float64, device=device)

    def __len__(self) -> int:
        return self._size

    def __getitem__(self, index):
        index = _i64_dev(index, self._value.device)
        return self._value[index + self._bound]

    def __setitem__(self, index, value) -> None:
        index = _i64_dev(index, self._value.device).reshape(-1)
        if not isinstance(value, torch.Tensor):
            value = np.broadcast_to(np.asarray(value, dtype=np.float64), (index.numel(),))
        elif value.numel() == 1 and index.numel() != 1:
            value = value.expand(index.numel())
        _setitem(self._value, index + self._bound, value)

    def reduce(self, start: int = 0, end: int | None = None) -> torch.Tensor:
        if start == 0 and end is None:
            return self._value[1]
        if end is None:
            end = self._size
        if end < 0:
            end += self._size
        return _reduce(self._value, start + self._bound - 1, end + self._bound)

    def get_prefix_sum_idx(self, value) -> torch.Tensor:
        if not isinstance(value, torch.Tensor):
            value = torch.as_tensor(np.atleast_1d(np.asarray(value, dtype=np.float64)),
                                    device=self._value.device)
        value = value.to(torch.float64).contiguous().clone()
        return _get_prefix_sum_idx(value, self._bound, self._value)


class PrioritizedWeights:
    """The PER state of PrioritizedReplayBuffer (prio.py:25-47): sum tree + running
    max/min priority, all on the device."""

    def __init__(self, size: int, alpha: float, beta: float, weight_norm: bool = True,
                 device="cuda") -> None:
        assert alpha > 0.0 and beta >= 0.0
        self._alpha, self._beta, self._weight_norm = alpha, beta, weight_norm
        self.weight = SegmentTree(size, device)
        # {max_prio, min_prio}, prio.py:36
        self.prio_minmax = torch.ones(2, dtype=torch.float64, device=device)

    @property
    def _ws(self):
        # the scratch of the CURRENT stream (leaf ids / winner table of a priority update): a priority update issued on a
        # replay stream beside an update's backward pass must not share the update's workspace
        return _lib.default_workspace(_dev_index(self.weight._value))

    def init_weight(self, index) -> None:
        """prio.py:46-47: new transitions get max_prio ** alpha."""
        index = _i64_dev(index, self.weight._value.device).reshape(-1)
        v = (self.prio_minmax[0] ** self._alpha).expand(index.numel())
        self.weight[index] = v

    def sample(self, uniform):
        """prio.py:63-67 + :69-79 + :104-106.  `uniform` = the np.random.rand(batch_size) draws of
        the reference (host-supplied for parity).  Returns (indices int64, weight float64)."""
        tree = self.weight._value
        dev = tree.device
        u = torch.as_tensor(np.asarray(uniform, dtype=np.float64), device=dev) \
            if not isinstance(uniform, torch.Tensor) else uniform.to(dev, torch.float64)
        u = u.contiguous()
        K = u.numel()
        idx = torch.empty(K, dtype=torch.int64, device=dev)
        w = torch.empty(K, dtype=torch.float64, device=dev)
        _lib.check(_lib.load().ts_per_sample(
            self._ws.handle, _lib.ptr(tree), _lib.i64(self.weight._bound), _lib.ptr(u), _lib.i64(K),
            _lib.ptr(self.prio_minmax), _lib.f64(self._beta), int(self._weight_norm), _lib.ptr(idx),
            _lib.ptr(w), _lib.current_stream(dev)))
        return idx, w

    def update_weight(self, index, new_weight) -> None:
        """prio.py:81-90 with new_weight = TD errors (float32)."""
        tree = self.weight._value
        dev = tree.device
        index = _i64_dev(index, dev).reshape(-1)
        nw = new_weight.detach().to(dev, torch.float32).reshape(-1).contiguous()
        if nw.numel() != index.numel():
            raise ValueError("index / new_weight size mismatch")
        _lib.check(_lib.load().ts_per_update_weight(
            self._ws.handle, _lib.ptr(tree), _lib.i64(self.weight._bound), _lib.ptr(index),
            _lib.ptr(nw), _lib.i64(index.numel()), _lib.f64(self._alpha),
            _lib.ptr(self.prio_minmax), _lib.current_stream(dev)))
