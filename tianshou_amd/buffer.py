"""Device-resident mirror of the replay-buffer state the learn() path reads.

Mirrors the read side of ``ReplayBuffer`` / ``ReplayBufferManager`` / ``VectorReplayBuffer``
(tianshou/data/buffer/buffer_base.py, manager.py, vecbuf.py): same method names, same index
semantics, torch tensors on an MI355X instead of NumPy arrays.  The write side (``add``) stays
with the reference's collector; ``from_arrays`` / ``from_tianshou`` take a snapshot.

Layout in HBM (structure of arrays, one allocation per key, row-major):
    obs, obs_next  [B, ...]   dtype as collected (f32 / u8)
    act            [B, ...]
    rew            [B]        float64 (buffer_base.py:492 stores rewards as float)
    terminated, truncated, done   [B] uint8
    offset [E+1], last_index [E], lengths [E], insertion [E]   int64  (manager.py:50-52)
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _dev_index(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise RuntimeError("tianshou_amd kernels need tensors on an MI355X (cuda/hip device); "
                           "there is no CPU fallback")
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _i64_dev(x, device) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.int64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.int64)), device=device)


def _u8_dev(x, device) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x.to(device=device).to(torch.uint8).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x).astype(bool).astype(np.uint8)), device=device)


def _next_index(index, offset, done, last_index, lengths) -> torch.Tensor:
    """manager.py:339-363, same argument order; int64 device tensors in and out."""
    return _step(True, index, offset, done, last_index, lengths)


def _prev_index(index, offset, done, last_index, lengths) -> torch.Tensor:
    """manager.py:311-336."""
    return _step(False, index, offset, done, last_index, lengths)


def _step(nxt: bool, index, offset, done, last_index, lengths) -> torch.Tensor:
    dev = done.device
    index = _i64_dev(index, dev)
    shape = index.shape
    index = index.reshape(-1)
    out = torch.empty_like(index)
    E = offset.numel() - 1
    if lengths.numel() != E or last_index.numel() != E:
        raise ValueError("offset / last_index / lengths size mismatch")
    fn = _lib.load().ts_next_index if nxt else _lib.load().ts_prev_index
    _lib.check(fn(_lib.ptr(index), _lib.i64(index.numel()), _lib.ptr(offset), _lib.i64(E),
                  _lib.ptr(done), _lib.ptr(last_index), _lib.ptr(lengths), _lib.ptr(out),
                  _lib.current_stream(dev)))
    return out.reshape(shape)


def gather_rows(src: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """src[index] for a contiguous [B, ...] tensor (buffer_base.py:605-649 fancy-index gather)."""
    if not src.is_contiguous():
        raise ValueError("gather_rows needs a contiguous source")
    index = _i64_dev(index, src.device).reshape(-1)
    row_bytes = src.element_size() * int(np.prod(src.shape[1:], dtype=np.int64))
    out = torch.empty((index.numel(), *src.shape[1:]), dtype=src.dtype, device=src.device)
    _lib.check(_lib.load().ts_gather_rows(_lib.ptr(src), _lib.i64(src.shape[0]), _lib.i64(row_bytes),
                                          _lib.ptr(index), _lib.i64(index.numel()), _lib.ptr(out),
                                          _lib.current_stream(src.device)))
    return out


def gather_rows_multi(srcs, index: torch.Tensor) -> list[torch.Tensor]:
    """[src[index] for src in srcs] -- Batch.__getitem__ over several keys (batch.py:714-738) -- in ONE launch for up to 8
    contiguous tensors with the same number of rows whose rows are whole 4-byte words (ts_gather_rows_multi); other inputs
    fall back to gather_rows per key."""
    srcs = list(srcs)
    if not srcs:
        return []
    dev = srcs[0].device
    index = _i64_dev(index, dev).reshape(-1)
    n = srcs[0].shape[0]
    rb = [s.element_size() * int(np.prod(s.shape[1:], dtype=np.int64)) for s in srcs]
    ok = (len(srcs) <= 8 and all(s.is_contiguous() and s.shape[0] == n and s.device == dev and s.data_ptr() % 4 == 0 for s in srcs)
          and all(b >= 4 and b % 4 == 0 for b in rb))
    if not ok:
        return [gather_rows(s, index) for s in srcs]
    outs = [torch.empty((index.numel(), *s.shape[1:]), dtype=s.dtype, device=dev) for s in srcs]
    k = len(srcs)
    h_src = (C.c_void_p * k)(*[s.data_ptr() for s in srcs])
    h_out = (C.c_void_p * k)(*[o.data_ptr() for o in outs])
    h_rb = (C.c_int64 * k)(*rb)
    _lib.check(_lib.load().ts_gather_rows_multi(_lib.i64(k), h_src, h_rb, _lib.i64(n), _lib.ptr(index), _lib.i64(index.numel()),
                                                h_out, _lib.current_stream(dev)))
    return outs


def random_permutation(n: int, seed: int, device="cuda") -> torch.Tensor:
    """Device-side replacement for np.random.permutation(n) in Batch.split (batch.py:1209): a keyed
    bijection of range(n), int64 device tensor (ts_random_permutation)."""
    out = torch.empty(n, dtype=torch.int64, device=device)
    _lib.check(_lib.load().ts_random_permutation(_lib.ptr(out), _lib.i64(n), C.c_uint64(seed & (2**64 - 1)),
                                                  _lib.current_stream(out.device)))
    return out


def normal_noise(shape, seed: int, offset: int, device="cuda") -> torch.Tensor:
    """eps ~ N(0, 1), float32 `shape`, from the engine's counter-based generator (ts_normal_fill: Philox-4x32-10 keyed by
    (seed, offset) + Box-Muller) -- the device-side stand-in for the `rsample()` / exploration noise the reference draws
    with torch's generator.  `offset` = a per-call counter."""
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    _lib.check(_lib.load().ts_normal_fill(_lib.ptr(out), _lib.i64(out.numel()), C.c_uint64(seed & (2**64 - 1)),
                                          C.c_uint64(offset & (2**64 - 1)), _lib.current_stream(out.device)))
    return out


class AddTracker:
    """Exact write log of a reference replay buffer: per sub-buffer the number of transitions
    `ReplayBufferManager.add` (manager.py:131-198: one slot in each of `buffer_ids`, all sub-buffers when None) /
    `ReplayBuffer.add` (buffer_base.py:420-501) wrote since the mirror last looked, and an `everything` flag for the
    calls that rewrite the storage wholesale (`reset`, `update`, `set_batch`).  The reference keeps no such counter,
    and `_insertion_idx` / `len()` are unchanged after exactly `size` adds or after `reset()` + an equal refill (the
    on-policy pattern, trainer.py:1116-1130).

    Mechanism: the write methods of the buffer's classes are wrapped ONCE per class (every class of the MRO that
    defines them; nested `super().add(...)` calls are counted once through a depth guard); a buffer takes part only
    while it carries the plain-data record `_hip_writes` in its `__dict__` (a dict of an int64 array, a bool and an
    int: survives pickling, deepcopy and the buffer's HDF5 export; nothing callable is stored on the instance).
    The patch a Tianshou maintainer would add instead is the three counting lines inside those methods
    (INTEGRATION.md)."""
    _ATTR = "_hip_writes"
    _MARK = "_hip_tracked_methods"

    @classmethod
    def _wrap_class(cls, klass) -> None:
        done = klass.__dict__.get(cls._MARK, ())
        for name in ("add", "reset", "update", "set_batch"):
            fn = klass.__dict__.get(name)
            if fn is None or name in done or not callable(fn):
                continue
            setattr(klass, name, cls._counted(fn, name))
            done = done + (name,)
        setattr(klass, cls._MARK, done)

    @classmethod
    def _counted(cls, fn, name):
        attr = cls._ATTR

        def method(self, *args, **kwargs):
            st = self.__dict__.get(attr)
            if st is None:
                return fn(self, *args, **kwargs)
            st["depth"] += 1
            try:
                out = fn(self, *args, **kwargs)
            finally:
                st["depth"] -= 1
            if st["depth"] == 0:
                if name != "add" or st["coarse"]:
                    st["everything"] = True
                else:
                    ids = kwargs.get("buffer_ids", args[1] if len(args) > 1 else None)
                    counts = st["counts"]
                    if counts.size == 1:
                        counts[0] += 1
                    elif ids is None:
                        counts += 1
                    else:
                        np.add.at(counts, np.asarray(ids, dtype=np.int64).reshape(-1), 1)
            return out

        method.__name__, method.__doc__, method.__wrapped__ = name, fn.__doc__, fn
        return method

    @classmethod
    def of(cls, buffer) -> dict:
        """The buffer's record, created (with `everything` set: nothing is known about earlier writes) on first use."""
        st = buffer.__dict__.get(cls._ATTR)
        if st is None:
            for klass in type(buffer).__mro__:
                if klass is not object:
                    cls._wrap_class(klass)
            n = len(buffer.buffers) if hasattr(buffer, "buffers") else 1
            # CachedReplayBuffer (cached.py) moves whole episodes between sub-buffers inside add(): not slot-countable
            st = {"counts": np.zeros(n, dtype=np.int64), "everything": True, "depth": 0,
                  "coarse": "Cached" in type(buffer).__name__}
            buffer.__dict__[cls._ATTR] = st
        return st

    @classmethod
    def take(cls, buffer):
        """-> (counts copy, everything flag) since the previous take(); resets both."""
        st = cls.of(buffer)
        out = (st["counts"].copy(), bool(st["everything"]))
        st["counts"][:] = 0
        st["everything"] = False
        return out


class DeviceReplayBuffer:
    """Read-side mirror of ReplayBufferManager on one GPU."""

    def __init__(self, *, offset, last_index, lengths, insertion, rew, terminated, truncated,
                 obs=None, act=None, obs_next=None, device="cuda"):
        device = torch.device(device)
        self.device = device
        # host copy of the (tiny) manager state: the host decides launch sizes from it
        self.h_offset = np.ascontiguousarray(np.asarray(offset, dtype=np.int64))
        self.h_last_index = np.ascontiguousarray(np.asarray(last_index, dtype=np.int64))
        self.h_lengths = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64))
        self.h_insertion = np.ascontiguousarray(np.asarray(insertion, dtype=np.int64))
        self.buffer_num = int(self.h_offset.size - 1)
        self.maxsize = int(self.h_offset[-1])
        self.offset = _i64_dev(self.h_offset, device)
        self.last_index = _i64_dev(self.h_last_index, device)
        self.lengths = _i64_dev(self.h_lengths, device)
        self.insertion = _i64_dev(self.h_insertion, device)
        self.rew = (rew.to(device=device, dtype=torch.float64) if isinstance(rew, torch.Tensor)
                    else torch.as_tensor(np.asarray(rew, dtype=np.float64), device=device)).contiguous()
        self.terminated = _u8_dev(terminated, device)
        self.truncated = _u8_dev(truncated, device)
        self.done = (self.terminated | self.truncated).contiguous()       # manager.py:150
        to = lambda x: None if x is None else (  # noqa: E731
            x.to(device) if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x), device=device)
        ).contiguous()
        self.obs, self.act, self.obs_next = to(obs), to(act), to(obs_next)
        self._ws_cache = None
        self._ep = None              # (ep_return f64[E], ep_len i64[E], ep_start i64[E] relative), created by add()
        self._host_stale = False

    @property
    def _ws(self):
        if self._ws_cache is None:         # created on first kernel use: raises on a CPU mirror (no CPU fallback)
            self._ws_cache = _lib.default_workspace(_dev_index(self.done))
        return self._ws_cache

    # -- constructors -------------------------------------------------------------------------
    @classmethod
    def from_vector_fill(cls, n_env: int, **arrays):
        """A VectorReplayBuffer(B, n_env) whose every sub-buffer was written exactly once from
        slot 0 to T-1 (synthetic rollouts)."""
        B = int(arrays["rew"].shape[0])
        T = B // n_env
        if T * n_env != B:
            raise ValueError("buffer size must be divisible by n_env")
        offset = np.arange(n_env + 1, dtype=np.int64) * T
        return cls(offset=offset, last_index=offset[:-1] + T - 1,
                   lengths=np.full(n_env, T, np.int64), insertion=np.zeros(n_env, np.int64), **arrays)

    @classmethod
    def empty(cls, total_size: int, n_env: int, obs_shape, act_shape=(), *, obs_dtype=torch.float32,
              act_dtype=torch.float32, save_obs_next: bool = True, device="cuda"):
        """A fresh VectorReplayBuffer(total_size, n_env) (vecbuf.py: equal sub-buffers of
        ceil(total_size / n_env) slots), to be filled with add()."""
        size = -(-total_size // n_env)
        B = size * n_env
        offset = np.arange(n_env + 1, dtype=np.int64) * size
        obs = torch.zeros((B, *tuple(obs_shape)), dtype=obs_dtype, device=device)
        act = torch.zeros((B, *tuple(act_shape)), dtype=act_dtype, device=device)
        return cls(offset=offset, last_index=offset[:-1].copy(), lengths=np.zeros(n_env, np.int64),
                   insertion=np.zeros(n_env, np.int64), rew=torch.zeros(B, dtype=torch.float64, device=device),
                   terminated=torch.zeros(B, dtype=torch.uint8, device=device),
                   truncated=torch.zeros(B, dtype=torch.uint8, device=device), obs=obs, act=act,
                   obs_next=torch.zeros_like(obs) if save_obs_next else None, device=device)

    # -- write side -----------------------------------------------------------------
    def add(self, obs, act, rew, terminated, truncated, obs_next=None, buffer_ids=None):
        """ReplayBufferManager.add (manager.py:131-198) on device tensors: one transition for each of the
        distinct sub-buffers `buffer_ids` (None = all, in order).  Returns the reference's tuple
        (current_index int64, episode_reward float64, episode_length int64, episode_start_index int64) as device
        tensors; the host copies of the manager state are refreshed lazily (see _sync_host)."""
        dev = self.device
        if self._ep is None:
            E = self.buffer_num
            self._ep = (torch.zeros(E, dtype=torch.float64, device=dev), torch.zeros(E, dtype=torch.int64, device=dev),
                        (self.insertion * 0 + torch.as_tensor(self.h_insertion, device=dev)).contiguous())
        ids = None if buffer_ids is None else _i64_dev(buffer_ids, dev).reshape(-1)
        rew = torch.as_tensor(rew, device=dev).to(torch.float64).reshape(-1).contiguous()
        K = rew.numel()
        if ids is not None and ids.numel() != K:
            raise ValueError("buffer_ids / batch length mismatch")
        if ids is None and K != self.buffer_num:
            raise ValueError("without buffer_ids the batch must have one row per sub-buffer")
        term, trunc = _u8_dev(terminated, dev).reshape(-1), _u8_dev(truncated, dev).reshape(-1)
        keys = []
        rows = {"obs": obs, "act": act, "obs_next": obs_next}
        hold = []
        for name, val in rows.items():
            dst = getattr(self, name)
            if dst is None or val is None:
                continue
            src = torch.as_tensor(val, device=dev).to(dst.dtype).reshape(K, -1).contiguous()
            hold.append(src)
            rb = dst.element_size() * int(np.prod(dst.shape[1:], dtype=np.int64))
            if src.shape[1] * src.element_size() != rb:
                raise ValueError(f"{name}: row shape does not match the buffer")
            keys.append((dst.data_ptr(), src.data_ptr(), rb))

        class _Key(C.Structure):
            _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("row_bytes", C.c_int64)]

        arr = (_Key * max(len(keys), 1))(*[_Key(*k) for k in keys])
        idx = torch.empty(K, dtype=torch.int64, device=dev)
        ep_ret = torch.empty(K, dtype=torch.float64, device=dev)
        ep_len, ep_start = torch.empty_like(idx), torch.empty_like(idx)
        _lib.check(_lib.load().ts_buffer_add(
            _lib.ptr(ids), _lib.i64(K), _lib.ptr(rew), _lib.ptr(term), _lib.ptr(trunc), _lib.ptr(self.offset),
            _lib.i64(self.buffer_num), _lib.ptr(self.insertion), _lib.ptr(self.lengths), _lib.ptr(self.last_index),
            _lib.ptr(self._ep[0]), _lib.ptr(self._ep[1]), _lib.ptr(self._ep[2]), _lib.ptr(self.rew),
            _lib.ptr(self.terminated), _lib.ptr(self.truncated), _lib.ptr(self.done), arr, C.c_int(len(keys)),
            _lib.ptr(idx), _lib.ptr(ep_ret), _lib.ptr(ep_len), _lib.ptr(ep_start), _lib.current_stream(dev)))
        self._host_stale = True
        return idx, ep_ret, ep_len, ep_start

    def _sync_host(self) -> None:
        """Refreshes the host copies of the (tiny) manager state after device-side add() calls."""
        if self._host_stale:
            self.h_last_index = self.last_index.cpu().numpy()
            self.h_lengths = self.lengths.cpu().numpy()
            self.h_insertion = self.insertion.cpu().numpy()
            self._host_stale = False

    @classmethod
    def from_tianshou(cls, buffer, device="cuda", env_range: tuple[int, int] | None = None):
        """Snapshot of a reference ReplayBuffer / ReplayBufferManager (duck-typed; only reads
        public attributes plus _extend_offset/_lengths/_insertion_idx, manager.py:50-52).

        `env_range = (lo, hi)`: mirror only the sub-buffers [lo, hi) -- this rank's shard of a buffer that is sharded by
        env id across the GPUs of a node (`tianshou_amd.distributed.shard_envs`).  Episodes never span
        sub-buffers (manager.py:50-60 offsets), so sampling, index math, GAE and n-step returns of the shard need nothing
        from the others.  Indices of the mirror are LOCAL (slot 0 = first slot of sub-buffer lo); `to_global` adds the
        shard's base."""
        if hasattr(buffer, "buffers"):
            offset = np.asarray(buffer._extend_offset, dtype=np.int64)
            lengths = np.asarray(buffer._lengths, dtype=np.int64)
            insertion = np.asarray([b._insertion_idx for b in buffer.buffers], dtype=np.int64)
        else:
            offset = np.asarray([0, buffer.maxsize], dtype=np.int64)
            lengths = np.asarray([len(buffer)], dtype=np.int64)
            insertion = np.asarray([buffer._insertion_idx], dtype=np.int64)
        n_env = offset.size - 1
        lo, hi = (0, n_env) if env_range is None else (int(env_range[0]), int(env_range[1]))
        if not 0 <= lo < hi <= n_env:
            raise ValueError(f"env_range {env_range} outside the {n_env} sub-buffers")
        base, top = int(offset[lo]), int(offset[hi])
        sl = slice(base, top)
        meta = buffer._meta
        has = lambda k: k in meta.get_keys()  # noqa: E731
        AddTracker.take(buffer)                         # the snapshot below covers everything written so far
        m = cls(offset=offset[lo:hi + 1] - base, last_index=np.asarray(buffer.last_index, dtype=np.int64).reshape(-1)[lo:hi] - base,
                lengths=lengths[lo:hi], insertion=insertion[lo:hi], rew=np.asarray(buffer.rew)[sl],
                terminated=np.asarray(buffer.terminated)[sl], truncated=np.asarray(buffer.truncated)[sl],
                obs=np.asarray(buffer.obs)[sl], act=np.asarray(buffer.act)[sl],
                obs_next=np.asarray(buffer.obs_next)[sl] if has("obs_next") else None, device=device)
        m.env_range, m.base, m.n_env_total = (lo, hi), base, n_env
        m.is_manager = hasattr(buffer, "buffers")
        return m

    is_manager = True                             # False: the mirror of a plain ReplayBuffer (sample_indices(None) differs)
    env_range: tuple[int, int] | None = None      # (lo, hi) of a shard mirror; None: the whole buffer
    base = 0                                      # first global slot of the mirror
    n_env_total: int | None = None

    def to_global(self, index):
        """Local mirror indices -> indices of the host buffer the mirror was taken from."""
        return index + self.base

    def sync_from_tianshou(self, buffer) -> int:
        """Incremental refresh from the reference buffer this mirror was created from: copies only the slots
        written since the last sync (ring order per sub-buffer, manager.py:162-177) plus the tiny manager state.
        How many slots each sub-buffer received comes from the `AddTracker` that `from_tianshou` installed on the
        buffer object (exact for any number of adds, including whole multiples of the sub-buffer size, and for
        `reset()` + refill, which leave `_insertion_idx` / `len` unchanged); adjacent ranges of neighbouring
        sub-buffers are merged so that a fully rewritten VectorReplayBuffer is one copy per key.  A shard mirror
        (`env_range`) looks at its own sub-buffers only.  Returns the number of slots copied."""
        subs = buffer.buffers if hasattr(buffer, "buffers") else [buffer]
        lo_e, hi_e = self.env_range if self.env_range is not None else (0, len(subs))
        if hi_e - lo_e != self.buffer_num or len(subs) != (self.n_env_total or len(subs)):
            raise ValueError("buffer layout changed since the mirror was created")
        counts, everything = AddTracker.take(buffer)
        keys = [k for k in ("obs", "act", "obs_next") if getattr(self, k) is not None]
        host = {k: np.asarray(getattr(buffer, k)) for k in keys}
        host.update(rew=np.asarray(buffer.rew), terminated=np.asarray(buffer.terminated),
                    truncated=np.asarray(buffer.truncated))
        base = self.base
        ranges = []
        for e in range(self.buffer_num):
            sb = subs[lo_e + e]
            start, size = int(self.h_offset[e]), int(self.h_offset[e + 1] - self.h_offset[e])
            new_ins, new_len = int(sb._insertion_idx), len(sb)
            k = size if everything else min(int(counts[lo_e + e]), size)
            if k >= size > 0:
                ranges.append((start, start + size))
            elif k > 0:                      # the k slots before the insertion point, ring order
                a = (new_ins - k) % size
                ranges.append((start + a, start + min(a + k, size)))
                if a + k > size:
                    ranges.append((start, start + a + k - size))
            self.h_insertion[e], self.h_lengths[e] = new_ins, new_len
        ranges.sort()
        merged = []
        for lo, hi in ranges:
            if merged and lo <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], hi)
            else:
                merged.append([lo, hi])
        copied = 0
        for lo, hi in merged:
            sl, hs = slice(lo, hi), slice(base + lo, base + hi)
            for key, arr in host.items():
                dst = getattr(self, key)
                src = torch.from_numpy(np.ascontiguousarray(arr[hs]))
                dst[sl].copy_(src if src.dtype == dst.dtype else src.to(dst.dtype), non_blocking=True)
            self.done[sl] = self.terminated[sl] | self.truncated[sl]
            copied += hi - lo
        self.h_last_index = np.ascontiguousarray(
            np.asarray(buffer.last_index, dtype=np.int64).reshape(-1)[lo_e:hi_e] - base)
        self.last_index.copy_(torch.as_tensor(self.h_last_index))
        self.lengths.copy_(torch.as_tensor(self.h_lengths))
        self.insertion.copy_(torch.as_tensor(self.h_insertion))
        return copied

    # -- checkpoints (tianshou_amd/persist.py) ------------------------------------------
    def to_tianshou(self, buffer) -> None:
        """Writes the mirror back into a reference buffer of the same layout (then `buffer.save_hdf5(...)` / pickle give
        the reference's own files)."""
        from . import persist

        persist.to_tianshou(self, buffer)

    def save_hdf5(self, path: str, compression: str | None = None) -> None:
        """buffer_base.py:252-256 for the mirror itself, converter.py's dataset / attribute conventions."""
        from . import persist

        persist.save_hdf5(self, path, compression)

    @classmethod
    def load_hdf5(cls, path: str, device="cuda"):
        """buffer_base.py:258-263."""
        from . import persist

        return persist.load_hdf5(path, device)

    # -- reference API ------------------------------------------------------------------------
    def __len__(self) -> int:
        self._sync_host()
        return int(self.h_lengths.sum())

    def next(self, index) -> torch.Tensor:
        return _next_index(index, self.offset, self.done, self.last_index, self.lengths)

    def prev(self, index) -> torch.Tensor:
        return _prev_index(index, self.offset, self.done, self.last_index, self.lengths)

    def obs_next_rows(self, index) -> torch.Tensor:
        """The `obs_next` column of `buffer[index]` (buffer_base.py:622-626): the stored rows when the buffer saves them,
        otherwise `obs[next(index)]` (save_obs_next=False: the next slot of the episode; an episode's last / newest
        transition maps to itself, as in the reference)."""
        if self.obs_next is not None:
            return gather_rows(self.obs_next, index)
        if self.obs is None:
            raise ValueError("the device buffer holds neither obs_next nor obs")
        return gather_rows(self.obs, self.next(index))

    def unfinished_index(self) -> torch.Tensor:
        """manager.py:85-91 -> int64[<=E] device tensor (one tiny D2H for the count)."""
        out, n = self._unfinished_raw()
        return out[: int(n.item())]

    def _unfinished_raw(self):
        out = torch.empty(self.buffer_num, dtype=torch.int64, device=self.device)
        n = torch.zeros(1, dtype=torch.int64, device=self.device)
        _lib.check(_lib.load().ts_unfinished_index(
            _lib.ptr(self.offset), _lib.i64(self.buffer_num), _lib.ptr(self.done),
            _lib.ptr(self.last_index), _lib.ptr(self.lengths), _lib.ptr(out), _lib.ptr(n),
            _lib.current_stream(self.device)))
        return out, n

    def sample_indices_stacked(self, batch_size: int, stack_num: int, *, positions=None, generator=None) -> torch.Tensor:
        """ReplayBufferManager.sample_indices for a frame-stacking buffer (`stack_num > 1 and sample_avail`,
        manager.py:205-216 -> buffer_base.py:532-545): only indices with stack_num - 1 earlier frames in their episode are
        available -- all indices in sub-buffer / ring order, minus those whose (stack_num - 2)-fold predecessor equals its
        own predecessor (the prev() index kernel; the compaction is a device boolean select).
        batch_size == 0 -> all available indices; > 0 -> `RandomState.choice(all_indices, batch_size)`: pass the reference's
        draws as `positions` (int64[bs], positions into the available indices) to replay a seeded run, otherwise they come
        from torch's device generator.  None -> len(all_indices) draws (manager.py:213-214); negative -> no indices
        (manager.py:202-204); batch_size > 0 with nothing available raises ValueError like RandomState.choice([], bs)."""
        if stack_num < 2:
            raise ValueError("sample_indices_stacked is the stack_num > 1 branch")
        if batch_size is not None and batch_size < 0:
            return torch.empty(0, dtype=torch.int64, device=self.device)
        all_idx = self.sample_indices(0)
        p = all_idx
        for _ in range(stack_num - 2):
            p = self.prev(p)
        avail = all_idx[p != self.prev(p)]
        if batch_size == 0:
            return avail
        if batch_size is None:
            batch_size = int(avail.numel())
            if batch_size == 0:
                return avail
        if avail.numel() == 0:
            raise ValueError("sample_indices: no index has stack_num - 1 earlier frames yet (a must be non-empty)")
        if positions is None:
            positions = torch.randint(0, avail.numel(), (int(batch_size),), device=self.device, generator=generator)
        else:
            positions = _i64_dev(positions, self.device).reshape(-1)
            if positions.numel() != batch_size or int(positions.min()) < 0 or int(positions.max()) >= avail.numel():
                raise ValueError("positions must be batch_size draws inside the available indices")
        return avail[positions]

    def sample_indices(self, batch_size: int | None, *, u_buffer=None, within=None, generator=None, seed=None) -> torch.Tensor:
        """ReplayBufferManager.sample_indices (manager.py:200-234) for stack_num == 1 (frame-stacking buffers with
        `sample_avail`: `sample_indices_stacked`).

        batch_size == 0 -> every valid index, sub-buffer-major and time-ordered.
        batch_size > 0  -> sub-buffer with probability proportional to its length, then uniform inside it, concatenated
        in sub-buffer order (ts_sample_indices_random).  The reference draws from the buffers' own RandomStates
        (buffer_base.py:98); to reproduce a seeded reference run pass its draws: `u_buffer` float64[bs] (the uniforms
        `RandomState.choice(E, bs, p=...)` consumes) and `within` int64[bs] (the children's `choice(len_e, n_e)` values,
        concatenated in sub-buffer order).  Without them the draws come from torch's device generator (`generator`), or --
        `seed=(key, counter)` -- from the engine's counter-based generator inside the sampling kernel (one launch).
        batch_size None -> the manager passes 0 to every child (manager.py:217-218): all indices in order, like 0; the mirror
        of a PLAIN ReplayBuffer (`is_manager` False) follows buffer_base.py:513-517 instead: batch_size = len(self), i.e.
        len(self) random draws with replacement (through the same draw arguments as any batch_size > 0);
        batch_size < 0  -> an empty index array (manager.py:202-204)."""
        if batch_size is not None and batch_size < 0:
            return torch.empty(0, dtype=torch.int64, device=self.device)
        if batch_size is None:
            batch_size = 0 if self.is_manager else len(self)
        if batch_size > 0:
            dev = self.device
            bs = int(batch_size)
            if len(self) == 0:                               # buffer_base.py:512-513: an empty buffer yields no indices
                return torch.empty(0, dtype=torch.int64, device=dev)
            if (u_buffer is None) != (within is None):
                raise ValueError("pass both u_buffer and within (the reference's draws) or neither")
            if u_buffer is None and seed is not None:
                # the engine's own counter-based draws (`seed` = (key, counter), e.g. (0x5EED, update number)): one launch
                key, counter = seed
                out = torch.empty(bs, dtype=torch.int64, device=dev)
                if getattr(self, "_err", None) is None:
                    self._err = torch.zeros(1, dtype=torch.int32, device=dev)
                _lib.check(_lib.load().ts_sample_indices_seeded(
                    _lib.ptr(self.offset), _lib.i64(self.buffer_num), _lib.ptr(self.lengths), C.c_uint64(int(key) & (2**64 - 1)),
                    C.c_uint64(int(counter) & (2**64 - 1)), _lib.i64(bs), _lib.ptr(out), _lib.ptr(self._err),
                    _lib.current_stream(dev)))
                return out
            if u_buffer is None:
                r = torch.rand(2, bs, dtype=torch.float64, device=dev, generator=generator)
                u, w_i, w_u = r[0].contiguous(), None, r[1].contiguous()
            else:
                u = torch.as_tensor(np.asarray(u_buffer, dtype=np.float64), device=dev).reshape(-1).contiguous()
                w_i, w_u = _i64_dev(within, dev).reshape(-1), None
                if u.numel() != bs or w_i.numel() != bs:
                    raise ValueError("u_buffer / within must have batch_size entries")
            out = torch.empty(bs, dtype=torch.int64, device=dev)
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(_lib.load().ts_sample_indices_random(
                _lib.ptr(self.offset), _lib.i64(self.buffer_num), _lib.ptr(self.lengths), _lib.ptr(u), _lib.ptr(w_i),
                _lib.ptr(w_u), _lib.i64(bs), _lib.ptr(out), _lib.ptr(err), _lib.current_stream(dev)))
            if w_i is not None and int(err.item()):            # host-supplied draws are checked (one tiny D2H)
                raise ValueError("sample_indices: empty buffer or a `within` draw outside its sub-buffer")
            return out
        total = len(self)
        out = torch.empty(total, dtype=torch.int64, device=self.device)
        _lib.check(_lib.load().ts_sample_indices_all(
            self._ws.handle, _lib.ptr(self.offset), _lib.i64(self.buffer_num), _lib.ptr(self.lengths),
            _lib.ptr(self.insertion), _lib.i64(total), _lib.ptr(out), _lib.current_stream(self.device)))
        return out

    def indices_are_identity(self) -> bool:
        """True when sample_indices(0) == arange(B): every sub-buffer full and unwrapped."""
        self._sync_host()
        size = np.diff(self.h_offset)
        return bool(np.all(self.h_lengths == size) and np.all(self.h_insertion % size == 0))

    def gather(self, key: str, index) -> torch.Tensor:
        return gather_rows(getattr(self, key), index)

    @staticmethod
    def gather_tensor(src: torch.Tensor, index) -> torch.Tensor:
        return gather_rows(src, index)
