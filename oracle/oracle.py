"""TEST INFRASTRUCTURE ONLY - Python face of the CPU parity oracle.

Two layers:

1. ctypes bindings to ``oracle/ts_oracle.c`` (plain-C restatement of the reference's numba
   leaf functions: ``_gae``, ``_nstep_return``, ``_next_index``/``_prev_index``, the
   sum-tree kernels, PER weights, ``RunningMeanStd.update``), and
2. a torch-fp32 (CPU, autograd) restatement of the floating-point part of the path that
   lives in third-party ``torch`` in the reference: the MLP actor/critic forward, the PPO
   ``_preprocess_batch`` / ``_update_with_batch`` loop, ``clip_grad_norm_`` and Adam
   (``oracle_ppo.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product package ``tianshou_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libts_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ts_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_segtree_reduce.restype = C.c_double
        _lib.oracle_unfinished_index.restype = C.c_int64
        _lib.oracle_sample_indices_all.restype = C.c_int64
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.float64)


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.int64)


def _u8(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a).astype(bool), dtype=np.uint8)


# ---------------------------------------------------------------------------------------------
# leaf functions (same names / argument order as the reference's njit functions)
# ---------------------------------------------------------------------------------------------
def _gae(v_s, v_s_, rew, end_flag, gamma: float, gae_lambda: float) -> np.ndarray:
    """algorithm_base.py:1085-1140 -> float64[N]."""
    v_s, v_s_, rew, end = _f64(v_s), _f64(v_s_), _f64(rew), _u8(end_flag)
    out = np.zeros(rew.shape, np.float64)
    lib().oracle_gae(_p(v_s), _p(v_s_), _p(rew), _p(end), C.c_int64(rew.size),
                     C.c_double(gamma), C.c_double(gae_lambda), _p(out))
    return out


def episode_mc_return_to_go(rewards, gamma: float = 0.99) -> np.ndarray:
    """algorithm_base.py:1143-1157."""
    r = _f64(rewards)
    out = np.zeros(r.shape, np.float64)
    lib().oracle_episode_mc_return_to_go(_p(r), C.c_int64(r.size), C.c_double(gamma), _p(out))
    return out


def _nstep_return(rew_B, end_flag_B, target_q_IA, stacked_indices_NI, gamma: float,
                  n_step: int) -> np.ndarray:
    """algorithm_base.py:1160-1222 -> float64[I, A]."""
    rew, end = _f64(rew_B), _u8(end_flag_B)
    tq = np.ascontiguousarray(np.asarray(target_q_IA), dtype=np.float32)
    I = tq.shape[0]
    tq2 = tq.reshape(I, -1)
    A = tq2.shape[1]
    idx = _i64(stacked_indices_NI)
    assert idx.shape == (n_step, I)
    out = np.zeros((I, A), np.float64)
    lib().oracle_nstep_return(_p(rew), _p(end), _p(tq2), _p(idx), C.c_int64(I), C.c_int64(A),
                              C.c_int64(n_step), C.c_double(gamma), _p(out))
    return out.reshape(tq.shape)


def _prev_index(index, offset, done, last_index, lengths) -> np.ndarray:
    """manager.py:311-336."""
    index, offset, last_index, lengths = _i64(index), _i64(offset), _i64(last_index), _i64(lengths)
    done = _u8(done)
    out = np.zeros_like(index)
    lib().oracle_prev_index(_p(index), C.c_int64(index.size), _p(offset),
                            C.c_int64(offset.size - 1), _p(done), _p(last_index), _p(lengths),
                            _p(out))
    return out


def _next_index(index, offset, done, last_index, lengths) -> np.ndarray:
    """manager.py:339-363."""
    index, offset, last_index, lengths = _i64(index), _i64(offset), _i64(last_index), _i64(lengths)
    done = _u8(done)
    out = np.zeros_like(index)
    lib().oracle_next_index(_p(index), C.c_int64(index.size), _p(offset),
                            C.c_int64(offset.size - 1), _p(done), _p(last_index), _p(lengths),
                            _p(out))
    return out


def unfinished_index(offset, done, last_index, lengths) -> np.ndarray:
    """buffer_base.py:314-317 + manager.py:85-91."""
    offset, last_index, lengths = _i64(offset), _i64(last_index), _i64(lengths)
    done = _u8(done)
    E = offset.size - 1
    out = np.zeros(max(E, 1), np.int64)
    k = lib().oracle_unfinished_index(_p(offset), C.c_int64(E), _p(done), _p(last_index),
                                      _p(lengths), _p(out))
    return out[:k].copy()


def sample_indices_all(offset, lengths, insertion_idx) -> np.ndarray:
    """manager.py:216-234 with batch_size == 0 (buffer_base.py:518-525 per sub-buffer)."""
    offset, lengths, insertion_idx = _i64(offset), _i64(lengths), _i64(insertion_idx)
    E = offset.size - 1
    out = np.zeros(max(int(lengths.sum()), 1), np.int64)
    k = lib().oracle_sample_indices_all(_p(offset), C.c_int64(E), _p(lengths),
                                        _p(insertion_idx), _p(out))
    return out[:k].copy()


def _setitem(tree: np.ndarray, index, value) -> None:
    """segtree.py:95-101 (tree float64[2*bound] mutated in place; index includes +bound)."""
    assert tree.dtype == np.float64 and tree.flags.c_contiguous
    index, value = _i64(index), _f64(value)
    lib().oracle_segtree_setitem(_p(tree), _p(index), _p(value), C.c_int64(index.size))


def _reduce(tree: np.ndarray, start: int, end: int) -> float:
    """segtree.py:104-116."""
    assert tree.dtype == np.float64 and tree.flags.c_contiguous
    return float(lib().oracle_segtree_reduce(_p(tree), C.c_int64(start), C.c_int64(end)))


def _get_prefix_sum_idx(value: np.ndarray, bound: int, sums: np.ndarray) -> np.ndarray:
    """segtree.py:119-134 (mutates ``value`` in place like the reference)."""
    assert value.dtype == np.float64 and value.flags.c_contiguous
    assert sums.dtype == np.float64 and sums.flags.c_contiguous
    out = np.zeros(value.shape, np.int64)
    lib().oracle_segtree_prefix_sum_idx(_p(value), C.c_int64(value.size), C.c_int64(bound),
                                        _p(sums), _p(out))
    return out


def per_get_weight(tree, bound: int, index, min_prio: float, beta: float,
                   weight_norm: bool = True) -> np.ndarray:
    """prio.py:69-79 + :104-106."""
    index = _i64(index)
    out = np.zeros(index.shape, np.float64)
    lib().oracle_per_get_weight(_p(tree), C.c_int64(bound), _p(index), C.c_int64(index.size),
                                C.c_double(min_prio), C.c_double(beta), C.c_int(int(weight_norm)),
                                _p(out))
    return out


def per_update_weight(tree, bound: int, index, new_weight, alpha: float, max_prio: float,
                      min_prio: float, f32_math: bool = True):
    """prio.py:81-90 -> (max_prio, min_prio); tree mutated in place."""
    index, nw = _i64(index), _f64(new_weight)
    mx, mn = C.c_double(max_prio), C.c_double(min_prio)
    eps = float(np.finfo(np.float32).eps)
    lib().oracle_per_update_weight(_p(tree), C.c_int64(bound), _p(index), _p(nw),
                                   C.c_int64(index.size), C.c_double(alpha), C.c_double(eps),
                                   C.c_int(int(f32_math)), C.byref(mx), C.byref(mn))
    return mx.value, mn.value


def rms_update(mean: float, var: float, count: float, x) -> tuple[float, float, float]:
    """statistics.py:99-114."""
    st = np.array([mean, var, count], np.float64)
    x = _f64(x)
    lib().oracle_rms_update(_p(st), _p(x), C.c_int64(x.size))
    return float(st[0]), float(st[1]), float(st[2])


# ---------------------------------------------------------------------------------------------
# NumPy glue around the leaves (same structure as the reference's static methods)
# ---------------------------------------------------------------------------------------------
class BufferState:
    """The replay-buffer state the hot path reads (ReplayBufferManager fields).

    offset      int64[E+1]  manager.py:50  (_extend_offset)
    last_index  int64[E]    manager.py:52,176
    lengths     int64[E]    manager.py:51,177
    insertion   int64[E]    child ReplayBuffer._insertion_idx (buffer_base.py:375)
    done/terminated/truncated  bool[B];  rew float64[B]  (buffer_base.py:492)
    """

    def __init__(self, offset, last_index, lengths, insertion, rew, terminated, truncated,
                 done=None):
        self.offset = _i64(offset)
        self.last_index = _i64(last_index)
        self.lengths = _i64(lengths)
        self.insertion = _i64(insertion)
        self.rew = _f64(rew)
        self.terminated = np.asarray(terminated).astype(bool)
        self.truncated = np.asarray(truncated).astype(bool)
        self.done = (self.terminated | self.truncated) if done is None else np.asarray(done).astype(bool)

    @classmethod
    def from_vector_fill(cls, rew, terminated, truncated, n_env: int):
        """A VectorReplayBuffer(B, n_env) filled completely, in time order, exactly once
        (each sub-buffer written slots 0..T-1; SURVEY 8d C2 layout)."""
        B = len(rew)
        T = B // n_env
        assert T * n_env == B
        offset = np.arange(n_env + 1) * T
        return cls(offset, offset[:-1] + T - 1, np.full(n_env, T), np.zeros(n_env, np.int64),
                   rew, terminated, truncated)

    def unfinished_index(self):
        return unfinished_index(self.offset, self.done, self.last_index, self.lengths)

    def sample_indices_all(self):
        return sample_indices_all(self.offset, self.lengths, self.insertion)

    def next(self, index):
        return _next_index(index, self.offset, self.done, self.last_index, self.lengths)

    def prev(self, index):
        return _prev_index(index, self.offset, self.done, self.last_index, self.lengths)


class BufferWriter:
    """Write-side state of a ReplayBufferManager (manager.py:131-198, buffer_base.py:360-418): per sub-buffer
    insertion slot, length, last index, running episode return / length / start; plus the scalar columns."""

    def __init__(self, offset):
        self.offset = _i64(offset)
        E, B = self.offset.size - 1, int(self.offset[-1])
        self.insertion, self.lengths = np.zeros(E, np.int64), np.zeros(E, np.int64)
        self.last_index = self.offset[:-1].copy()                       # manager.py:52
        self.ep_return, self.ep_len, self.ep_start = np.zeros(E), np.zeros(E, np.int64), np.zeros(E, np.int64)
        self.rew, self.terminated = np.zeros(B), np.zeros(B, np.uint8)
        self.truncated, self.done = np.zeros(B, np.uint8), np.zeros(B, np.uint8)

    def add(self, rew, terminated, truncated, buffer_ids=None):
        """-> (insertion index, episode return, episode length, episode start index), manager.py:193-198."""
        rew, term, trunc = _f64(rew), _u8(terminated), _u8(truncated)
        K = rew.size
        ids = None if buffer_ids is None else _i64(buffer_ids)
        idx, er = np.zeros(K, np.int64), np.zeros(K)
        el, es = np.zeros(K, np.int64), np.zeros(K, np.int64)
        lib().oracle_buffer_add(_p(ids) if ids is not None else None, C.c_int64(K), _p(rew), _p(term), _p(trunc),
                                _p(self.offset), _p(self.insertion), _p(self.lengths), _p(self.last_index),
                                _p(self.ep_return), _p(self.ep_len), _p(self.ep_start), _p(self.rew),
                                _p(self.terminated), _p(self.truncated), _p(self.done), _p(idx), _p(er), _p(el),
                                _p(es))
        return idx, er, el, es


def compute_episodic_return(rew, terminated, truncated, indices, unfinished, v_s_, v_s,
                            gamma: float = 0.99, gae_lambda: float = 0.95):
    """algorithm_base.py:653-719 on raw arrays -> (returns, advantage) float64[N].

    ``rew/terminated/truncated`` are the batch arrays (= buffer arrays gathered at
    ``indices``); ``unfinished`` = buffer.unfinished_index()."""
    rew = _f64(rew)
    n = rew.size
    if v_s_ is None:
        assert np.isclose(gae_lambda, 1.0)
        v_s_ = np.zeros_like(rew)
        vmasked = v_s_
    else:
        vmasked = np.asarray(v_s_).reshape(-1)
    if v_s is None:
        # np.roll(v_s_ * mask, 1)  (algorithm_base.py:712)
        v_s = np.roll(_f64(vmasked) * (~np.asarray(terminated).astype(bool)), 1)
    v_s = _f64(np.asarray(v_s).reshape(-1))
    vnext = _f64(vmasked)
    term, trunc = _u8(terminated), _u8(truncated)
    idx, unf = _i64(indices), _i64(unfinished)
    ret = np.zeros(n, np.float64)
    adv = np.zeros(n, np.float64)
    lib().oracle_compute_episodic_return(_p(v_s), _p(vnext), _p(rew), _p(term), _p(trunc),
                                         _p(idx), C.c_int64(n), _p(unf), C.c_int64(unf.size),
                                         C.c_double(gamma), C.c_double(gae_lambda), _p(ret),
                                         _p(adv))
    return ret, adv


def compute_nstep_return(state: BufferState, indices, target_q_fn, gamma: float = 0.99,
                         n_step: int = 1):
    """algorithm_base.py:721-817 on a BufferState -> (returns float64[I,A], indices_after_n)."""
    indices = _i64(indices)
    stack = [indices]
    for _ in range(n_step - 1):
        stack.append(state.next(stack[-1]))
    stacked = np.stack(stack)
    after = stacked[-1]
    tq = np.asarray(target_q_fn(after), dtype=np.float32).reshape(len(indices), -1)
    tq = tq * (~state.terminated[after]).reshape(-1, 1)            # value_mask, :798
    end_flag = state.done.copy()                                       # :799
    end_flag[state.unfinished_index()] = True                          # :800
    return _nstep_return(state.rew, end_flag, tq.astype(np.float32), stacked, gamma, n_step), after


def sample_indices_random(offset, lengths, u_buffer, within) -> np.ndarray:
    """ReplayBufferManager.sample_indices(batch_size > 0), stack_num == 1 (manager.py:216-234; children:
    buffer_base.py:514-517) with the random draws as inputs: `u_buffer` = the uniforms RandomState.choice(E, bs, p)
    consumes (legacy choice: cdf = p.cumsum(); cdf /= cdf[-1]; cdf.searchsorted(u, side="right")), `within` = the
    children's choice(len_e, n_e) draws concatenated in sub-buffer order."""
    offset, lengths = _i64(offset), _i64(lengths)
    E = lengths.size
    p = lengths / lengths.sum()
    cdf = p.cumsum()
    cdf /= cdf[-1]
    buffer_idx = cdf.searchsorted(np.asarray(u_buffer, np.float64), side="right")
    sample_num = np.bincount(buffer_idx, minlength=E)
    within = _i64(within)
    out, pos = [], 0
    for e in range(E):
        n = int(sample_num[e])
        if n == 0:                       # manager.py:227-228: -1 -> the child returns an empty array
            continue
        out.append(offset[e] + within[pos:pos + n])
        pos += n
    return np.concatenate(out) if out else np.array([], np.int64)


def sample_indices_stack(state: "BufferState", insertion_idx, stack_num: int, positions=None) -> np.ndarray:
    """ReplayBufferManager.sample_indices with `stack_num > 1 and sample_avail` (manager.py:205-216; per sub-buffer
    buffer_base.py:532-545): all indices in ring order, minus those whose (stack_num - 2)-fold predecessor equals its own
    predecessor, i.e. that have fewer than stack_num - 1 earlier frames in their episode.  `positions` (None: batch_size
    0) = the draws of `RandomState.choice(all_indices, bs)` as positions into the available indices."""
    all_idx = sample_indices_all(state.offset, state.lengths, insertion_idx)
    p = all_idx
    for _ in range(stack_num - 2):
        p = state.prev(p)
    avail = all_idx[p != state.prev(p)]
    return avail if positions is None else avail[_i64(positions)]
