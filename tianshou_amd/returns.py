"""GAE / n-step returns on the MI355X: host-side mirror of the reference interface.

Same names, argument meaning and error behaviour as
    Algorithm.compute_episodic_return / _gae        tianshou/algorithm/algorithm_base.py:653-719, 1085-1140
    Algorithm.compute_nstep_return / _nstep_return  tianshou/algorithm/algorithm_base.py:721-817, 1160-1222
with torch device tensors instead of NumPy arrays and a DeviceReplayBuffer instead of a
ReplayBuffer.  All arithmetic happens in libtsengine's HIP kernels (tianshou_amd/csrc/ts_returns.hip).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .buffer import DeviceReplayBuffer, _dev_index, _i64_dev, _u8_dev


def _f32(x, device) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float32).reshape(-1).contiguous()
    return torch.as_tensor(np.asarray(x, dtype=np.float32).reshape(-1), device=device)


def gae_scan(v_s, v_s_next, rew, terminated, truncated, cut_pos=None, *, gamma=0.99,
             gae_lambda=0.95, v_scale=1.0, ret_div=1.0, want_f64=False, want_ret_stats=False,
             d_n_cut=None, ws=None):
    """Fused compute_episodic_return (ts_gae_scan).  Returns dict(adv, returns[, adv64, ret64,
    ret_sum, ret_sumsq]); adv/returns are float32 like the reference's to_torch_as casts."""
    dev = v_s.device
    v_s = _f32(v_s, dev)
    v_s_next = _f32(v_s_next, dev)
    n = v_s.numel()
    if isinstance(rew, torch.Tensor):
        rew = rew.to(dev).reshape(-1)
        if rew.dtype not in (torch.float32, torch.float64):
            rew = rew.to(torch.float64)
        rew = rew.contiguous()
    else:
        rew = torch.as_tensor(np.asarray(rew, dtype=np.float64).reshape(-1), device=dev)
    terminated, truncated = _u8_dev(terminated, dev).reshape(-1), _u8_dev(truncated, dev).reshape(-1)
    if not (v_s_next.numel() == rew.numel() == terminated.numel() == truncated.numel() == n):
        raise ValueError("gae_scan: input length mismatch")
    n_cut = 0
    if cut_pos is not None:
        cut_pos = _i64_dev(cut_pos, dev).reshape(-1)
        n_cut = cut_pos.numel()
    adv = torch.empty(n, dtype=torch.float32, device=dev)
    ret = torch.empty(n, dtype=torch.float32, device=dev)
    adv64 = torch.empty(n, dtype=torch.float64, device=dev) if want_f64 else None
    ret64 = torch.empty(n, dtype=torch.float64, device=dev) if want_f64 else None
    lib = _lib.load()
    parts = None
    if want_ret_stats:
        parts = torch.zeros(2 * max(int(lib.ts_gae_num_tiles(n)), 1), dtype=torch.float64, device=dev)
    if ws is None:
        ws = _lib.default_workspace(_dev_index(v_s))
    _lib.check(lib.ts_gae_scan(
        ws.handle, _lib.ptr(v_s), _lib.ptr(v_s_next), _lib.ptr(rew),
        0 if rew.dtype == torch.float32 else 1, _lib.ptr(terminated), _lib.ptr(truncated),
        _lib.ptr(cut_pos), _lib.i64(n_cut), _lib.ptr(d_n_cut), _lib.i64(n), _lib.f64(gamma),
        _lib.f64(gae_lambda), _lib.f64(v_scale), _lib.f64(ret_div), _lib.ptr(adv), _lib.ptr(ret),
        _lib.ptr(adv64), _lib.ptr(ret64), _lib.ptr(parts), _lib.current_stream(dev)))
    out = {"adv": adv, "returns": ret}
    if want_f64:
        out["adv64"], out["ret64"] = adv64, ret64
    if want_ret_stats:
        p = parts.view(-1, 2).sum(dim=0)
        out["ret_sum"], out["ret_sumsq"] = p[0], p[1]
    return out


def _gae(v_s, v_s_, rew, end_flag, gamma: float, gae_lambda: float) -> torch.Tensor:
    """Drop-in for the njit leaf (algorithm_base.py:1085-1140): float64 advantages.
    `end_flag` already contains every cut, so it is passed as `truncated` with no terminations
    (the leaf does not mask v_s_)."""
    dev = v_s.device if isinstance(v_s, torch.Tensor) else torch.device("cuda")
    end = _u8_dev(end_flag, dev)
    zeros = torch.zeros_like(end)
    return gae_scan(_f32(v_s, dev), _f32(v_s_, dev), rew, zeros, end, gamma=gamma,
                    gae_lambda=gae_lambda, want_f64=True)["adv64"]


def cut_positions(buffer: DeviceReplayBuffer, indices: torch.Tensor):
    """Batch positions p with indices[p] in buffer.unfinished_index() (algorithm_base.py:715).
    Returns (cut_pos int64[E] device, d_n_cut int64[1] device) without a host round trip."""
    dev = buffer.device
    unf, n_unf = buffer._unfinished_raw()
    indices = _i64_dev(indices, dev).reshape(-1)
    # unfinished_index() is ascending; only the first *n_unf entries are valid, the rest is
    # padded with a value no index can equal so that the binary search stays correct.
    pad = torch.full_like(unf, torch.iinfo(torch.int64).max)
    ar = torch.arange(unf.numel(), device=dev)
    unf_sorted = torch.where(ar < n_unf, unf, pad)
    cut = torch.empty(max(unf.numel(), 1), dtype=torch.int64, device=dev)
    d_n = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(_lib.load().ts_isin_positions(
        _lib.ptr(indices), _lib.i64(indices.numel()), _lib.ptr(unf_sorted), _lib.i64(unf.numel()),
        _lib.ptr(cut), _lib.i64(cut.numel()), _lib.ptr(d_n), _lib.current_stream(dev)))
    return cut, d_n


def compute_episodic_return(batch, buffer: DeviceReplayBuffer, indices, v_s_=None, v_s=None,
                            gamma: float = 0.99, gae_lambda: float = 0.95):
    """Algorithm.compute_episodic_return (algorithm_base.py:653-719).

    `batch` needs `.rew`, `.terminated`, `.truncated` (device tensors in batch order, i.e.
    buffer arrays gathered at `indices`).  Returns (returns, advantage) as float64 device
    tensors, like the reference's float64 NumPy arrays."""
    dev = buffer.device
    rew = batch.rew
    n = rew.numel()
    if v_s_ is None:
        if not np.isclose(gae_lambda, 1.0):
            raise AssertionError("v_s_ is None requires gae_lambda == 1.0")  # :706
        v_next = torch.zeros(n, dtype=torch.float32, device=dev)
    else:
        v_next = _f32(v_s_, dev)
    term = _u8_dev(batch.terminated, dev).reshape(-1)
    if v_s is None:
        # np.roll(v_s_ * value_mask, 1)  (:711-712)
        v_cur = torch.roll(v_next * (term == 0).to(torch.float32), 1)
    else:
        v_cur = _f32(v_s, dev)
    cut, d_n = cut_positions(buffer, indices)
    out = gae_scan(v_cur, v_next, rew, term, batch.truncated, cut, gamma=gamma,
                   gae_lambda=gae_lambda, want_f64=True, d_n_cut=d_n)
    return out["ret64"], out["adv64"]


def _nstep_return(rew_B, end_flag_B, target_q_IA, stacked_indices_NI, gamma: float,
                  n_step: int, want_f64: bool = False):
    """Drop-in for the njit leaf (algorithm_base.py:1160-1222).  float32 [I, A] (+ float64)."""
    dev = target_q_IA.device
    tq = target_q_IA.to(torch.float32).contiguous()
    I = tq.shape[0]
    A = tq.numel() // max(I, 1)
    rew = rew_B.to(device=dev, dtype=torch.float64).contiguous()
    end = _u8_dev(end_flag_B, dev)
    idx = _i64_dev(stacked_indices_NI, dev)
    if idx.shape != (n_step, I):
        raise ValueError(f"stacked indices shape {tuple(idx.shape)} != ({n_step}, {I})")
    out = torch.empty_like(tq)
    out64 = torch.empty(tq.shape, dtype=torch.float64, device=dev) if want_f64 else None
    _lib.check(_lib.load().ts_nstep_return(
        _lib.ptr(rew), _lib.ptr(end), _lib.ptr(tq), _lib.ptr(idx), _lib.i64(I), _lib.i64(A),
        _lib.i64(n_step), _lib.i64(rew.numel()), _lib.f64(gamma), _lib.ptr(out), _lib.ptr(out64),
        _lib.current_stream(dev)))
    return (out, out64) if want_f64 else out


def nstep_indices(buffer: DeviceReplayBuffer, indices, n_step: int, want_stack: bool = False):
    """indices_after_n_steps (algorithm_base.py:772-791)."""
    dev = buffer.device
    indices = _i64_dev(indices, dev).reshape(-1)
    I = indices.numel()
    after = torch.empty(I, dtype=torch.int64, device=dev)
    stack = torch.empty((n_step, I), dtype=torch.int64, device=dev) if want_stack else None
    _lib.check(_lib.load().ts_nstep_indices(
        _lib.ptr(indices), _lib.i64(I), _lib.i64(n_step), _lib.ptr(buffer.offset),
        _lib.i64(buffer.buffer_num), _lib.ptr(buffer.done), _lib.ptr(buffer.last_index),
        _lib.ptr(buffer.lengths), _lib.ptr(after), _lib.ptr(stack), _lib.current_stream(dev)))
    return (after, stack) if want_stack else after


def nstep_return_from_target_q(buffer: DeviceReplayBuffer, indices, tq: torch.Tensor, gamma: float, n_step: int):
    """The arithmetic half of compute_nstep_return (algorithm_base.py:793-812) for a caller that already holds
    target_q(s_{t+n}): float32 tensor shaped like `tq`.  The kernel re-walks next() from `indices` itself."""
    dev = buffer.device
    indices = _i64_dev(indices, dev).reshape(-1)
    I = indices.numel()
    tq2 = tq.to(torch.float32).reshape(I, -1).contiguous()
    out = torch.empty_like(tq2)
    _lib.check(_lib.load().ts_nstep_return_fused(
        _lib.ptr(indices), _lib.i64(I), _lib.i64(n_step), _lib.ptr(buffer.offset),
        _lib.i64(buffer.buffer_num), _lib.ptr(buffer.done), _lib.ptr(buffer.terminated),
        _lib.ptr(buffer.last_index), _lib.ptr(buffer.lengths), _lib.ptr(buffer.rew), _lib.ptr(tq2),
        _lib.i64(tq2.shape[1]), _lib.f64(gamma), _lib.ptr(out), None, _lib.current_stream(dev)))
    return out.reshape(tq.shape)


def nstep_coefficients(buffer: DeviceReplayBuffer, indices, gamma: float, n_step: int):
    """(mask float32[I], gpow float64[I], mc float64[I]) with returns = float(double(target_q * mask) * gpow + mc): the part
    of compute_nstep_return (algorithm_base.py:772-811) that does not need target_q (ts_nstep_coefficients)."""
    dev = buffer.device
    indices = _i64_dev(indices, dev).reshape(-1)
    I = indices.numel()
    mask = torch.empty(I, dtype=torch.float32, device=dev)
    gpow = torch.empty(I, dtype=torch.float64, device=dev)
    mc = torch.empty(I, dtype=torch.float64, device=dev)
    _lib.check(_lib.load().ts_nstep_coefficients(
        _lib.ptr(indices), _lib.i64(I), _lib.i64(n_step), _lib.ptr(buffer.offset), _lib.i64(buffer.buffer_num),
        _lib.ptr(buffer.done), _lib.ptr(buffer.terminated), _lib.ptr(buffer.last_index), _lib.ptr(buffer.lengths),
        _lib.ptr(buffer.rew), _lib.f64(gamma), _lib.ptr(mask), _lib.ptr(gpow), _lib.ptr(mc), _lib.current_stream(dev)))
    return mask, gpow, mc


def compute_nstep_return(batch, buffer: DeviceReplayBuffer, indices, target_q_fn,
                         gamma: float = 0.99, n_step: int = 1, want_f64: bool = False):
    """Algorithm.compute_nstep_return (algorithm_base.py:721-817): sets batch.returns (float32
    tensor shaped like target_q_fn's output) and returns the batch.

    target_q_fn(buffer, indices_after_n) -> float32 device tensor [I] or [I, A]."""
    dev = buffer.device
    indices = _i64_dev(indices, dev).reshape(-1)
    I = indices.numel()
    if hasattr(batch, "__len__") and len(batch) != I:
        raise ValueError(f"Batch size {len(batch)} and indices size {I} mismatch.")  # :757-758
    after = nstep_indices(buffer, indices, n_step)
    with torch.no_grad():
        tq = target_q_fn(buffer, after)
    tq2 = tq.to(torch.float32).reshape(I, -1).contiguous()
    A = tq2.shape[1]
    out = torch.empty_like(tq2)
    out64 = torch.empty(tq2.shape, dtype=torch.float64, device=dev) if want_f64 else None
    _lib.check(_lib.load().ts_nstep_return_fused(
        _lib.ptr(indices), _lib.i64(I), _lib.i64(n_step), _lib.ptr(buffer.offset),
        _lib.i64(buffer.buffer_num), _lib.ptr(buffer.done), _lib.ptr(buffer.terminated),
        _lib.ptr(buffer.last_index), _lib.ptr(buffer.lengths), _lib.ptr(buffer.rew), _lib.ptr(tq2),
        _lib.i64(A), _lib.f64(gamma), _lib.ptr(out), _lib.ptr(out64), _lib.current_stream(dev)))
    batch.returns = out.reshape(tq.shape)
    if want_f64:
        batch.returns64 = out64.reshape(tq.shape)
    if hasattr(batch, "weight") and batch.weight is not None:
        batch.weight = torch.as_tensor(batch.weight, dtype=torch.float32, device=dev)  # :814-815
    return batch
