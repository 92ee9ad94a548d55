"""Target-network updates on flat parameter vectors (tianshou/utils/lagged_network.py:8-18, 81-87)."""
from __future__ import annotations

import torch

from . import _lib


def polyak_parameter_update(tgt: torch.Tensor, src: torch.Tensor, tau: float) -> None:
    """tgt <- tau * src + (1 - tau) * tgt, in place (float32 device tensors, same rounding as torch)."""
    if tgt.shape != src.shape or tgt.dtype != torch.float32 or src.dtype != torch.float32:
        raise ValueError("tgt / src must be float32 tensors of the same shape")
    if not (tgt.is_contiguous() and src.is_contiguous()):
        raise ValueError("tgt / src must be contiguous")
    _lib.check(_lib.load().ts_polyak_update(_lib.ptr(tgt), _lib.ptr(src), _lib.i64(tgt.numel()),
                                            _lib.f64(tau), _lib.current_stream(tgt.device)))


def full_parameter_update(tgt: torch.Tensor, src: torch.Tensor) -> None:
    polyak_parameter_update(tgt, src, 1.0)
