"""Optimizer-state fidelity between the engines' flat vectors and the reference's torch objects.

`Algorithm.state_dict()` (tianshou/algorithm/algorithm_base.py:523-543) covers the nn.Modules and every
torch.optim optimizer the algorithm created.  The engines keep parameters and Adam moments as flat vectors in
kernel-friendly layouts; these helpers move the Adam state of an ordered parameter list in and out of a
torch.optim.Adam so that a run can be resumed on the engine from a reference checkpoint and vice versa.
Layout permutations themselves live next to each engine (`flat_from_torch` / `flat_to_torch`).
"""
from __future__ import annotations

import torch


def adam_state(opt: torch.optim.Optimizer, params: list[torch.nn.Parameter]):
    """-> (exp_avg tensors, exp_avg_sq tensors, step) of `params` in order; zeros / 0 for a fresh optimizer.
    All parameters of one optimizer step together (torch.optim.Adam), so a single step count is returned."""
    ms, vs, step = [], [], 0
    for p in params:
        st = opt.state.get(p, {})
        ms.append(st["exp_avg"].detach().clone() if "exp_avg" in st else torch.zeros_like(p))
        vs.append(st["exp_avg_sq"].detach().clone() if "exp_avg_sq" in st else torch.zeros_like(p))
        if "step" in st:
            step = max(step, int(float(st["step"])))
    return ms, vs, step


def store_adam_state(opt: torch.optim.Optimizer, params: list[torch.nn.Parameter], ms, vs, step: int) -> None:
    """Writes Adam moments / step of `params` (in order) into `opt.state` (tensors are copied to each parameter's
    device and shape)."""
    for p, m, v in zip(params, ms, vs):
        st = opt.state[p]
        st["step"] = torch.tensor(float(step))
        st["exp_avg"] = m.detach().reshape(p.shape).to(p.device, p.dtype).clone()
        st["exp_avg_sq"] = v.detach().reshape(p.shape).to(p.device, p.dtype).clone()


def params_by_keys(module: torch.nn.Module, keys) -> list[torch.nn.Parameter]:
    named = dict(module.named_parameters())
    return [named[k] for k in keys]
