"""Optimizer-state fidelity between the engines' flat vectors and the reference's torch objects.

`Algorithm.state_dict()` (tianshou/algorithm/algorithm_base.py:523-543) covers the nn.Modules and every
torch.optim optimizer the algorithm created.  The engines keep parameters and Adam moments as flat vectors in
kernel-friendly layouts; these helpers move the Adam state of an ordered parameter list in and out of a
torch.optim.Adam so that a run can be resumed on the engine from a reference checkpoint and vice versa.
Layout permutations themselves live next to each engine (`flat_from_torch` / `flat_to_torch`).
"""
from __future__ import annotations

import torch


def _state_keys(opt: torch.optim.Optimizer) -> tuple[str | None, str]:
    """Names of the (auxiliary, second-moment) state tensors the engines keep in their `adam_m` / `adam_v` vectors:
    torch.optim.Adam: exp_avg / exp_avg_sq; torch.optim.RMSprop: momentum_buffer (momentum > 0) or grad_avg (centered) /
    square_avg (torch/optim/rmsprop.py `_init_group`)."""
    if type(opt).__name__ == "RMSprop":
        g = opt.param_groups[0]
        return ("momentum_buffer" if g.get("momentum", 0) > 0 else ("grad_avg" if g.get("centered", False) else None)), "square_avg"
    return "exp_avg", "exp_avg_sq"


def adam_state(opt: torch.optim.Optimizer, params: list[torch.nn.Parameter]):
    """-> (exp_avg tensors, exp_avg_sq tensors, step) of `params` in order; zeros / 0 for a fresh optimizer.
    All parameters of one optimizer step together (torch.optim.Adam), so a single step count is returned.
    For torch.optim.RMSprop the pair is (momentum_buffer or grad_avg, square_avg), see `_state_keys`."""
    ms, vs, step = [], [], 0
    km, kv = _state_keys(opt)
    for p in params:
        st = opt.state.get(p, {})
        ms.append(st[km].detach().clone() if km in st else torch.zeros_like(p))
        vs.append(st[kv].detach().clone() if kv in st else torch.zeros_like(p))
        if "step" in st:
            step = max(step, int(float(st["step"])))
    return ms, vs, step


def store_adam_state(opt: torch.optim.Optimizer, params: list[torch.nn.Parameter], ms, vs, step: int) -> None:
    """Writes Adam moments / step of `params` (in order) into `opt.state` (tensors are copied to each parameter's
    device and shape)."""
    km, kv = _state_keys(opt)
    for p, m, v in zip(params, ms, vs):
        st = opt.state[p]
        st["step"] = torch.tensor(float(step))
        if km is not None:
            st[km] = m.detach().reshape(p.shape).to(p.device, p.dtype).clone()
        st[kv] = v.detach().reshape(p.shape).to(p.device, p.dtype).clone()


def params_by_keys(module: torch.nn.Module, keys) -> list[torch.nn.Parameter]:
    named = dict(module.named_parameters())
    return [named[k] for k in keys]
