"""TEST INFRASTRUCTURE ONLY - CPU (torch fp32) restatement of the reference's DQN learn() path.

Never imported by the product (`tianshou_amd/`); only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg use it, as the checker.  Pinned against the UNMODIFIED reference
through tests/golden/dqn_*.npz (oracle/gen_golden.py::gen_dqn).

Follows, line by line:
  DQNet                       tianshou/env/atari/atari_network.py:60-122 (NatureCNN, no /255)
  DiscreteQLearningPolicy.forward   algorithm/modelfree/dqn.py:101-143   (act = argmax_a Q)
  DQN._target_q               dqn.py:365-379   (double-Q / max over the target net)
  DQN._update_with_batch      dqn.py:381-404   (Huber mean | (td^2 * w).mean(); batch.weight = td)
  periodic hard sync          dqn.py:277-285   (iter % target_update_freq == 0, BEFORE the step)
  Optimizer.step              algorithm_base.py:484-500 (optional clip_grad_norm_, then Adam)
  ReplayBuffer.get (frame stack)  data/buffer/buffer_base.py:557-603
  compute_nstep_return        algorithm_base.py:721-817 (via oracle.compute_nstep_return)
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O

PARAM_ORDER = ["conv1.w", "conv1.b", "conv2.w", "conv2.b", "conv3.w", "conv3.b",
               "fc1.w", "fc1.b", "fc2.w", "fc2.b"]
# reference state_dict keys (nn.Sequential nesting of DQNet.net, atari_network.py:79-98)
TIANSHOU_KEYS = ["net.0.0.weight", "net.0.0.bias", "net.0.2.weight", "net.0.2.bias",
                 "net.0.4.weight", "net.0.4.bias", "net.1.weight", "net.1.bias",
                 "net.3.weight", "net.3.bias"]
CONVS = [(32, 8, 4), (64, 4, 2), (64, 3, 1)]          # (out channels, kernel, stride)
HIDDEN = 512


def conv_out_hw(h: int, w: int) -> list[tuple[int, int]]:
    out = []
    for _, k, s in CONVS:
        h, w = (h - k) // s + 1, (w - k) // s + 1
        out.append((h, w))
    return out


def param_shapes(c: int, h: int, w: int, n_act: int) -> dict[str, tuple[int, ...]]:
    oh, ow = conv_out_hw(h, w)[-1]
    return {
        "conv1.w": (32, c, 8, 8), "conv1.b": (32,),
        "conv2.w": (64, 32, 4, 4), "conv2.b": (64,),
        "conv3.w": (64, 64, 3, 3), "conv3.b": (64,),
        "fc1.w": (HIDDEN, 64 * oh * ow), "fc1.b": (HIDDEN,),
        "fc2.w": (n_act, HIDDEN), "fc2.b": (n_act,),
    }


def param_count(c: int, h: int, w: int, n_act: int) -> int:
    return int(sum(np.prod(s) for s in param_shapes(c, h, w, n_act).values()))


def init_params(c: int, h: int, w: int, n_act: int, seed: int = 0) -> dict[str, torch.Tensor]:
    """Same RNG consumption as `torch.manual_seed(seed); DQNet(c, h, w, n_act)` with the default
    (identity) layer_init: five torch modules constructed in this order (atari_network.py:79-98)."""
    torch.manual_seed(seed)
    oh, ow = conv_out_hw(h, w)[-1]
    mods = [torch.nn.Conv2d(c, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1),
            torch.nn.Linear(64 * oh * ow, HIDDEN), torch.nn.Linear(HIDDEN, n_act)]
    p = {}
    for name, m in zip(["conv1", "conv2", "conv3", "fc1", "fc2"], mods):
        p[name + ".w"] = m.weight.detach().clone()
        p[name + ".b"] = m.bias.detach().clone()
    return p


def flatten_params(p: dict[str, torch.Tensor]) -> torch.Tensor:
    return torch.cat([p[k].reshape(-1) for k in PARAM_ORDER])


def unflatten_params(flat: torch.Tensor, c: int, h: int, w: int, n_act: int) -> dict[str, torch.Tensor]:
    out, off = {}, 0
    for k, s in param_shapes(c, h, w, n_act).items():
        n = int(np.prod(s))
        out[k] = flat[off:off + n].reshape(s).clone()
        off += n
    return out


def forward(p: dict[str, torch.Tensor], obs) -> torch.Tensor:
    """DQNet.forward: obs u8/f32 [B, C, H, W] -> Q [B, A] (atari_network.py:111-122)."""
    x = torch.as_tensor(np.asarray(obs) if not isinstance(obs, torch.Tensor) else obs, dtype=torch.float32)
    x = F.relu(F.conv2d(x, p["conv1.w"], p["conv1.b"], stride=4))
    x = F.relu(F.conv2d(x, p["conv2.w"], p["conv2.b"], stride=2))
    x = F.relu(F.conv2d(x, p["conv3.w"], p["conv3.b"], stride=1))
    x = F.relu(F.linear(x.flatten(1), p["fc1.w"], p["fc1.b"]))
    return F.linear(x, p["fc2.w"], p["fc2.b"])


@dataclass
class DQNConfig:
    gamma: float = 0.99
    n_step: int = 1
    target_update_freq: int = 0
    is_double: bool = True
    huber_delta: float | None = None
    lr: float = 1e-3
    betas: tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8
    max_grad_norm: float | None = None


@dataclass
class DQNState:
    params: dict[str, torch.Tensor]
    params_old: dict[str, torch.Tensor] | None = None
    adam_m: dict[str, torch.Tensor] = field(default_factory=dict)
    adam_v: dict[str, torch.Tensor] = field(default_factory=dict)
    adam_step: int = 0
    iter: int = 0

    @classmethod
    def create(cls, params, cfg: DQNConfig):
        old = {k: v.clone() for k, v in params.items()} if cfg.target_update_freq > 0 else None
        return cls(params={k: v.clone() for k, v in params.items()}, params_old=old,
                   adam_m={k: torch.zeros_like(v) for k, v in params.items()},
                   adam_v={k: torch.zeros_like(v) for k, v in params.items()})


def stacked_frames(state: O.BufferState, frames: np.ndarray, index, stack_num: int) -> np.ndarray:
    """ReplayBuffer.get(index, "obs") with stack_num > 1 (buffer_base.py:586-596):
    [obs[prev^(s-1)(i)], ..., obs[prev(i)], obs[i]] stacked on axis 1."""
    idx = np.asarray(index, np.int64)
    if stack_num == 1:
        return frames[idx]
    stack = []
    for _ in range(stack_num):
        stack = [frames[idx], *stack]
        idx = state.prev(idx)
    return np.stack(stack, axis=1)


def target_q(st: DQNState, cfg: DQNConfig, obs_next) -> torch.Tensor:
    """dqn.py:365-379 -> [B] float32."""
    with torch.no_grad():
        q_online = forward(st.params, obs_next)
        act = q_online.argmax(dim=1)                                     # dqn.py:141
        tq = forward(st.params_old, obs_next) if st.params_old is not None else q_online
        if cfg.is_double:
            return tq[torch.arange(len(act)), act]
        return tq.max(dim=1)[0]


def _adam(st: DQNState, cfg: DQNConfig, grads: dict[str, torch.Tensor]) -> None:
    """clip_grad_norm_ (algorithm_base.py:496-499) + torch.optim.Adam single-tensor math."""
    if cfg.max_grad_norm:
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
        coef = torch.clamp(cfg.max_grad_norm / (total + 1e-6), max=1.0)
        grads = {k: g * coef for k, g in grads.items()}
    st.adam_step += 1
    b1, b2 = cfg.betas
    bc1 = 1 - b1 ** st.adam_step
    bc2 = 1 - b2 ** st.adam_step
    for k, g in grads.items():
        m, v = st.adam_m[k], st.adam_v[k]
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / np.sqrt(bc2)).add_(cfg.adam_eps)
        st.params[k] = st.params[k] - (cfg.lr / bc1) * (m / denom)


def update_with_batch(st: DQNState, cfg: DQNConfig, obs, act, returns, weight=None,
                      collect: dict | None = None):
    """dqn.py:381-404 -> (loss float, td_error float32[B])."""
    if st.params_old is not None and st.iter % cfg.target_update_freq == 0:     # dqn.py:283-285
        st.params_old = {k: v.clone() for k, v in st.params.items()}
    st.iter += 1
    p = {k: v.clone().requires_grad_(True) for k, v in st.params.items()}
    q_all = forward(p, obs)
    act_t = torch.as_tensor(np.asarray(act), dtype=torch.int64)
    q = q_all[torch.arange(len(act_t)), act_t]
    ret = torch.as_tensor(np.asarray(returns), dtype=torch.float32).flatten()
    td = ret - q
    if cfg.huber_delta is not None:
        loss = F.huber_loss(q.reshape(-1, 1), ret.reshape(-1, 1), delta=cfg.huber_delta, reduction="mean")
    else:
        w = 1.0 if weight is None else torch.as_tensor(np.asarray(weight), dtype=torch.float32)
        loss = (td.pow(2) * w).mean()
    loss.backward()
    grads = {k: v.grad for k, v in p.items()}
    if collect is not None:
        collect["q_all"] = q_all.detach().clone()
        collect["grads"] = {k: g.clone() for k, g in grads.items()}
    _adam(st, cfg, grads)
    return float(loss.item()), td.detach().clone()


def preprocess(st: DQNState, cfg: DQNConfig, bstate: O.BufferState, frames: np.ndarray, indices,
               stack_num: int, obs_next_frames: np.ndarray | None = None):
    """DQN._preprocess_batch (dqn.py:257-275): n-step returns with target_q_fn = _target_q.
    obs_next is read through next() when the buffer ignores obs_next (buffer_base.py:624-626)."""

    def tq_fn(after):
        if obs_next_frames is None:
            on = stacked_frames(bstate, frames, bstate.next(after), stack_num)
        else:
            on = stacked_frames(bstate, obs_next_frames, after, stack_num)
        return target_q(st, cfg, on).numpy().reshape(-1, 1)

    ret, _ = O.compute_nstep_return(bstate, indices, tq_fn, cfg.gamma, cfg.n_step)
    return ret.astype(np.float32).reshape(-1)        # to_torch_as(target_q_IA, ...) :811
