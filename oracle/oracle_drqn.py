"""TEST INFRASTRUCTURE ONLY - CPU (torch fp32 autograd) restatement of the reference's DQN on a recurrent Q network (DRQN).

Never imported by the product (`tianshou_amd/`).  Pinned against the UNMODIFIED reference through tests/golden/drqn_*.npz
(oracle/gen_golden.py::gen_drqn).

Follows:
  net         Recurrent.forward utils/net/common.py:400-452: fc1 -> nn.LSTM(H, H, L, batch_first) from a zero state -> fc2 on
              the last step; obs [B, T, dim] with T = the buffer's stack_num (test/discrete/test_drqn.py:79-101).  The LSTM cell
              is written out (torch's gate order i, f, g, o): gates = x W_ih^T + b_ih + h W_hh^T + b_hh,
              c' = sigmoid(f) c + sigmoid(i) tanh(g), h' = sigmoid(o) tanh(c')
  stacking    ReplayBuffer.get with stack_num buffer_base.py:586-596 (oracle_dqn.stacked_frames); obs_next through next()
              when the buffer ignores obs_next (:624-626)
  target      DQN._target_q dqn.py:365-379; n-step returns algorithm_base.py:721-817 (oracle.compute_nstep_return)
  update      DQN._update_with_batch dqn.py:381-404 + periodic hard sync :277-285; Optimizer.step algorithm_base.py:484-500
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O
from . import oracle_dqn as OD


def param_keys(layers: int) -> list[str]:
    """Recurrent.state_dict() order: the LSTM is constructed first (common.py:386-393)."""
    ks = []
    for k in range(layers):
        ks += [f"nn.weight_ih_l{k}", f"nn.weight_hh_l{k}", f"nn.bias_ih_l{k}", f"nn.bias_hh_l{k}"]
    return ks + ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]


def param_shapes(obs_dim: int, hidden: int, layers: int, n_act: int) -> dict[str, tuple[int, ...]]:
    s = {}
    for k in range(layers):
        s[f"nn.weight_ih_l{k}"], s[f"nn.weight_hh_l{k}"] = (4 * hidden, hidden), (4 * hidden, hidden)
        s[f"nn.bias_ih_l{k}"], s[f"nn.bias_hh_l{k}"] = (4 * hidden,), (4 * hidden,)
    s["fc1.weight"], s["fc1.bias"], s["fc2.weight"], s["fc2.bias"] = (hidden, obs_dim), (hidden,), (n_act, hidden), (n_act,)
    return s


def unflatten(flat, obs_dim: int, hidden: int, layers: int, n_act: int) -> dict[str, torch.Tensor]:
    flat = torch.as_tensor(np.asarray(flat), dtype=torch.float32)
    shapes, out, off = param_shapes(obs_dim, hidden, layers, n_act), {}, 0
    for k in param_keys(layers):
        n = int(np.prod(shapes[k]))
        out[k] = flat[off:off + n].reshape(shapes[k]).clone()
        off += n
    assert off == flat.numel()
    return out


def flatten(p: dict, layers: int) -> torch.Tensor:
    return torch.cat([p[k].detach().reshape(-1) for k in param_keys(layers)])


def n_layers(p: dict) -> int:
    return sum(1 for k in p if k.startswith("nn.weight_ih_l"))


def forward(p: dict, obs, state=None, want_state: bool = False):
    """obs [B, T, dim] (or [B, dim]) -> Q [B, n_act]; state = (hidden, cell) each [L, B, H]."""
    x = torch.as_tensor(np.asarray(obs) if not isinstance(obs, torch.Tensor) else obs, dtype=torch.float32)
    if x.dim() == 2:
        x = x.unsqueeze(-2)
    x = F.linear(x, p["fc1.weight"], p["fc1.bias"])
    B, T, H = x.shape
    hs, cs = [], []
    for k in range(n_layers(p)):
        h = torch.zeros(B, H) if state is None else state[0][k]
        c = torch.zeros(B, H) if state is None else state[1][k]
        outs = []
        for t in range(T):
            gates = F.linear(x[:, t], p[f"nn.weight_ih_l{k}"], p[f"nn.bias_ih_l{k}"]) + \
                F.linear(h, p[f"nn.weight_hh_l{k}"], p[f"nn.bias_hh_l{k}"])
            i, f, g, o = gates.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        x = torch.stack(outs, dim=1)
        hs.append(h)
        cs.append(c)
    q = F.linear(x[:, -1], p["fc2.weight"], p["fc2.bias"])
    return (q, (torch.stack(hs), torch.stack(cs))) if want_state else q


def target_q(st: OD.DQNState, cfg: OD.DQNConfig, obs_next) -> torch.Tensor:
    with torch.no_grad():
        q_online = forward(st.params, obs_next)
        act = q_online.argmax(dim=1)
        tq = forward(st.params_old, obs_next) if st.params_old is not None else q_online
        return tq[torch.arange(len(act)), act] if cfg.is_double else tq.max(dim=1)[0]


def preprocess(st: OD.DQNState, cfg: OD.DQNConfig, bstate: O.BufferState, obs_rows: np.ndarray, indices, stack_num: int):
    """DQN._preprocess_batch (dqn.py:257-275) on a buffer with ignore_obs_next (test_drqn.py:103-108)."""

    def tq_fn(after):
        on = OD.stacked_frames(bstate, obs_rows, bstate.next(after), stack_num)
        return target_q(st, cfg, on).numpy().reshape(-1, 1)

    ret, _ = O.compute_nstep_return(bstate, indices, tq_fn, cfg.gamma, cfg.n_step)
    return ret.astype(np.float32).reshape(-1)


def update_with_batch(st: OD.DQNState, cfg: OD.DQNConfig, obs, act, returns, weight=None, collect: dict | None = None):
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
    OD._adam(st, cfg, grads)
    return float(loss.item()), td.detach().clone()
