"""TEST INFRASTRUCTURE ONLY - CPU (torch fp32 autograd) restatement of the reference's recurrent actor and critic.

Never imported by the product (`tianshou_amd/`).  Pinned against the UNMODIFIED reference classes through
tests/golden/recurrent_*.npz (oracle/gen_golden.py::gen_recurrent).

Follows:
  actor   RecurrentActorProb.forward utils/net/continuous.py:276-322: nn.LSTM(obs_dim, H, L, batch_first) on the observation
          ([B, T, dim], or [B, dim] -> one step), optional carried state {"hidden", "cell"} stored [B, L, H]; mu = Linear(H, A)
          on the last step, mu = max_action * tanh(mu) unless unbounded; sigma = exp(sigma_param) broadcast ([A, 1] parameter)
  critic  RecurrentCritic.forward continuous.py:346-380: the same trunk from a zero state, fc2 = Linear(H + A, 1) on
          cat([h_T, act])
The LSTM cell is written out (torch's gate order i, f, g, o), as in oracle_drqn.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def lstm_keys(layers: int) -> list[str]:
    ks = []
    for k in range(layers):
        ks += [f"nn.weight_ih_l{k}", f"nn.weight_hh_l{k}", f"nn.bias_ih_l{k}", f"nn.bias_hh_l{k}"]
    return ks


def actor_keys(layers: int) -> list[str]:
    """RecurrentActorProb.state_dict() order: sigma_param (a Parameter of the module itself) first, then nn.*, then mu.*."""
    return ["sigma_param", *lstm_keys(layers), "mu.weight", "mu.bias"]


def critic_keys(layers: int) -> list[str]:
    return [*lstm_keys(layers), "fc2.weight", "fc2.bias"]


def init(kind: str, obs_dim: int, act_dim: int, hidden: int, layers: int, seed: int = 0) -> dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / np.sqrt(hidden)
    u = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * k  # noqa: E731
    p = {}
    for l in range(layers):
        in_dim = obs_dim if l == 0 else hidden
        p[f"nn.weight_ih_l{l}"], p[f"nn.weight_hh_l{l}"] = u(4 * hidden, in_dim), u(4 * hidden, hidden)
        p[f"nn.bias_ih_l{l}"], p[f"nn.bias_hh_l{l}"] = u(4 * hidden), u(4 * hidden)
    if kind == "actor":
        p["mu.weight"], p["mu.bias"] = u(act_dim, hidden), u(act_dim)
        p["sigma_param"] = (torch.rand(act_dim, 1, generator=g) - 0.5)
    else:
        p["fc2.weight"], p["fc2.bias"] = u(1, hidden + act_dim), u(1)
    return p


def trunk(p: dict, obs, state=None):
    """-> (h_T [B, H], (hidden, cell) each [L, B, H])."""
    x = torch.as_tensor(obs, dtype=torch.float32)
    if x.dim() == 2:
        x = x.unsqueeze(-2)
    B, T, _ = x.shape
    layers = sum(1 for k in p if k.startswith("nn.weight_ih_l"))
    hs, cs = [], []
    for k in range(layers):
        H = p[f"nn.weight_hh_l{k}"].shape[1]
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
    return x[:, -1], (torch.stack(hs), torch.stack(cs))


def actor_forward(p: dict, obs, state=None, max_action: float = 1.0, unbounded: bool = False):
    """-> (mu [B, A], sigma [B, A], (hidden, cell) each [B, L, H] as the reference returns them)."""
    if state is not None:                                        # stored [B, L, H] (continuous.py:298-304)
        state = tuple(torch.as_tensor(s, dtype=torch.float32).transpose(0, 1) for s in state)
    h_t, (hid, cell) = trunk(p, obs, state)
    mu = F.linear(h_t, p["mu.weight"], p["mu.bias"])
    if not unbounded:
        mu = max_action * torch.tanh(mu)
    sigma = (p["sigma_param"].view(1, -1) + torch.zeros_like(mu)).exp()
    return mu, sigma, (hid.transpose(0, 1), cell.transpose(0, 1))


def critic_forward(p: dict, obs, act=None):
    h_t, _ = trunk(p, obs)
    if act is not None:
        h_t = torch.cat([h_t, torch.as_tensor(act, dtype=torch.float32)], dim=1)
    return F.linear(h_t, p["fc2.weight"], p["fc2.bias"])
