"""Recurrent actor and critic of the continuous-control nets on the MI355X engine.

Mirrors, on device tensors:
    RecurrentActorProb.forward    tianshou/utils/net/continuous.py:276-322 (nn.LSTM on the observation, `mu` head on the last
                                  step, max_action * tanh unless unbounded, sigma = exp(sigma_param), carried state [B, L, H])
    RecurrentCritic.forward       continuous.py:346-380 (same trunk, fc2 on cat([h_T, act]))
plus the backward pass of both (d loss / d parameters for a given d loss / d output; BPTT on the GEMM kernels of
ts_rnnq.hip) and clip + Adam (ts_adam_step), so that an actor-critic learner can train them.  There is no CPU path: every
function calls libtsengine.so.

Flat layout (include/tsengine.h, ts_lstm_net_layout): layer 0: W_ih [k0 + 1, 4H] | W_hh [H + 1, 4H] | layers >= 1:
W_ih [H + 1, 4H] | W_hh [H + 1, 4H] | head [head_in + 1, 32] (k0 = obs_dim rounded up to 32; head_in = H, or H + act_dim
rounded up to 32 for the critic; last row of a block = its bias).  `*_flat_from_torch` / `*_flat_to_torch` convert from / to
the modules' state_dict() order; the actor's sigma_param travels beside the flat vector (it does not touch the LSTM).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

HEAD = 32


def lstm_keys(layers: int) -> list[str]:
    ks = []
    for k in range(layers):
        ks += [f"nn.weight_ih_l{k}", f"nn.weight_hh_l{k}", f"nn.bias_ih_l{k}", f"nn.bias_hh_l{k}"]
    return ks


def actor_state_dict_keys(layers: int) -> list[str]:
    """RecurrentActorProb.state_dict() order (sigma_param is a Parameter of the module itself: first)."""
    return ["sigma_param", *lstm_keys(layers), "mu.weight", "mu.bias"]


def critic_state_dict_keys(layers: int) -> list[str]:
    return [*lstm_keys(layers), "fc2.weight", "fc2.bias"]


def layout(obs_dim: int, hidden: int, layers: int, out_dim: int, extra_dim: int = 0, has_fc1: bool = False) -> dict:
    out = (C.c_int64 * (5 + 2 * layers))()
    _lib.check(_lib.load().ts_lstm_net_layout(_lib.i64(obs_dim), _lib.i64(hidden), _lib.i64(layers), _lib.i64(out_dim),
                                              _lib.i64(int(has_fc1)), _lib.i64(extra_dim), out))
    return {"k0": int(out[0]), "count": int(out[1]), "fc1": int(out[2]), "ih": [int(out[3 + 2 * l]) for l in range(layers)],
            "hh": [int(out[4 + 2 * l]) for l in range(layers)], "head": int(out[3 + 2 * layers]), "head_in": int(out[4 + 2 * layers])}


def _block(w: torch.Tensor, b: torch.Tensor, k_pad: int, n_pad: int) -> torch.Tensor:
    """nn.Linear-layout weight [n, k] + bias [n] -> [k_pad + 1, n_pad] (zero padding, bias in the last row)."""
    n, k = w.shape
    m = torch.zeros((k_pad + 1, n_pad), dtype=torch.float32)
    m[:k, :n] = w.detach().float().cpu().t()
    m[k_pad, :n] = b.detach().float().cpu().reshape(-1)
    return m.reshape(-1)


def _flat(lstm: list[torch.Tensor], head_w, head_b, obs_dim: int, hidden: int, layers: int, head_in: int, device) -> torch.Tensor:
    k0 = (obs_dim + 31) // 32 * 32
    parts = []
    for k in range(layers):
        w_ih, w_hh, b_ih, b_hh = lstm[4 * k:4 * k + 4]
        parts += [_block(w_ih, b_ih, k0 if k == 0 else hidden, 4 * hidden), _block(w_hh, b_hh, hidden, 4 * hidden)]
    parts.append(_block(head_w, head_b, head_in, HEAD))
    return torch.cat(parts).to(device).contiguous()


def _unflat(flat: torch.Tensor, obs_dim: int, hidden: int, layers: int, head_in: int, head_k: int, head_n: int):
    k0 = (obs_dim + 31) // 32 * 32
    f, off = flat.detach(), 0

    def take(k_pad, n_pad, k, n):
        nonlocal off
        m = f[off:off + (k_pad + 1) * n_pad].reshape(k_pad + 1, n_pad)
        off += (k_pad + 1) * n_pad
        return m[:k, :n].t().contiguous(), m[k_pad, :n].clone()

    lstm = []
    for l in range(layers):
        in_pad, in_k = (k0, obs_dim) if l == 0 else (hidden, hidden)
        (w_ih, b_ih), (w_hh, b_hh) = take(in_pad, 4 * hidden, in_k, 4 * hidden), take(hidden, 4 * hidden, hidden, 4 * hidden)
        lstm += [w_ih, w_hh, b_ih, b_hh]
    head_w, head_b = take(head_in, HEAD, head_k, head_n)
    return lstm, head_w, head_b


def actor_flat_from_torch(t: list[torch.Tensor], obs_dim: int, act_dim: int, hidden: int, layers: int, device="cuda"):
    """Tensors in actor_state_dict_keys(layers) order -> (flat vector, sigma_param float32[act_dim])."""
    sigma = t[0].detach().float().reshape(-1).to(device).contiguous()
    return _flat(t[1:1 + 4 * layers], t[1 + 4 * layers], t[2 + 4 * layers], obs_dim, hidden, layers, hidden, device), sigma


def actor_flat_to_torch(flat: torch.Tensor, sigma_param: torch.Tensor, obs_dim: int, act_dim: int, hidden: int, layers: int):
    lstm, w, b = _unflat(flat, obs_dim, hidden, layers, hidden, hidden, act_dim)
    return [sigma_param.detach().reshape(-1, 1).clone(), *lstm, w, b]


def critic_flat_from_torch(t: list[torch.Tensor], obs_dim: int, act_dim: int, hidden: int, layers: int, device="cuda"):
    head_in = (hidden + act_dim + 31) // 32 * 32 if act_dim else hidden
    return _flat(t[:4 * layers], t[4 * layers], t[4 * layers + 1], obs_dim, hidden, layers, head_in, device)


def critic_flat_to_torch(flat: torch.Tensor, obs_dim: int, act_dim: int, hidden: int, layers: int):
    head_in = (hidden + act_dim + 31) // 32 * 32 if act_dim else hidden
    lstm, w, b = _unflat(flat, obs_dim, hidden, layers, head_in, hidden + act_dim, 1)
    return [*lstm, w, b]


class _LstmNetEngine:
    """Flat parameters + Adam moments of one LSTM-trunk network on one GPU."""

    def __init__(self, obs_dim: int, hidden: int, layers: int, out_dim: int, extra_dim: int, flat_params: torch.Tensor,
                 lr: float = 1e-3, betas=(0.9, 0.999), adam_eps: float = 1e-8, max_grad_norm: float | None = None):
        if not flat_params.is_cuda:
            raise RuntimeError("the recurrent engines need parameters on an MI355X (no CPU fallback)")
        self.obs_dim, self.hidden, self.layers, self.out_dim, self.extra_dim = obs_dim, hidden, layers, out_dim, extra_dim
        self.lay = layout(obs_dim, hidden, layers, out_dim, extra_dim)
        self.P = self.lay["count"]
        if flat_params.numel() != self.P:
            raise ValueError(f"expected {self.P} parameters, got {flat_params.numel()}")
        self.device = flat_params.device
        self.params = flat_params.detach().float().contiguous().clone()
        self.adam_m, self.adam_v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.adam_step = 0
        self.lr, self.betas, self.adam_eps, self.max_grad_norm = lr, betas, adam_eps, max_grad_norm
        self._ws = _lib.default_workspace(self.device.index or 0)

    def _dims(self):
        return (_lib.i64(self.obs_dim), _lib.i64(self.hidden), _lib.i64(self.layers), _lib.i64(self.out_dim), _lib.i64(0),
                _lib.i64(self.extra_dim))

    def _obs(self, obs) -> torch.Tensor:
        obs = torch.as_tensor(obs, device=self.device).to(torch.float32)
        if obs.dim() == 2:
            obs = obs.unsqueeze(1)                                   # evaluation mode (continuous.py:292-293)
        if obs.dim() != 3 or obs.shape[2] != self.obs_dim:
            raise ValueError(f"obs must be [B, T, {self.obs_dim}] or [B, {self.obs_dim}]")
        return obs.contiguous()

    def _state_in(self, state, b):
        if state is None:
            return None, None
        h, c = (torch.as_tensor(s, device=self.device).to(torch.float32).transpose(0, 1).contiguous() for s in state)
        if h.shape != (self.layers, b, self.hidden) or c.shape != h.shape:
            raise ValueError("state tensors must be [B, layers, hidden]")
        return h, c

    def _forward(self, obs, extra, state, tanh_scale: float, want_state: bool):
        obs = self._obs(obs)
        b, t = obs.shape[:2]
        h_in, c_in = self._state_in(state, b)
        out = torch.empty((b, self.out_dim), dtype=torch.float32, device=self.device)
        h_out = c_out = None
        if want_state:
            h_out = torch.empty((self.layers, b, self.hidden), dtype=torch.float32, device=self.device)
            c_out = torch.empty_like(h_out)
        _lib.check(_lib.load().ts_lstm_net_forward(
            self._ws.handle, _lib.ptr(self.params), *self._dims(), _lib.ptr(obs), _lib.ptr(extra), _lib.i64(b), _lib.i64(t),
            _lib.ptr(h_in), _lib.ptr(c_in), _lib.f64(tanh_scale), _lib.ptr(out), _lib.ptr(h_out), _lib.ptr(c_out),
            _lib.current_stream(self.device)))
        if want_state:
            return out, (h_out.transpose(0, 1).contiguous(), c_out.transpose(0, 1).contiguous())
        return out, None

    def _gradient(self, obs, extra, state, tanh_scale: float, d_out) -> torch.Tensor:
        obs = self._obs(obs)
        b, t = obs.shape[:2]
        h_in, c_in = self._state_in(state, b)
        d_out = torch.as_tensor(d_out, device=self.device).to(torch.float32).reshape(b, self.out_dim).contiguous()
        grad = torch.empty(self.P, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().ts_lstm_net_backward(
            self._ws.handle, _lib.ptr(self.params), *self._dims(), _lib.ptr(obs), _lib.ptr(extra), _lib.i64(b), _lib.i64(t),
            _lib.ptr(h_in), _lib.ptr(c_in), _lib.f64(tanh_scale), _lib.ptr(d_out), _lib.ptr(None), _lib.ptr(grad),
            _lib.current_stream(self.device)))
        return grad

    def apply_gradient(self, grad: torch.Tensor) -> None:
        """clip_grad_norm_ + Adam on a flat gradient (algorithm_base.py:496-500)."""
        self.adam_step += 1
        _lib.check(_lib.load().ts_adam_step(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v), _lib.ptr(grad),
            _lib.i64(self.P), _lib.i64(self.adam_step), _lib.f64(self.lr), _lib.f64(self.betas[0]), _lib.f64(self.betas[1]),
            _lib.f64(self.adam_eps), _lib.f64(self.max_grad_norm or 0.0), _lib.current_stream(self.device)))


class RecurrentActorProbEngine(_LstmNetEngine):
    """RecurrentActorProb (continuous.py:241-322) with a state-independent sigma (`conditioned_sigma=False`, the default)."""

    def __init__(self, obs_dim: int, act_dim: int, hidden: int, layers: int, flat_params: torch.Tensor, sigma_param: torch.Tensor,
                 max_action: float = 1.0, unbounded: bool = False, **adam):
        if act_dim > HEAD:
            raise NotImplementedError(f"act_dim {act_dim} > {HEAD} is not supported by the head kernels")
        super().__init__(obs_dim, hidden, layers, act_dim, 0, flat_params, **adam)
        self.act_dim = act_dim
        self.max_action = 1.0 if unbounded else float(max_action)     # continuous.py:259-261
        self.unbounded = unbounded
        self.sigma_param = torch.as_tensor(sigma_param, device=self.device).to(torch.float32).reshape(-1).contiguous().clone()
        if self.sigma_param.numel() != act_dim:
            raise ValueError("sigma_param must have act_dim entries")

    @classmethod
    def from_module(cls, module, device="cuda", **adam):
        """From a `RecurrentActorProb` instance (duck-typed: `.nn` = the nn.LSTM, `.mu`, `.sigma_param`, `._c_sigma`,
        `.max_action`, `._unbounded`, continuous.py:262-274)."""
        if getattr(module, "_c_sigma", False):
            raise NotImplementedError("RecurrentActorProb(conditioned_sigma=True) is not supported: sigma must be the "
                                      "state-independent sigma_param")
        lstm = module.nn
        layers, hidden, obs_dim = int(lstm.num_layers), int(lstm.hidden_size), int(lstm.input_size)
        act_dim = int(module.mu.out_features)
        sd = module.state_dict()
        flat, sigma = actor_flat_from_torch([sd[k] for k in actor_state_dict_keys(layers)], obs_dim, act_dim, hidden, layers, device)
        return cls(obs_dim, act_dim, hidden, layers, flat, sigma, max_action=float(module.max_action),
                   unbounded=bool(module._unbounded), **adam)

    def to_module(self, module) -> None:
        """Writes the engine's parameters back into the torch module (checkpoints, the collector's policy)."""
        t = actor_flat_to_torch(self.params, self.sigma_param, self.obs_dim, self.act_dim, self.hidden, self.layers)
        sd = module.state_dict()
        with torch.no_grad():
            for k, v in zip(actor_state_dict_keys(self.layers), t):
                sd[k].copy_(v.reshape(sd[k].shape))

    @property
    def _scale(self) -> float:
        return 0.0 if self.unbounded else self.max_action

    def forward(self, obs, state=None):
        """-> ((mu, sigma) float32[B, A] each, {"hidden", "cell"} float32[B, L, H]) as the reference returns them."""
        mu, st = self._forward(obs, None, None if state is None else (state["hidden"], state["cell"]), self._scale, True)
        sigma = self.sigma_param.exp().unsqueeze(0).expand_as(mu).contiguous()      # continuous.py:318-320
        return (mu, sigma), {"hidden": st[0], "cell": st[1]}

    def gradient(self, obs, d_mu, state=None) -> torch.Tensor:
        """d loss / d (LSTM + mu head parameters) for d loss / d mu (the bounded mean); sigma_param's gradient is
        (d loss / d sigma * sigma).sum(0) and needs no network pass."""
        return self._gradient(obs, None, None if state is None else (state["hidden"], state["cell"]), self._scale, d_mu)


class RecurrentCriticEngine(_LstmNetEngine):
    """RecurrentCritic (continuous.py:325-380): V(s) (act_dim = 0) or Q(s, a)."""

    def __init__(self, obs_dim: int, act_dim: int, hidden: int, layers: int, flat_params: torch.Tensor, **adam):
        super().__init__(obs_dim, hidden, layers, 1, act_dim, flat_params, **adam)
        self.act_dim = act_dim

    @classmethod
    def from_module(cls, module, device="cuda", **adam):
        """From a `RecurrentCritic` instance (`.nn` = the nn.LSTM, `.fc2`, continuous.py:335-344)."""
        lstm = module.nn
        layers, hidden, obs_dim = int(lstm.num_layers), int(lstm.hidden_size), int(lstm.input_size)
        act_dim = int(module.fc2.in_features) - hidden
        sd = module.state_dict()
        flat = critic_flat_from_torch([sd[k] for k in critic_state_dict_keys(layers)], obs_dim, act_dim, hidden, layers, device)
        return cls(obs_dim, act_dim, hidden, layers, flat, **adam)

    def to_module(self, module) -> None:
        t = critic_flat_to_torch(self.params, self.obs_dim, self.act_dim, self.hidden, self.layers)
        sd = module.state_dict()
        with torch.no_grad():
            for k, v in zip(critic_state_dict_keys(self.layers), t):
                sd[k].copy_(v.reshape(sd[k].shape))

    def _act(self, act, b):
        if self.act_dim == 0:
            return None
        if act is None:
            raise ValueError("this critic was built with an action input")
        act = torch.as_tensor(act, device=self.device).to(torch.float32).reshape(b, self.act_dim).contiguous()
        return act

    def forward(self, obs, act=None) -> torch.Tensor:
        """-> float32[B, 1]; obs [B, T, dim] (the reference asserts three dimensions, continuous.py:365)."""
        obs = self._obs(obs)
        return self._forward(obs, self._act(act, obs.shape[0]), None, 0.0, False)[0]

    def gradient(self, obs, act, d_value) -> torch.Tensor:
        obs = self._obs(obs)
        return self._gradient(obs, self._act(act, obs.shape[0]), None, 0.0, d_value)
