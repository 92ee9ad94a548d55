"""DiscreteSAC learn() path on the MI355X engine.

Mirrors, on device tensors:
    DiscreteSACPolicy.forward             tianshou/algorithm/modelfree/discrete_sac.py:53-67 (logits; sampling is torch's)
    _target_q / _target_q_compute_value   ddpg.py:327-339, discrete_sac.py:147-155 (n-step via tianshou_amd.returns)
    DiscreteSAC._update_with_batch        discrete_sac.py:157-196 (critic x2, actor, AutoAlpha, Polyak)
Networks: test/discrete/test_discrete_sac.py:88-97 (Net(obs, [h, h]) ReLU under DiscreteActor / DiscreteCritic).
There is no CPU path: every function calls libtsengine.so and raises when it is missing.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .buffer import DeviceReplayBuffer, _i64_dev, gather_rows
from .returns import compute_nstep_return
from .sac import SACConfig, SACStateC, keys_depth, mlp_layout, trunk_flat, trunk_keys, trunk_unflat, use_hidden  # noqa: F401

TIANSHOU_KEYS = ["preprocess.model.model.0.weight", "preprocess.model.model.0.bias",
                 "preprocess.model.model.2.weight", "preprocess.model.model.2.bias",
                 "last.model.0.weight", "last.model.0.bias"]


def net_keys(depth: int = 2) -> list[str]:
    return trunk_keys(depth, ("last",))


def layout(obs_dim: int, n_act: int, hidden: int, depth: int = 2) -> dict[str, int]:
    _lib.check(_lib.load().ts_dsac_layout(_lib.i64(obs_dim), _lib.i64(n_act), _lib.i64(hidden), (C.c_int64 * 3)()))      # argument checks
    k, offs = mlp_layout(obs_dim, hidden, depth, (n_act + 31) // 32 * 32)
    return {"ka": k, "hw": (n_act + 31) // 32 * 32, "count": offs[-1], "offs": offs}


def _block(w: torch.Tensor, b: torch.Tensor, k_pad: int, n_pad: int) -> torch.Tensor:
    """nn.Linear (weight [out, in], bias [out]) -> matrix [k_pad + 1, n_pad] (rows = inputs, last row = bias)."""
    wb = torch.zeros((k_pad + 1, n_pad), dtype=torch.float32)
    wb[: w.shape[1], : w.shape[0]] = w.detach().float().cpu().t()
    wb[k_pad, : b.shape[0]] = b.detach().float().cpu()
    return wb.reshape(-1)


def net_flat_from_torch(t: list[torch.Tensor], obs_dim: int, n_act: int, hidden: int, device="cuda") -> torch.Tensor:
    """[w1, b1, ..., wd, bd, w_head, b_head] (torch nn.Linear layout; also valid for Adam moments) -> flat vector.  Widths other
    than [hidden] * d are embedded by zero padding (`tianshou_amd.widths`)."""
    from . import widths as W

    d = W.depth_of(t, 1)
    t = W.pad_layers(t, hidden, 1)
    lay = layout(obs_dim, n_act, hidden, d)
    return torch.cat(trunk_flat(t, d, lay["ka"]) + [_block(t[2 * d], t[2 * d + 1], hidden, lay["hw"])]).to(device).contiguous()


def net_flat_to_torch(flat: torch.Tensor, obs_dim: int, n_act: int, hidden: int, sizes=None, depth: int | None = None) -> list[torch.Tensor]:
    from . import widths as W

    d = len(sizes) if sizes is not None else int(depth or 2)
    lay = layout(obs_dim, n_act, hidden, d)
    f, offs = flat.detach(), lay["offs"]
    hd = f[offs[d]: offs[d + 1]].reshape(hidden + 1, lay["hw"])
    out = trunk_unflat(f, obs_dim, lay["ka"], hidden, d, offs) + [hd[:hidden, :n_act].t().contiguous(), hd[hidden, :n_act].clone()]
    return W.unpad_layers(out, sizes) if sizes is not None else out


class DiscreteSACEngine:
    """State of one DiscreteSAC learner on one GPU (hyper-parameters: tianshou_amd.sac.SACConfig)."""

    def __init__(self, obs_dim: int, n_act: int, hidden: int, actor: torch.Tensor, critic1: torch.Tensor,
                 critic2: torch.Tensor, cfg: SACConfig, depth: int = 2, activation: str = "relu"):
        if not actor.is_cuda:
            raise RuntimeError("DiscreteSACEngine needs parameters on an MI355X (no CPU fallback)")
        self.depth, self.activation = int(depth), activation
        lay = layout(obs_dim, n_act, hidden, self.depth)
        if any(t.numel() != lay["count"] for t in (actor, critic1, critic2)):
            raise ValueError("flat parameter vectors do not match ts_dsac_layout")
        self.obs_dim, self.n_act, self.hidden, self.cfg, self.lay = obs_dim, n_act, hidden, cfg, lay
        self.device = actor.device
        cl = lambda t: t.detach().float().contiguous().clone()  # noqa: E731
        self.actor, self.critic1, self.critic2 = cl(actor), cl(critic1), cl(critic2)
        self.critic1_old, self.critic2_old = cl(critic1), cl(critic2)              # td3.py:90-91
        z = torch.zeros_like
        self.actor_m, self.actor_v = z(self.actor), z(self.actor)
        self.critic1_m, self.critic1_v = z(self.critic1), z(self.critic1)
        self.critic2_m, self.critic2_v = z(self.critic2), z(self.critic2)
        self.log_alpha = torch.full((1,), cfg.log_alpha0, dtype=torch.float32, device=self.device)
        self.log_alpha_m, self.log_alpha_v = z(self.log_alpha), z(self.log_alpha)
        self.adam_step = 0
        self._ws = _lib.default_workspace(self.device.index or 0)

    def _state_c(self) -> SACStateC:
        return SACStateC(*[getattr(self, n).data_ptr() for n, _ in SACStateC._fields_])

    def _f32(self, x, shape=None) -> torch.Tensor:
        t = torch.as_tensor(x, device=self.device).to(torch.float32).contiguous()
        return t if shape is None else t.reshape(shape)

    def _dims(self):
        use_hidden(self._ws, self.hidden, self.depth, 0.0, self.activation)          # (the entry points read the depth from the workspace)
        return _lib.i64(self.obs_dim), _lib.i64(self.n_act), _lib.i64(self.hidden)

    @property
    def alpha(self) -> torch.Tensor:
        if self.cfg.auto_alpha:
            return self.log_alpha.exp()
        return torch.full((1,), self.cfg.alpha, dtype=torch.float32, device=self.device)

    def policy_forward(self, obs) -> torch.Tensor:
        """-> logits float32[B, n_act] (the input of Categorical(logits=...), discrete_sac.py:60-61)."""
        obs = self._f32(obs).reshape(-1, self.obs_dim)
        b = obs.shape[0]
        out = torch.empty((b, self.n_act), dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().ts_dsac_policy_forward(
            self._ws.handle, _lib.ptr(self.actor), _lib.ptr(obs), _lib.i64(b), *self._dims(), _lib.ptr(out),
            _lib.current_stream(self.device)))
        return out

    def target_q(self, obs_next) -> torch.Tensor:
        obs_next = self._f32(obs_next).reshape(-1, self.obs_dim)
        b = obs_next.shape[0]
        out = torch.empty(b, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().ts_dsac_target_q(
            self._ws.handle, _lib.ptr(self.actor), _lib.ptr(self.critic1_old), _lib.ptr(self.critic2_old),
            _lib.ptr(self.log_alpha if self.cfg.auto_alpha else None), _lib.f64(self.cfg.alpha), _lib.ptr(obs_next),
            _lib.i64(b), *self._dims(), _lib.ptr(out), _lib.current_stream(self.device)))
        return out

    def preprocess(self, buffer: DeviceReplayBuffer, indices) -> torch.Tensor:
        """n-step returns float32[I] with target_q_fn = _target_q (ddpg.py:287-301); obs_next from the buffer's stored column or obs[next(index)] (buffer_base.py:622-626)."""

        def tq_fn(buf, after):
            return self.target_q(buf.obs_next_rows(after))

        class _B:
            pass

        return compute_nstep_return(_B(), buffer, indices, tq_fn, self.cfg.gamma, self.cfg.n_step).returns.reshape(-1)

    def update_with_batch(self, obs, act, returns, weight=None, grads_out: torch.Tensor | None = None,
                          lr_scale: float = 1.0):
        """-> (stats float32[5] device = {actor_loss, critic1_loss, critic2_loss, alpha, alpha_loss},
        weight float32[B] = (td1 + td2) / 2)."""
        obs = self._f32(obs)
        b = obs.shape[0]
        act = _i64_dev(act, self.device).reshape(-1)
        returns = self._f32(returns, (-1,))
        weight = None if weight is None else self._f32(weight, (-1,))
        if obs.shape != (b, self.obs_dim) or act.numel() != b or returns.numel() != b \
                or (weight is not None and weight.numel() != b):
            raise ValueError("obs / act / returns / weight shapes do not match the engine")
        self.adam_step += 1
        stats = torch.empty(5, dtype=torch.float32, device=self.device)
        w_out = torch.empty(b, dtype=torch.float32, device=self.device)
        st, hp = self._state_c(), self.cfg.to_c(lr_scale)
        _lib.check(_lib.load().ts_dsac_update(
            self._ws.handle, C.byref(st), _lib.i64(self.adam_step), _lib.ptr(obs), _lib.ptr(act), _lib.ptr(returns),
            _lib.ptr(weight), _lib.i64(b), *self._dims(), C.byref(hp), _lib.ptr(stats), _lib.ptr(w_out),
            _lib.ptr(grads_out), _lib.current_stream(self.device)))
        return stats, w_out
