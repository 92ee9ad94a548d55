"""REDQ learn() path on the MI355X engine.

Mirrors, on device tensors:
    REDQPolicy.forward                     tianshou/algorithm/modelfree/redq.py:103-131 (= SAC's tanh-Gaussian policy)
    _target_q / _target_q_compute_value    ddpg.py:327-339, redq.py:248-261 (n-step via tianshou_amd.returns)
    REDQ._update_with_batch                redq.py:263-304
Networks: test/continuous/test_redq.py:86-107 with hidden [256, 256]: SAC's actor; one critic module built from
EnsembleLinear layers (utils/net/common.py:518-550).  The rsample() noise and the np.random.choice subset are supplied
by the caller, which is what makes the path reproducible against the reference.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .buffer import DeviceReplayBuffer, gather_rows
from .returns import compute_nstep_return
from .sac import SACConfig, SACEngine, critic_flat_from_torch, critic_flat_to_torch, layout, mlp_layout, use_hidden  # noqa: F401

TIANSHOU_CRITIC_KEYS = ["preprocess.model.model.0.weight", "preprocess.model.model.0.bias_weights",
                        "preprocess.model.model.2.weight", "preprocess.model.model.2.bias_weights",
                        "last.model.0.weight", "last.model.0.bias_weights"]


class REDQStateC(C.Structure):
    """struct ts_redq_state (include/tsengine.h)."""

    _fields_ = [(n, C.c_void_p) for n in ("actor", "actor_m", "actor_v", "critics", "critics_m", "critics_v", "critics_old",
                                          "log_alpha", "log_alpha_m", "log_alpha_v")]


def critic_keys(depth: int = 2) -> list[str]:
    """state_dict keys of the EnsembleLinear critic (utils/net/common.py:518-550: `weight`, `bias_weights`)."""
    ks = []
    for i in range(depth):
        ks += [f"preprocess.model.model.{2 * i}.weight", f"preprocess.model.model.{2 * i}.bias_weights"]
    return ks + ["last.model.0.weight", "last.model.0.bias_weights"]


def keys_depth(keys) -> int | None:
    keys = list(keys)
    for d in range(1, 7):
        if keys == critic_keys(d):
            return d
    return None


def ensemble_flat_from_torch(t: list[torch.Tensor], obs_dim: int, act_dim: int, device="cuda", hidden: int | None = None) -> torch.Tensor:
    """[w1 [E, in, h1], b1 [E, 1, h1], ..., wd [E, h(d-1), hd], bd, wq [E, hd, 1], bq [E, 1, 1]] (EnsembleLinear layout; also valid
    for the matching Adam moments) -> E consecutive critic blocks of ts_mlp_layout (any widths: embedded by zero padding into
    Net[hidden] * d, `tianshou_amd.widths`)."""
    E = t[0].shape[0]
    blocks = [critic_flat_from_torch([x[e].t() if i % 2 == 0 else x[e, 0] for i, x in enumerate(t)],
                                     obs_dim, act_dim, "cpu", hidden=hidden) for e in range(E)]
    return torch.cat(blocks).to(device).contiguous()


def ensemble_flat_to_torch(flat: torch.Tensor, E: int, obs_dim: int, act_dim: int, hidden: int = 256, sizes=None,
                           depth: int | None = None) -> list[torch.Tensor]:
    d = len(sizes) if sizes is not None else int(depth or 2)
    pc = mlp_layout(obs_dim + act_dim, hidden, d, 32)[1][-1]
    per = [critic_flat_to_torch(flat[e * pc:(e + 1) * pc], obs_dim, act_dim, hidden, sizes=sizes, depth=d) for e in range(E)]
    st = lambda i, f: torch.stack([f(p[i]) for p in per])  # noqa: E731
    out = []
    for i in range(d):
        out += [st(2 * i, lambda w: w.t()), st(2 * i + 1, lambda b: b[None, :])]
    return out + [st(2 * d, lambda w: w.t()), st(2 * d + 1, lambda b: b.reshape(1, 1))]


@dataclass
class REDQConfig(SACConfig):
    """REDQ's hyper-parameters (redq.py:133-246) on top of SAC's."""

    ensemble_size: int = 10
    subset_size: int = 2
    actor_delay: int = 20
    target_mode: str = "min"


class REDQEngine:
    """State of one REDQ learner on one GPU."""

    def __init__(self, obs_dim: int, act_dim: int, actor: torch.Tensor, critics: torch.Tensor, cfg: REDQConfig,
                 hidden: int = 256, depth: int = 2, max_action: float = 0.0, activation: str = "relu"):
        """`hidden` / `depth`: width h and number of hidden layers of the actor's Net[h] * depth and of the EnsembleLinear critics
        (test_redq.py uses [256, 256]; any multiple of 32 up to 1024, 1 .. 6 layers, utils/net/common.py:246-369)."""
        if not actor.is_cuda:
            raise RuntimeError("REDQEngine needs parameters on an MI355X (no CPU fallback)")
        if cfg.target_mode not in ("min", "mean") or not 0 < cfg.subset_size <= cfg.ensemble_size <= 64:
            raise ValueError("target_mode must be 'min' or 'mean' and 0 < subset_size <= ensemble_size <= 64")
        self.depth, self.max_action, self.activation = int(depth), float(max_action), activation      # (as in SACEngine)
        n_actor, n_critic = mlp_layout(obs_dim, hidden, self.depth, 64)[1][-1], mlp_layout(obs_dim + act_dim, hidden, self.depth, 32)[1][-1]
        if actor.numel() != n_actor or critics.numel() != cfg.ensemble_size * n_critic:
            raise ValueError("flat parameter vectors do not match ts_mlp_layout / the ensemble size")
        self.obs_dim, self.act_dim, self.cfg, self.hidden = obs_dim, act_dim, cfg, int(hidden)
        self.lay = layout(obs_dim, act_dim, hidden) if self.depth == 2 else {"actor_count": n_actor, "critic_count": n_critic}
        self.device = actor.device
        cl = lambda t: t.detach().float().contiguous().clone()  # noqa: E731
        self.actor, self.critics = cl(actor), cl(critics)
        self.critics_old = cl(critics)                                            # ddpg.py:262
        z = torch.zeros_like
        self.actor_m, self.actor_v = z(self.actor), z(self.actor)
        self.critics_m, self.critics_v = z(self.critics), z(self.critics)
        self.log_alpha = torch.full((1,), cfg.log_alpha0, dtype=torch.float32, device=self.device)
        self.log_alpha_m, self.log_alpha_v = z(self.log_alpha), z(self.log_alpha)
        self.critic_gradient_step = 0                                             # redq.py:241
        self.actor_steps = 0
        self._stats = torch.zeros(4, dtype=torch.float32, device=self.device)     # actor_loss keeps its last value
        self._ws = _lib.default_workspace(self.device.index or 0)

    _f32 = SACEngine._f32
    alpha = SACEngine.alpha
    policy_forward = SACEngine.policy_forward

    def target_q(self, obs_next, noise, subset) -> torch.Tensor:
        obs_next = self._f32(obs_next)
        b = obs_next.shape[0]
        noise = self._f32(noise, (b, self.act_dim))
        sub = np.ascontiguousarray(np.asarray(subset, dtype=np.int32).reshape(-1))
        if sub.size != self.cfg.subset_size:
            raise ValueError("subset must hold subset_size member indices")
        out = torch.empty(b, dtype=torch.float32, device=self.device)
        use_hidden(self._ws, self.hidden, self.depth, self.max_action, self.activation)
        _lib.check(_lib.load().ts_redq_target_q(
            self._ws.handle, _lib.ptr(self.actor), _lib.ptr(self.critics_old), _lib.i64(self.cfg.ensemble_size),
            sub.ctypes.data_as(C.POINTER(C.c_int32)), _lib.i64(sub.size), C.c_int(int(self.cfg.target_mode == "mean")),
            _lib.ptr(self.log_alpha if self.cfg.auto_alpha else None), _lib.f64(self.cfg.alpha), _lib.ptr(obs_next),
            _lib.ptr(noise), _lib.i64(b), _lib.i64(self.obs_dim), _lib.i64(self.act_dim), _lib.ptr(out),
            _lib.current_stream(self.device)))
        return out

    def preprocess(self, buffer: DeviceReplayBuffer, indices, noise, subset) -> torch.Tensor:
        """n-step returns float32[I] with target_q_fn = _target_q (ddpg.py:287-301); obs_next from the buffer's stored column or obs[next(index)] (buffer_base.py:622-626)."""

        def tq_fn(buf, after):
            return self.target_q(buf.obs_next_rows(after), noise, subset)

        class _B:
            pass

        return compute_nstep_return(_B(), buffer, indices, tq_fn, self.cfg.gamma, self.cfg.n_step).returns.reshape(-1)

    def will_update_actor(self) -> bool:
        """Whether the NEXT update_with_batch performs the actor step (and therefore needs rsample() noise)."""
        return (self.critic_gradient_step + 1) % self.cfg.actor_delay == 0

    def update_with_batch(self, obs, act, returns, noise=None, weight=None, grads_out: torch.Tensor | None = None,
                          lr_scale: float = 1.0):
        """-> (stats float32[4] device = {actor_loss (of the latest actor update), critic_loss, alpha, alpha_loss (nan when
        this update had no alpha step)}, weight float32[B] = mean_e td_e)."""
        obs, act = self._f32(obs), self._f32(act)
        b = obs.shape[0]
        returns = self._f32(returns, (b,))
        weight = None if weight is None else self._f32(weight, (b,))
        if act.shape != (b, self.act_dim) or obs.shape != (b, self.obs_dim):
            raise ValueError("obs / act shapes do not match the engine")
        do_actor = self.will_update_actor()
        if do_actor and noise is None:
            raise ValueError("this update performs the actor step: rsample() noise needed")
        noise = None if noise is None else self._f32(noise, (b, self.act_dim))
        self.critic_gradient_step += 1
        if do_actor:
            self.actor_steps += 1
        w_out = torch.empty(b, dtype=torch.float32, device=self.device)
        st = REDQStateC(*[getattr(self, n).data_ptr() for n, _ in REDQStateC._fields_])
        hp = self.cfg.to_c(lr_scale)
        use_hidden(self._ws, self.hidden, self.depth, self.max_action, self.activation)
        _lib.check(_lib.load().ts_redq_update(
            self._ws.handle, C.byref(st), _lib.i64(self.cfg.ensemble_size), _lib.i64(self.critic_gradient_step),
            _lib.i64(max(self.actor_steps, 1)), C.c_int(int(do_actor)), _lib.ptr(obs), _lib.ptr(act), _lib.ptr(returns),
            _lib.ptr(weight), _lib.ptr(noise), _lib.i64(b), _lib.i64(self.obs_dim), _lib.i64(self.act_dim), C.byref(hp),
            _lib.ptr(self._stats), _lib.ptr(w_out), _lib.ptr(grads_out), _lib.current_stream(self.device)))
        stats = self._stats.clone()
        stats[2] = self.alpha[0]                                                  # Alpha.value after this update
        if not (do_actor and self.cfg.auto_alpha):                                # Alpha.update returns None then
            stats[3] = float("nan")
        return stats, w_out
