"""TD3 / DDPG learn() path on the MI355X engine.

Mirrors ContinuousDeterministicPolicy.forward (tianshou/algorithm/modelfree/ddpg.py:162-180),
ActorCriticOffPolicyAlgorithm._target_q (ddpg.py:327-339) with DDPG's / TD3's lagged-actor action
(ddpg.py:397-399, td3.py:190-202), and DDPG / TD3 `_update_with_batch` (ddpg.py:401-411, td3.py:204-226)
for the networks of examples/mujoco/mujoco_td3.py:85-103.  The critics share SAC's layout and kernels
(tianshou_amd/sac.py); TD3's target-smoothing noise is an argument.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib
from . import widths as W
from .buffer import DeviceReplayBuffer, gather_rows
from .returns import compute_nstep_return
from .sac import (HID, _dense, _l1, critic_flat_from_torch, critic_flat_to_torch, critic_keys, mlp_layout,  # noqa: F401
                  keys_depth, trunk_flat, trunk_keys, trunk_unflat, use_hidden)

TIANSHOU_ACTOR_KEYS = ["preprocess.model.model.0.weight", "preprocess.model.model.0.bias",
                       "preprocess.model.model.2.weight", "preprocess.model.model.2.bias",
                       "last.model.0.weight", "last.model.0.bias"]


class TD3HParams(C.Structure):
    """struct ts_td3_hparams (include/tsengine.h)."""

    _fields_ = [("actor_lr", C.c_double), ("critic_lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("adam_eps", C.c_double), ("tau", C.c_double), ("max_action", C.c_double), ("update_actor", C.c_int32),
                ("reserved", C.c_int32)]


class TD3StateC(C.Structure):
    """struct ts_td3_state (include/tsengine.h)."""

    _fields_ = [(n, C.c_void_p) for n in ("actor", "actor_m", "actor_v", "critic1", "critic1_m", "critic1_v", "critic2",
                                          "critic2_m", "critic2_v", "actor_old", "critic1_old", "critic2_old")]


def layout(obs_dim: int, act_dim: int, hidden: int = HID) -> dict[str, int]:
    out = (C.c_int64 * 4)()
    _lib.check(_lib.load().ts_td3_layout_h(_lib.i64(obs_dim), _lib.i64(act_dim), _lib.i64(hidden), out))
    return dict(zip(["ka", "kc", "actor_count", "critic_count"], (int(v) for v in out)))


def actor_keys(depth: int = 2) -> list[str]:
    return trunk_keys(depth, ("last",))


def actor_flat_from_torch(t: list[torch.Tensor], obs_dim: int, act_dim: int, device="cuda", hidden: int | None = None) -> torch.Tensor:
    """[w1, b1, ..., wd, bd, wa, ba] in torch nn.Linear layout -> flat engine vector (depth and widths read off the tensors; unequal
    widths / no multiple of 32: embedded by zero padding into Net[hidden] * d, `tianshou_amd.widths`)."""
    d = W.depth_of(t, 1)
    H = int(hidden or W.engine_hidden([W.layer_widths(t, 1)]))
    t = W.pad_layers(t, H, 1)
    k, _ = mlp_layout(obs_dim, H, d, 32)
    head = torch.zeros((H + 1, 32), dtype=torch.float32)
    head[:H, :act_dim] = t[2 * d].detach().float().cpu().t()
    head[H, :act_dim] = t[2 * d + 1].detach().float().cpu()
    return torch.cat(trunk_flat(t, d, k) + [head.reshape(-1)]).to(device).contiguous()


def actor_flat_to_torch(flat: torch.Tensor, obs_dim: int, act_dim: int, hidden: int = HID, sizes=None, depth: int | None = None) -> list[torch.Tensor]:
    d = len(sizes) if sizes is not None else int(depth or 2)
    H = int(hidden)
    k, offs = mlp_layout(obs_dim, H, d, 32)
    f = flat.detach()
    hd = f[offs[d]: offs[d + 1]].reshape(H + 1, 32)
    out = trunk_unflat(f, obs_dim, k, H, d, offs) + [hd[:H, :act_dim].t().contiguous(), hd[H, :act_dim].clone()]
    return W.unpad_layers(out, sizes) if sizes is not None else out


@dataclass
class TD3Config:
    """Hyper-parameters of the reference TD3 (td3.py:110-188); twin=False gives DDPG (ddpg.py:346-395)."""

    gamma: float = 0.99
    tau: float = 0.005
    n_step: int = 1
    twin: bool = True
    policy_noise: float = 0.2
    noise_clip: float = 0.5
    update_actor_freq: int = 2
    max_action: float = 1.0
    actor_lr: float = 1e-3
    critic_lr: float = 1e-3
    betas: tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8


class TD3Engine:
    """State of one TD3 / DDPG learner on one GPU."""

    def __init__(self, obs_dim: int, act_dim: int, actor: torch.Tensor, critic1: torch.Tensor,
                 critic2: torch.Tensor | None, cfg: TD3Config, hidden: int = HID, depth: int = 2, activation: str = "relu"):
        """`hidden` / `depth`: the Net[hidden] * depth trunks (any multiple of 32 up to 1024, 1 .. 6 hidden layers; [256, 256]
        in the examples)."""
        if not actor.is_cuda:
            raise RuntimeError("TD3Engine needs parameters on an MI355X (no CPU fallback)")
        if cfg.twin != (critic2 is not None):
            raise ValueError("cfg.twin and critic2 disagree")
        self.hidden, self.depth, self.activation = int(hidden), int(depth), activation
        if actor.numel() != mlp_layout(obs_dim, self.hidden, self.depth, 32)[1][-1] \
                or critic1.numel() != mlp_layout(obs_dim + act_dim, self.hidden, self.depth, 32)[1][-1]:
            raise ValueError("flat parameter vectors do not match ts_mlp_layout")
        self.obs_dim, self.act_dim, self.cfg, self.device = obs_dim, act_dim, cfg, actor.device
        self.lay = layout(obs_dim, act_dim, self.hidden) if self.depth == 2 else None       # (ts_td3_layout_h: two hidden layers)
        cl = lambda t: None if t is None else t.detach().float().contiguous().clone()   # noqa: E731
        z = lambda t: None if t is None else torch.zeros_like(t)                         # noqa: E731
        self.actor, self.critic1, self.critic2 = cl(actor), cl(critic1), cl(critic2)
        self.actor_old, self.critic1_old, self.critic2_old = cl(actor), cl(critic1), cl(critic2)
        self.actor_m, self.actor_v = z(self.actor), z(self.actor)
        self.critic1_m, self.critic1_v = z(self.critic1), z(self.critic1)
        self.critic2_m, self.critic2_v = z(self.critic2), z(self.critic2)
        self.cnt = 0                 # TD3._cnt
        self.actor_steps = 0
        self._ws = _lib.default_workspace(self.device.index or 0)

    def _f32(self, x, shape=None) -> torch.Tensor:
        t = torch.as_tensor(x, device=self.device).to(torch.float32).contiguous()
        return t if shape is None else t.reshape(shape)

    def policy_forward(self, obs) -> torch.Tensor:
        obs = self._f32(obs)
        act = torch.empty((obs.shape[0], self.act_dim), dtype=torch.float32, device=self.device)
        use_hidden(self._ws, self.hidden, self.depth, 0.0, self.activation)
        _lib.check(_lib.load().ts_td3_policy_forward(
            self._ws.handle, _lib.ptr(self.actor), _lib.ptr(obs), _lib.i64(obs.shape[0]), _lib.i64(self.obs_dim),
            _lib.i64(self.act_dim), _lib.f64(self.cfg.max_action), _lib.ptr(act), _lib.current_stream(self.device)))
        return act

    def target_q(self, obs_next, noise=None) -> torch.Tensor:
        cfg = self.cfg
        obs_next = self._f32(obs_next)
        b = obs_next.shape[0]
        if cfg.twin and noise is None:
            raise ValueError("TD3 needs the target-smoothing noise (the torch.randn draws of td3.py:196)")
        noise = self._f32(noise, (b, self.act_dim)) if cfg.twin else None
        out = torch.empty(b, dtype=torch.float32, device=self.device)
        use_hidden(self._ws, self.hidden, self.depth, 0.0, self.activation)
        _lib.check(_lib.load().ts_td3_target_q(
            self._ws.handle, _lib.ptr(self.actor_old), _lib.ptr(self.critic1_old), _lib.ptr(self.critic2_old),
            _lib.ptr(obs_next), _lib.ptr(noise), _lib.i64(b), _lib.i64(self.obs_dim), _lib.i64(self.act_dim),
            _lib.f64(cfg.max_action), _lib.f64(cfg.policy_noise), _lib.f64(cfg.noise_clip), _lib.ptr(out),
            _lib.current_stream(self.device)))
        return out

    def preprocess(self, buffer: DeviceReplayBuffer, indices, noise=None) -> torch.Tensor:
        """n-step returns float32[I] with target_q_fn = _target_q (ddpg.py:287-301)."""

        class _B:
            pass

        fn = lambda buf, after: self.target_q(buf.obs_next_rows(after), noise)  # noqa: E731
        return compute_nstep_return(_B(), buffer, indices, fn, self.cfg.gamma, self.cfg.n_step).returns.reshape(-1)

    def update_with_batch(self, obs, act, returns, weight=None, grads_out=None, lr_scale: float = 1.0):
        """-> (stats float32[3] = {actor_loss (of the latest actor update), critic1_loss, critic2_loss}, weight)."""
        cfg = self.cfg
        obs, act = self._f32(obs), self._f32(act)
        b = obs.shape[0]
        returns = self._f32(returns, (b,))
        weight = None if weight is None else self._f32(weight, (b,))
        if obs.shape != (b, self.obs_dim) or act.shape != (b, self.act_dim):
            raise ValueError("obs / act shapes do not match the engine")
        upd = self.cnt % (cfg.update_actor_freq if cfg.twin else 1) == 0            # td3.py:215
        self.cnt += 1
        if upd:
            self.actor_steps += 1
        if not hasattr(self, "_stats"):
            self._stats = torch.zeros(3, dtype=torch.float32, device=self.device)
        w_out = torch.empty(b, dtype=torch.float32, device=self.device)
        names = [n for n, _ in TD3StateC._fields_]
        st = TD3StateC(*[None if getattr(self, n) is None else getattr(self, n).data_ptr() for n in names])
        hp = TD3HParams(cfg.actor_lr * lr_scale, cfg.critic_lr * lr_scale, cfg.betas[0], cfg.betas[1], cfg.adam_eps,
                        cfg.tau, cfg.max_action, int(upd), 0)
        use_hidden(self._ws, self.hidden, self.depth, 0.0, self.activation)
        _lib.check(_lib.load().ts_td3_update(
            self._ws.handle, C.byref(st), _lib.i64(self.cnt), _lib.i64(max(self.actor_steps, 1)), _lib.ptr(obs),
            _lib.ptr(act), _lib.ptr(returns), _lib.ptr(weight), _lib.i64(b), _lib.i64(self.obs_dim),
            _lib.i64(self.act_dim), C.byref(hp), _lib.ptr(self._stats), _lib.ptr(w_out), _lib.ptr(grads_out),
            _lib.current_stream(self.device)))
        return self._stats.clone(), w_out
