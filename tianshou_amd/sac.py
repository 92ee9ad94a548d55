"""SAC learn() path on the MI355X engine.

Mirrors, on device tensors:
    SACPolicy.forward                     tianshou/algorithm/modelfree/sac.py:108-131
    ActorCriticOffPolicyAlgorithm._target_q / _preprocess_batch   ddpg.py:287-339 (n-step via returns.py)
    SAC._update_with_batch                sac.py:298-336 (critic x2, actor, AutoAlpha, Polyak)
Networks: examples/mujoco/mujoco_sac.py:82-104 (Net[256, 256] ReLU; actor with state-conditioned sigma,
unbounded; critics on concat(obs, act)).  rsample() noise is supplied by the caller (host- or
device-generated), which is what makes the path reproducible against the reference.
There is no CPU path: every function calls libtsengine.so and raises when it is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from . import widths as W
from .buffer import DeviceReplayBuffer, _i64_dev, gather_rows
from .returns import compute_nstep_return

TIANSHOU_ACTOR_KEYS = ["preprocess.model.model.0.weight", "preprocess.model.model.0.bias",
                       "preprocess.model.model.2.weight", "preprocess.model.model.2.bias",
                       "mu.model.0.weight", "mu.model.0.bias", "sigma.model.0.weight", "sigma.model.0.bias"]
TIANSHOU_CRITIC_KEYS = ["preprocess.model.model.0.weight", "preprocess.model.model.0.bias",
                        "preprocess.model.model.2.weight", "preprocess.model.model.2.bias",
                        "last.model.0.weight", "last.model.0.bias"]          # (two hidden layers: actor_keys(2) / critic_keys(2))
HID = 256


class SACHParams(C.Structure):
    """struct ts_sac_hparams (include/tsengine.h)."""

    _fields_ = [("actor_lr", C.c_double), ("critic_lr", C.c_double), ("alpha_lr", C.c_double),
                ("beta1", C.c_double), ("beta2", C.c_double), ("adam_eps", C.c_double), ("tau", C.c_double),
                ("alpha", C.c_double), ("target_entropy", C.c_double), ("auto_alpha", C.c_int32),
                ("reserved", C.c_int32)]


class SACStateC(C.Structure):
    """struct ts_sac_state (include/tsengine.h)."""

    _fields_ = [(n, C.c_void_p) for n in (
        "actor", "actor_m", "actor_v", "critic1", "critic1_m", "critic1_v", "critic2", "critic2_m", "critic2_v",
        "critic1_old", "critic2_old", "log_alpha", "log_alpha_m", "log_alpha_v")]


class SACReplayC(C.Structure):
    """struct ts_sac_replay (include/tsengine.h): the replay buffer's columns as ts_sac_learn_rows reads them."""

    _fields_ = [(n, C.c_void_p) for n in ("obs", "act", "obs_next", "rew", "terminated")]


def trunk_keys(depth: int, heads: tuple[str, ...]) -> list[str]:
    """state_dict keys of Net(hidden_sizes=[...] * depth) (Sequential(Linear, ReLU, ...): the Linears sit at even positions,
    utils/net/common.py:90-178) under an actor / critic with the given single-Linear heads."""
    ks = []
    for i in range(depth):
        ks += [f"preprocess.model.model.{2 * i}.weight", f"preprocess.model.model.{2 * i}.bias"]
    for h in heads:
        ks += [f"{h}.model.0.weight", f"{h}.model.0.bias"]
    return ks


def keys_depth(keys, heads: tuple[str, ...]) -> int | None:
    """Number of hidden layers if `keys` are exactly those of a trunk + the heads, else None."""
    keys = list(keys)
    for d in range(1, W.MAX_DEPTH + 1):
        if keys == trunk_keys(d, heads):
            return d
    return None


def actor_keys(depth: int = 2) -> list[str]:
    return trunk_keys(depth, ("mu", "sigma"))


def critic_keys(depth: int = 2) -> list[str]:
    return trunk_keys(depth, ("last",))


def layout(obs_dim: int, act_dim: int, hidden: int = HID) -> dict[str, int]:
    """ts_sac_layout_h: offsets / counts of the flat vectors for Net[hidden, hidden] (a multiple of 32 up to 1024)."""
    out = (C.c_int64 * 8)()
    _lib.check(_lib.load().ts_sac_layout_h(_lib.i64(obs_dim), _lib.i64(act_dim), _lib.i64(hidden), out))
    keys = ["ka", "kc", "actor_count", "critic_count", "actor_l2", "actor_head", "critic_l2", "critic_head"]
    return dict(zip(keys, (int(v) for v in out)))


def mlp_layout(in_dim: int, hidden: int, depth: int, head_cols: int) -> tuple[int, list[int]]:
    """ts_mlp_layout: (k = in_dim rounded up to 32, offsets of the depth + 1 linear layers followed by the element count) of a
    flat Net[hidden] * depth vector with a `head_cols`-column head block."""
    out = (C.c_int64 * (depth + 3))()
    _lib.check(_lib.load().ts_mlp_layout(_lib.i64(in_dim), _lib.i64(hidden), _lib.i64(depth), _lib.i64(head_cols), out))
    return int(out[0]), [int(v) for v in out[1:]]


ACTIVATIONS = {"relu": 1, "tanh": 0}      # TS_NET_ACT_* (include/tsengine.h)


def use_hidden(ws, hidden: int, depth: int = 2, max_action: float = 0.0, activation: str = "relu") -> None:
    """Hidden width, depth and the Gaussian actor's tanh bound are properties of the workspace (ts_mlp_set_trunk,
    ts_sac_set_actor_bound): every SAC / TD3 / DDPG / REDQ / DiscreteSAC engine sets its own before each call, since engines of
    different networks may share the device's default workspace."""
    lib = _lib.load()
    _lib.check(lib.ts_mlp_set_trunk(ws.handle, _lib.i64(hidden), _lib.i64(depth)))
    _lib.check(lib.ts_sac_set_actor_bound(ws.handle, _lib.f64(max_action)))
    if activation != "relu":                  # (ts_mlp_set_trunk has just reset it to ReLU)
        _lib.check(lib.ts_mlp_set_activation(ws.handle, C.c_int(ACTIVATIONS[activation])))


def _l1(w: torch.Tensor, b: torch.Tensor, k_pad: int) -> torch.Tensor:
    wb = torch.zeros((k_pad + 1, w.shape[0]), dtype=torch.float32)
    wb[: w.shape[1]] = w.detach().float().cpu().t()
    wb[k_pad] = b.detach().float().cpu()
    return wb.reshape(-1)


def _dense(w, b) -> torch.Tensor:
    return torch.cat([w.detach().float().cpu().t().reshape(-1), b.detach().float().cpu().reshape(-1)])


def trunk_flat(t: list[torch.Tensor], depth: int, k_pad: int) -> list[torch.Tensor]:
    """The hidden layers' wb blocks of [w1, b1, ..., wd, bd, ...] (all of one width)."""
    return [_l1(t[0], t[1], k_pad)] + [_dense(t[2 * i], t[2 * i + 1]) for i in range(1, depth)]


def trunk_unflat(f: torch.Tensor, in_dim: int, k_pad: int, H: int, depth: int, offs: list[int]) -> list[torch.Tensor]:
    l1 = f[: offs[1]].reshape(k_pad + 1, H)
    out = [l1[:in_dim].t().contiguous(), l1[k_pad].clone()]
    for i in range(1, depth):
        li = f[offs[i]: offs[i + 1]].reshape(H + 1, H)
        out += [li[:H].t().contiguous(), li[H].clone()]
    return out


def actor_flat_from_torch(t: list[torch.Tensor], obs_dim: int, act_dim: int, device="cuda", hidden: int | None = None) -> torch.Tensor:
    """[w1, b1, ..., wd, bd, wmu, bmu, wsig, bsig] (torch nn.Linear layout; also valid for Adam moments).  Depth and widths are
    read off the tensors; unequal widths / widths that are no multiple of 32 are embedded by zero padding into
    Net[hidden] * d (`tianshou_amd.widths`; hidden = the largest width rounded up to 32 unless given)."""
    d = W.depth_of(t, 2)
    H = int(hidden or W.engine_hidden([W.layer_widths(t, 2)]))
    t = W.pad_layers(t, H, 2)
    k, _ = mlp_layout(obs_dim, H, d, 64)
    head = torch.zeros((H + 1, 64), dtype=torch.float32)
    head[:H, :act_dim] = t[2 * d].detach().float().cpu().t()
    head[H, :act_dim] = t[2 * d + 1].detach().float().cpu()
    head[:H, 32:32 + act_dim] = t[2 * d + 2].detach().float().cpu().t()
    head[H, 32:32 + act_dim] = t[2 * d + 3].detach().float().cpu()
    return torch.cat(trunk_flat(t, d, k) + [head.reshape(-1)]).to(device).contiguous()


def critic_flat_from_torch(t: list[torch.Tensor], obs_dim: int, act_dim: int, device="cuda", hidden: int | None = None) -> torch.Tensor:
    """[w1, b1, ..., wd, bd, wq, bq] (depth / widths as in `actor_flat_from_torch`)."""
    d = W.depth_of(t, 1)
    H = int(hidden or W.engine_hidden([W.layer_widths(t, 1)]))
    t = W.pad_layers(t, H, 1)
    k, _ = mlp_layout(obs_dim + act_dim, H, d, 32)
    head = torch.zeros((H + 1, 32), dtype=torch.float32)
    head[:H, 0] = t[2 * d].detach().float().cpu().reshape(-1)
    head[H, 0] = t[2 * d + 1].detach().float().cpu().reshape(())
    return torch.cat(trunk_flat(t, d, k) + [head.reshape(-1)]).to(device).contiguous()


def actor_flat_to_torch(flat: torch.Tensor, obs_dim: int, act_dim: int, hidden: int = HID, sizes=None, depth: int | None = None) -> list[torch.Tensor]:
    """sizes = (h1, ..., hd): the widths of the torch network embedded in Net[hidden] * d (`tianshou_amd.widths`); without
    sizes the depth is `depth` (default 2)."""
    d = len(sizes) if sizes is not None else int(depth or 2)
    H = int(hidden)
    k, offs = mlp_layout(obs_dim, H, d, 64)
    f = flat.detach()
    hd = f[offs[d]: offs[d + 1]].reshape(H + 1, 64)
    out = trunk_unflat(f, obs_dim, k, H, d, offs) + [hd[:H, :act_dim].t().contiguous(), hd[H, :act_dim].clone(),
                                                     hd[:H, 32:32 + act_dim].t().contiguous(), hd[H, 32:32 + act_dim].clone()]
    return W.unpad_layers(out, sizes) if sizes is not None else out


def critic_flat_to_torch(flat: torch.Tensor, obs_dim: int, act_dim: int, hidden: int = HID, sizes=None, depth: int | None = None) -> list[torch.Tensor]:
    d = len(sizes) if sizes is not None else int(depth or 2)
    H = int(hidden)
    k, offs = mlp_layout(obs_dim + act_dim, H, d, 32)
    f = flat.detach()
    hd = f[offs[d]: offs[d + 1]].reshape(H + 1, 32)
    out = trunk_unflat(f, obs_dim + act_dim, k, H, d, offs) + [hd[:H, 0].reshape(1, H).clone(), hd[H, 0].reshape(1).clone()]
    return W.unpad_layers(out, sizes) if sizes is not None else out


@dataclass
class SACConfig:
    """Hyper-parameters of the reference SAC (sac.py:222-283) + Adam factories."""

    gamma: float = 0.99
    tau: float = 0.005
    n_step: int = 1
    alpha: float = 0.2
    auto_alpha: bool = False
    target_entropy: float = 0.0
    log_alpha0: float = 0.0
    actor_lr: float = 1e-3
    critic_lr: float = 1e-3
    alpha_lr: float = 3e-4
    betas: tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8

    def to_c(self, lr_scale: float = 1.0) -> SACHParams:
        return SACHParams(self.actor_lr * lr_scale, self.critic_lr * lr_scale, self.alpha_lr * lr_scale,
                          self.betas[0], self.betas[1], self.adam_eps, self.tau, self.alpha, self.target_entropy,
                          int(self.auto_alpha), 0)


class SACEngine:
    """State of one SAC learner on one GPU."""

    def __init__(self, obs_dim: int, act_dim: int, actor: torch.Tensor, critic1: torch.Tensor,
                 critic2: torch.Tensor, cfg: SACConfig, hidden: int = HID, depth: int = 2, max_action: float = 0.0,
                 activation: str = "relu"):
        """`hidden`: width of the Net[hidden] * depth trunks (utils/net/common.py:246-369; [256, 256] in mujoco_sac.py): any
        multiple of 32 up to 1024, 1 .. 6 hidden layers -- [256, 256] runs on the fused three-layer kernels, everything else on the
        per-layer GEMMs.  `max_action` > 0: the class-default BOUNDED actor, mu = max_action * tanh(mu) (continuous.py:230-231);
        0 = `unbounded=True` as in the examples."""
        if not actor.is_cuda:
            raise RuntimeError("SACEngine needs parameters on an MI355X (no CPU fallback)")
        if activation not in ACTIVATIONS:
            raise NotImplementedError("activation must be 'relu' (Net's default) or 'tanh'")
        self.hidden, self.depth, self.max_action, self.activation = int(hidden), int(depth), float(max_action), activation
        n_actor, n_critic = mlp_layout(obs_dim, self.hidden, self.depth, 64)[1][-1], mlp_layout(obs_dim + act_dim, self.hidden, self.depth, 32)[1][-1]
        if actor.numel() != n_actor or critic1.numel() != n_critic or critic2.numel() != n_critic:
            raise ValueError("flat parameter vectors do not match ts_mlp_layout")
        self.obs_dim, self.act_dim, self.cfg = obs_dim, act_dim, cfg
        self.lay = layout(obs_dim, act_dim, self.hidden) if self.depth == 2 else None       # (ts_sac_layout_h: two hidden layers)
        self.device = actor.device
        cl = lambda t: t.detach().float().contiguous().clone()  # noqa: E731
        self.actor, self.critic1, self.critic2 = cl(actor), cl(critic1), cl(critic2)
        self.critic1_old, self.critic2_old = cl(critic1), cl(critic2)              # ddpg.py:262, td3.py:90-91
        z = torch.zeros_like
        self.actor_m, self.actor_v = z(self.actor), z(self.actor)
        self.critic1_m, self.critic1_v = z(self.critic1), z(self.critic1)
        self.critic2_m, self.critic2_v = z(self.critic2), z(self.critic2)
        self.log_alpha = torch.full((1,), cfg.log_alpha0, dtype=torch.float32, device=self.device)
        self.log_alpha_m, self.log_alpha_v = z(self.log_alpha), z(self.log_alpha)
        self.adam_step = 0
        self._ws = _lib.default_workspace(self.device.index or 0)

    def _state_c(self) -> SACStateC:
        names = [n for n, _ in SACStateC._fields_]
        return SACStateC(*[getattr(self, n).data_ptr() for n in names])

    def _f32(self, x, shape=None) -> torch.Tensor:
        t = torch.as_tensor(x, device=self.device).to(torch.float32).contiguous()
        return t if shape is None else t.reshape(shape)

    @property
    def alpha(self) -> torch.Tensor:
        """Alpha.value (sac.py:164-201) as a device scalar."""
        if self.cfg.auto_alpha:
            return self.log_alpha.exp()
        return torch.full((1,), self.cfg.alpha, dtype=torch.float32, device=self.device)

    # -- SACPolicy.forward ---------------------------------------------------------------------------------
    def policy_forward(self, obs, noise=None):
        """-> (act float32[B, A] tanh-squashed, log_prob float32[B, 1]); noise None = deterministic mode."""
        obs = self._f32(obs)
        b = obs.shape[0]
        noise = None if noise is None else self._f32(noise, (b, self.act_dim))
        act = torch.empty((b, self.act_dim), dtype=torch.float32, device=self.device)
        logp = torch.empty(b, dtype=torch.float32, device=self.device)
        use_hidden(self._ws, self.hidden, self.depth, self.max_action, self.activation)
        _lib.check(_lib.load().ts_sac_policy_forward(
            self._ws.handle, _lib.ptr(self.actor), _lib.ptr(obs), _lib.ptr(noise), _lib.i64(b), _lib.i64(self.obs_dim),
            _lib.i64(self.act_dim), _lib.ptr(act), _lib.ptr(logp), None, _lib.current_stream(self.device)))
        return act, logp.unsqueeze(-1)

    # -- _target_q ---------------------------------------------------------------------------------------------
    def target_q(self, obs_next, noise) -> torch.Tensor:
        obs_next = self._f32(obs_next)
        b = obs_next.shape[0]
        noise = self._f32(noise, (b, self.act_dim))
        out = torch.empty(b, dtype=torch.float32, device=self.device)
        use_hidden(self._ws, self.hidden, self.depth, self.max_action, self.activation)
        _lib.check(_lib.load().ts_sac_target_q(
            self._ws.handle, _lib.ptr(self.actor), _lib.ptr(self.critic1_old), _lib.ptr(self.critic2_old),
            _lib.ptr(self.log_alpha if self.cfg.auto_alpha else None), _lib.f64(self.cfg.alpha), _lib.ptr(obs_next),
            _lib.ptr(noise), _lib.i64(b), _lib.i64(self.obs_dim), _lib.i64(self.act_dim), _lib.ptr(out),
            _lib.current_stream(self.device)))
        return out

    # -- _preprocess_batch ---------------------------------------------------------------------------------------
    def _rows_ok(self, buffer: DeviceReplayBuffer) -> bool:
        """The entry points that read their rows straight from the buffer columns (ts_sac_*_rows) need float32 columns of
        the engine's widths, float64 rewards and uint8 flags -- what DeviceReplayBuffer keeps."""
        o, a, n = buffer.obs, buffer.act, buffer.obs_next
        return (o is not None and a is not None and n is not None and o.dtype == a.dtype == n.dtype == torch.float32
                and o.is_contiguous() and a.is_contiguous() and n.is_contiguous() and o.dim() == 2 and a.dim() == 2
                and o.shape[1] == self.obs_dim and a.shape[1] == self.act_dim and buffer.rew.dtype == torch.float64
                and buffer.terminated.dtype == torch.uint8 and not os.environ.get("TS_SAC_NO_ROWS"))

    def preprocess(self, buffer: DeviceReplayBuffer, indices, noise) -> torch.Tensor:
        """n-step returns float32[I] with target_q_fn = _target_q (ddpg.py:287-301); obs_next from the buffer's stored column or obs[next(index)] (buffer_base.py:622-626).
        n_step = 1 (SAC's default): one launch sequence gathers obs_next inside the input packing and ends with the 1-step
        return (ts_sac_returns_rows), bit-identical to the general path below."""
        if self.cfg.n_step == 1 and self._rows_ok(buffer):
            idx = _i64_dev(indices, self.device).reshape(-1).contiguous()
            b = idx.numel()
            noise = self._f32(noise, (b, self.act_dim))
            out = torch.empty(b, dtype=torch.float32, device=self.device)
            use_hidden(self._ws, self.hidden, self.depth, self.max_action, self.activation)
            _lib.check(_lib.load().ts_sac_returns_rows(
                self._ws.handle, _lib.ptr(self.actor), _lib.ptr(self.critic1_old), _lib.ptr(self.critic2_old),
                _lib.ptr(self.log_alpha if self.cfg.auto_alpha else None), _lib.f64(self.cfg.alpha), _lib.ptr(buffer.obs_next),
                _lib.ptr(buffer.rew), _lib.ptr(buffer.terminated), _lib.ptr(idx), _lib.ptr(noise), _lib.i64(b),
                _lib.i64(self.obs_dim), _lib.i64(self.act_dim), _lib.f64(self.cfg.gamma), _lib.ptr(out),
                _lib.current_stream(self.device)))
            return out

        def tq_fn(buf, after):
            return self.target_q(buf.obs_next_rows(after), noise)

        class _B:
            pass

        return compute_nstep_return(_B(), buffer, indices, tq_fn, self.cfg.gamma, self.cfg.n_step).returns.reshape(-1)

    # -- SAC._update_with_batch -----------------------------------------------------------------------------------
    def update_with_batch(self, obs, act, returns, noise, weight=None, grads_out: torch.Tensor | None = None,
                          lr_scale: float = 1.0):
        """-> (stats float32[5] device = {actor_loss, critic1_loss, critic2_loss, alpha, alpha_loss},
        weight float32[B] = (td1 + td2) / 2)."""
        obs, act = self._f32(obs), self._f32(act)
        b = obs.shape[0]
        returns, noise = self._f32(returns, (b,)), self._f32(noise, (b, self.act_dim))
        weight = None if weight is None else self._f32(weight, (b,))
        if act.shape != (b, self.act_dim) or obs.shape != (b, self.obs_dim):
            raise ValueError("obs / act shapes do not match the engine")
        self.adam_step += 1
        stats = torch.empty(5, dtype=torch.float32, device=self.device)
        w_out = torch.empty(b, dtype=torch.float32, device=self.device)
        st, hp = self._state_c(), self.cfg.to_c(lr_scale)
        use_hidden(self._ws, self.hidden, self.depth, self.max_action, self.activation)
        _lib.check(_lib.load().ts_sac_update(
            self._ws.handle, C.byref(st), _lib.i64(self.adam_step), _lib.ptr(obs), _lib.ptr(act), _lib.ptr(returns),
            _lib.ptr(weight), _lib.ptr(noise), _lib.i64(b), _lib.i64(self.obs_dim), _lib.i64(self.act_dim),
            C.byref(hp), _lib.ptr(stats), _lib.ptr(w_out), _lib.ptr(grads_out), _lib.current_stream(self.device)))
        return stats, w_out

    def update_with_rows(self, buffer: DeviceReplayBuffer, indices, returns, noise, weight=None, lr_scale: float = 1.0):
        """update_with_batch(buffer.obs[indices], buffer.act[indices], ...) without the two gather launches: the input
        packing kernel reads the rows (ts_sac_update_rows).  Bit-identical."""
        if not self._rows_ok(buffer):
            return self.update_with_batch(gather_rows(buffer.obs, indices), gather_rows(buffer.act, indices), returns, noise,
                                          weight, lr_scale=lr_scale)
        idx = _i64_dev(indices, self.device).reshape(-1).contiguous()
        b = idx.numel()
        returns, noise = self._f32(returns, (b,)), self._f32(noise, (b, self.act_dim))
        weight = None if weight is None else self._f32(weight, (b,))
        self.adam_step += 1
        stats = torch.empty(5, dtype=torch.float32, device=self.device)
        w_out = torch.empty(b, dtype=torch.float32, device=self.device)
        st, hp = self._state_c(), self.cfg.to_c(lr_scale)
        use_hidden(self._ws, self.hidden, self.depth, self.max_action, self.activation)
        _lib.check(_lib.load().ts_sac_update_rows(
            self._ws.handle, C.byref(st), _lib.i64(self.adam_step), _lib.ptr(buffer.obs), _lib.ptr(buffer.act), _lib.ptr(idx),
            _lib.ptr(returns), _lib.ptr(weight), _lib.ptr(noise), _lib.i64(b), _lib.i64(self.obs_dim), _lib.i64(self.act_dim),
            C.byref(hp), _lib.ptr(stats), _lib.ptr(w_out), _lib.current_stream(self.device)))
        return stats, w_out

    def learn_rows(self, buffer: DeviceReplayBuffer, indices, noise=None, noise_key=None, weight=None, lr_scale: float = 1.0,
                   noise_streams: int = 1):
        """`preprocess(buffer, indices, noise[0])` + `update_with_rows(buffer, indices, returns, noise[1])` as ONE library call
        (ts_sac_learn_rows; n_step = 1): -> (stats float32[5], weight float32[B], returns float32[B], noise float32[2, B, act]),
        bit-identical to the two calls.  `noise`: float32[2, B, act_dim], or None with `noise_key = (seed, offset)`: the engine
        draws `normal_noise((2, B, act_dim), seed, offset)` inside its first launch (`noise_streams=2`: the two halves as
        `normal_noise((B, act_dim), seed, offset)` and `(..., offset + 1)`, the hooks' two consecutive draws).  On the one-launch chains the call saves three
        of the update's 22 launches (one packing pass, the noise draw, the return kernel)."""
        if self.cfg.n_step != 1 or not self._rows_ok(buffer):
            raise NotImplementedError("learn_rows: n_step = 1 on float32 replay columns (use preprocess + update_with_rows)")
        idx = _i64_dev(indices, self.device).reshape(-1).contiguous()
        b = idx.numel()
        if noise is None:
            if noise_key is None:
                raise ValueError("learn_rows: pass noise or noise_key=(seed, offset)")
            noise2 = torch.empty((2, b, self.act_dim), dtype=torch.float32, device=self.device)
            fill, seed, off = (2 if noise_streams == 2 else 1), int(noise_key[0]) & (2**64 - 1), int(noise_key[1]) & (2**64 - 1)
        else:
            noise2, fill, seed, off = self._f32(noise, (2, b, self.act_dim)), 0, 0, 0
        weight = None if weight is None else self._f32(weight, (b,))
        self.adam_step += 1
        stats = torch.empty(5, dtype=torch.float32, device=self.device)
        w_out = torch.empty(b, dtype=torch.float32, device=self.device)
        ret = torch.empty(b, dtype=torch.float32, device=self.device)
        st, hp = self._state_c(), self.cfg.to_c(lr_scale)
        rp = SACReplayC(buffer.obs.data_ptr(), buffer.act.data_ptr(), buffer.obs_next.data_ptr(), buffer.rew.data_ptr(),
                        buffer.terminated.data_ptr())
        use_hidden(self._ws, self.hidden, self.depth, self.max_action, self.activation)
        _lib.check(_lib.load().ts_sac_learn_rows(
            self._ws.handle, C.byref(st), _lib.i64(self.adam_step), C.byref(rp), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(noise2),
            C.c_int(fill), C.c_uint64(seed), C.c_uint64(off), _lib.i64(b), _lib.i64(self.obs_dim), _lib.i64(self.act_dim),
            C.byref(hp), _lib.f64(self.cfg.gamma), _lib.ptr(ret), _lib.ptr(stats), _lib.ptr(w_out), _lib.current_stream(self.device)))
        return stats, w_out, ret, noise2

    # -- the same update in four phases (data-parallel replicas all-reduce between "grad" and "apply") -----------
    PHASE_CRITIC_GRAD, PHASE_CRITIC_APPLY, PHASE_ACTOR_GRAD, PHASE_ACTOR_APPLY = 1, 2, 4, 8

    def begin_phased_update(self, obs, act, returns, noise, weight=None, lr_scale: float = 1.0) -> dict:
        """Checks / converts the minibatch once and advances the optimizer step; the returned context goes to the
        four `update_phase` calls of this update (order 1, 2, 4, 8; nothing else may use the workspace in between)."""
        obs, act = self._f32(obs), self._f32(act)
        b = obs.shape[0]
        if act.shape != (b, self.act_dim) or obs.shape != (b, self.obs_dim):
            raise ValueError("obs / act shapes do not match the engine")
        self.adam_step += 1
        return {"obs": obs, "act": act, "returns": self._f32(returns, (b,)), "noise": self._f32(noise, (b, self.act_dim)),
                "weight": None if weight is None else self._f32(weight, (b,)), "b": b, "hp": self.cfg.to_c(lr_scale),
                "stats": torch.zeros(5, dtype=torch.float32, device=self.device),
                "w_out": torch.empty(b, dtype=torch.float32, device=self.device)}

    def exchange_floats(self) -> tuple[int, int]:
        """Sizes of the two exchange buffers: ([critic1 | critic2], [actor | -mean(log_prob)])."""
        return 2 * self.critic1.numel(), self.actor.numel() + 1

    def update_phase(self, ctx: dict, phase: int, grads: torch.Tensor) -> None:
        st = self._state_c()
        use_hidden(self._ws, self.hidden, self.depth, self.max_action, self.activation)
        _lib.check(_lib.load().ts_sac_update_phase(
            self._ws.handle, C.byref(st), _lib.i64(self.adam_step), _lib.ptr(ctx["obs"]), _lib.ptr(ctx["act"]),
            _lib.ptr(ctx["returns"]), _lib.ptr(ctx["weight"]), _lib.ptr(ctx["noise"]), _lib.i64(ctx["b"]),
            _lib.i64(self.obs_dim), _lib.i64(self.act_dim), C.byref(ctx["hp"]), C.c_int(phase), _lib.ptr(ctx["stats"]),
            _lib.ptr(ctx["w_out"]), _lib.ptr(grads), _lib.current_stream(self.device)))
