"""Reinforce learn() on one MI355X: discounted returns (ts_gae_scan, lambda = 1) + vanilla policy-gradient minibatch steps
(ts_npg_actor_grad + ts_adam_step) for the reference's Reinforce (tianshou/algorithm/modelfree/reinforce.py) with the actor of
examples/mujoco/mujoco_reinforce.py:84-103 (Net[h, h] tanh, unbounded Gaussian, state-independent sigma_param).

The actor uses the flat layout of ts_npg_layout (tianshou_amd.npg.actor_flat_from_torch).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from . import ppo as _ppo
from .npg import actor_flat_from_torch, layout
from .ppo_cnn import run_minibatches
from .returns import _i64_dev, gae_scan


@dataclass
class ReinforceConfig:
    gamma: float = 0.99
    return_standardization: bool = False
    lr: float = 1e-3
    betas: tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8
    max_grad_norm: float | None = None


def rms_merge_tensors(rms: torch.Tensor, s1: torch.Tensor, s2: torch.Tensor, n: int) -> torch.Tensor:
    """ppo.rms_merge (RunningMeanStd.update, utils/statistics.py:99-114) on float64 tensors: rms = [mean, var, count], s1 / s2
    = sum and sum of squares of the batch (0-dim) -> the new [mean, var, count].  The same IEEE operations in the same order
    as the host version, wherever the tensors live."""
    mean, var, count = rms[0], rms[1], rms[2]
    b_mean = s1 / n
    b_var = torch.clamp(s2 / n - b_mean * b_mean, min=0.0)
    delta, tot = b_mean - mean, count + n
    return torch.stack([mean + delta * n / tot, (var * count + b_var * n + delta * delta * count * n / tot) / tot, tot])


class ReinforceEngine:
    """State of one Reinforce learner on one GPU: flat actor + Adam moments + the running return statistics."""

    def __init__(self, obs_dim: int, act_dim: int, hidden: int, actor: torch.Tensor, cfg: ReinforceConfig):
        if not actor.is_cuda:
            raise RuntimeError("ReinforceEngine needs parameters on an MI355X (no CPU fallback)")
        lay = layout(obs_dim, hidden, act_dim)
        if actor.numel() != lay["actor_count"]:
            raise ValueError("flat actor vector does not match ts_npg_layout")
        self.obs_dim, self.act_dim, self.hidden, self.cfg, self.lay = obs_dim, act_dim, hidden, cfg, lay
        self.device = actor.device
        self.actor = actor.detach().float().contiguous().clone()
        self.adam_m, self.adam_v = torch.zeros_like(self.actor), torch.zeros_like(self.actor)
        self.adam_step = 0
        self._rms_host, self._rms_dev = [0.0, 1.0, 0.0], None
        self._grad = torch.empty_like(self.actor)
        self._ws = _lib.default_workspace(self.device.index or 0)
        self._fused = None               # (PPOEngine, positions in the actor vector, indices in its parameter vector)

    # RunningMeanStd (mean, var, count; statistics.py:60-114).  `preprocess` keeps it on the device -- reading it here is the
    # only host synchronisation of an update loop; assigning it (hooks, checkpoints) replaces the device copy.
    @property
    def ret_rms(self) -> list:
        if self._rms_dev is not None:
            self._rms_host = [float(x) for x in self._rms_dev.tolist()]
        return list(self._rms_host)

    @ret_rms.setter
    def ret_rms(self, value) -> None:
        self._rms_host, self._rms_dev = [float(x) for x in value], None

    def _f32(self, x, shape=None) -> torch.Tensor:
        t = torch.as_tensor(x, device=self.device).to(torch.float32).contiguous()
        return t if shape is None else t.reshape(shape)

    # -- DiscountedReturnComputation.add_discounted_returns (reinforce.py:266-310) -------------------------------------
    def preprocess(self, rew, terminated, truncated, cut_pos=None) -> torch.Tensor:
        """Batch-order arrays of the whole buffer -> batch.returns float32[N] (standardised when configured)."""
        cfg = self.cfg
        term = torch.as_tensor(terminated, device=self.device).reshape(-1)
        n = term.numel()
        if self._rms_dev is None:
            self._rms_dev = torch.tensor(self._rms_host, dtype=torch.float64, device=self.device)
        mean, var, count = self._rms_dev[0], self._rms_dev[1], self._rms_dev[2]           # float64 device scalars
        v_next = torch.where(term.bool(), 0.0, mean).to(torch.float32)              # full(ret_rms.mean) * value_mask
        v_s = torch.roll(v_next, 1)                                                 # algorithm_base.py:712
        cut = None if cut_pos is None else _i64_dev(cut_pos, self.device)
        out = gae_scan(v_s, v_next, torch.as_tensor(rew, device=self.device), term,
                       torch.as_tensor(truncated, device=self.device), cut, gamma=cfg.gamma, gae_lambda=1.0,
                       want_f64=cfg.return_standardization, want_ret_stats=cfg.return_standardization)
        if not cfg.return_standardization:
            return out["returns"]
        ret = ((out["ret64"] - mean) / torch.sqrt(var + 1e-8)).to(torch.float32)          # reinforce.py:305-307
        # ret_rms.update(unnormalised returns), statistics.py:99-114: the same float64 operations in the same order, on the
        # device (no host round trip between two updates)
        self._rms_dev = rms_merge_tensors(self._rms_dev, out["ret_sum"], out["ret_sumsq"], n)
        return ret

    # -- one minibatch (reinforce.py:371-380) ----------------------------------------------------------------------------
    def gradient(self, obs, act, returns, grad_out: torch.Tensor | None = None) -> torch.Tensor:
        """loss float32[1]; the flat gradient lands in grad_out (default: the engine's scratch vector)."""
        obs = self._f32(obs).reshape(-1, self.obs_dim)
        b = obs.shape[0]
        act, returns = self._f32(act, (b, self.act_dim)), self._f32(returns, (b,))
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        g = self._grad if grad_out is None else grad_out
        _lib.check(_lib.load().ts_npg_actor_grad(
            self._ws.handle, _lib.ptr(self.actor), _lib.i64(self.obs_dim), _lib.i64(self.hidden), _lib.i64(self.act_dim),
            _lib.ptr(obs), _lib.ptr(act), _lib.ptr(returns), _lib.i64(b), _lib.ptr(loss), _lib.ptr(g),
            _lib.current_stream(self.device)))
        return loss

    def apply_gradient(self, grad: torch.Tensor | None = None) -> None:
        """clip_grad_norm_ + Adam on the flat gradient (algorithm_base.py:496-500)."""
        cfg = self.cfg
        self.adam_step += 1
        g = self._grad if grad is None else grad
        _lib.check(_lib.load().ts_adam_step(
            self._ws.handle, _lib.ptr(self.actor), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v), _lib.ptr(g),
            _lib.i64(self.actor.numel()), _lib.i64(self.adam_step), _lib.f64(cfg.lr), _lib.f64(cfg.betas[0]),
            _lib.f64(cfg.betas[1]), _lib.f64(cfg.adam_eps), _lib.f64(cfg.max_grad_norm or 0.0),
            _lib.current_stream(self.device)))

    def step(self, obs, act, returns) -> torch.Tensor:
        loss = self.gradient(obs, act, returns)
        self.apply_gradient()
        return loss

    # -- the whole minibatch loop on the fused actor-critic step kernel ------------------------------------------------------
    def fused_supported(self) -> bool:
        """Net[64, 64], obs_dim <= 31, act_dim <= 8: the shapes of ts_ppo.hip's step kernel (TS_REINFORCE_GEMM=1 keeps the
        per-layer GEMM path; read per call)."""
        return (self.hidden == _ppo.HIDDEN and self.obs_dim <= 31 and self.act_dim <= 8
                and not os.environ.get("TS_REINFORCE_GEMM"))

    def _fused_engine(self):
        """Reinforce's loss -(log_prob * returns).mean() (reinforce.py:371-380) is A2C's actor loss (a2c.py:266-267) with
        adv := returns, vf_coef = ent_coef = 0 and no advantage normalisation: the fused step kernel (ts_ppo.hip, algo a2c)
        computes it, its gradient, the global-norm clip and Adam in three launches per minibatch instead of ~30 per-layer
        ones.  The critic half of that engine is a block of zeros whose gradient is exactly zero (vf_coef = 0), so its Adam
        moments and parameters stay zero.  The actor and its moments are copied between the two flat layouts around every
        update (one indexed copy each: `pos` / `idx` come from running the layout converter on an index vector)."""
        if self._fused is None:
            cfg = self.cfg
            pc = _ppo.PPOConfig(algo="a2c", vf_coef=0.0, ent_coef=0.0, advantage_normalization=False,
                                max_grad_norm=cfg.max_grad_norm, lr=cfg.lr, betas=cfg.betas, adam_eps=cfg.adam_eps,
                                nets=0 if os.environ.get("TS_REINFORCE_BOTH_NETS") else 1)    # 1: the actor's half of the step only
            eng = _ppo.PPOEngine(self.obs_dim, self.act_dim,
                                 torch.zeros(_ppo.param_count(self.obs_dim, self.act_dim), device=self.device), pc)
            shapes, off, t = _ppo.param_shapes(self.obs_dim, self.act_dim), 0, []
            for k in _ppo.PARAM_ORDER[:7]:                       # the actor's seven tensors, numbered 1 .. in flat order
                n = int(np.prod(shapes[k]))
                t.append((torch.arange(off, off + n, dtype=torch.float32) + 1.0).reshape(shapes[k]))
                off += n
            marks = actor_flat_from_torch(t, self.obs_dim, self.hidden, self.act_dim, device="cpu")
            pos = torch.nonzero(marks > 0).reshape(-1)
            idx = (marks[pos] - 1.0).long()
            self._fused = (eng, pos.to(self.device), idx.to(self.device))
        return self._fused

    # -- Reinforce._update_with_batch ------------------------------------------------------------------------------------
    def update(self, obs, act, returns, batch_size: int | None, repeat: int, perms=None):
        """-> (losses float32[steps, 1], steps)."""
        obs = self._f32(obs).reshape(-1, self.obs_dim)
        n = obs.shape[0]
        act, returns = self._f32(act, (n, self.act_dim)), self._f32(returns, (n,))
        if self.fused_supported():
            eng, pos, idx = self._fused_engine()
            for src, dst in ((self.actor, eng.params), (self.adam_m, eng.adam_m), (self.adam_v, eng.adam_v)):
                dst[idx] = src[pos]
            eng.adam_step, eng.cfg.lr = self.adam_step, self.cfg.lr
            zeros = torch.zeros(n, dtype=torch.float32, device=self.device)
            b = {"obs": obs, "act": act, "adv": returns, "returns": zeros, "logp_old": zeros, "v_s": zeros}
            losses, steps = eng.update(b, batch_size, repeat, perms)[:2]
            for dst, src in ((self.actor, eng.params), (self.adam_m, eng.adam_m), (self.adam_v, eng.adam_v)):
                dst[pos] = src[idx]
            self.adam_step = eng.adam_step
            return losses[:, :1].contiguous(), steps
        return run_minibatches(self.device, n, batch_size, repeat, perms,
                               lambda rows: self.step(obs[rows], act[rows], returns[rows]))


class NetReinforceEngine(ReinforceEngine):
    """ReinforceEngine's interface for actors outside Net[h, h] tanh (round 6): any `Net(hidden_sizes=[...], activation=Tanh |
    ReLU | None)` trunk (utils/net/common.py:90-178, 246-369), the reference's DEFAULT bounded actor (max_action * tanh on mu,
    utils/net/continuous.py:194, 230-231), Adam with weight decay or RMSprop (algorithm/optim.py:89-140).

    Reinforce's loss -(log_prob * returns).mean() (reinforce.py:371-380) is A2C's actor loss (a2c.py:266-267) with adv := returns,
    vf_coef = ent_coef = 0 and no advantage normalisation, so the minibatch loop runs on the per-layer actor-critic engine
    (ppo_wide.NetPPOEngine: ts_ppo_net_step) beside a one-layer critic of zeros whose gradient is exactly zero (vf_coef = 0,
    returns = 0): its parameters and optimizer state stay zero under Adam and RMSprop alike, the joint gradient norm is the
    actor's.  `actor` / `adam_m` / `adam_v` are views of that engine's vectors (ts_net_layout order)."""

    def __init__(self, obs_dim: int, act_dim: int, hidden, activation: str, actor: torch.Tensor, cfg: ReinforceConfig,
                 max_action: float | None = None, optimizer: dict | None = None, layer_norm: bool = False, ln_eps: float = 1e-5):
        from .ppo_wide import NetPPOEngine

        if not actor.is_cuda:
            raise RuntimeError("NetReinforceEngine needs parameters on an MI355X (no CPU fallback)")
        self.obs_dim, self.act_dim, self.hidden, self.cfg = obs_dim, act_dim, None, cfg
        self.hidden_sizes, self.activation = [int(h) for h in hidden], activation
        self.device = actor.device
        opt = dict(optimizer or {})
        pc = _ppo.PPOConfig(algo="a2c", vf_coef=0.0, ent_coef=0.0, advantage_normalization=False, max_grad_norm=cfg.max_grad_norm,
                            lr=cfg.lr, betas=cfg.betas, adam_eps=cfg.adam_eps, max_action=max_action,
                            optimizer=opt.get("optimizer", "adam"), weight_decay=opt.get("weight_decay", 0.0),
                            rms_alpha=opt.get("rms_alpha", 0.99), rms_momentum=opt.get("rms_momentum", 0.0),
                            rms_centered=opt.get("rms_centered", False))
        # (layer_norm: MLP(norm_layer=nn.LayerNorm) trunks, TS_NET_LAYERNORM -- the zero critic then carries a zero gamma / beta
        # pair as well: its output and every gradient stay exactly zero)
        probe = _lib.NetDesc.make(obs_dim, [32], activation, _lib.NetDesc.LAYERNORM if layer_norm else 0, ln_eps=ln_eps)
        out = (C.c_int64 * 3)()
        _lib.check(_lib.load().ts_net_layout(C.byref(probe), _lib.i64(act_dim), out))
        n_critic = int(out[2])
        flat = torch.cat([actor.detach().float().reshape(-1), torch.zeros(n_critic, device=self.device)]).contiguous()
        self._net = NetPPOEngine(obs_dim, act_dim, self.hidden_sizes, [32], activation, flat, pc, layer_norm=layer_norm, ln_eps=ln_eps)
        if self._net.n_actor != actor.numel():
            raise ValueError("flat actor vector does not match ts_net_layout")
        self._rms_host, self._rms_dev = [0.0, 1.0, 0.0], None
        self._ws = self._net._ws

    # the learner's state lives in the per-layer engine's vectors
    @property
    def actor(self) -> torch.Tensor:
        return self._net.params[: self._net.n_actor]

    @property
    def adam_m(self) -> torch.Tensor:
        return self._net.adam_m[: self._net.n_actor]

    @adam_m.setter
    def adam_m(self, v) -> None:
        self._net.adam_m[: self._net.n_actor] = v

    @property
    def adam_v(self) -> torch.Tensor:
        return self._net.adam_v[: self._net.n_actor]

    @adam_v.setter
    def adam_v(self, v) -> None:
        self._net.adam_v[: self._net.n_actor] = v

    @property
    def adam_step(self) -> int:
        return self._net.adam_step

    @adam_step.setter
    def adam_step(self, v) -> None:
        self._net.adam_step = int(v)

    def fused_supported(self) -> bool:
        return False

    def update(self, obs, act, returns, batch_size: int | None, repeat: int, perms=None):
        """-> (losses float32[steps, 1], steps)."""
        obs = self._f32(obs).reshape(-1, self.obs_dim)
        n = obs.shape[0]
        act, returns = self._f32(act, (n, self.act_dim)), self._f32(returns, (n,))
        self._net.cfg.lr = self.cfg.lr                                    # (schedulers: the hooks refresh cfg.lr per update)
        zeros = torch.zeros(n, dtype=torch.float32, device=self.device)
        b = {"obs": obs, "act": act, "adv": returns, "returns": zeros, "logp_old": zeros, "v_s": zeros}
        losses, steps = self._net.update(b, batch_size, repeat, perms)[:2]
        return losses[:, :1].contiguous(), steps
