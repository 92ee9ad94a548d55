"""NPG / TRPO learn() path on the MI355X engine.

Mirrors, on device tensors:
    NPG._preprocess_batch                tianshou/algorithm/modelfree/npg.py:123-138 (a2c.py:115-153 + log pi_old + whole-batch
                                         advantage normalisation)
    NPG._update_with_batch               npg.py:140-193 (vanilla gradient, conjugate gradients on Fisher-vector products,
                                         fixed-size natural-gradient step, critic iterations)
    TRPO._update_with_batch              trpo.py:123-214 (ratio surrogate, step size from the KL bound, backtracking line search)
Networks: examples/mujoco/mujoco_npg.py:103-128 (the PPO nets).  No CPU path.
"""
from __future__ import annotations

import os

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from . import widths as W
from .buffer import _i64_dev, gather_rows_multi
from .ppo_cnn import run_minibatches
from .returns import gae_scan

HEAD = 32


_CRITIC_STREAMS: dict = {}     # device index -> the side stream of the critic iterations (NPGEngine._critic_stream)

class NPGHParams(C.Structure):
    """struct ts_npg_hparams (include/tsengine.h)."""

    _fields_ = [("damping", C.c_double), ("trust_region_size", C.c_double), ("max_kl", C.c_double),
                ("backtrack_coeff", C.c_double), ("residual_tol", C.c_double), ("algo", C.c_int32), ("cg_iters", C.c_int32),
                ("max_backtracks", C.c_int32), ("reserved", C.c_int32)]


def layout(obs_dim: int, hidden: int, act_dim: int) -> dict[str, int]:
    out = (C.c_int64 * 3)()
    _lib.check(_lib.load().ts_npg_layout(_lib.i64(obs_dim), _lib.i64(hidden), _lib.i64(act_dim), out))
    return dict(zip(["k0", "actor_count", "critic_count"], (int(v) for v in out)))


def _block(w: torch.Tensor, b: torch.Tensor, k_pad: int, n_pad: int) -> torch.Tensor:
    wb = torch.zeros((k_pad + 1, n_pad), dtype=torch.float32)
    wb[: w.shape[1], : w.shape[0]] = w.detach().float().cpu().t()
    wb[k_pad, : b.shape[0]] = b.detach().float().cpu()
    return wb.reshape(-1)


def actor_flat_from_torch(t: list[torch.Tensor], obs_dim: int, hidden: int, act_dim: int, device="cuda") -> torch.Tensor:
    """[w1, b1, w2, b2, w_mu, b_mu, sigma_param] (nn.Linear layout; sigma_param of any shape with act_dim elements).  Widths
    other than [hidden, hidden] are embedded by zero padding (`tianshou_amd.widths`: tanh(0) = 0, so a padding unit is inert)."""
    t = W.pad_two_layer(list(t[:6]), hidden) + list(t[6:])
    k0 = layout(obs_dim, hidden, act_dim)["k0"]
    ls = torch.zeros(HEAD, dtype=torch.float32)
    ls[:act_dim] = t[6].detach().float().cpu().reshape(-1)
    return torch.cat([_block(t[0], t[1], k0, hidden), _block(t[2], t[3], hidden, hidden), _block(t[4], t[5], hidden, HEAD),
                      ls]).to(device).contiguous()


def critic_flat_from_torch(t: list[torch.Tensor], obs_dim: int, hidden: int, device="cuda") -> torch.Tensor:
    """[w1, b1, w2, b2, w_v [1, h], b_v [1]] (widths as in `actor_flat_from_torch`)."""
    t = W.pad_two_layer(list(t), hidden)
    k0 = layout(obs_dim, hidden, 1)["k0"]
    return torch.cat([_block(t[0], t[1], k0, hidden), _block(t[2], t[3], hidden, hidden),
                      _block(t[4], t[5], hidden, HEAD)]).to(device).contiguous()


def _unblocks(flat: torch.Tensor, obs_dim: int, hidden: int, k0: int, n_out: int):
    n1, n2 = (k0 + 1) * hidden, (hidden + 1) * hidden
    l1 = flat[:n1].reshape(k0 + 1, hidden)
    l2 = flat[n1:n1 + n2].reshape(hidden + 1, hidden)
    hd = flat[n1 + n2:n1 + n2 + (hidden + 1) * HEAD].reshape(hidden + 1, HEAD)
    return [l1[:obs_dim].t().contiguous(), l1[k0].clone(), l2[:hidden].t().contiguous(), l2[hidden].clone(),
            hd[:hidden, :n_out].t().contiguous(), hd[hidden, :n_out].clone()]


def actor_flat_to_torch(flat: torch.Tensor, obs_dim: int, hidden: int, act_dim: int, sizes=None) -> list[torch.Tensor]:
    """sizes = (h1, h2): the widths of the torch network embedded in Net[hidden, hidden]."""
    k0 = layout(obs_dim, hidden, act_dim)["k0"]
    f = flat.detach()
    body = _unblocks(f, obs_dim, hidden, k0, act_dim)
    return (W.unpad_two_layer(body, *sizes) if sizes is not None else body) + [f[-HEAD:][:act_dim].clone()]


def critic_flat_to_torch(flat: torch.Tensor, obs_dim: int, hidden: int, sizes=None) -> list[torch.Tensor]:
    k0 = layout(obs_dim, hidden, 1)["k0"]
    body = _unblocks(flat.detach(), obs_dim, hidden, k0, 1)
    return W.unpad_two_layer(body, *sizes) if sizes is not None else body


@dataclass
class NPGConfig:
    """Hyper-parameters of the reference NPG (npg.py:33-121) / TRPO (trpo.py:31-121) + the critic's Adam."""

    algo: str = "npg"                  # "npg" | "trpo"
    gamma: float = 0.99
    gae_lambda: float = 0.95
    optim_critic_iters: int = 5
    trust_region_size: float = 0.5
    advantage_normalization: bool = True
    return_scaling: bool = False
    damping: float = 0.1
    max_kl: float = 0.01
    backtrack_coeff: float = 0.8
    max_backtracks: int = 10
    lr: float = 1e-3
    betas: tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8
    max_grad_norm: float | None = None

    def to_c(self) -> NPGHParams:
        return NPGHParams(self.damping, self.trust_region_size, self.max_kl, self.backtrack_coeff, 1e-10,
                          {"npg": 0, "trpo": 1}[self.algo], 10, self.max_backtracks, 0)


class NPGEngine:
    """State of one NPG / TRPO learner on one GPU."""

    def __init__(self, obs_dim: int, act_dim: int, hidden: int, actor: torch.Tensor, critic: torch.Tensor, cfg: NPGConfig):
        if not actor.is_cuda:
            raise RuntimeError("NPGEngine needs parameters on an MI355X (no CPU fallback)")
        lay = layout(obs_dim, hidden, act_dim)
        if actor.numel() != lay["actor_count"] or critic.numel() != lay["critic_count"]:
            raise ValueError("flat parameter vectors do not match ts_npg_layout")
        if cfg.algo not in ("npg", "trpo"):
            raise ValueError("algo must be 'npg' or 'trpo'")
        self.obs_dim, self.act_dim, self.hidden, self.cfg, self.lay = obs_dim, act_dim, hidden, cfg, lay
        self.device = actor.device
        cl = lambda t: t.detach().float().contiguous().clone()  # noqa: E731
        self.actor, self.critic = cl(actor), cl(critic)
        self.critic_m, self.critic_v = torch.zeros_like(self.critic), torch.zeros_like(self.critic)
        self.adam_step = 0
        self.ret_rms = [0.0, 1.0, 0.0]
        self._ws = _lib.default_workspace(self.device.index or 0)

    def _dims(self):
        return _lib.i64(self.obs_dim), _lib.i64(self.hidden), _lib.i64(self.act_dim)

    # -- the critic iterations of a minibatch --------------------------------------------------------------------------------
    def critic_steps(self, obs, returns, iters: int) -> torch.Tensor:
        """`iters` critic iterations on one minibatch (npg.py:179-183) in one library call -> the last vf_loss float32[1].
        Net[64, 64] with obs_dim <= 32: one kernel + one small sum per gradient (csrc/ts_npg_q.h; TS_NPG_FVP=0 keeps the
        per-layer GEMM passes)."""
        obs = self._f32(obs).reshape(-1, self.obs_dim)
        b = obs.shape[0]
        returns = self._f32(returns, (b,))
        cfg = self.cfg
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        ws = _lib.default_workspace(self.device.index or 0)         # the CALLING stream's workspace (update() runs this on a side stream)
        _lib.check(_lib.load().ts_npg_critic_steps(
            ws.handle, _lib.ptr(self.critic), _lib.ptr(self.critic_m), _lib.ptr(self.critic_v),
            _lib.i64(self.adam_step + 1), _lib.i64(self.obs_dim), _lib.i64(self.hidden), _lib.ptr(obs), _lib.ptr(returns),
            _lib.i64(b), _lib.i64(iters), _lib.f64(cfg.lr), _lib.f64(cfg.betas[0]), _lib.f64(cfg.betas[1]),
            _lib.f64(cfg.adam_eps), _lib.f64(cfg.max_grad_norm or 0.0), _lib.ptr(loss), _lib.ptr(None),
            _lib.current_stream(self.device)))
        self.adam_step += iters
        return loss

    def _f32(self, x, shape=None) -> torch.Tensor:
        t = torch.as_tensor(x, device=self.device).to(torch.float32).contiguous()
        return t if shape is None else t.reshape(shape)

    def infer(self, obs, act=None, want_v: bool = True, want_mu: bool = False):
        """-> (V float32[B] or None, log_prob float32[B] or None[, mu float32[B, A]])."""
        obs = self._f32(obs).reshape(-1, self.obs_dim)
        b = obs.shape[0]
        act = None if act is None else self._f32(act, (b, self.act_dim))
        v = torch.empty(b, dtype=torch.float32, device=self.device) if want_v else None
        logp = torch.empty(b, dtype=torch.float32, device=self.device) if act is not None else None
        mu = torch.empty((b, self.act_dim), dtype=torch.float32, device=self.device) if want_mu else None
        _lib.check(_lib.load().ts_npg_infer(
            self._ws.handle, _lib.ptr(self.actor), _lib.ptr(self.critic), *self._dims(), _lib.ptr(obs), _lib.ptr(act),
            _lib.i64(b), _lib.ptr(v), _lib.ptr(logp), _lib.ptr(mu), _lib.current_stream(self.device)))
        return (v, logp, mu) if want_mu else (v, logp)

    # -- NPG._preprocess_batch -----------------------------------------------------------------------------------
    def preprocess(self, obs, obs_next, act, rew, terminated, truncated, cut_pos=None) -> dict:
        """Batch-order tensors of the whole buffer (sample_indices(0) order) -> dict(obs, act, v_s, returns, adv, logp_old).
        cut_pos: batch positions of the unfinished slots (algorithm_base.py:715)."""
        cfg = self.cfg
        obs, obs_next = self._f32(obs).reshape(-1, self.obs_dim), self._f32(obs_next).reshape(-1, self.obs_dim)
        n = obs.shape[0]
        act = self._f32(act, (n, self.act_dim))
        v_s, logp_old = self.infer(obs, act)
        v_next, _ = self.infer(obs_next)
        scale = float(np.sqrt(self.ret_rms[1] + 1e-8)) if cfg.return_scaling else 1.0
        cut = None if cut_pos is None else _i64_dev(cut_pos, self.device)
        out = gae_scan(v_s, v_next, torch.as_tensor(rew, device=self.device), torch.as_tensor(terminated, device=self.device),
                       torch.as_tensor(truncated, device=self.device), cut, gamma=cfg.gamma, gae_lambda=cfg.gae_lambda,
                       v_scale=scale, ret_div=scale, want_ret_stats=cfg.return_scaling)
        if cfg.return_scaling:                                                   # statistics.py:99-114
            s1, s2 = float(out["ret_sum"]), float(out["ret_sumsq"])
            b_mean = s1 / n
            b_var = max(s2 / n - b_mean * b_mean, 0.0)
            mean, var, count = self.ret_rms
            delta, tot = b_mean - mean, count + n
            self.ret_rms = [mean + delta * n / tot, (var * count + b_var * n + delta * delta * count * n / tot) / tot, tot]
        adv = out["adv"]
        if cfg.advantage_normalization:                                          # npg.py:136-137 (no epsilon)
            adv = (adv - adv.mean()) / adv.std()
        return {"obs": obs, "act": act, "v_s": v_s, "returns": out["returns"], "adv": adv, "logp_old": logp_old}

    # -- one minibatch --------------------------------------------------------------------------------------------------
    def actor_step(self, obs, act, adv, logp_old=None, want_debug: bool = False):
        """-> stats float32[3] = {actor_loss, kl, step_size}[, debug float32[3, P]]."""
        obs = self._f32(obs).reshape(-1, self.obs_dim)
        b = obs.shape[0]
        act, adv = self._f32(act, (b, self.act_dim)), self._f32(adv, (b,))
        logp_old = None if logp_old is None else self._f32(logp_old, (b,))
        if self.cfg.algo == "trpo" and logp_old is None:
            raise ValueError("TRPO needs logp_old")
        stats = torch.empty(3, dtype=torch.float32, device=self.device)
        dbg = torch.empty((3, self.lay["actor_count"]), dtype=torch.float32, device=self.device) if want_debug else None
        hp = self.cfg.to_c()
        _lib.check(_lib.load().ts_npg_actor_step(
            self._ws.handle, _lib.ptr(self.actor), *self._dims(), _lib.ptr(obs), _lib.ptr(act), _lib.ptr(adv), _lib.ptr(logp_old),
            _lib.i64(b), C.byref(hp), _lib.ptr(stats), _lib.ptr(dbg), _lib.current_stream(self.device)))
        return (stats, dbg) if want_debug else stats

    def critic_step(self, obs, returns, grad_out: torch.Tensor | None = None, apply: bool = True) -> torch.Tensor:
        """One iteration of npg.py:180-183 -> vf_loss float32[1]."""
        obs = self._f32(obs).reshape(-1, self.obs_dim)
        b = obs.shape[0]
        returns = self._f32(returns, (b,))
        cfg = self.cfg
        if apply:
            self.adam_step += 1
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().ts_npg_critic_step(
            self._ws.handle, _lib.ptr(self.critic), _lib.ptr(self.critic_m), _lib.ptr(self.critic_v),
            _lib.i64(max(self.adam_step, 1)), _lib.i64(self.obs_dim), _lib.i64(self.hidden), _lib.ptr(obs), _lib.ptr(returns),
            _lib.i64(b), _lib.f64(cfg.lr if apply else -1.0), _lib.f64(cfg.betas[0]), _lib.f64(cfg.betas[1]),
            _lib.f64(cfg.adam_eps), _lib.f64(cfg.max_grad_norm or 0.0), _lib.ptr(loss), _lib.ptr(grad_out),
            _lib.current_stream(self.device)))
        return loss

    # -- _update_with_batch ------------------------------------------------------------------------------------------------
    def update(self, pre: dict, batch_size: int | None, repeat: int, perms=None):
        """npg.py:140-193 / trpo.py:123-214 -> (stats float32[steps, 4] = {actor_loss, vf_loss, kl, step_size}, steps)."""

        # The actor's step and the critic's iterations of a minibatch share nothing but their inputs (npg.py:149-183: two
        # networks, two optimisers), and half of the actor's launches are one-workgroup kernels (conjugate-gradient updates,
        # slab sums) that leave the chip idle: the critic iterations run on a second stream beside them, with that stream's
        # own workspace (TS_NPG_ONE_STREAM=1: one after the other on the caller's stream).  65,536-row minibatches: NPG 1,057 -> 1,107,
        # TRPO 887 -> 904 update-steps/s -- the chip-filling kernels of the two chains still take turns.
        # (one stream too while the workspace is timing its kernels: the event pairs of two streams would overlap)
        side = None if os.environ.get("TS_NPG_ONE_STREAM") or self._ws.profiling else self._critic_stream()

        def step_rows(rows):
            obs, ret, act, adv, lpo = gather_rows_multi([pre[k] for k in ("obs", "returns", "act", "adv", "logp_old")], rows)
            if side is None:
                st = self.actor_step(obs, act, adv, lpo)
                vf = self.critic_steps(obs, ret, self.cfg.optim_critic_iters)
            else:
                main = torch.cuda.current_stream(self.device)
                side.wait_stream(main)                      # the minibatch's rows are gathered
                with torch.cuda.stream(side):
                    vf = self.critic_steps(obs, ret, self.cfg.optim_critic_iters)
                vf.record_stream(main)                      # allocated on the side stream, read on the caller's
                st = self.actor_step(obs, act, adv, lpo)
                main.wait_stream(side)                      # (also keeps obs / ret alive until the side stream is done)
            return torch.stack([st[0], vf[0], st[1], st[2]])

        return run_minibatches(self.device, pre["obs"].shape[0], batch_size, repeat, perms, step_rows)

    def _critic_stream(self):
        # ONE side stream per device for every engine of the process: the library keeps a workspace per launch stream
        # (_lib.default_workspace), so a stream per engine would leave a workspace per engine behind
        idx = self.device.index or 0
        side = _CRITIC_STREAMS.get(idx)
        if side is None:
            side = _CRITIC_STREAMS[idx] = torch.cuda.Stream(device=self.device)
        return side


class NetNPGEngine(NPGEngine):
    """NPGEngine's interface for actor / critic trunks of any depth, widths and activation (round 6): `Net(hidden_sizes=[...],
    activation=nn.Tanh | nn.ReLU | None)` (utils/net/common.py:246-369), actor and critic trunks independent.  Runs layer by layer on
    the GEMM kernels (ts_npg_net_actor_step / ts_npg_net_critic_steps / ts_ppo_net_infer); the Fisher-vector product is one
    forward-mode and one reverse pass through the trunk.  Flat vectors: `ppo_wide.net_flat_from_tensors` (ts_net_layout)."""

    def __init__(self, obs_dim: int, act_dim: int, hidden_actor, hidden_critic, activation: str, actor: torch.Tensor,
                 critic: torch.Tensor, cfg: NPGConfig):
        if not actor.is_cuda:
            raise RuntimeError("NetNPGEngine needs parameters on an MI355X (no CPU fallback)")
        if cfg.algo not in ("npg", "trpo"):
            raise ValueError("algo must be 'npg' or 'trpo'")
        self.hidden_actor, self.hidden_critic, self.activation = [int(h) for h in hidden_actor], [int(h) for h in hidden_critic], activation
        self._na = _lib.NetDesc.make(obs_dim, self.hidden_actor, activation)
        self._nc = _lib.NetDesc.make(obs_dim, self.hidden_critic, activation)
        out = (C.c_int64 * 3)()
        _lib.check(_lib.load().ts_net_layout(C.byref(self._na), _lib.i64(act_dim), out))
        k0, n_actor = int(out[0]), int(out[1])
        _lib.check(_lib.load().ts_net_layout(C.byref(self._nc), _lib.i64(act_dim), out))
        n_critic = int(out[2])
        if actor.numel() != n_actor or critic.numel() != n_critic:
            raise ValueError("flat parameter vectors do not match ts_net_layout")
        self.obs_dim, self.act_dim, self.hidden, self.cfg = obs_dim, act_dim, None, cfg
        self.lay = {"k0": k0, "actor_count": n_actor, "critic_count": n_critic}
        self.device = actor.device
        cl = lambda t: t.detach().float().contiguous().clone()  # noqa: E731
        self.actor, self.critic = cl(actor), cl(critic)
        self.critic_m, self.critic_v = torch.zeros_like(self.critic), torch.zeros_like(self.critic)
        self.adam_step = 0
        self.ret_rms = [0.0, 1.0, 0.0]
        self._ws = _lib.default_workspace(self.device.index or 0)

    # -- converters (nn.Linear-layout tensor lists <-> flat vectors) ----------------------------------------------------------
    def actor_from_tensors(self, t, device=None) -> torch.Tensor:
        from .ppo_wide import net_flat_from_tensors

        return net_flat_from_tensors(t, self.obs_dim, self.hidden_actor, self.act_dim, device or self.device)

    def critic_from_tensors(self, t, device=None) -> torch.Tensor:
        from .ppo_wide import net_flat_from_tensors

        return net_flat_from_tensors(t, self.obs_dim, self.hidden_critic, None, device or self.device)

    def actor_to_tensors(self, flat):
        from .ppo_wide import net_flat_to_tensors

        return net_flat_to_tensors(flat, self.obs_dim, self.hidden_actor, self.act_dim, True)

    def critic_to_tensors(self, flat):
        from .ppo_wide import net_flat_to_tensors

        return net_flat_to_tensors(flat, self.obs_dim, self.hidden_critic, 1, False)

    def infer(self, obs, act=None, want_v: bool = True, want_mu: bool = False):
        obs = self._f32(obs).reshape(-1, self.obs_dim)
        b = obs.shape[0]
        act = None if act is None else self._f32(act, (b, self.act_dim))
        v = torch.empty(b, dtype=torch.float32, device=self.device) if want_v else None
        logp = torch.empty(b, dtype=torch.float32, device=self.device) if act is not None else None
        mu = torch.empty((b, self.act_dim), dtype=torch.float32, device=self.device) if want_mu else None
        _lib.check(_lib.load().ts_ppo_net_infer(
            self._ws.handle, _lib.ptr(self.actor), _lib.ptr(self.critic), C.byref(self._na), C.byref(self._nc), _lib.i64(self.act_dim),
            _lib.ptr(obs), _lib.ptr(act), _lib.i64(b), _lib.ptr(v), _lib.ptr(logp), _lib.ptr(mu), _lib.current_stream(self.device)))
        return (v, logp, mu) if want_mu else (v, logp)

    def _critic_call(self, ws, obs, returns, iters: int, first_step: int, apply: bool, grad_out=None) -> torch.Tensor:
        obs = self._f32(obs).reshape(-1, self.obs_dim)
        b = obs.shape[0]
        returns = self._f32(returns, (b,))
        cfg = self.cfg
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().ts_npg_net_critic_steps(
            ws.handle, _lib.ptr(self.critic), _lib.ptr(self.critic_m), _lib.ptr(self.critic_v), _lib.i64(first_step), C.byref(self._nc),
            _lib.ptr(obs), _lib.ptr(returns), _lib.i64(b), _lib.i64(iters), _lib.f64(cfg.lr if apply else -1.0), _lib.f64(cfg.betas[0]),
            _lib.f64(cfg.betas[1]), _lib.f64(cfg.adam_eps), _lib.f64(cfg.max_grad_norm or 0.0), _lib.ptr(loss), _lib.ptr(grad_out),
            _lib.current_stream(self.device)))
        return loss

    def critic_steps(self, obs, returns, iters: int) -> torch.Tensor:
        ws = _lib.default_workspace(self.device.index or 0)         # the CALLING stream's workspace (update() runs this on a side stream)
        loss = self._critic_call(ws, obs, returns, iters, self.adam_step + 1, True)
        self.adam_step += iters
        return loss

    def critic_step(self, obs, returns, grad_out: torch.Tensor | None = None, apply: bool = True) -> torch.Tensor:
        if apply:
            self.adam_step += 1
        return self._critic_call(self._ws, obs, returns, 1, max(self.adam_step, 1), apply, grad_out)

    def actor_step(self, obs, act, adv, logp_old=None, want_debug: bool = False):
        obs = self._f32(obs).reshape(-1, self.obs_dim)
        b = obs.shape[0]
        act, adv = self._f32(act, (b, self.act_dim)), self._f32(adv, (b,))
        logp_old = None if logp_old is None else self._f32(logp_old, (b,))
        if self.cfg.algo == "trpo" and logp_old is None:
            raise ValueError("TRPO needs logp_old")
        stats = torch.empty(3, dtype=torch.float32, device=self.device)
        dbg = torch.empty((3, self.lay["actor_count"]), dtype=torch.float32, device=self.device) if want_debug else None
        hp = self.cfg.to_c()
        _lib.check(_lib.load().ts_npg_net_actor_step(
            self._ws.handle, _lib.ptr(self.actor), C.byref(self._na), _lib.i64(self.act_dim), _lib.ptr(obs), _lib.ptr(act), _lib.ptr(adv),
            _lib.ptr(logp_old), _lib.i64(b), C.byref(hp), _lib.ptr(stats), _lib.ptr(dbg), _lib.current_stream(self.device)))
        return (stats, dbg) if want_debug else stats
