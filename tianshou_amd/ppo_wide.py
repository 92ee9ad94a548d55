"""PPO / A2C learn() path for actor-critic MLPs outside the fused kernels' envelope.

`tianshou_amd.ppo.PPOEngine` drives kernels specialised to the MuJoCo example nets (hidden 64 x 64, obs <= 31,
act <= 8).  The reference's `Net` / `MLP` accept any `hidden_sizes` (utils/net/common.py:90-178, 246-369) and e.g.
Humanoid is obs 376 / act 17; this engine runs the same hooks
    ActorCriticOnPolicyAlgorithm._add_returns_and_advantages   modelfree/a2c.py:115-153
    PPO._preprocess_batch / PPO._update_with_batch             modelfree/ppo.py:146-224   (A2C: a2c.py:239-290)
    Algorithm.Optimizer.step                                   algorithm_base.py:484-500
for Net[h, h] (tanh, h a multiple of 32 up to 1024), act_dim <= 32, on the implicit-GEMM layer kernels
(ts_ppo_wide_step / ts_npg_infer in csrc/ts_npg.hip over csrc/ts_conv.hip).  Same interface as PPOEngine, so the
`HipPPO` hooks and `DataParallelPPO`'s preprocessing work with either.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from . import npg as NG
from .buffer import gather_rows
from .ppo import PPOConfig, rms_merge, split_offsets
from .returns import gae_scan


def flat_from_tensors(actor_t: list[torch.Tensor], critic_t: list[torch.Tensor], obs_dim: int, hidden: int, act_dim: int,
                      device="cuda") -> torch.Tensor:
    """[actor | critic] in the ts_npg_layout order from nn.Linear-layout tensors
    ([w1, b1, w2, b2, w_mu, b_mu, sigma_param], [w1, b1, w2, b2, w_v, b_v])."""
    return torch.cat([NG.actor_flat_from_torch(actor_t, obs_dim, hidden, act_dim, device),
                      NG.critic_flat_from_torch(critic_t, obs_dim, hidden, device)]).contiguous()


def flat_to_tensors(flat: torch.Tensor, obs_dim: int, hidden: int, act_dim: int):
    """-> (actor tensors [7], critic tensors [6]) in nn.Linear layout (sigma_param as a flat [act_dim] vector)."""
    na = NG.layout(obs_dim, hidden, act_dim)["actor_count"]
    return (NG.actor_flat_to_torch(flat[:na], obs_dim, hidden, act_dim), NG.critic_flat_to_torch(flat[na:], obs_dim, hidden))


class WidePPOEngine:
    """Device-resident state of one PPO / A2C learner on the GEMM path (interface of ppo.PPOEngine)."""

    def __init__(self, obs_dim: int, act_dim: int, hidden: int, flat_params: torch.Tensor, cfg: PPOConfig):
        if not flat_params.is_cuda:
            raise RuntimeError("WidePPOEngine needs its parameters on an MI355X; there is no CPU fallback")
        if hidden % 32 or not 32 <= hidden <= 1024 or not 1 <= act_dim <= 32:
            raise NotImplementedError("WidePPOEngine: hidden a multiple of 32 in [32, 1024], act_dim <= 32")
        if cfg.max_action:
            raise NotImplementedError("WidePPOEngine: unbounded actors only (a bounded actor runs on NetPPOEngine)")
        self.obs_dim, self.act_dim, self.hidden, self.cfg = obs_dim, act_dim, hidden, cfg
        lay = NG.layout(obs_dim, hidden, act_dim)
        self.n_actor, self.n_critic = lay["actor_count"], lay["critic_count"]
        self.P = self.n_actor + self.n_critic
        if flat_params.numel() != self.P:
            raise ValueError(f"flat_params has {flat_params.numel()} entries, layout needs {self.P}")
        self.params = flat_params.detach().to(torch.float32).contiguous().clone()
        self.adam_m, self.adam_v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.adam_step = 0
        self.device = self.params.device
        self.ret_rms = [0.0, 1.0, 0.0]
        self._eps = 1e-8
        self._ws = _lib.default_workspace(self.device.index or 0)

    @property
    def actor(self) -> torch.Tensor:
        return self.params[: self.n_actor]

    @property
    def critic(self) -> torch.Tensor:
        return self.params[self.n_actor:]

    def check(self) -> None:
        if self._ws.gae_check():
            raise _lib.EngineError(-1, "gae_single_pass: a tile hand-off timed out; advantages / returns are invalid")

    def _f32(self, x) -> torch.Tensor:
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x), device=self.device)
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    def _dims(self):
        return _lib.i64(self.obs_dim), _lib.i64(self.hidden), _lib.i64(self.act_dim)

    def infer(self, obs, act=None, want_v=True):
        """-> (V float32[B] or None, log pi(act | obs) float32[B] or None) for the whole array."""
        b = obs.shape[0]
        v = torch.empty(b, dtype=torch.float32, device=self.device) if want_v else None
        logp = torch.empty(b, dtype=torch.float32, device=self.device) if act is not None else None
        _lib.check(_lib.load().ts_npg_infer(
            self._ws.handle, _lib.ptr(self.actor), _lib.ptr(self.critic), *self._dims(), _lib.ptr(obs), _lib.ptr(act),
            _lib.i64(b), _lib.ptr(v), _lib.ptr(logp), None, _lib.current_stream(self.device)))
        return v, logp

    # ------------------------------------------------------------------ preprocess
    def add_returns_and_advantages(self, obs, obs_next, rew, terminated, truncated, cut_pos, d_n_cut=None,
                                   reduce_stats=None):
        """a2c.py:115-153 -> (v_s, returns, adv) float32 device tensors (see PPOEngine.add_returns_and_advantages)."""
        cfg = self.cfg
        v_s, _ = self.infer(obs)
        v_next, _ = self.infer(obs_next)
        scale = math.sqrt(self.ret_rms[1] + self._eps) if cfg.return_scaling else 1.0
        out = gae_scan(v_s, v_next, rew, terminated, truncated, cut_pos, gamma=cfg.gamma, gae_lambda=cfg.gae_lambda,
                       v_scale=scale, ret_div=scale, want_ret_stats=cfg.return_scaling, d_n_cut=d_n_cut, ws=self._ws)
        if cfg.return_scaling:
            n = float(v_s.numel())
            s1, s2 = float(out["ret_sum"]), float(out["ret_sumsq"])
            if reduce_stats is not None:
                s1, s2, n = reduce_stats(s1, s2, n)
            self.ret_rms = rms_merge(self.ret_rms, s1, s2, n)
        return v_s, out["returns"], out["adv"]

    def preprocess(self, obs, obs_next, act, rew, terminated, truncated, cut_pos, d_n_cut=None, reduce_stats=None):
        """PPO._preprocess_batch (ppo.py:146-162) on batch-order device arrays."""
        obs, obs_next = self._f32(obs).reshape(-1, self.obs_dim), self._f32(obs_next).reshape(-1, self.obs_dim)
        act = self._f32(act).reshape(obs.shape[0], self.act_dim)
        v_s, returns, adv = self.add_returns_and_advantages(obs, obs_next, rew, terminated, truncated, cut_pos, d_n_cut,
                                                            reduce_stats)
        if self.cfg.algo == "a2c":          # A2C._preprocess_batch (a2c.py:239-247): no logp_old
            logp_old = torch.zeros_like(adv)
        else:
            _, logp_old = self.infer(obs, act, want_v=False)
        return {"obs": obs, "obs_next": obs_next, "act": act, "rew": rew, "terminated": terminated, "truncated": truncated,
                "cut_pos": cut_pos, "d_n_cut": d_n_cut, "v_s": v_s, "returns": returns, "adv": adv, "logp_old": logp_old}

    # ------------------------------------------------------------------ update
    def step(self, b: dict, rows: torch.Tensor | None, losses_out: torch.Tensor, grad_out: torch.Tensor | None = None,
             apply: bool = True, global_batch: int | None = None, adv_stats: torch.Tensor | None = None):
        """One minibatch (rows of `b`; None = all): forward, loss, backward, joint clip + Adam.  `global_batch` /
        `adv_stats`: the data-parallel path (rows = this rank's share of a global minibatch, {mean, std} of the GLOBAL
        minibatch's advantages); apply=False leaves the gradient of the global-mean loss in grad_out."""
        cfg = self.cfg
        take = (lambda t: t) if rows is None else (lambda t: gather_rows(t, rows))       # noqa: E731
        obs, act = take(b["obs"]), take(b["act"])
        adv, ret, lp_old, v_old = take(b["adv"]), take(b["returns"]), take(b["logp_old"]), take(b["v_s"])
        n = obs.shape[0]
        stats = adv_stats
        if stats is None and cfg.advantage_normalization and cfg.algo != "a2c":     # ppo.py:184-186 (unbiased std)
            a64 = adv.double()
            stats = torch.stack([a64.mean(), a64.std()]).float().contiguous()
        hp = cfg.to_c()
        if not apply:
            hp.lr = -1.0
        else:
            self.adam_step += 1
        _lib.check(_lib.load().ts_ppo_wide_step(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v), _lib.i64(max(self.adam_step, 1)),
            *self._dims(), _lib.ptr(obs), _lib.ptr(act), _lib.ptr(adv), _lib.ptr(ret), _lib.ptr(lp_old), _lib.ptr(v_old),
            _lib.i64(n), _lib.i64(global_batch or n), _lib.ptr(stats), C.byref(hp), _lib.ptr(losses_out), _lib.ptr(grad_out),
            _lib.current_stream(self.device)))

    def update(self, b: dict, batch_size: int | None, repeat: int, perms=None, want_grad=False):
        """PPO._update_with_batch (ppo.py:164-224); same contract as PPOEngine.update."""
        n = b["obs"].shape[0]
        cfg = self.cfg
        if perms is None:
            perms = [np.random.permutation(n) for _ in range(repeat)]
        offs = split_offsets(n, batch_size, merge_last=True)
        n_steps = repeat * (len(offs) - 1)
        losses = torch.empty((n_steps, 4), dtype=torch.float32, device=self.device)
        grads = torch.empty(self.P, dtype=torch.float32, device=self.device) if want_grad else None
        k = 0
        for r in range(repeat):
            if cfg.recompute_advantage and r > 0:                                   # ppo.py:174-178
                v_s, returns, adv = self.add_returns_and_advantages(b["obs"], b["obs_next"], b["rew"], b["terminated"],
                                                                    b["truncated"], b["cut_pos"], b.get("d_n_cut"))
                b = dict(b, v_s=v_s, returns=returns, adv=adv)
            perm = (perms[r].to(device=self.device, dtype=torch.int64) if isinstance(perms[r], torch.Tensor)
                    else torch.as_tensor(np.asarray(perms[r], dtype=np.int64), device=self.device))
            for lo, hi in zip(offs[:-1], offs[1:]):
                self.step(b, perm[lo:hi], losses[k], grads if k == n_steps - 1 else None)
                k += 1
        return (losses, n_steps, grads) if want_grad else (losses, n_steps)


# ---------------------------------------------------------------------------------------------------------------------
# Trunks of any depth / width / activation (ts_net_desc): Net(hidden_sizes=[...], activation=Tanh | ReLU | None),
# utils/net/common.py:90-178, 246-369 -- the same hooks on ts_ppo_net_step / ts_ppo_net_infer.

def _pad32(n: int) -> int:
    return (int(n) + 31) // 32 * 32


CS_COL = 16          # conditioned sigma: the head block's columns [16, 16 + act) are the sigma head (csrc/ts_npg.hip)


def net_flat_from_tensors(t: list[torch.Tensor], obs_dim: int, hidden: list[int], n_out: int | None, device="cuda",
                          conditioned_sigma: bool = False, layer_norm: bool = False) -> torch.Tensor:
    """nn.Linear-layout tensors [w1, b1, ..., w_L, b_L, w_head, b_head(, sigma_param)] -> the ts_net_layout vector: per layer
    one block [K_pad + 1, N_pad] (last row = bias; widths padded to 32 with zeros), the head padded to 32 columns, and for
    an actor (n_out = act_dim) log_sigma padded to 32.  conditioned_sigma: the tensors end with [..., w_mu, b_mu, w_sigma,
    b_sigma] instead (ContinuousActorProbabilistic(conditioned_sigma=True), continuous.py:212-218): the sigma head goes into
    the head block's columns [16, 16 + act), the log_sigma block stays zero.  layer_norm (MLP(norm_layer=nn.LayerNorm),
    utils/net/common.py:25-39): every hidden layer brings four tensors [w, b, gamma, beta] (module order) and its block is
    followed by gamma[N_pad] | beta[N_pad] (padding entries zero)."""
    nl = len(hidden)
    per = 4 if layer_norm else 2
    dims = [_pad32(obs_dim)] + [_pad32(h) for h in hidden] + [NG.HEAD]
    parts = []
    for i in range(nl):
        parts.append(NG._block(t[per * i], t[per * i + 1], dims[i], dims[i + 1]))
        if layer_norm:
            gb = torch.zeros(2, dims[i + 1], dtype=torch.float32)
            gb[0, : hidden[i]] = t[per * i + 2].detach().float().cpu().reshape(-1)
            gb[1, : hidden[i]] = t[per * i + 3].detach().float().cpu().reshape(-1)
            parts.append(gb.reshape(-1))
    o = per * nl
    head = NG._block(t[o], t[o + 1], dims[nl], NG.HEAD).reshape(dims[nl] + 1, NG.HEAD)
    ls = torch.zeros(NG.HEAD, dtype=torch.float32)
    if n_out is not None and conditioned_sigma:
        if n_out > CS_COL:
            raise NotImplementedError("conditioned_sigma: at most 16 actions")
        w_s, b_s = t[o + 2].detach().float().cpu(), t[o + 3].detach().float().cpu()
        head[: w_s.shape[1], CS_COL:CS_COL + n_out] = w_s.t()
        head[dims[nl], CS_COL:CS_COL + n_out] = b_s
    elif n_out is not None:
        ls[:n_out] = t[o + 2].detach().float().cpu().reshape(-1)
    parts.append(head.reshape(-1))
    if n_out is not None:
        parts.append(ls)
    return torch.cat(parts).to(device).contiguous()


def net_flat_to_tensors(flat: torch.Tensor, obs_dim: int, hidden: list[int], n_head: int, actor: bool,
                        conditioned_sigma: bool = False, layer_norm: bool = False) -> list[torch.Tensor]:
    """Inverse of net_flat_from_tensors (nn.Linear layout; n_head = act_dim for an actor, 1 for a critic)."""
    true = [obs_dim] + list(hidden) + [n_head]
    dims = [_pad32(obs_dim)] + [_pad32(h) for h in hidden] + [NG.HEAD]
    f, off, out = flat.detach(), 0, []
    for i in range(len(hidden) + 1):
        n = (dims[i] + 1) * dims[i + 1]
        blk = f[off:off + n].reshape(dims[i] + 1, dims[i + 1])
        out += [blk[: true[i], : true[i + 1]].t().contiguous(), blk[dims[i], : true[i + 1]].clone()]
        if actor and conditioned_sigma and i == len(hidden):
            out += [blk[: true[i], CS_COL:CS_COL + n_head].t().contiguous(), blk[dims[i], CS_COL:CS_COL + n_head].clone()]
        off += n
        if layer_norm and i < len(hidden):
            out += [f[off: off + true[i + 1]].clone(), f[off + dims[i + 1]: off + dims[i + 1] + true[i + 1]].clone()]
            off += 2 * dims[i + 1]
    if actor and not conditioned_sigma:
        out.append(f[off:off + NG.HEAD][:n_head].clone())
    return out


class NetPPOEngine(WidePPOEngine):
    """WidePPOEngine's interface (preprocess / step / update / infer) for actor-critics whose trunks have any number of
    hidden layers of any widths and a tanh / ReLU / no activation; actor and critic trunks may differ."""

    def __init__(self, obs_dim: int, act_dim: int, hidden_actor, hidden_critic, activation: str, flat_params: torch.Tensor,
                 cfg: PPOConfig, conditioned_sigma: bool = False, layer_norm: bool = False, ln_eps: float = 1e-5):
        if not flat_params.is_cuda:
            raise RuntimeError("NetPPOEngine needs its parameters on an MI355X; there is no CPU fallback")
        if not 1 <= act_dim <= (CS_COL if conditioned_sigma else 32):
            raise NotImplementedError("NetPPOEngine: act_dim <= 32 (<= 16 with conditioned_sigma)")
        self.obs_dim, self.act_dim, self.cfg = obs_dim, act_dim, cfg
        self.hidden_actor, self.hidden_critic, self.activation = [int(h) for h in hidden_actor], [int(h) for h in hidden_critic], activation
        self.hidden = None
        self.conditioned_sigma = bool(conditioned_sigma)
        self.entropy_is_batch_sum = self.conditioned_sigma           # (DataParallelPPO: the entropy part is summed, not repeated)
        self.layer_norm, self.ln_eps = bool(layer_norm), float(ln_eps)
        ln = _lib.NetDesc.LAYERNORM if layer_norm else 0
        self._na = _lib.NetDesc.make(obs_dim, self.hidden_actor, activation,
                                     (_lib.NetDesc.CONDITIONED_SIGMA if conditioned_sigma else 0) | ln,
                                     max_action=cfg.max_action or 0.0, ln_eps=ln_eps)
        self._nc = _lib.NetDesc.make(obs_dim, self.hidden_critic, activation, ln, ln_eps=ln_eps)
        out = (C.c_int64 * 3)()
        _lib.check(_lib.load().ts_net_layout(C.byref(self._na), _lib.i64(act_dim), out))
        self.n_actor = int(out[1])
        _lib.check(_lib.load().ts_net_layout(C.byref(self._nc), _lib.i64(act_dim), out))
        self.n_critic = int(out[2])
        self.P = self.n_actor + self.n_critic
        if flat_params.numel() != self.P:
            raise ValueError(f"flat_params has {flat_params.numel()} entries, layout needs {self.P}")
        self.params = flat_params.detach().to(torch.float32).contiguous().clone()
        self.adam_m, self.adam_v = torch.zeros_like(self.params), torch.zeros_like(self.params)
        self.adam_step = 0
        self.device = self.params.device
        self.ret_rms = [0.0, 1.0, 0.0]
        self._eps = 1e-8
        self._ws = _lib.default_workspace(self.device.index or 0)

    def flat_from_tensors(self, actor_t, critic_t) -> torch.Tensor:
        return torch.cat([net_flat_from_tensors(actor_t, self.obs_dim, self.hidden_actor, self.act_dim, self.device,
                                                conditioned_sigma=self.conditioned_sigma, layer_norm=self.layer_norm),
                          net_flat_from_tensors(critic_t, self.obs_dim, self.hidden_critic, None, self.device,
                                                layer_norm=self.layer_norm)]).contiguous()

    def flat_to_tensors(self, flat: torch.Tensor):
        return (net_flat_to_tensors(flat[: self.n_actor], self.obs_dim, self.hidden_actor, self.act_dim, True, self.conditioned_sigma,
                                    self.layer_norm),
                net_flat_to_tensors(flat[self.n_actor:], self.obs_dim, self.hidden_critic, 1, False, layer_norm=self.layer_norm))

    def infer(self, obs, act=None, want_v=True):
        b = obs.shape[0]
        v = torch.empty(b, dtype=torch.float32, device=self.device) if want_v else None
        logp = torch.empty(b, dtype=torch.float32, device=self.device) if act is not None else None
        _lib.check(_lib.load().ts_ppo_net_infer(
            self._ws.handle, _lib.ptr(self.actor), _lib.ptr(self.critic), C.byref(self._na), C.byref(self._nc),
            _lib.i64(self.act_dim), _lib.ptr(obs), _lib.ptr(act), _lib.i64(b), _lib.ptr(v), _lib.ptr(logp), None,
            _lib.current_stream(self.device)))
        return v, logp

    def step(self, b: dict, rows, losses_out: torch.Tensor, grad_out=None, apply: bool = True, global_batch=None, adv_stats=None):
        """One minibatch (see WidePPOEngine.step) on ts_ppo_net_step."""
        cfg = self.cfg
        take = (lambda t: t) if rows is None else (lambda t: gather_rows(t, rows))       # noqa: E731
        obs, act = take(b["obs"]), take(b["act"])
        adv, ret, lp_old, v_old = take(b["adv"]), take(b["returns"]), take(b["logp_old"]), take(b["v_s"])
        n = obs.shape[0]
        stats = adv_stats
        if stats is None and cfg.advantage_normalization and cfg.algo != "a2c":     # ppo.py:184-186 (unbiased std)
            a64 = adv.double()
            stats = torch.stack([a64.mean(), a64.std()]).float().contiguous()
        hp = cfg.to_c()
        if not apply:
            hp.lr = -1.0
        else:
            self.adam_step += 1
        _lib.check(_lib.load().ts_ppo_net_step(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v), _lib.i64(max(self.adam_step, 1)),
            C.byref(self._na), C.byref(self._nc), _lib.i64(self.act_dim), _lib.ptr(obs), _lib.ptr(act), _lib.ptr(adv), _lib.ptr(ret),
            _lib.ptr(lp_old), _lib.ptr(v_old), _lib.i64(n), _lib.i64(global_batch or n), _lib.ptr(stats), C.byref(hp),
            _lib.ptr(losses_out), _lib.ptr(grad_out), _lib.current_stream(self.device)))
