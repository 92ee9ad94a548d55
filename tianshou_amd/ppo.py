"""PPO learn() path on the MI355X: host-side mirror of the reference's PPO/A2C hooks.

Mirrors (same names, argument meaning and control flow)
    ActorCriticOnPolicyAlgorithm._add_returns_and_advantages   tianshou/algorithm/modelfree/a2c.py:115-153
    PPO._preprocess_batch / PPO._update_with_batch             tianshou/algorithm/modelfree/ppo.py:146-224
    Algorithm.Optimizer.step                                   tianshou/algorithm/algorithm_base.py:484-500
for the MLP actor-critic of examples/mujoco/mujoco_ppo.py.  Every floating-point operation runs
in libtsengine's HIP kernels (tianshou_amd/csrc/ts_ppo.hip, ts_returns.hip); this module only
owns tensors, hyper-parameters, the host-supplied minibatch permutations and the RunningMeanStd
scalars.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from .buffer import _dev_index
from .returns import gae_scan

HIDDEN = 64
PARAM_ORDER = ("a_w1", "a_b1", "a_w2", "a_b2", "a_wmu", "a_bmu", "a_sigma",
               "c_w1", "c_b1", "c_w2", "c_b2", "c_wv", "c_bv")


def param_shapes(obs_dim: int, act_dim: int) -> dict[str, tuple[int, ...]]:
    """Flat fp32 parameter layout of include/tsengine.h."""
    return {
        "a_w1": (HIDDEN, obs_dim), "a_b1": (HIDDEN,), "a_w2": (HIDDEN, HIDDEN), "a_b2": (HIDDEN,),
        "a_wmu": (act_dim, HIDDEN), "a_bmu": (act_dim,), "a_sigma": (act_dim,),
        "c_w1": (HIDDEN, obs_dim), "c_b1": (HIDDEN,), "c_w2": (HIDDEN, HIDDEN), "c_b2": (HIDDEN,),
        "c_wv": (1, HIDDEN), "c_bv": (1,),
    }


def param_count(obs_dim: int, act_dim: int) -> int:
    return int(_lib.load().ts_ppo_param_count(_lib.i64(obs_dim), _lib.i64(act_dim)))


# names of the reference modules' state_dict entries, in flat-layout order
# (ContinuousActorProbabilistic / ContinuousCritic over Net, tianshou/utils/net/continuous.py)
TIANSHOU_ACTOR_KEYS = ("preprocess.model.model.0.weight", "preprocess.model.model.0.bias",
                       "preprocess.model.model.2.weight", "preprocess.model.model.2.bias",
                       "mu.model.0.weight", "mu.model.0.bias", "sigma_param")
TIANSHOU_CRITIC_KEYS = ("preprocess.model.model.0.weight", "preprocess.model.model.0.bias",
                        "preprocess.model.model.2.weight", "preprocess.model.model.2.bias",
                        "last.model.0.weight", "last.model.0.bias")


def flat_from_modules(actor, critic, device="cuda") -> torch.Tensor:
    sa, sc = actor.state_dict(), critic.state_dict()
    parts = [sa[k] for k in TIANSHOU_ACTOR_KEYS] + [sc[k] for k in TIANSHOU_CRITIC_KEYS]
    return torch.cat([p.detach().reshape(-1).to(torch.float32) for p in parts]).to(device).contiguous()


def flat_to_modules(flat: torch.Tensor, actor, critic) -> None:
    """Writes the engine's parameters back into the reference's nn.Parameters (state_dict keeps
    working, algorithm_base.py:523-543)."""
    off = 0
    with torch.no_grad():
        for mod, keys in ((actor, TIANSHOU_ACTOR_KEYS), (critic, TIANSHOU_CRITIC_KEYS)):
            sd = dict(mod.named_parameters())
            for k in keys:
                p = sd[k]
                n = p.numel()
                p.copy_(flat[off:off + n].reshape(p.shape).to(p.device))
                off += n
    assert off == flat.numel()


@dataclass
class PPOConfig:
    """Hyper-parameters of PPO.__init__ (ppo.py:24-36) / A2C.__init__ (a2c.py:163-237) +
    AdamOptimizerFactory (optim.py:89-110)."""
    algo: str = "ppo"            # "ppo" or "a2c"
    gamma: float = 0.99
    gae_lambda: float = 0.95
    eps_clip: float = 0.2
    dual_clip: float | None = None
    value_clip: bool = False
    advantage_normalization: bool = True
    recompute_advantage: bool = False
    vf_coef: float = 0.5
    ent_coef: float = 0.01
    max_grad_norm: float | None = None
    return_scaling: bool = False
    lr: float = 1e-3
    betas: tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8
    nets: int = 0                # 0: actor and critic; 1 / 2: only the actor's / the critic's half of every step (the other
                                 # network is a stand-in and takes a zero gradient: reinforce.py / npg.py; ts_ppo_hparams.nets)
    # optimizer factories of tianshou/algorithm/optim.py: "adam" (AdamOptimizerFactory, :89-110) or "rmsprop"
    # (RMSpropOptimizerFactory, :113-140 -- examples/mujoco/mujoco_a2c.py:117); lr / adam_eps (= eps) are shared.
    # State vectors: `adam_v` holds RMSprop's square_avg, `adam_m` its momentum buffer (or grad_avg when centered).
    optimizer: str = "adam"
    weight_decay: float = 0.0
    rms_alpha: float = 0.99
    rms_momentum: float = 0.0
    rms_centered: bool = False
    max_action: float | None = None   # ContinuousActorProbabilistic(unbounded=False): mu = max_action * tanh(.) (continuous.py:230-231)

    def to_c(self) -> _lib.PPOHParams:
        if self.optimizer not in ("adam", "rmsprop"):
            raise NotImplementedError(f"optimizer {self.optimizer!r}: the engines implement torch.optim.Adam and torch.optim.RMSprop")
        if self.optimizer == "rmsprop" and self.rms_centered and self.rms_momentum > 0:
            raise NotImplementedError("RMSprop(centered=True, momentum > 0) needs two auxiliary state vectors; one is provided")
        return _lib.PPOHParams(
            eps_clip=self.eps_clip, dual_clip=self.dual_clip or 0.0, vf_coef=self.vf_coef,
            ent_coef=self.ent_coef, max_grad_norm=self.max_grad_norm or 0.0, lr=self.lr,
            beta1=self.betas[0], beta2=self.betas[1], adam_eps=self.adam_eps,
            value_clip=int(self.value_clip), adv_norm=int(self.advantage_normalization),
            algo={"ppo": 0, "a2c": 1}[self.algo], nets=int(self.nets),
            optimizer={"adam": 0, "rmsprop": 1}[self.optimizer], rms_centered=int(self.rms_centered),
            weight_decay=float(self.weight_decay), rms_alpha=float(self.rms_alpha), rms_momentum=float(self.rms_momentum),
            max_action=float(self.max_action or 0.0))

    @property
    def plain_adam(self) -> bool:
        """torch.optim.Adam without weight decay: what the one-launch update kernels (ts_mlp_small.hip) have built in."""
        return self.optimizer == "adam" and not self.weight_decay


def rms_merge(rms, s1: float, s2: float, n: float) -> list[float]:
    """RunningMeanStd.update (utils/statistics.py:99-114) from the batch moments (sum, sum of squares, count):
    [mean, var, count] -> [mean', var', count']."""
    b_mean = s1 / n
    b_var = max(s2 / n - b_mean * b_mean, 0.0)
    mean, var, count = rms
    delta = b_mean - mean
    tot = count + n
    m2 = var * count + b_var * n + delta * delta * count * n / tot
    return [mean + delta * n / tot, m2 / tot, tot]


def split_offsets(n: int, size: int | None, merge_last: bool = True) -> list[int]:
    """Chunk boundaries of Batch.split(size, merge_last) (tianshou/data/batch.py:1205-1215)."""
    if not size or size == -1:
        size = n
    if size < 1:
        raise ValueError("batch size must be >= 1")
    merge = merge_last and n % size > 0
    offs = [0]
    for idx in range(0, n, size):
        if merge and idx + size + size >= n:
            offs.append(n)
            break
        offs.append(min(idx + size, n))
    return offs


def infer(params: torch.Tensor, obs_dim: int, act_dim: int, obs: torch.Tensor, act=None, *,
          want_v: bool = True, want_logp: bool = False, max_action: float | None = None):
    """V(obs) and/or log pi(act|obs) for the whole array in one launch (a2c.py:122-129,
    ppo.py:157-161 without the max_batchsize chunk loop).  max_action: the tanh bound of a bounded actor."""
    n = obs.shape[0]
    dev = obs.device
    v = torch.empty(n, dtype=torch.float32, device=dev) if want_v else None
    lp = torch.empty(n, dtype=torch.float32, device=dev) if want_logp else None
    ws = _lib.default_workspace(_dev_index(params))
    _lib.check(_lib.load().ts_ppo_infer_bounded(
        ws.handle, _lib.ptr(params), _lib.i64(obs_dim), _lib.i64(act_dim), _lib.f64(max_action or 0.0), _lib.ptr(obs),
        _lib.ptr(act), _lib.i64(n), _lib.ptr(v), _lib.ptr(lp), None, _lib.current_stream(dev)))
    return v, lp


def policy_forward(params: torch.Tensor, obs_dim: int, act_dim: int, obs: torch.Tensor, noise=None, *,
                   bound_method: str | None = "clip", low=None, high=None, max_action: float | None = None,
                   want_mu: bool = False):
    """The collector's inference step (collector.py:707-772): ProbabilisticActorPolicy.forward
    (reinforce.py:167-192) with dist.sample() = mu + sigma * noise (noise None = dist.mode) followed by
    Algorithm.map_action (algorithm_base.py:254-287).  -> (act for the buffer, mapped act for the env[, mu]).
    max_action: the tanh bound on mu of ContinuousActorProbabilistic(unbounded=False) (continuous.py:230-231)."""
    n, dev = obs.shape[0], obs.device
    obs = obs.to(torch.float32).contiguous()
    noise = None if noise is None else torch.as_tensor(noise, device=dev).to(torch.float32).reshape(n, act_dim).contiguous()
    f = lambda x: None if x is None else torch.as_tensor(x, dtype=torch.float32, device=dev).reshape(act_dim).contiguous()  # noqa: E731
    low, high = f(low), f(high)
    act = torch.empty((n, act_dim), dtype=torch.float32, device=dev)
    mapped = torch.empty_like(act)
    mu = torch.empty_like(act) if want_mu else None
    ws = _lib.default_workspace(_dev_index(params))
    _lib.check(_lib.load().ts_ppo_policy_forward_bounded(
        ws.handle, _lib.ptr(params), _lib.i64(obs_dim), _lib.i64(act_dim), _lib.f64(max_action or 0.0), _lib.ptr(obs),
        _lib.ptr(noise), _lib.i64(n), C.c_int({None: 0, "clip": 1, "tanh": 2}[bound_method]), _lib.ptr(low), _lib.ptr(high),
        _lib.ptr(act), _lib.ptr(mapped), _lib.ptr(mu), _lib.current_stream(dev)))
    return (act, mapped, mu) if want_mu else (act, mapped)


def pack_batch(b: dict, obs_dim: int, act_dim: int) -> torch.Tensor:
    """[n, W] packed per-sample records (obs | act | adv ret logp_old v_s | pad), see
    include/tsengine.h (ts_ppo_pack_batch); used by the data-parallel path."""
    n = b["obs"].shape[0]
    lib = _lib.load()
    lib.ts_ppo_record_width.restype = C.c_int64
    w = int(lib.ts_ppo_record_width(_lib.i64(obs_dim), _lib.i64(act_dim)))
    rec = torch.empty((n, w), dtype=torch.float32, device=b["obs"].device)
    _lib.check(lib.ts_ppo_pack_batch(
        _lib.ptr(b["obs"]), _lib.ptr(b["act"]), _lib.ptr(b["adv"]), _lib.ptr(b["returns"]),
        _lib.ptr(b["logp_old"]), _lib.ptr(b["v_s"]), _lib.i64(n), _lib.i64(obs_dim), _lib.i64(act_dim),
        _lib.ptr(rec), _lib.current_stream(rec.device)))
    return rec


class PPOEngine:
    """Device-resident state of one PPO learner: flat params, Adam moments, ret_rms scalars."""

    def __init__(self, obs_dim: int, act_dim: int, flat_params: torch.Tensor, cfg: PPOConfig):
        if not flat_params.is_cuda:
            raise RuntimeError("PPOEngine needs its parameters on an MI355X; there is no CPU fallback")
        self.obs_dim, self.act_dim, self.cfg = obs_dim, act_dim, cfg
        self.P = param_count(obs_dim, act_dim)
        if flat_params.numel() != self.P:
            raise ValueError(f"flat_params has {flat_params.numel()} entries, layout needs {self.P}")
        self.params = flat_params.detach().to(torch.float32).contiguous().clone()
        self.adam_m = torch.zeros_like(self.params)
        self.adam_v = torch.zeros_like(self.params)
        self.adam_step = 0
        self.device = self.params.device
        # RunningMeanStd (utils/statistics.py:81-91): mean 0, var 1, count 0
        self.ret_rms = [0.0, 1.0, 0.0]
        self._eps = 1e-8
        self._ws = _lib.default_workspace(_dev_index(self.params))

    def check(self) -> None:
        """Raises if a single-pass GAE scan on this engine's workspace gave up on a tile hand-off (its outputs would be
        garbage).  Synchronises the stream: call it where a D2H already happened (after reading the losses)."""
        if self._ws.gae_check():
            raise _lib.EngineError(-1, "gae_single_pass: a tile hand-off timed out; advantages / returns are invalid")

    # ------------------------------------------------------------------ preprocess
    def _f32(self, x) -> torch.Tensor:
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x), device=self.device)
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    def add_returns_and_advantages(self, obs, obs_next, rew, terminated, truncated, cut_pos,
                                   d_n_cut=None, reduce_stats=None, v_s=None):
        """a2c.py:115-153 -> (v_s, returns, adv) float32 device tensors.

        `reduce_stats(sum, sumsq, count) -> (sum, sumsq, count)`: data-parallel hook - the three float64 moments of the
        unnormalised returns summed over all ranks, so that every replica updates `ret_rms` with the statistics of the
        GLOBAL batch (what `ret_rms.update(unnormalized_returns)` sees in a single process, a2c.py:148)."""
        cfg = self.cfg
        if v_s is None:                    # preprocess() passes V(s) from the launch that also produced log pi_old
            v_s, _ = infer(self.params, self.obs_dim, self.act_dim, obs)
        v_next, _ = infer(self.params, self.obs_dim, self.act_dim, obs_next)     # (critic only: the actor's bound is not involved)
        scale = math.sqrt(self.ret_rms[1] + self._eps) if cfg.return_scaling else 1.0
        out = gae_scan(v_s, v_next, rew, terminated, truncated, cut_pos, gamma=cfg.gamma,
                       gae_lambda=cfg.gae_lambda, v_scale=scale, ret_div=scale,
                       want_ret_stats=cfg.return_scaling, d_n_cut=d_n_cut, ws=self._ws)
        if cfg.return_scaling:
            n = float(v_s.numel())
            s1, s2 = float(out["ret_sum"]), float(out["ret_sumsq"])       # one small D2H
            if reduce_stats is not None:
                s1, s2, n = reduce_stats(s1, s2, n)
            self.ret_rms = rms_merge(self.ret_rms, s1, s2, n)
        return v_s, out["returns"], out["adv"]

    def preprocess(self, obs, obs_next, act, rew, terminated, truncated, cut_pos, d_n_cut=None, reduce_stats=None):
        """PPO._preprocess_batch (ppo.py:146-162) on batch-order device arrays."""
        obs, obs_next, act = self._f32(obs), self._f32(obs_next), self._f32(act)
        if self.cfg.algo == "a2c":      # A2C._preprocess_batch (a2c.py:239-247): no logp_old
            v_s, returns, adv = self.add_returns_and_advantages(obs, obs_next, rew, terminated,
                                                                truncated, cut_pos, d_n_cut, reduce_stats)
            logp_old = torch.zeros_like(adv)
        else:
            # V(s) and log pi_old(a | s) read the same observations: one launch with both networks resident (the
            # parameters do not change between ppo.py:157 and :160, so the order of the two passes is immaterial)
            v_s0, logp_old = infer(self.params, self.obs_dim, self.act_dim, obs, act, want_v=True, want_logp=True,
                                   max_action=self.cfg.max_action)
            v_s, returns, adv = self.add_returns_and_advantages(obs, obs_next, rew, terminated, truncated, cut_pos,
                                                                d_n_cut, reduce_stats, v_s=v_s0)
        return {"obs": obs, "obs_next": obs_next, "act": act, "rew": rew, "terminated": terminated,
                "truncated": truncated, "cut_pos": cut_pos, "d_n_cut": d_n_cut,
                "v_s": v_s, "returns": returns, "adv": adv, "logp_old": logp_old}

    # ------------------------------------------------------------------ update
    def _run_steps(self, b: dict, perm: torch.Tensor | None, offsets: list[int], want_grad=False):
        n_steps = len(offsets) - 1
        losses = torch.empty((n_steps, 4), dtype=torch.float32, device=self.device)
        grads = torch.empty(self.P, dtype=torch.float32, device=self.device) if want_grad else None
        h_off = (C.c_int64 * len(offsets))(*offsets)
        hp = self.cfg.to_c()
        _lib.check(_lib.load().ts_ppo_update(
            self._ws.handle, _lib.ptr(self.params), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v),
            _lib.i64(self.adam_step), _lib.i64(self.obs_dim), _lib.i64(self.act_dim),
            _lib.ptr(b["obs"]), _lib.ptr(b["act"]), _lib.ptr(b["adv"]), _lib.ptr(b["returns"]),
            _lib.ptr(b["logp_old"]), _lib.ptr(b["v_s"]), _lib.i64(b["obs"].shape[0]), _lib.ptr(perm),
            h_off, _lib.i64(n_steps), C.byref(hp), _lib.ptr(losses), _lib.ptr(grads),
            _lib.current_stream(self.device)))
        self.adam_step += n_steps
        return losses, grads

    def update(self, b: dict, batch_size: int | None, repeat: int, perms=None, want_grad=False):
        """PPO._update_with_batch (ppo.py:164-224).

        `perms`: sequence of `repeat` permutations of range(N) - the np.random.permutation draws
        of Batch.split (batch.py:1209); entries may be NumPy arrays (parity with a seeded
        reference run) or int64 device tensors (e.g. torch.randperm on the GPU, which keeps the
        host off the critical path).  None draws them from the global NumPy RNG exactly like the
        reference.  Returns (losses float32[steps, 4] device tensor with columns
        (loss, clip_loss, vf_loss, ent_loss), gradient_steps[, last unclipped gradient])."""
        n = b["obs"].shape[0]
        cfg = self.cfg
        if perms is None:
            perms = [np.random.permutation(n) for _ in range(repeat)]
        offs1 = split_offsets(n, batch_size, merge_last=True)
        out, grads = [], None
        if not cfg.recompute_advantage:
            if all(isinstance(p, torch.Tensor) for p in perms):
                perm = torch.cat([p.to(device=self.device, dtype=torch.int64) for p in perms])
            else:
                perm = torch.as_tensor(np.concatenate([np.asarray(p, dtype=np.int64) for p in perms]),
                                       device=self.device)
            offsets = [r * n + o for r in range(repeat) for o in offs1[:-1]] + [repeat * n]
            losses, grads = self._run_steps(b, perm, offsets, want_grad)
            out.append(losses)
        else:
            for r in range(repeat):
                if r > 0:                                                      # ppo.py:174-178
                    v_s, returns, adv = self.add_returns_and_advantages(
                        b["obs"], b["obs_next"], b["rew"], b["terminated"], b["truncated"],
                        b["cut_pos"], b.get("d_n_cut"))
                    b = dict(b, v_s=v_s, returns=returns, adv=adv)
                perm = (perms[r].to(device=self.device, dtype=torch.int64) if isinstance(perms[r], torch.Tensor)
                        else torch.as_tensor(np.asarray(perms[r], dtype=np.int64), device=self.device))
                losses, grads = self._run_steps(b, perm, offs1, want_grad)
                out.append(losses)
        losses = torch.cat(out, dim=0)
        steps = losses.shape[0]
        return (losses, steps, grads) if want_grad else (losses, steps)


def summary_stats(seq: np.ndarray) -> dict[str, float]:
    """SequenceSummaryStats.from_sequence (tianshou/data/stats.py:26-43): population std."""
    seq = np.asarray(seq, dtype=np.float64)
    return {"mean": float(seq.mean()), "std": float(seq.std()), "max": float(seq.max()),
            "min": float(seq.min())}
