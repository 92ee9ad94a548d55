"""TEST INFRASTRUCTURE ONLY - CPU (torch fp32 autograd) restatement of the reference's REDQ learn() path.

Never imported by the product (`tianshou_amd/`).  Pinned against the UNMODIFIED reference through
tests/golden/redq_*.npz (oracle/gen_golden.py::gen_redq).

Follows:
  nets      test/continuous/test_redq.py:86-107: SAC's actor (Net + ContinuousActorProbabilistic, unbounded, conditioned
            sigma); one critic module whose Linear layers are EnsembleLinear(E, in, out) (utils/net/common.py:518-550):
            weight [E, in, out], bias [E, 1, out], output [E, B, 1]
  policy    REDQPolicy.forward modelfree/redq.py:103-131 (= SACPolicy.forward: tanh-squashed Gaussian, eps = float32 eps)
  target    _target_q ddpg.py:327-339 + _target_q_compute_value redq.py:248-261: a random subset of the lagged ensemble
            (np.random.choice), min or mean over it, minus alpha * log_prob
  update    _update_with_batch redq.py:263-304: one loss over the whole ensemble, one Adam step; every actor_delay-th
            update the actor step on (alpha log_prob - mean_e Q_e).mean() and AutoAlpha.update(-log_prob); Polyak
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle_sac as OS

CRITIC_ORDER = ["w1", "b1", "w2", "b2", "wq", "bq"]
TIANSHOU_CRITIC_KEYS = ["preprocess.model.model.0.weight", "preprocess.model.model.0.bias_weights",
                        "preprocess.model.model.2.weight", "preprocess.model.model.2.bias_weights",
                        "last.model.0.weight", "last.model.0.bias_weights"]


@dataclass
class REDQConfig(OS.SACConfig):
    ensemble_size: int = 10
    subset_size: int = 2
    actor_delay: int = 20
    target_mode: str = "min"


def _ensemble_linear(E: int, fin: int, fout: int):
    k = np.sqrt(1.0 / fin)                                     # common.py:536-544
    w = torch.rand((E, fin, fout)) * 2 * k - k
    b = torch.rand((E, 1, fout)) * 2 * k - k
    return w, b


def init_params(obs_dim: int, act_dim: int, E: int, seed: int, hidden=256):
    """Same RNG consumption as torch.manual_seed(seed) followed by test_redq.py:86-107.  `hidden`: see oracle_sac.layer_sizes
    (any depth since round 6)."""
    torch.manual_seed(seed)
    L = torch.nn.Linear
    sa, sc = OS.layer_sizes(hidden)
    mods = OS._linears(obs_dim, sa) + [L(sa[-1], act_dim), L(sa[-1], act_dim)]
    actor = dict(zip(OS.actor_order(len(sa)), [t.detach().clone() for m in mods for t in (m.weight, m.bias)]))
    dims = list(zip([obs_dim + act_dim] + list(sc[:-1]), sc)) + [(sc[-1], 1)]
    ts = [t for dm in dims for t in _ensemble_linear(E, *dm)]
    return actor, dict(zip(OS.critic_order(len(sc)), ts))


def critic_forward(p, obs, act) -> torch.Tensor:
    """-> [E, B, 1]"""
    h = torch.cat([obs.flatten(1), act.flatten(1)], dim=1)
    for i in range(1, OS.depth_of(p) + 1):
        h = OS._ACT["fn"](torch.matmul(h, p[f"w{i}"]) + p[f"b{i}"])
    return torch.matmul(h, p["wq"]) + p["bq"]


@dataclass
class REDQState:
    actor: dict
    critic: dict
    critic_old: dict
    log_alpha: torch.Tensor
    opt_actor: OS.Adam
    opt_critic: OS.Adam
    opt_alpha: OS.Adam
    critic_gradient_step: int = 0
    last_actor_loss: float = 0.0

    @classmethod
    def create(cls, actor, critic, cfg: REDQConfig):
        cp = lambda d: {k: v.clone() for k, v in d.items()}  # noqa: E731
        mk = lambda lr: OS.Adam(lr, cfg.betas, cfg.adam_eps)   # noqa: E731
        return cls(cp(actor), cp(critic), cp(critic),
                   torch.tensor(cfg.log_alpha0 if cfg.auto_alpha else float(np.log(cfg.alpha))),
                   mk(cfg.actor_lr), mk(cfg.critic_lr), mk(cfg.alpha_lr))


def target_q(st: REDQState, cfg: REDQConfig, obs_next, noise, subset) -> torch.Tensor:
    with torch.no_grad():
        obs_next = torch.as_tensor(obs_next, dtype=torch.float32)
        act, logp, _, _ = OS.policy_forward(st.actor, obs_next, torch.as_tensor(noise, dtype=torch.float32), cfg.max_action)
        qs = critic_forward(st.critic_old, obs_next, act)[np.asarray(subset), ...]
        tq = torch.min(qs, dim=0)[0] if cfg.target_mode == "min" else torch.mean(qs, dim=0)
        return tq - OS.alpha_value(st, cfg) * logp


def update_with_batch(st: REDQState, cfg: REDQConfig, obs, act, returns, noise=None, weight=None, collect=None):
    """redq.py:263-304 -> dict(actor_loss, critic_loss, alpha, alpha_loss, weight)."""
    obs, act = torch.as_tensor(obs, dtype=torch.float32), torch.as_tensor(act, dtype=torch.float32)
    ret = torch.as_tensor(returns, dtype=torch.float32).flatten()
    w = 1.0 if weight is None else torch.as_tensor(weight, dtype=torch.float32)
    p = {k: v.clone().requires_grad_(True) for k, v in st.critic.items()}
    td = critic_forward(p, obs, act).flatten(1) - ret
    critic_loss = (td.pow(2) * w).mean()
    g = OS._grads(critic_loss, p)
    if collect is not None:
        collect["critic_grads"] = g
    st.critic = st.opt_critic.apply(st.critic, g)
    out = {"critic_loss": float(critic_loss.item()), "weight": torch.mean(td.detach(), dim=0), "alpha_loss": None}
    st.critic_gradient_step += 1
    if st.critic_gradient_step % cfg.actor_delay == 0:
        alpha = OS.alpha_value(st, cfg)
        pa = {k: v.clone().requires_grad_(True) for k, v in st.actor.items()}
        a, logp, _, _ = OS.policy_forward(pa, obs, torch.as_tensor(noise, dtype=torch.float32), cfg.max_action)
        qa = critic_forward(st.critic, obs, a).mean(dim=0).flatten()
        actor_loss = (alpha * logp.flatten() - qa).mean()
        ga = OS._grads(actor_loss, pa)
        if collect is not None:
            collect["actor_grads"] = ga
        st.actor = st.opt_actor.apply(st.actor, ga)
        st.last_actor_loss = float(actor_loss.item())
        if cfg.auto_alpha:
            la = st.log_alpha.clone().requires_grad_(True)
            alpha_loss = -(la * (cfg.target_entropy + logp.detach())).mean()
            (gl,) = torch.autograd.grad(alpha_loss, [la])
            st.log_alpha = st.opt_alpha.apply({"a": st.log_alpha}, {"a": gl})["a"]
            out["alpha_loss"] = float(alpha_loss.item())
    for k in st.critic_old:
        st.critic_old[k] = cfg.tau * st.critic[k] + (1 - cfg.tau) * st.critic_old[k]
    out["actor_loss"] = st.last_actor_loss
    out["alpha"] = OS.alpha_value(st, cfg)
    return out
