"""TEST INFRASTRUCTURE ONLY - CPU (torch fp32 autograd) restatement of the reference's SAC learn() path.

Never imported by the product (`tianshou_amd/`).  Pinned against the UNMODIFIED reference through
tests/golden/sac_*.npz (oracle/gen_golden.py::gen_sac), with the reference's rsample() noise recorded
and replayed ("noise tensors host-supplied").

Follows:
  ContinuousActorProbabilistic.forward  tianshou/utils/net/continuous.py:220-238 (conditioned sigma,
                                        unbounded; clamp(SIGMA_MIN=-20, SIGMA_MAX=2).exp(), :22-23)
  ContinuousCritic.forward              continuous.py:144-169 (concat(obs, act) -> Net -> Linear)
  Net / MLP (ReLU)                      utils/net/common.py:90-178, 246-369
  SACPolicy.forward                     algorithm/modelfree/sac.py:108-131
  correct_log_prob_gaussian_tanh        sac.py:25-39
  SAC._target_q_compute_value           sac.py:290-296 ; td3.py:94-102 (min of the lagged critics)
  _minimize_critic_squared_loss         ddpg.py:267-285
  SAC._update_with_batch                sac.py:298-336
  AutoAlpha.update                      sac.py:203-209
  polyak_parameter_update               utils/lagged_network.py:8-18
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F
from torch.distributions import Independent, Normal

ACTOR_ORDER = ["w1", "b1", "w2", "b2", "wmu", "bmu", "wsig", "bsig"]
CRITIC_ORDER = ["w1", "b1", "w2", "b2", "wq", "bq"]
TIANSHOU_ACTOR_KEYS = ["preprocess.model.model.0.weight", "preprocess.model.model.0.bias",
                       "preprocess.model.model.2.weight", "preprocess.model.model.2.bias",
                       "mu.model.0.weight", "mu.model.0.bias", "sigma.model.0.weight", "sigma.model.0.bias"]
TIANSHOU_CRITIC_KEYS = ["preprocess.model.model.0.weight", "preprocess.model.model.0.bias",
                        "preprocess.model.model.2.weight", "preprocess.model.model.2.bias",
                        "last.model.0.weight", "last.model.0.bias"]
SIGMA_MIN, SIGMA_MAX = -20.0, 2.0
TANH_EPS = float(np.finfo(np.float32).eps)


def actor_shapes(obs_dim: int, act_dim: int, hidden: int = 256):
    return {"w1": (hidden, obs_dim), "b1": (hidden,), "w2": (hidden, hidden), "b2": (hidden,),
            "wmu": (act_dim, hidden), "bmu": (act_dim,), "wsig": (act_dim, hidden), "bsig": (act_dim,)}


def critic_shapes(obs_dim: int, act_dim: int, hidden: int = 256):
    return {"w1": (hidden, obs_dim + act_dim), "b1": (hidden,), "w2": (hidden, hidden), "b2": (hidden,),
            "wq": (1, hidden), "bq": (1,)}


def init_params(shapes: dict, seed: int) -> dict[str, torch.Tensor]:
    """torch.nn.Linear default init for every (w, b) pair, in order."""
    torch.manual_seed(seed)
    p = {}
    names = list(shapes)
    for wn, bn in zip(names[0::2], names[1::2]):
        lin = torch.nn.Linear(shapes[wn][1], shapes[wn][0])
        p[wn], p[bn] = lin.weight.detach().clone(), lin.bias.detach().clone()
    return p


def trunk_order(depth: int, heads: tuple[str, ...]) -> list[str]:
    """Parameter names of an MLP trunk of `depth` hidden layers + single-Linear heads: w1, b1, ..., wd, bd, then w<head>, b<head>."""
    ks = []
    for i in range(1, depth + 1):
        ks += [f"w{i}", f"b{i}"]
    for h in heads:
        ks += [f"w{h}", f"b{h}"]
    return ks


def trunk_keys(depth: int, heads: tuple[str, ...]) -> list[str]:
    """The reference's state_dict keys of the same network: Net's Sequential holds its Linears at even positions
    (utils/net/common.py:90-178), heads are one-Linear MLPs."""
    ks = []
    for i in range(depth):
        ks += [f"preprocess.model.model.{2 * i}.weight", f"preprocess.model.model.{2 * i}.bias"]
    for h in heads:
        ks += [f"{h}.model.0.weight", f"{h}.model.0.bias"]
    return ks


def actor_order(depth: int = 2) -> list[str]:
    return trunk_order(depth, ("mu", "sig"))


def critic_order(depth: int = 2) -> list[str]:
    return trunk_order(depth, ("q",))


def det_actor_order(depth: int = 2) -> list[str]:
    return trunk_order(depth, ("a",))


def depth_of(p: dict) -> int:
    """Number of hidden layers of a parameter dict (w1 .. wd)."""
    d = 0
    while f"w{d + 1}" in p:
        d += 1
    return d


def order_of(p: dict) -> list[str]:
    """The dict's own parameter order (dicts are built in order: trunk, then heads)."""
    return list(p.keys())


_ACT = {"fn": F.relu}


class activation:
    """`with oracle_sac.activation("tanh"):` -- Net(activation=nn.Tanh) instead of the default nn.ReLU for every trunk evaluated
    inside the block (test infrastructure: the parameter dicts carry no module structure)."""

    def __init__(self, name: str):
        self.fn = {"relu": F.relu, "tanh": torch.tanh}[name]

    def __enter__(self):
        self.old, _ACT["fn"] = _ACT["fn"], self.fn

    def __exit__(self, *exc):
        _ACT["fn"] = self.old


def trunk_forward(p, x):
    """Net / MLP: the activation (default ReLU) after every hidden Linear (utils/net/common.py:90-178)."""
    for i in range(1, depth_of(p) + 1):
        x = _ACT["fn"](F.linear(x, p[f"w{i}"], p[f"b{i}"]))
    return x


def layer_sizes(hidden) -> tuple[list[int], list[int]]:
    """(actor hidden sizes, critic hidden sizes) from an int (Net[h, h] everywhere), a flat pair (actor = critics), four
    widths (actor h1, h2, critic h1, h2) or a nested pair (actor sizes, critic sizes) of any depth >= 1."""
    if isinstance(hidden, (int, np.integer)):
        return [int(hidden)] * 2, [int(hidden)] * 2
    h = tuple(hidden)
    if len(h) == 2 and not isinstance(h[0], (int, np.integer)):
        return [int(x) for x in h[0]], [int(x) for x in h[1]]
    h = tuple(int(x) for x in h)
    return (list(h), list(h)) if len(h) == 2 else (list(h[:2]), list(h[2:]))


def _linears(sizes_in: int, sizes: list[int]):
    L, out, k = torch.nn.Linear, [], sizes_in
    for h in sizes:
        out.append(L(k, h))
        k = h
    return out


def hidden_widths(hidden) -> tuple[int, int, int, int]:
    """(actor h1, actor h2, critic h1, critic h2) from an int (all equal), a pair (actor = critics) or four widths:
    Net(hidden_sizes=[h1, h2]) of the reference takes any widths (utils/net/common.py:246-369)."""
    if isinstance(hidden, (int, np.integer)):
        return (int(hidden),) * 4
    h = tuple(int(x) for x in hidden)
    return h + h if len(h) == 2 else h


def init_sac_params(obs_dim: int, act_dim: int, seed: int, hidden=256):
    """Same RNG consumption as examples/mujoco/mujoco_sac.py:82-104 after torch.manual_seed(seed):
    Net(actor), actor mu / sigma heads, Net(critic1), Net(critic2), critic1.last, critic2.last
    -> (actor, critic1, critic2) parameter dicts.  `hidden`: see `layer_sizes` (any depth since round 6)."""
    torch.manual_seed(seed)
    L = torch.nn.Linear
    sa, sc = layer_sizes(hidden)
    na = _linears(obs_dim, sa)
    mu, sig = L(sa[-1], act_dim), L(sa[-1], act_dim)
    n1, n2 = _linears(obs_dim + act_dim, sc), _linears(obs_dim + act_dim, sc)
    q1, q2 = L(sc[-1], 1), L(sc[-1], 1)
    wb = lambda m: (m.weight.detach().clone(), m.bias.detach().clone())  # noqa: E731
    flat = lambda ms: [t for m in ms for t in wb(m)]                      # noqa: E731
    actor = dict(zip(actor_order(len(sa)), flat(na + [mu, sig])))
    critic1 = dict(zip(critic_order(len(sc)), flat(n1 + [q1])))
    critic2 = dict(zip(critic_order(len(sc)), flat(n2 + [q2])))
    return actor, critic1, critic2


def flatten(p: dict, order: list[str]) -> torch.Tensor:
    return torch.cat([p[k].reshape(-1) for k in order])


def actor_forward(p, obs, max_action: float = 0.0):
    """max_action > 0: ContinuousActorProbabilistic(unbounded=False), the class default -- mu = max_action * tanh(mu)
    (continuous.py:230-231); 0: unbounded=True as in examples/mujoco/mujoco_sac.py."""
    h = trunk_forward(p, obs)
    mu = F.linear(h, p["wmu"], p["bmu"])
    if max_action > 0.0:
        mu = max_action * torch.tanh(mu)
    sigma = torch.clamp(F.linear(h, p["wsig"], p["bsig"]), min=SIGMA_MIN, max=SIGMA_MAX).exp()
    return mu, sigma


def critic_forward(p, obs, act):
    x = torch.cat([obs.flatten(1), act.flatten(1)], dim=1)
    return F.linear(trunk_forward(p, x), p["wq"], p["bq"])


def policy_forward(p, obs, noise, max_action: float = 0.0):
    """SACPolicy.forward in training mode with rsample() = loc + noise * scale (sac.py:108-131)
    -> (squashed action [B, A], log_prob [B, 1], mu, sigma)."""
    mu, sigma = actor_forward(p, obs, max_action)
    dist = Independent(Normal(loc=mu, scale=sigma), 1)
    act = mu + noise * sigma                                   # Normal.rsample
    log_prob = dist.log_prob(act).unsqueeze(-1)
    squashed = torch.tanh(act)
    log_prob = log_prob - torch.log(1 - squashed.pow(2) + TANH_EPS).sum(-1, keepdim=True)
    return squashed, log_prob, mu, sigma


@dataclass
class SACConfig:
    gamma: float = 0.99
    tau: float = 0.005
    n_step: int = 1
    alpha: float = 0.2                  # used when auto_alpha is False
    auto_alpha: bool = False
    target_entropy: float = 0.0
    log_alpha0: float = 0.0
    actor_lr: float = 1e-3
    critic_lr: float = 1e-3
    alpha_lr: float = 3e-4
    betas: tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8
    max_action: float = 0.0             # > 0: bounded actor (unbounded=False, the class default); 0: unbounded=True


@dataclass
class Adam:
    lr: float
    betas: tuple[float, float]
    eps: float
    m: dict = field(default_factory=dict)
    v: dict = field(default_factory=dict)
    step: int = 0

    def apply(self, params: dict, grads: dict) -> dict:
        """torch.optim.Adam single-tensor arithmetic (optim.py:89-110 -> torch/optim/adam.py)."""
        self.step += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.step, 1 - b2 ** self.step
        out = {}
        for k, g in grads.items():
            m = self.m.setdefault(k, torch.zeros_like(g))
            v = self.v.setdefault(k, torch.zeros_like(g))
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / np.sqrt(bc2)).add_(self.eps)
            out[k] = params[k].addcdiv(m, denom, value=-(self.lr / bc1))
        return out


@dataclass
class SACState:
    actor: dict
    critic1: dict
    critic2: dict
    critic1_old: dict
    critic2_old: dict
    log_alpha: torch.Tensor
    opt_actor: Adam
    opt_c1: Adam
    opt_c2: Adam
    opt_alpha: Adam

    @classmethod
    def create(cls, actor, critic1, critic2, cfg: SACConfig):
        cp = lambda d: {k: v.clone() for k, v in d.items()}  # noqa: E731
        mk = lambda lr: Adam(lr, cfg.betas, cfg.adam_eps)    # noqa: E731
        return cls(cp(actor), cp(critic1), cp(critic2), cp(critic1), cp(critic2),
                   torch.tensor(cfg.log_alpha0 if cfg.auto_alpha else float(np.log(cfg.alpha))),
                   mk(cfg.actor_lr), mk(cfg.critic_lr), mk(cfg.critic_lr), mk(cfg.alpha_lr))


def alpha_value(st: SACState, cfg: SACConfig) -> float:
    return float(st.log_alpha.detach().exp().item()) if cfg.auto_alpha else cfg.alpha     # sac.py:199-201


def target_q(st: SACState, cfg: SACConfig, obs_next, noise) -> torch.Tensor:
    """ddpg.py:327-339 + sac.py:290-296 -> [B, 1]."""
    with torch.no_grad():
        act, logp, _, _ = policy_forward(st.actor, obs_next, noise, cfg.max_action)
        q = torch.min(critic_forward(st.critic1_old, obs_next, act), critic_forward(st.critic2_old, obs_next, act))
        return q - alpha_value(st, cfg) * logp


def _grads(loss, p: dict) -> dict:
    gs = torch.autograd.grad(loss, list(p.values()))
    return dict(zip(p.keys(), gs))


def update_with_batch(st: SACState, cfg: SACConfig, obs, act, returns, noise, weight=None, collect=None):
    """sac.py:298-336 -> dict(actor_loss, critic1_loss, critic2_loss, alpha, alpha_loss, weight)."""
    obs = torch.as_tensor(obs, dtype=torch.float32)
    act = torch.as_tensor(act, dtype=torch.float32)
    ret = torch.as_tensor(returns, dtype=torch.float32).flatten()
    noise = torch.as_tensor(noise, dtype=torch.float32)
    w = 1.0 if weight is None else torch.as_tensor(weight, dtype=torch.float32)
    out = {}
    tds = []
    for name, opt in (("critic1", st.opt_c1), ("critic2", st.opt_c2)):        # ddpg.py:279-285
        p = {k: v.clone().requires_grad_(True) for k, v in getattr(st, name).items()}
        td = critic_forward(p, obs, act).flatten() - ret
        loss = (td.pow(2) * w).mean()
        g = _grads(loss, p)
        if collect is not None:
            collect[name + "_grads"] = g
        setattr(st, name, opt.apply(getattr(st, name), g))
        tds.append(td.detach())
        out[name + "_loss"] = float(loss.item())
    out["weight"] = (tds[0] + tds[1]) / 2.0                                       # sac.py:306
    alpha = alpha_value(st, cfg)
    p = {k: v.clone().requires_grad_(True) for k, v in st.actor.items()}
    a, logp, _, _ = policy_forward(p, obs, noise, cfg.max_action)
    q1a = critic_forward(st.critic1, obs, a).flatten()
    q2a = critic_forward(st.critic2, obs, a).flatten()
    actor_loss = (alpha * logp.flatten() - torch.min(q1a, q2a)).mean()
    g = _grads(actor_loss, p)
    if collect is not None:
        collect["actor_grads"] = g
    st.actor = st.opt_actor.apply(st.actor, g)
    out["actor_loss"] = float(actor_loss.item())
    out["alpha_loss"] = None
    if cfg.auto_alpha:                                                            # sac.py:203-209
        entropy = -logp.detach()
        la = st.log_alpha.clone().requires_grad_(True)
        alpha_loss = -(la * (cfg.target_entropy - entropy)).mean()
        (ga,) = torch.autograd.grad(alpha_loss, [la])
        st.log_alpha = st.opt_alpha.apply({"a": st.log_alpha}, {"a": ga})["a"]
        out["alpha_loss"] = float(alpha_loss.item())
    for old, new in ((st.critic1_old, st.critic1), (st.critic2_old, st.critic2)):  # lagged_network.py:17-18
        for k in old:
            old[k] = cfg.tau * new[k] + (1 - cfg.tau) * old[k]
    out["alpha"] = alpha_value(st, cfg)
    return out


def gradients(actor, critic1, critic2, alpha: float, obs, act, returns, noise, weight=None, dtype=torch.float32,
              max_action: float = 0.0):
    """The three loss gradients of one update with the critics held fixed (what update_with_batch computes when
    every learning rate is 0), evaluated in `dtype` -- float64 gives the yardstick for float32 rounding noise."""
    cast = lambda d: {k: v.to(dtype).clone().requires_grad_(True) for k, v in d.items()}  # noqa: E731
    t = lambda x: torch.as_tensor(x).to(dtype)                                             # noqa: E731
    obs, act, ret, noise = t(obs), t(act), t(returns).flatten(), t(noise)
    w = 1.0 if weight is None else t(weight)
    out = {}
    for name, p in (("critic1", cast(critic1)), ("critic2", cast(critic2))):
        td = critic_forward(p, obs, act).flatten() - ret
        out[name + "_grads"] = _grads((td.pow(2) * w).mean(), p)
    p = cast(actor)
    a, logp, _, _ = policy_forward(p, obs, noise, max_action)
    c1 = {k: v.to(dtype) for k, v in critic1.items()}
    c2 = {k: v.to(dtype) for k, v in critic2.items()}
    loss = (alpha * logp.flatten() - torch.min(critic_forward(c1, obs, a).flatten(),
                                               critic_forward(c2, obs, a).flatten())).mean()
    out["actor_grads"] = _grads(loss, p)
    return out


# =====================================================================================================
# TD3 / DDPG (deterministic actor) on the same networks -- SURVEY 8f N3
#   ContinuousActorDeterministic.forward   utils/net/continuous.py:70-85   (max_action * tanh(last(h)))
#   ActorCriticOffPolicyAlgorithm._target_q      modelfree/ddpg.py:327-339
#   DDPG._target_q_compute_action / _update_with_batch   ddpg.py:397-411
#   TD3._target_q_compute_action / _update_with_batch    td3.py:190-226 (policy smoothing noise, delayed actor)
# =====================================================================================================
DET_ACTOR_ORDER = ["w1", "b1", "w2", "b2", "wa", "ba"]
TIANSHOU_DET_ACTOR_KEYS = ["preprocess.model.model.0.weight", "preprocess.model.model.0.bias",
                           "preprocess.model.model.2.weight", "preprocess.model.model.2.bias",
                           "last.model.0.weight", "last.model.0.bias"]


def init_td3_params(obs_dim: int, act_dim: int, seed: int, twin: bool = True, hidden=256):
    """RNG consumption of examples/mujoco/mujoco_td3.py:85-103 (mujoco_ddpg.py without the second critic):
    Net(actor), actor.last, Net(critic1)[, Net(critic2)], critic1.last[, critic2.last].  `hidden`: see `layer_sizes`."""
    torch.manual_seed(seed)
    L = torch.nn.Linear
    wb = lambda m: (m.weight.detach().clone(), m.bias.detach().clone())  # noqa: E731
    flat = lambda ms: [t for m in ms for t in wb(m)]                      # noqa: E731
    sa, sc = layer_sizes(hidden)
    a = _linears(obs_dim, sa) + [L(sa[-1], act_dim)]
    n1 = _linears(obs_dim + act_dim, sc)
    n2 = _linears(obs_dim + act_dim, sc) if twin else None
    q1 = L(sc[-1], 1)
    q2 = L(sc[-1], 1) if twin else None
    return (dict(zip(det_actor_order(len(sa)), flat(a))), dict(zip(critic_order(len(sc)), flat(n1 + [q1]))),
            dict(zip(critic_order(len(sc)), flat(n2 + [q2]))) if twin else None)


def det_actor_forward(p, obs, max_action: float = 1.0):
    return max_action * torch.tanh(F.linear(trunk_forward(p, obs), p["wa"], p["ba"]))


@dataclass
class TD3Config:
    gamma: float = 0.99
    tau: float = 0.005
    n_step: int = 1
    twin: bool = True               # False = DDPG
    policy_noise: float = 0.2
    noise_clip: float = 0.5
    update_actor_freq: int = 2
    max_action: float = 1.0
    actor_lr: float = 1e-3
    critic_lr: float = 1e-3
    betas: tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8


@dataclass
class TD3State:
    actor: dict
    critic1: dict
    critic2: dict | None
    actor_old: dict
    critic1_old: dict
    critic2_old: dict | None
    opt_actor: Adam
    opt_c1: Adam
    opt_c2: Adam
    cnt: int = 0
    last_actor_loss: float = 0.0

    @classmethod
    def create(cls, actor, critic1, critic2, cfg: TD3Config):
        cp = lambda d: None if d is None else {k: v.clone() for k, v in d.items()}  # noqa: E731
        mk = lambda lr: Adam(lr, cfg.betas, cfg.adam_eps)                            # noqa: E731
        return cls(cp(actor), cp(critic1), cp(critic2), cp(actor), cp(critic1), cp(critic2),
                   mk(cfg.actor_lr), mk(cfg.critic_lr), mk(cfg.critic_lr))


def td3_target_q(st: TD3State, cfg: TD3Config, obs_next, noise=None) -> torch.Tensor:
    """ddpg.py:327-339 with DDPG's lagged actor (:397-399) or TD3's smoothed one (td3.py:190-202) -> [B, 1]."""
    with torch.no_grad():
        act = det_actor_forward(st.actor_old, obs_next, cfg.max_action)
        if cfg.twin:
            n = torch.as_tensor(noise, dtype=torch.float32) * cfg.policy_noise
            if cfg.noise_clip > 0.0:
                n = n.clamp(-cfg.noise_clip, cfg.noise_clip)
            act = act + n
            return torch.min(critic_forward(st.critic1_old, obs_next, act), critic_forward(st.critic2_old, obs_next, act))
        return critic_forward(st.critic1_old, obs_next, act)


def td3_update_with_batch(st: TD3State, cfg: TD3Config, obs, act, returns, weight=None, collect=None):
    """td3.py:204-226 (twin) / ddpg.py:401-411 -> dict(actor_loss, critic1_loss[, critic2_loss], weight)."""
    obs = torch.as_tensor(obs, dtype=torch.float32)
    act = torch.as_tensor(act, dtype=torch.float32)
    ret = torch.as_tensor(returns, dtype=torch.float32).flatten()
    w = 1.0 if weight is None else torch.as_tensor(weight, dtype=torch.float32)
    out, tds = {}, []
    for name, opt in (("critic1", st.opt_c1), ("critic2", st.opt_c2)):
        if getattr(st, name) is None:
            continue
        p = {k: v.clone().requires_grad_(True) for k, v in getattr(st, name).items()}
        td = critic_forward(p, obs, act).flatten() - ret
        loss = (td.pow(2) * w).mean()
        g = _grads(loss, p)
        if collect is not None:
            collect[name + "_grads"] = g
        setattr(st, name, opt.apply(getattr(st, name), g))
        tds.append(td.detach())
        out[name + "_loss"] = float(loss.item())
    out["weight"] = (tds[0] + tds[1]) / 2.0 if cfg.twin else tds[0]
    freq = cfg.update_actor_freq if cfg.twin else 1
    if st.cnt % freq == 0:
        p = {k: v.clone().requires_grad_(True) for k, v in st.actor.items()}
        actor_loss = -critic_forward(st.critic1, obs, det_actor_forward(p, obs, cfg.max_action)).mean()
        g = _grads(actor_loss, p)
        if collect is not None:
            collect["actor_grads"] = g
        st.actor = st.opt_actor.apply(st.actor, g)
        st.last_actor_loss = float(actor_loss.item())
        pairs = [(st.actor_old, st.actor), (st.critic1_old, st.critic1)]
        if cfg.twin:
            pairs.append((st.critic2_old, st.critic2))
        for old, new in pairs:                                                  # lagged_network.py:17-18
            for k in old:
                old[k] = cfg.tau * new[k] + (1 - cfg.tau) * old[k]
    st.cnt += 1
    out["actor_loss"] = st.last_actor_loss
    return out
