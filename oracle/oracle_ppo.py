"""TEST INFRASTRUCTURE ONLY - torch-fp32 (CPU) restatement of the PPO learn() path.

The floating-point part of the reference's PPO path lives in third-party ``torch``
(nn.Linear / tanh / torch.distributions.Normal / autograd / clip_grad_norm_ / Adam).  This
file restates the *sequence of operations* the reference issues on that library, on plain
tensors, so that the HIP engine can be checked step by step on the GPU box where
/root/reference does not exist.  It is pinned against the unmodified reference by
``oracle/gen_golden.py`` -> ``tests/golden/ppo_*.npz`` (tests/test_oracle_golden.py).

Followed reference code (paths relative to the reference checkout):
  actor/critic forward   tianshou/utils/net/common.py:172-178, 343-369;
                         tianshou/utils/net/continuous.py:144-169, 220-238
  distribution           examples/mujoco/mujoco_ppo.py:131-133  (Independent(Normal(mu, sigma), 1))
  preprocess             tianshou/algorithm/modelfree/a2c.py:115-153;  ppo.py:146-162
  update loop            tianshou/algorithm/modelfree/ppo.py:164-224
  optimizer step         tianshou/algorithm/algorithm_base.py:484-500;  optim.py:89-110
  minibatch order        tianshou/data/batch.py:1199-1215  (np.random.permutation, merge_last)
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
from torch.distributions import Independent, Normal

from . import oracle as _o

PARAM_ORDER = (
    "a_w1", "a_b1", "a_w2", "a_b2", "a_wmu", "a_bmu", "a_sigma",
    "c_w1", "c_b1", "c_w2", "c_b2", "c_wv", "c_bv",
)


def param_shapes(obs_dim: int, act_dim: int, hidden: int = 64) -> dict[str, tuple[int, ...]]:
    """Flat parameter layout shared with the engine (see include/tsengine.h)."""
    return {
        "a_w1": (hidden, obs_dim), "a_b1": (hidden,),
        "a_w2": (hidden, hidden), "a_b2": (hidden,),
        "a_wmu": (act_dim, hidden), "a_bmu": (act_dim,),
        "a_sigma": (act_dim,),
        "c_w1": (hidden, obs_dim), "c_b1": (hidden,),
        "c_w2": (hidden, hidden), "c_b2": (hidden,),
        "c_wv": (1, hidden), "c_bv": (1,),
    }


def init_params(obs_dim: int, act_dim: int, hidden: int = 64, seed: int = 0,
                sigma_init: float = -0.5) -> dict[str, torch.Tensor]:
    """examples/mujoco/mujoco_ppo.py:108-120: orthogonal(gain sqrt 2), zero bias,
    mu head x0.01, sigma_param = -0.5."""
    g = torch.Generator().manual_seed(seed)
    shapes = param_shapes(obs_dim, act_dim, hidden)
    p: dict[str, torch.Tensor] = {}
    for name, shp in shapes.items():
        t = torch.zeros(shp, dtype=torch.float32)
        if name.endswith(("w1", "w2", "wmu", "wv")):
            tmp = torch.empty(shp, dtype=torch.float32)
            # orthogonal_ without a generator argument on older torch: emulate
            a = torch.randn(shp, generator=g, dtype=torch.float32)
            rows, cols = shp
            flat = a if rows >= cols else a.t()
            q, r = torch.linalg.qr(flat)
            q = q * torch.sign(torch.diagonal(r)).unsqueeze(0)
            tmp.copy_(q if rows >= cols else q.t())
            t = tmp * math.sqrt(2.0)
            if name == "a_wmu":
                t = t * 0.01
        elif name == "a_sigma":
            t = torch.full(shp, sigma_init, dtype=torch.float32)
        p[name] = t.contiguous()
    return p


def flatten_params(p: dict[str, torch.Tensor]) -> torch.Tensor:
    return torch.cat([p[k].detach().reshape(-1) for k in PARAM_ORDER]).contiguous()


def unflatten_params(flat: torch.Tensor, obs_dim: int, act_dim: int, hidden: int = 64):
    shapes = param_shapes(obs_dim, act_dim, hidden)
    out, off = {}, 0
    for k in PARAM_ORDER:
        n = int(np.prod(shapes[k]))
        out[k] = flat[off:off + n].reshape(shapes[k]).clone()
        off += n
    assert off == flat.numel()
    return out


def _trunk(obs, w1, b1, w2, b2):
    h = torch.tanh(torch.nn.functional.linear(obs, w1, b1))
    return torch.tanh(torch.nn.functional.linear(h, w2, b2))


def actor_forward(p, obs, max_action: float | None = None):
    """ContinuousActorProbabilistic.forward (continuous.py:220-238); max_action = None: unbounded=True."""
    h = _trunk(obs, p["a_w1"], p["a_b1"], p["a_w2"], p["a_b2"])
    mu = torch.nn.functional.linear(h, p["a_wmu"], p["a_bmu"])
    if max_action is not None:
        mu = max_action * torch.tanh(mu)                     # continuous.py:230-231 (unbounded=False, the default)
    # continuous.py:236-238: sigma = (sigma_param.view(1,-1) + zeros_like(mu)).exp()
    sigma = (p["a_sigma"].view(1, -1) + torch.zeros_like(mu)).exp()
    return mu, sigma


def critic_forward(p, obs):
    h = _trunk(obs, p["c_w1"], p["c_b1"], p["c_w2"], p["c_b2"])
    return torch.nn.functional.linear(h, p["c_wv"], p["c_bv"])


def dist_of(mu, sigma):
    return Independent(Normal(mu, sigma), 1)


@dataclass
class PPOConfig:
    algo: str = "ppo"          # "ppo" (ppo.py:164-224) or "a2c" (a2c.py:249-290)
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
    max_batchsize: int = 256
    # optim.py:89-140: "adam" (AdamOptimizerFactory) or "rmsprop" (RMSpropOptimizerFactory; lr / adam_eps shared)
    optimizer: str = "adam"
    weight_decay: float = 0.0
    rms_alpha: float = 0.99
    rms_momentum: float = 0.0
    rms_centered: bool = False
    max_action: float | None = None    # the actor's tanh bound (None: unbounded=True)


@dataclass
class RMS:
    """statistics.py:81-91 initial state (mean 0, var 1, count 0)."""
    mean: float = 0.0
    var: float = 1.0
    count: float = 0.0


@dataclass
class PPOState:
    params: dict
    adam_m: dict = field(default_factory=dict)
    adam_v: dict = field(default_factory=dict)
    adam_step: int = 0
    ret_rms: RMS = field(default_factory=RMS)


def split_slices(n: int, size: int, merge_last: bool) -> list[tuple[int, int]]:
    """batch.py:1205-1215: chunk boundaries of Batch.split()."""
    if size == -1:
        size = n
    assert size >= 1
    merge_last = merge_last and n % size > 0
    out = []
    for idx in range(0, n, size):
        if merge_last and idx + size + size >= n:
            out.append((idx, n))
            break
        out.append((idx, min(idx + size, n)))
    return out


def add_returns_and_advantages(state: PPOState, cfg: PPOConfig, obs, obs_next, rew,
                               terminated, truncated, indices, unfinished):
    """a2c.py:115-153 -> (v_s f32, returns f32, adv f32) torch tensors."""
    p = state.params
    n = obs.shape[0]
    v_s, v_s_ = [], []
    with torch.no_grad():
        for lo, hi in split_slices(n, cfg.max_batchsize, merge_last=True):
            v_s.append(critic_forward(p, obs[lo:hi]))
            v_s_.append(critic_forward(p, obs_next[lo:hi]))
    v_s_t = torch.cat(v_s, dim=0).flatten()
    dev = v_s_t.device              # CPU in the parity tests; "cuda" in bench.py's ROCm-eager baseline leg, where - as in
    v_s_np = v_s_t.cpu().numpy()    # the reference (to_numpy, a2c.py:130-131) - the value estimates cross to the host
    v_next_np = torch.cat(v_s_, dim=0).flatten().cpu().numpy()
    if cfg.return_scaling:
        scale = np.sqrt(state.ret_rms.var + 1e-8)
        v_s_np = v_s_np * scale
        v_next_np = v_next_np * scale
    ret, adv = _o.compute_episodic_return(rew, terminated, truncated, indices, unfinished,
                                          v_next_np, v_s_np, cfg.gamma, cfg.gae_lambda)
    if cfg.return_scaling:
        returns = ret / np.sqrt(state.ret_rms.var + 1e-8)
        m, v, c = _o.rms_update(state.ret_rms.mean, state.ret_rms.var, state.ret_rms.count, ret)
        state.ret_rms = RMS(m, v, c)
    else:
        returns = ret
    return (v_s_t, torch.from_numpy(returns.astype(np.float32)).to(dev),       # to_torch_as(..., batch.v_s), a2c.py:151-152
            torch.from_numpy(adv.astype(np.float32)).to(dev))


def preprocess(state: PPOState, cfg: PPOConfig, obs, obs_next, act, rew, terminated, truncated,
               indices, unfinished):
    """ppo.py:146-162 -> dict(v_s, returns, adv, logp_old)."""
    v_s, returns, adv = add_returns_and_advantages(state, cfg, obs, obs_next, rew, terminated,
                                                   truncated, indices, unfinished)
    if cfg.algo == "a2c":      # A2C._preprocess_batch (a2c.py:239-247) has no logp_old
        return {"v_s": v_s, "returns": returns, "adv": adv, "logp_old": torch.zeros_like(adv)}
    logp = []
    with torch.no_grad():
        for lo, hi in split_slices(obs.shape[0], cfg.max_batchsize, merge_last=True):
            mu, sigma = actor_forward(state.params, obs[lo:hi], cfg.max_action)
            logp.append(dist_of(mu, sigma).log_prob(act[lo:hi]))
    return {"v_s": v_s, "returns": returns, "adv": adv, "logp_old": torch.cat(logp).flatten()}


def _adam_step(state: PPOState, cfg: PPOConfig, grads: dict):
    """torch.optim.Adam (optim.py:104-110: lr, betas, eps, weight_decay=0; no amsgrad),
    single-tensor formulation: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    if cfg.optimizer == "rmsprop":
        return _rmsprop_step(state, cfg, grads)
    b1, b2 = cfg.betas
    state.adam_step += 1
    t = state.adam_step
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    step_size = cfg.lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    for k in PARAM_ORDER:
        g = grads[k]
        if cfg.weight_decay != 0:
            g = g.add(state.params[k], alpha=cfg.weight_decay)      # torch/optim/adam.py _single_tensor_adam
        if k not in state.adam_m:
            state.adam_m[k] = torch.zeros_like(g)
            state.adam_v[k] = torch.zeros_like(g)
        m, v = state.adam_m[k], state.adam_v[k]
        m.lerp_(g, 1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (v.sqrt() / bc2_sqrt).add_(cfg.adam_eps)
        state.params[k] = state.params[k].addcdiv(m, denom, value=-step_size)


def _rmsprop_step(state: PPOState, cfg: PPOConfig, grads: dict):
    """torch.optim.RMSprop (optim.py:113-140), torch/optim/rmsprop.py `_single_tensor_rmsprop` operation by operation;
    square_avg lives in state.adam_v, the momentum buffer (momentum > 0) or grad_avg (centered) in state.adam_m."""
    assert not (cfg.rms_centered and cfg.rms_momentum > 0), "one auxiliary state vector"
    state.adam_step += 1
    alpha = cfg.rms_alpha
    for k in PARAM_ORDER:
        g = grads[k]
        if cfg.weight_decay != 0:
            g = g.add(state.params[k], alpha=cfg.weight_decay)
        if k not in state.adam_v:
            state.adam_m[k] = torch.zeros_like(g)
            state.adam_v[k] = torch.zeros_like(g)
        sq = state.adam_v[k]
        sq.mul_(alpha).addcmul_(g, g, value=1 - alpha)
        if cfg.rms_centered:
            ga = state.adam_m[k]
            ga.lerp_(g, 1 - alpha)
            avg = sq.addcmul(ga, ga, value=-1).sqrt_()
        else:
            avg = sq.sqrt()
        avg = avg.add_(cfg.adam_eps)
        if cfg.rms_momentum > 0:
            buf = state.adam_m[k]
            buf.mul_(cfg.rms_momentum).addcdiv_(g, avg)
            state.params[k] = state.params[k].add(buf, alpha=-cfg.lr)
        else:
            state.params[k] = state.params[k].addcdiv(g, avg, value=-cfg.lr)


def ppo_minibatch_loss(p, cfg: PPOConfig, obs, act, adv, returns, logp_old, v_s):
    """ppo.py:181-211 on one minibatch -> (loss, clip_loss, vf_loss, ent_loss)."""
    mu, sigma = actor_forward(p, obs, cfg.max_action)
    dist = dist_of(mu, sigma)
    if cfg.advantage_normalization:
        mean, std = adv.mean(), adv.std()
        adv = (adv - mean) / (std + 1e-8)
    ratios = (dist.log_prob(act) - logp_old).exp().float()
    ratios = ratios.reshape(ratios.size(0), -1).transpose(0, 1)
    surr1 = ratios * adv
    surr2 = ratios.clamp(1.0 - cfg.eps_clip, 1.0 + cfg.eps_clip) * adv
    if cfg.dual_clip:
        clip1 = torch.min(surr1, surr2)
        clip2 = torch.max(clip1, cfg.dual_clip * adv)
        clip_loss = -torch.where(adv < 0, clip2, clip1).mean()
    else:
        clip_loss = -torch.min(surr1, surr2).mean()
    value = critic_forward(p, obs).flatten()
    if cfg.value_clip:
        v_clip = v_s + (value - v_s).clamp(-cfg.eps_clip, cfg.eps_clip)
        vf1 = (returns - value).pow(2)
        vf2 = (returns - v_clip).pow(2)
        vf_loss = torch.max(vf1, vf2).mean()
    else:
        vf_loss = (returns - value).pow(2).mean()
    ent_loss = dist.entropy().mean()
    loss = clip_loss + cfg.vf_coef * vf_loss - cfg.ent_coef * ent_loss
    return loss, clip_loss, vf_loss, ent_loss


def a2c_minibatch_loss(p, cfg: PPOConfig, obs, act, adv, returns):
    """a2c.py:262-273 on one minibatch -> (loss, actor_loss, vf_loss, ent_loss)."""
    mu, sigma = actor_forward(p, obs, cfg.max_action)
    dist = dist_of(mu, sigma)
    log_prob = dist.log_prob(act)
    log_prob = log_prob.reshape(len(adv), -1).transpose(0, 1)
    actor_loss = -(log_prob * adv).mean()
    value = critic_forward(p, obs).flatten()
    vf_loss = torch.nn.functional.mse_loss(returns, value)
    ent_loss = dist.entropy().mean()
    loss = actor_loss + cfg.vf_coef * vf_loss - cfg.ent_coef * ent_loss
    return loss, actor_loss, vf_loss, ent_loss


def update(state: PPOState, cfg: PPOConfig, data: dict, pre: dict, batch_size: int | None,
           repeat: int, perms: list[np.ndarray], recompute=None, collect_grads: bool = False):
    """ppo.py:164-224.  ``perms[r]`` is the np.random.permutation(N) the reference draws in
    repeat r (batch.py:1209); the engine receives the same host-supplied permutations.
    Returns per-step arrays (loss, clip, vf, ent) [+ the flat UNCLIPPED gradient of the last step]."""
    obs, act = data["obs"], data["act"]
    n = obs.shape[0]
    size = batch_size or -1
    losses = []
    first_grads = None
    for r in range(repeat):
        if cfg.recompute_advantage and r > 0:
            assert recompute is not None
            v_s, returns, adv = recompute()
            pre = dict(pre, v_s=v_s, returns=returns, adv=adv)
        perm = torch.from_numpy(np.asarray(perms[r], dtype=np.int64)).to(obs.device)
        for lo, hi in split_slices(n, size, merge_last=True):
            idx = perm[lo:hi]
            p = {k: v.detach().clone().requires_grad_(True) for k, v in state.params.items()}
            if cfg.algo == "a2c":
                loss, clip_loss, vf_loss, ent_loss = a2c_minibatch_loss(
                    p, cfg, obs[idx], act[idx], pre["adv"][idx], pre["returns"][idx])
            else:
                loss, clip_loss, vf_loss, ent_loss = ppo_minibatch_loss(
                    p, cfg, obs[idx], act[idx], pre["adv"][idx], pre["returns"][idx],
                    pre["logp_old"][idx], pre["v_s"][idx])
            loss.backward()
            plist = [p[k] for k in PARAM_ORDER]
            for t in plist:
                if t.grad is None:
                    t.grad = torch.zeros_like(t)
            if collect_grads:  # unclipped gradient of the most recent step
                first_grads = torch.cat([p[k].grad.detach().reshape(-1) for k in PARAM_ORDER]).clone()
            if cfg.max_grad_norm is not None:
                torch.nn.utils.clip_grad_norm_(plist, max_norm=cfg.max_grad_norm)
            grads = {k: p[k].grad.detach() for k in PARAM_ORDER}
            _adam_step(state, cfg, grads)
            losses.append([loss.item(), clip_loss.item(), vf_loss.item(), ent_loss.item()])
    out = np.asarray(losses, dtype=np.float64).reshape(-1, 4)
    return (out, first_grads) if collect_grads else out
