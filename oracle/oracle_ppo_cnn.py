"""TEST INFRASTRUCTURE ONLY - CPU (torch fp32 autograd) restatement of the reference's PPO learn() path on
the Atari actor-critic (shared DQNet feature trunk, categorical policy).

Never imported by the product (`tianshou_amd/`).  Pinned against the UNMODIFIED reference through
tests/golden/ppo_cnn.npz (oracle/gen_golden.py::gen_ppo_cnn).

Follows:
  nets        examples/atari/atari_ppo.py:106-118: DQNet(features_only=True, output_dim_added_layer=512)
              (env/atari/atari_network.py:79-107) shared by DiscreteActor(softmax_output=False) and
              DiscreteCritic (utils/net/discrete.py:27-123)
  policy      DiscreteActorPolicy / dist_fn_categorical_from_logits (discrete.py:20-24,
              modelfree/reinforce.py:167-192): Categorical(logits=...)
  preprocess  _add_returns_and_advantages a2c.py:115-153, PPO._preprocess_batch ppo.py:146-162
  update      PPO._update_with_batch ppo.py:164-224, Optimizer.step algorithm_base.py:484-500
The reference runs the shared trunk twice per minibatch (policy forward and critic forward); the restatement
does the same so that autograd sums the two paths exactly as there.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from torch.distributions import Categorical

from . import oracle as O
from . import oracle_dqn as OD
from . import oracle_ppo as OP

PARAM_ORDER = ["conv1.w", "conv1.b", "conv2.w", "conv2.b", "conv3.w", "conv3.b", "fc.w", "fc.b",
               "actor.w", "actor.b", "critic.w", "critic.b"]
# ActorCritic(actor, critic).state_dict() keys (the shared trunk appears under "actor." only after de-duplication
# by parameter identity; the fixture generator asserts the mapping)
TRUNK_KEYS = ["preprocess.net.0.0.weight", "preprocess.net.0.0.bias", "preprocess.net.0.2.weight",
              "preprocess.net.0.2.bias", "preprocess.net.0.4.weight", "preprocess.net.0.4.bias",
              "preprocess.net.1.weight", "preprocess.net.1.bias"]
HEAD_KEYS = ["last.model.0.weight", "last.model.0.bias"]
HIDDEN = 512


def param_shapes(c: int, h: int, w: int, n_act: int) -> dict[str, tuple[int, ...]]:
    oh, ow = OD.conv_out_hw(h, w)[-1]
    return {"conv1.w": (32, c, 8, 8), "conv1.b": (32,), "conv2.w": (64, 32, 4, 4), "conv2.b": (64,),
            "conv3.w": (64, 64, 3, 3), "conv3.b": (64,), "fc.w": (HIDDEN, 64 * oh * ow), "fc.b": (HIDDEN,),
            "actor.w": (n_act, HIDDEN), "actor.b": (n_act,), "critic.w": (1, HIDDEN), "critic.b": (1,)}


def init_params(c: int, h: int, w: int, n_act: int, seed: int) -> dict[str, torch.Tensor]:
    """Same RNG consumption as torch.manual_seed(seed); DQNet(..., features_only=True,
    output_dim_added_layer=512); DiscreteActor(...); DiscreteCritic(...) with default (identity) layer_init."""
    torch.manual_seed(seed)
    oh, ow = OD.conv_out_hw(h, w)[-1]
    mods = [torch.nn.Conv2d(c, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1),
            torch.nn.Linear(64 * oh * ow, HIDDEN), torch.nn.Linear(HIDDEN, n_act), torch.nn.Linear(HIDDEN, 1)]
    p = {}
    for name, m in zip(["conv1", "conv2", "conv3", "fc", "actor", "critic"], mods):
        p[name + ".w"], p[name + ".b"] = m.weight.detach().clone(), m.bias.detach().clone()
    return p


def flatten_params(p) -> torch.Tensor:
    return torch.cat([p[k].reshape(-1) for k in PARAM_ORDER])


def features(p, obs) -> torch.Tensor:
    x = torch.as_tensor(np.asarray(obs) if not isinstance(obs, torch.Tensor) else obs, dtype=torch.float32)
    x = F.relu(F.conv2d(x, p["conv1.w"], p["conv1.b"], stride=4))
    x = F.relu(F.conv2d(x, p["conv2.w"], p["conv2.b"], stride=2))
    x = F.relu(F.conv2d(x, p["conv3.w"], p["conv3.b"], stride=1))
    return F.relu(F.linear(x.flatten(1), p["fc.w"], p["fc.b"]))


def actor_forward(p, obs) -> torch.Tensor:
    return F.linear(features(p, obs), p["actor.w"], p["actor.b"])          # logits (softmax_output=False)


def critic_forward(p, obs) -> torch.Tensor:
    return F.linear(features(p, obs), p["critic.w"], p["critic.b"])


def dist(p, obs) -> Categorical:
    return Categorical(logits=actor_forward(p, obs))          # dist_fn_categorical_from_logits, discrete.py:20-24


class _CnnNet:
    """The network-specific part of this restatement (oracle_ppo_discrete.MlpNet is the other implementation)."""
    dist = staticmethod(dist)
    critic_forward = staticmethod(critic_forward)


def _chunks(n, size):
    return OP.split_slices(n, size, merge_last=True)          # Batch.split(merge_last=True), a2c.py:125


def _clip_adam(st: OP.PPOState, cfg: OP.PPOConfig, grads: dict) -> None:
    """clip_grad_norm_ over all parameters (algorithm_base.py:496-499) + torch.optim.Adam single-tensor math."""
    if cfg.max_grad_norm:
        total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads.values()]))
        coef = torch.clamp(cfg.max_grad_norm / (total + 1e-6), max=1.0)
        grads = {k: g * coef for k, g in grads.items()}
    b1, b2 = cfg.betas
    st.adam_step += 1
    bc1, bc2 = 1.0 - b1 ** st.adam_step, 1.0 - b2 ** st.adam_step
    for k, g in grads.items():
        if k not in st.adam_m:
            st.adam_m[k], st.adam_v[k] = torch.zeros_like(g), torch.zeros_like(g)
        m, v = st.adam_m[k], st.adam_v[k]
        m.lerp_(g, 1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (v.sqrt() / np.sqrt(bc2)).add_(cfg.adam_eps)
        st.params[k] = st.params[k].addcdiv(m, denom, value=-(cfg.lr / bc1))


def preprocess(st: OP.PPOState, cfg: OP.PPOConfig, obs, obs_next, act, rew, terminated, truncated, indices,
               unfinished, net=_CnnNet):
    """a2c.py:115-153 + ppo.py:146-162 with the categorical policy -> dict(v_s, returns, adv, logp_old)."""
    p = st.params
    critic_forward = net.critic_forward
    with torch.no_grad():
        v_s = torch.cat([critic_forward(p, obs[a:b]) for a, b in _chunks(len(obs), cfg.max_batchsize)]).flatten()
        v_s_ = torch.cat([critic_forward(p, obs_next[a:b]) for a, b in _chunks(len(obs), cfg.max_batchsize)]).flatten()
    v_np, vn_np = v_s.numpy(), v_s_.numpy()
    if cfg.return_scaling:                                                   # a2c.py:134-136
        scale = np.sqrt(st.ret_rms.var + 1e-8)
        v_np, vn_np = v_np * scale, vn_np * scale
    vn_masked = vn_np * (~np.asarray(terminated).astype(bool))              # value_mask :711
    ret, adv = O.compute_episodic_return(rew, terminated, truncated, indices, unfinished, vn_masked, v_np,
                                         cfg.gamma, cfg.gae_lambda)
    if cfg.return_scaling:                                                   # a2c.py:146-148
        returns = ret / np.sqrt(st.ret_rms.var + 1e-8)
        st.ret_rms = OP.RMS(*O.rms_update(st.ret_rms.mean, st.ret_rms.var, st.ret_rms.count, ret))
    else:
        returns = ret
    with torch.no_grad():
        act_t = torch.as_tensor(np.asarray(act), dtype=torch.int64)
        logp = torch.cat([net.dist(p, obs[a:b]).log_prob(act_t[a:b])
                          for a, b in _chunks(len(obs), cfg.max_batchsize)])
    return {"v_s": v_s, "returns": torch.as_tensor(returns, dtype=torch.float32),
            "adv": torch.as_tensor(adv, dtype=torch.float32), "logp_old": logp}


def minibatch_loss(p, cfg: OP.PPOConfig, obs, act, adv, returns, logp_old, v_s, net=_CnnNet):
    """ppo.py:179-211 (cfg.algo "ppo") or a2c.py:262-273 ("a2c") -> (loss, clip / actor loss, vf_loss, ent_loss)."""
    if cfg.algo == "a2c":
        dist = net.dist(p, obs)
        log_prob = dist.log_prob(act).reshape(len(adv), -1).transpose(0, 1)
        actor_loss = -(log_prob * adv).mean()
        vf_loss = F.mse_loss(returns, net.critic_forward(p, obs).flatten())
        ent_loss = dist.entropy().mean()
        return actor_loss + cfg.vf_coef * vf_loss - cfg.ent_coef * ent_loss, actor_loss, vf_loss, ent_loss
    if cfg.advantage_normalization:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    dist = net.dist(p, obs)
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
    value = net.critic_forward(p, obs).flatten()
    if cfg.value_clip:
        v_clip = v_s + (value - v_s).clamp(-cfg.eps_clip, cfg.eps_clip)
        vf_loss = torch.max((returns - value).pow(2), (returns - v_clip).pow(2)).mean()
    else:
        vf_loss = (returns - value).pow(2).mean()
    ent_loss = dist.entropy().mean()
    loss = clip_loss + cfg.vf_coef * vf_loss - cfg.ent_coef * ent_loss
    return loss, clip_loss, vf_loss, ent_loss


def update(st: OP.PPOState, cfg: OP.PPOConfig, obs, act, pre: dict, batch_size, repeat: int, perms,
           collect: dict | None = None, net=_CnnNet, recompute=None) -> np.ndarray:
    """ppo.py:164-224 -> losses [steps, 4].  `recompute` (recompute_advantage, ppo.py:174-178): a callable that runs
    `_add_returns_and_advantages` again (e.g. `lambda: preprocess(st, cfg, ...)`) and returns the new v_s / returns / adv;
    it is called before every repeat after the first, log pi_old stays."""
    n = len(obs)
    act_t = torch.as_tensor(np.asarray(act), dtype=torch.int64)
    obs_t = torch.as_tensor(np.asarray(obs), dtype=torch.float32)
    out = []
    for r in range(repeat):
        if recompute is not None and r > 0:
            new = recompute()
            pre = dict(pre, v_s=new["v_s"], returns=new["returns"], adv=new["adv"])
        perm = torch.as_tensor(np.asarray(perms[r], dtype=np.int64))
        for lo, hi in OP.split_slices(n, batch_size or n, merge_last=True):
            rows = perm[lo:hi]
            p = {k: v.clone().requires_grad_(True) for k, v in st.params.items()}
            loss, clip, vf, ent = minibatch_loss(p, cfg, obs_t[rows], act_t[rows], pre["adv"][rows],
                                                 pre["returns"][rows], pre["logp_old"][rows], pre["v_s"][rows], net=net)
            loss.backward()
            grads = {k: v.grad for k, v in p.items()}
            if collect is not None:
                collect["grads"] = {k: g.clone() for k, g in grads.items()}
            _clip_adam(st, cfg, grads)
            out.append([loss.item(), clip.item(), vf.item(), ent.item()])
    return np.asarray(out, np.float64)
