"""TEST INFRASTRUCTURE ONLY - CPU (torch fp32 autograd) restatement of the reference's Reinforce learn() path.

Never imported by the product (`tianshou_amd/`).  Pinned against the UNMODIFIED reference through
tests/golden/reinforce_*.npz (oracle/gen_golden.py::gen_reinforce).

Follows:
  nets        the actor of oracle_ppo (examples/mujoco/mujoco_reinforce.py:84-103: Net[64, 64] tanh, unbounded Gaussian with a
              state-independent sigma_param); there is no critic
  preprocess  DiscountedReturnComputation.add_discounted_returns modelfree/reinforce.py:266-310: v_s_ = full(ret_rms.mean),
              compute_episodic_return(gae_lambda = 1) with v_s = roll(v_s_ * value_mask, 1) (algorithm_base.py:706-717),
              optional standardisation by the running mean / std and ret_rms.update (the SURVEY 8a row a7 call site)
  update      Reinforce._update_with_batch reinforce.py:363-382: loss = -(log_prob * returns).mean(); Optimizer.step
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import oracle as O
from . import oracle_ppo as OP

ACTOR_KEYS = ["a_w1", "a_b1", "a_w2", "a_b2", "a_wmu", "a_bmu", "a_sigma"]


@dataclass
class ReinforceConfig:
    gamma: float = 0.99
    return_standardization: bool = False
    lr: float = 1e-3
    betas: tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8
    max_grad_norm: float | None = None


def preprocess(st: OP.PPOState, cfg: ReinforceConfig, rew, terminated, truncated, indices, unfinished) -> torch.Tensor:
    """-> batch.returns (float64 in the reference; cast to float32 at reinforce.py:376)."""
    n = len(indices)
    v_next = np.full(n, st.ret_rms.mean) * (~np.asarray(terminated).astype(bool))       # value_mask, algorithm_base.py:711
    v_s = np.roll(v_next, 1)                                                           # :712
    ret, _ = O.compute_episodic_return(rew, terminated, truncated, indices, unfinished, v_next, v_s, cfg.gamma, 1.0)
    if cfg.return_standardization:
        out = (ret - st.ret_rms.mean) / np.sqrt(st.ret_rms.var + 1e-8)
        st.ret_rms = OP.RMS(*O.rms_update(st.ret_rms.mean, st.ret_rms.var, st.ret_rms.count, ret))
    else:
        out = ret
    return torch.as_tensor(out, dtype=torch.float32)


def update(st: OP.PPOState, cfg: ReinforceConfig, obs, act, returns: torch.Tensor, batch_size, repeat: int, perms,
           collect: dict | None = None) -> np.ndarray:
    from .oracle_npg import NPGConfig, _critic_adam          # the same clip + Adam arithmetic on a parameter subset

    acfg = NPGConfig(lr=cfg.lr, betas=cfg.betas, adam_eps=cfg.adam_eps, max_grad_norm=cfg.max_grad_norm)
    obs_t, act_t = torch.as_tensor(obs, dtype=torch.float32), torch.as_tensor(act, dtype=torch.float32)
    n, out = len(obs_t), []
    for r in range(repeat):
        perm = torch.as_tensor(np.asarray(perms[r], dtype=np.int64))
        for lo, hi in OP.split_slices(n, batch_size or n, merge_last=True):
            rows = perm[lo:hi]
            pa = {k: st.params[k].clone().requires_grad_(True) for k in ACTOR_KEYS}
            mu, sigma = OP.actor_forward({**st.params, **pa}, obs_t[rows])
            log_prob = OP.dist_of(mu, sigma).log_prob(act_t[rows]).reshape(len(rows), -1).transpose(0, 1)
            loss = -(log_prob * returns[rows]).mean()
            gs = dict(zip(ACTOR_KEYS, torch.autograd.grad(loss, [pa[k] for k in ACTOR_KEYS])))
            if collect is not None:
                collect["grads"] = {k: g.clone() for k, g in gs.items()}
            _critic_adam(st, acfg, gs)
            out.append(float(loss.item()))
    return np.asarray(out, np.float64)
