"""Binding a Tianshou maintainer adds: `HipPPO`, a `PPO` subclass whose learn() hooks run on the
MI355X engine.  Needs `tianshou` importable (it is not on the GPU test box; there the same engine
is driven through `tianshou_amd.ppo.PPOEngine` directly).

Overrides exactly the two hooks `Algorithm._update` calls (algorithm_base.py:622-627):
    _preprocess_batch(batch, buffer, indices) -> batch          (ppo.py:146-162)
    _update_with_batch(batch, batch_size, repeat) -> A2CTrainingStats   (ppo.py:164-224)
keeping the Policy / Algorithm API, the Batch fields (`v_s`, `returns`, `adv`, `logp_old`, `act`),
the stats dataclasses and `state_dict()` (parameters and Adam moments are copied back into the
torch modules / optimizer after every update()).  Supported net: the MuJoCo actor-critic of
examples/mujoco/mujoco_ppo.py (Net[64,64] tanh, ContinuousActorProbabilistic(unbounded=True) with a
state-independent sigma, ContinuousCritic); anything else raises at construction.
"""
from __future__ import annotations

import numpy as np
import torch

from .ppo import PPOConfig, PPOEngine, flat_from_modules, flat_to_modules, TIANSHOU_ACTOR_KEYS, TIANSHOU_CRITIC_KEYS


def ppo_config_from(algorithm) -> PPOConfig:
    """Reads the reference PPO's hyper-parameters (ppo.py:126-144, a2c.py:95-113, optim.py:89-110)."""
    opt = algorithm.optim._optim
    g = opt.param_groups[0]
    if type(opt).__name__ != "Adam" or g.get("weight_decay", 0) != 0 or g.get("amsgrad", False):
        raise NotImplementedError("HipPPO supports torch.optim.Adam without weight decay / amsgrad")
    return PPOConfig(
        gamma=algorithm.gamma, gae_lambda=algorithm.gae_lambda, eps_clip=algorithm.eps_clip,
        dual_clip=algorithm.dual_clip, value_clip=algorithm.value_clip,
        advantage_normalization=algorithm.advantage_normalization,
        recompute_advantage=algorithm.recompute_adv, vf_coef=algorithm.vf_coef,
        ent_coef=algorithm.ent_coef, max_grad_norm=algorithm.optim._max_grad_norm,
        return_scaling=algorithm.return_scaling, lr=g["lr"], betas=tuple(g["betas"]), adam_eps=g["eps"])


def _check_supported(actor, critic) -> tuple[int, int]:
    sa, sc = actor.state_dict(), critic.state_dict()
    for k in TIANSHOU_ACTOR_KEYS:
        if k not in sa:
            raise NotImplementedError(f"HipPPO: unsupported actor (missing {k}); see tianshou_amd/integration.py")
    for k in TIANSHOU_CRITIC_KEYS:
        if k not in sc:
            raise NotImplementedError(f"HipPPO: unsupported critic (missing {k})")
    if len(sa) != len(TIANSHOU_ACTOR_KEYS) or len(sc) != len(TIANSHOU_CRITIC_KEYS):
        raise NotImplementedError("HipPPO: only Net[64, 64] trunks with linear heads are supported")
    w1 = sa["preprocess.model.model.0.weight"]
    if w1.shape[0] != 64 or sa["preprocess.model.model.2.weight"].shape != (64, 64):
        raise NotImplementedError("HipPPO: hidden sizes must be [64, 64]")
    if not getattr(actor, "_unbounded", False) or getattr(actor, "_c_sigma", True):
        raise NotImplementedError("HipPPO: actor must be unbounded with a state-independent sigma_param")
    return int(w1.shape[1]), int(sa["mu.model.0.weight"].shape[0])


def make_hip_ppo():
    """Returns the HipPPO class (imports tianshou lazily)."""
    from tianshou.algorithm.modelfree.a2c import A2CTrainingStats
    from tianshou.algorithm.modelfree.ppo import PPO
    from tianshou.data import SequenceSummaryStats

    class HipPPO(PPO):
        def __init__(self, *args, device="cuda", **kwargs):
            super().__init__(*args, **kwargs)
            self._hip_device = torch.device(device)
            self._hip_dims = _check_supported(self.policy.actor, self.critic)
            self._hip_engine = None
            self._hip_batch = None

        # -- engine life cycle ------------------------------------------------------------------
        def _engine(self) -> PPOEngine:
            if self._hip_engine is None:
                obs_dim, act_dim = self._hip_dims
                flat = flat_from_modules(self.policy.actor, self.critic, self._hip_device)
                self._hip_engine = PPOEngine(obs_dim, act_dim, flat, ppo_config_from(self))
                self._hip_engine.ret_rms = [float(self.ret_rms.mean), float(self.ret_rms.var),
                                            float(self.ret_rms.count)]
            return self._hip_engine

        def _sync_back(self) -> None:
            """Engine state -> nn.Parameters, torch.optim.Adam.state, ret_rms (state_dict keeps working,
            algorithm_base.py:523-543)."""
            eng = self._hip_engine
            flat_to_modules(eng.params, self.policy.actor, self.critic)
            opt = self.optim._optim
            named = {**{"a." + k: v for k, v in self.policy.actor.named_parameters()},
                     **{"c." + k: v for k, v in self.critic.named_parameters()}}
            off = 0
            for key in ["a." + k for k in TIANSHOU_ACTOR_KEYS] + ["c." + k for k in TIANSHOU_CRITIC_KEYS]:
                p = named[key]
                n = p.numel()
                st = opt.state[p]
                st["step"] = torch.tensor(float(eng.adam_step))
                st["exp_avg"] = eng.adam_m[off:off + n].reshape(p.shape).to(p.device).clone()
                st["exp_avg_sq"] = eng.adam_v[off:off + n].reshape(p.shape).to(p.device).clone()
                off += n
            self.ret_rms.mean, self.ret_rms.var, self.ret_rms.count = eng.ret_rms

        # -- hooks ------------------------------------------------------------------------------------
        def _preprocess_batch(self, batch, buffer, indices):
            eng = self._engine()
            dev = self._hip_device
            t = lambda x, dt=None: torch.as_tensor(np.ascontiguousarray(x), device=dev) if dt is None \
                else torch.as_tensor(np.ascontiguousarray(x), device=dev).to(dt)  # noqa: E731
            cut = np.nonzero(np.isin(indices, buffer.unfinished_index()))[0]      # algorithm_base.py:715
            b = eng.preprocess(t(batch.obs, torch.float32), t(batch.obs_next, torch.float32),
                               t(batch.act, torch.float32), t(batch.rew, torch.float64),
                               t(batch.terminated), t(batch.truncated), t(cut))
            self._hip_batch = b
            batch.v_s, batch.returns, batch.adv = b["v_s"], b["returns"], b["adv"]
            batch.act, batch.logp_old = b["act"], b["logp_old"]
            return batch

        def _update_with_batch(self, batch, batch_size, repeat):
            eng = self._engine()
            n = len(batch)
            perms = [np.random.permutation(n) for _ in range(repeat)]    # Batch.split, batch.py:1209
            losses, steps = eng.update(self._hip_batch, batch_size, repeat, perms)
            arr = losses.cpu().numpy().astype(np.float64)              # one D2H per update()
            self._sync_back()
            return A2CTrainingStats(
                loss=SequenceSummaryStats.from_sequence(arr[:, 0]),
                actor_loss=SequenceSummaryStats.from_sequence(arr[:, 1]),
                vf_loss=SequenceSummaryStats.from_sequence(arr[:, 2]),
                ent_loss=SequenceSummaryStats.from_sequence(arr[:, 3]),
                gradient_steps=steps,
            )

    return HipPPO
