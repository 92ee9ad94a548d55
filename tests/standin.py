"""TEST INFRASTRUCTURE ONLY - minimal stand-ins for the reference classes the Hip* hook bodies touch.

/root/reference does not exist on the GPU box, so the `-m gpu` tests cannot subclass the real `PPO` / `SAC`.
These classes expose exactly the attribute surface `tianshou_amd/integration.py` reads or writes (names as in the
reference, cited per class) and nothing else: no losses, no network forward passes, no sampling logic beyond what
`Algorithm._update` / `VectorReplayBuffer` need to hand data to the hooks.  `tests/test_standin_surface.py` (CPU,
only where the reference is mounted) checks every stand-in against the real class it replaces: same state_dict keys,
same attribute names, same buffer bookkeeping after identical `add()` sequences.

Injected through the `ref=` argument of `make_hip_ppo` / `make_hip_sac` (`integration._ref`).
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn


# ------------------------------------------------------------------------------------------------ data / stats
class Batch(SimpleNamespace):
    """tianshou/data/batch.py:625 - only attribute assignment is used by the hooks."""

    def pop(self, key, default=None):
        return self.__dict__.pop(key, default)

    def __len__(self):
        for v in self.__dict__.values():
            if hasattr(v, "shape") and len(v.shape) > 0:
                return int(v.shape[0])
        return 0


@dataclass
class SequenceSummaryStats:
    """tianshou/data/stats.py:17-43 (population std)."""
    mean: float
    std: float
    max: float
    min: float

    @classmethod
    def from_sequence(cls, sequence):
        a = np.asarray(sequence, dtype=np.float64)
        return cls(mean=float(a.mean()), std=float(a.std()), max=float(a.max()), min=float(a.min()))


@dataclass
class A2CTrainingStats:
    """tianshou/algorithm/modelfree/a2c.py:23-30."""
    loss: SequenceSummaryStats
    actor_loss: SequenceSummaryStats
    vf_loss: SequenceSummaryStats
    ent_loss: SequenceSummaryStats
    gradient_steps: int = 0
    train_time: float = 0.0


@dataclass
class SACTrainingStats:
    """tianshou/algorithm/modelfree/sac.py:42-48."""
    actor_loss: float
    critic1_loss: float
    critic2_loss: float
    alpha: float | None = None
    alpha_loss: float | None = None
    train_time: float = 0.0


@dataclass
class TD3TrainingStats:
    """tianshou/algorithm/modelfree/td3.py:20-25."""
    actor_loss: float
    critic1_loss: float
    critic2_loss: float
    train_time: float = 0.0


@dataclass
class DDPGTrainingStats:
    """tianshou/algorithm/modelfree/ddpg.py:35-38."""
    actor_loss: float
    critic_loss: float
    train_time: float = 0.0


@dataclass
class SimpleLossTrainingStats:
    """tianshou/algorithm/modelfree/reinforce.py:63-66."""
    loss: float
    train_time: float = 0.0


@dataclass
class LossSequenceTrainingStats:
    """tianshou/algorithm/modelfree/reinforce.py:47-60 as the hooks build it (a single loss)."""
    loss: float
    train_time: float = 0.0


class RunningMeanStd:
    """tianshou/utils/statistics.py:81-91: the three scalars the wrappers mirror."""

    def __init__(self):
        self.mean, self.var, self.count = 0.0, 1.0, 0


# ------------------------------------------------------------------------------------------------ networks
class _MLP(nn.Module):
    """utils/net/common.py MLP: `.model` is the nn.Sequential."""

    def __init__(self, sizes, activation, linear_layer=nn.Linear, norm_layer=None, norm_args=None):
        super().__init__()
        layers = []
        for i in range(len(sizes) - 1):
            layers.append(linear_layer(sizes[i], sizes[i + 1]))
            if norm_layer is not None:                        # miniblock, common.py:25-39: Linear -> norm -> activation
                layers.append(norm_layer(sizes[i + 1], **(norm_args or {})))
            if activation is not None:
                layers.append(activation())
        self.model = nn.Sequential(*layers)


class Net(nn.Module):
    """utils/net/common.py:246-369: `.model` is an MLP -> state_dict keys `model.model.{0,2}.{weight,bias}`."""

    def __init__(self, in_dim, hidden_sizes, activation=nn.ReLU, linear_layer=nn.Linear, norm_layer=None, norm_args=None):
        super().__init__()
        self.model = _MLP([in_dim, *hidden_sizes], activation, linear_layer, norm_layer, norm_args)
        self.output_dim = hidden_sizes[-1]


class EnsembleLinear(nn.Module):
    """utils/net/common.py:518-550: `weight` [E, in, out] and `bias_weights` [E, 1, out], both U(-1/sqrt(in), 1/sqrt(in))."""

    def __init__(self, ensemble_size, in_feature, out_feature):
        super().__init__()
        k = (1.0 / in_feature) ** 0.5
        self.weight = nn.Parameter(torch.rand((ensemble_size, in_feature, out_feature)) * 2 * k - k)
        self.bias_weights = nn.Parameter(torch.rand((ensemble_size, 1, out_feature)) * 2 * k - k)


class ContinuousActorProbabilistic(nn.Module):
    """utils/net/continuous.py:172-238: preprocess + mu head (+ sigma head or sigma_param)."""

    def __init__(self, preprocess_net, action_dim, unbounded=True, conditioned_sigma=False, max_action=1.0):
        super().__init__()
        self.preprocess = preprocess_net
        self.mu = _MLP([preprocess_net.output_dim, action_dim], None)
        self._c_sigma = conditioned_sigma
        if conditioned_sigma:
            self.sigma = _MLP([preprocess_net.output_dim, action_dim], None)
        else:
            self.sigma_param = nn.Parameter(torch.zeros(action_dim, 1))
        self._unbounded = unbounded
        self.max_action = 1.0 if unbounded else max_action          # continuous.py:199-201: discarded when unbounded


class ContinuousCritic(nn.Module):
    """utils/net/continuous.py:88-169: preprocess + `last` head."""

    def __init__(self, preprocess_net, linear_layer=nn.Linear):
        super().__init__()
        self.preprocess = preprocess_net
        self.last = _MLP([preprocess_net.output_dim, 1], None, linear_layer)


def dist_fn_categorical_from_logits(logits):
    """utils/net/discrete.py: the default `dist_fn` of a discrete ProbabilisticActorPolicy (reinforce.py:200)."""
    return torch.distributions.Categorical(logits=logits)


class Box:
    """gymnasium.spaces.Box as far as Algorithm.map_action reads it: `low`, `high`, `shape`."""

    def __init__(self, low, high, shape):
        self.low = np.broadcast_to(np.asarray(low, np.float32), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, np.float32), shape).copy()
        self.shape = tuple(shape)


class Policy(nn.Module):
    """algorithm_base.py Policy: `actor`, `is_within_training_step`; `dist_fn` / `deterministic_eval` of ProbabilisticActorPolicy
    (reinforce.py:94-165); `action_space` / `action_scaling` / `action_bound_method` and `map_action` (algorithm_base.py:254-287,
    restated line by line: the Collector calls it right behind forward, collector.py:744)."""

    def __init__(self, actor, dist_fn=None, action_space=None, action_scaling=False, action_bound_method=None,
                 deterministic_eval=False):
        super().__init__()
        self.actor = actor
        self.dist_fn = dist_fn
        self.is_within_training_step = False
        self.action_space, self.action_scaling, self.action_bound_method = action_space, action_scaling, action_bound_method
        self.deterministic_eval = deterministic_eval

    def map_action(self, act):
        act = act.detach().cpu().numpy() if isinstance(act, torch.Tensor) else np.asarray(act)
        if isinstance(self.action_space, Box):
            if self.action_bound_method == "clip":
                act = np.clip(act, -1.0, 1.0)
            elif self.action_bound_method == "tanh":
                act = np.tanh(act)
            if self.action_scaling:
                assert np.min(act) >= -1.0 and np.max(act) <= 1.0
                low, high = self.action_space.low, self.action_space.high
                act = low + (high - low) * (act + 1.0) / 2.0
        return act


class EvalModeModuleWrapper(nn.Module):
    """utils/lagged_network.py:21-50: `.module` holds the lagged copy."""

    def __init__(self, module):
        super().__init__()
        self.module = module


# ------------------------------------------------------------------------------------------------ algorithm
class _Optimizer:
    """Algorithm.Optimizer (algorithm_base.py:463-507): `_optim`, `_max_grad_norm`, (load_)state_dict."""

    def __init__(self, optim, module, max_grad_norm=None):
        self._optim, self._module, self._max_grad_norm = optim, module, max_grad_norm

    def state_dict(self):
        return self._optim.state_dict()

    def load_state_dict(self, sd):
        self._optim.load_state_dict(sd)


class Algorithm(nn.Module):
    """algorithm_base.py:434-631: optimizer / scheduler bookkeeping, state_dict with the optimizers, `_update`."""
    _STATE_DICT_KEY_OPTIMIZERS = "_optimizers"

    def __init__(self, policy):
        super().__init__()
        self.policy = policy
        self.lr_schedulers = []
        self._optimizers = []

    def _create_optimizer(self, module, lr, max_grad_norm=None, lr_lambda=None, eps=1e-8, optim=None):
        # `optim` = (torch.optim class, kwargs): what an OptimizerFactory of tianshou/algorithm/optim.py:89-140 builds
        # (AdamOptimizerFactory incl. weight_decay, RMSpropOptimizerFactory); default AdamOptimizerFactory(lr, eps)
        opt = torch.optim.Adam(module.parameters(), lr=lr, eps=eps) if optim is None else optim[0](module.parameters(), lr=lr, **optim[1])
        if lr_lambda is not None:                         # optim.py:22-53 (LambdaLR)
            self.lr_schedulers.append(torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lr_lambda))
        o = _Optimizer(opt, module, max_grad_norm)
        self._optimizers.append(o)
        return o

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        d = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        d[prefix + self._STATE_DICT_KEY_OPTIMIZERS] = [o.state_dict() for o in self._optimizers]
        return d

    def load_state_dict(self, state_dict, strict=True, assign=False):
        state_dict = dict(state_dict)
        for o, sd in zip(self._optimizers, state_dict.pop(self._STATE_DICT_KEY_OPTIMIZERS)):
            o.load_state_dict(sd)
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def _postprocess_batch(self, batch, buffer, indices):
        if hasattr(buffer, "update_weight") and hasattr(batch, "weight"):
            buffer.update_weight(indices, batch.weight)

    def _update(self, sample_size, buffer, update_with_batch_fn):
        if not self.policy.is_within_training_step:
            raise RuntimeError("update() was called outside of a training step")
        batch, indices = buffer.sample(sample_size)
        batch = self._preprocess_batch(batch, buffer, indices)
        was = self.training
        try:
            self.train(True)
            stat = update_with_batch_fn(batch)
        finally:
            self.train(was)
        self._postprocess_batch(batch, buffer, indices)
        for s in self.lr_schedulers:
            s.step()
        return stat


class _ActorCritic(nn.Module):
    """utils/net/common.py:457: what `max_grad_norm` clips jointly."""

    def __init__(self, actor, critic):
        super().__init__()
        self.actor, self.critic = actor, critic


class PPO(Algorithm):
    """modelfree/ppo.py:24-144 over a2c.py:79-113: hyper-parameter attribute names as the reference stores them."""

    def __init__(self, *, policy, critic, lr=3e-4, lr_lambda=None, eps_clip=0.2, dual_clip=None, value_clip=False,
                 advantage_normalization=True, recompute_advantage=False, vf_coef=0.5, ent_coef=0.01,
                 max_grad_norm=None, gae_lambda=0.95, max_batchsize=256, gamma=0.99, return_scaling=False, optim=None):
        super().__init__(policy)
        self.critic = critic
        self.optim = self._create_optimizer(_ActorCritic(policy.actor, critic), lr, max_grad_norm, lr_lambda, optim=optim)
        self.eps_clip, self.dual_clip, self.value_clip = eps_clip, dual_clip, value_clip
        self.advantage_normalization, self.recompute_adv = advantage_normalization, recompute_advantage
        self.vf_coef, self.ent_coef, self.gae_lambda, self.gamma = vf_coef, ent_coef, gae_lambda, gamma
        self.max_batchsize, self.return_scaling = max_batchsize, return_scaling
        self.ret_rms = RunningMeanStd()
        self._eps = 1e-8

    def update(self, buffer, batch_size, repeat):
        return self._update(0, buffer, lambda batch: self._update_with_batch(batch, batch_size, repeat))


class A2C(PPO):
    """modelfree/a2c.py:163-247: no clipping attributes."""

    def __init__(self, **kw):
        super().__init__(**kw)
        for name in ("eps_clip", "dual_clip", "value_clip", "advantage_normalization", "recompute_adv"):
            delattr(self, name)


@dataclass
class NPGTrainingStats:
    """modelfree/npg.py:20-24."""
    actor_loss: SequenceSummaryStats
    vf_loss: SequenceSummaryStats
    kl: SequenceSummaryStats
    train_time: float = 0.0


@dataclass
class TRPOTrainingStats(NPGTrainingStats):
    """modelfree/trpo.py:18-20."""
    step_size: SequenceSummaryStats | None = None


class NPG(Algorithm):
    """modelfree/npg.py:27-118 over a2c.py:79-113 with optim_include_actor=False: the optimizer holds the critic only."""

    def __init__(self, *, policy, critic, lr=1e-3, optim_critic_iters=5, trust_region_size=0.5, advantage_normalization=True,
                 gae_lambda=0.95, max_batchsize=256, gamma=0.99, return_scaling=False):
        super().__init__(policy)
        self.critic = critic
        self.gae_lambda, self.max_batchsize, self.gamma, self.return_scaling = gae_lambda, max_batchsize, gamma, return_scaling
        self.optim = self._create_optimizer(critic, lr)
        self.ret_rms = RunningMeanStd()
        self._eps = 1e-8
        self.advantage_normalization, self.optim_critic_iters = advantage_normalization, optim_critic_iters
        self.trust_region_size = trust_region_size
        self._damping = 0.1

    def update(self, buffer, batch_size, repeat):
        return self._update(0, buffer, lambda batch: self._update_with_batch(batch, batch_size, repeat))


class TRPO(NPG):
    """modelfree/trpo.py:23-121: NPG's attributes without `trust_region_size`'s role, plus the line-search parameters."""

    def __init__(self, *, max_kl=0.01, backtrack_coeff=0.8, max_backtracks=10, **kw):
        super().__init__(**kw)
        self.max_backtracks, self.max_kl, self.backtrack_coeff = max_backtracks, max_kl, backtrack_coeff


class DiscountedReturnComputation:
    """modelfree/reinforce.py:249-271: `gamma`, `return_standardization`, `ret_rms`, `eps`."""

    def __init__(self, gamma=0.99, return_standardization=False):
        self.gamma, self.return_standardization = gamma, return_standardization
        self.ret_rms = RunningMeanStd()
        self.eps = 1e-8


class Reinforce(Algorithm):
    """modelfree/reinforce.py:312-347: `discounted_return_computation`, `optim` over the policy."""

    def __init__(self, *, policy, lr=1e-3, gamma=0.99, return_standardization=False, max_grad_norm=None, optim=None):
        super().__init__(policy)
        self.discounted_return_computation = DiscountedReturnComputation(gamma, return_standardization)
        self.optim = self._create_optimizer(policy, lr, max_grad_norm, optim=optim)

    def update(self, buffer, batch_size, repeat):
        return self._update(0, buffer, lambda batch: self._update_with_batch(batch, batch_size, repeat))


class FixedAlpha:
    """sac.py:161-172."""

    def __init__(self, alpha):
        self._value = alpha

    @property
    def value(self):
        return self._value


class AutoAlpha(nn.Module):
    """sac.py:175-210: `_target_entropy`, `_log_alpha`, `_optim` (a plain torch optimizer)."""

    def __init__(self, target_entropy, log_alpha, lr):
        super().__init__()
        self._target_entropy = target_entropy
        self._log_alpha = nn.Parameter(torch.tensor(float(log_alpha)))
        self._optim = torch.optim.Adam([self._log_alpha], lr=lr)

    @property
    def value(self):
        return self._log_alpha.detach().exp().item()


class SAC(Algorithm):
    """modelfree/sac.py:213-296 over ddpg.py ActorDualCriticsOffPolicyAlgorithm: attribute names as stored."""

    def __init__(self, *, policy, critic, critic2, lr=1e-3, critic_lr=None, lr_lambda=None, tau=0.005, gamma=0.99,
                 alpha=0.2, n_step_return_horizon=1):
        import copy

        super().__init__(policy)
        self.policy_optim = self._create_optimizer(policy, lr, None, lr_lambda)
        self.critic, self.critic2 = critic, critic2
        self.critic_old = EvalModeModuleWrapper(copy.deepcopy(critic))
        self.critic2_old = EvalModeModuleWrapper(copy.deepcopy(critic2))
        self.critic_optim = self._create_optimizer(critic, critic_lr or lr, None, lr_lambda)
        self.critic2_optim = self._create_optimizer(critic2, critic_lr or lr, None, lr_lambda)
        self.tau, self.gamma, self.n_step_return_horizon = tau, gamma, n_step_return_horizon
        self.alpha = FixedAlpha(alpha) if isinstance(alpha, float) else alpha

    def update(self, buffer, sample_size):
        return self._update(sample_size, buffer, lambda batch: self._update_with_batch(batch))


class DiscreteSACTrainingStats(SACTrainingStats):
    """modelfree/discrete_sac.py:23-25."""


class DiscreteSAC(SAC):
    """modelfree/discrete_sac.py:81-133 over ddpg.py ActorDualCriticsOffPolicyAlgorithm: the attributes of SAC."""


@dataclass
class REDQTrainingStats(DDPGTrainingStats):
    """modelfree/redq.py:26-31."""
    alpha: float | None = None
    alpha_loss: float | None = None


class REDQ(Algorithm):
    """modelfree/redq.py:131-246 over ddpg.py:213-264: one ensemble critic module, its lagged copy, the counters."""

    def __init__(self, *, policy, critic, lr=1e-3, critic_lr=None, ensemble_size=10, subset_size=2, tau=0.005, gamma=0.99,
                 alpha=0.2, n_step_return_horizon=1, actor_delay=20, target_mode="min"):
        import copy

        super().__init__(policy)
        self.policy_optim = self._create_optimizer(policy, lr)
        self.critic = critic
        self.critic_old = EvalModeModuleWrapper(copy.deepcopy(critic))
        self.critic_optim = self._create_optimizer(critic, critic_lr or lr)
        self.tau, self.gamma, self.n_step_return_horizon = tau, gamma, n_step_return_horizon
        self.ensemble_size, self.subset_size = ensemble_size, subset_size
        self.alpha = FixedAlpha(alpha) if isinstance(alpha, float) else alpha
        self.target_mode, self.critic_gradient_step, self.actor_delay = target_mode, 0, actor_delay
        self._last_actor_loss = 0.0

    def update(self, buffer, sample_size):
        return self._update(sample_size, buffer, lambda batch: self._update_with_batch(batch))


class ContinuousActorDeterministic(nn.Module):
    """utils/net/continuous.py:26-85: `preprocess` + `last` (MLP with one Linear), `max_action`."""

    def __init__(self, preprocess_net, act_dim, max_action=1.0):
        super().__init__()
        self.preprocess = preprocess_net
        self.last = _MLP([preprocess_net.output_dim, act_dim], None)
        self.max_action = max_action


class TD3(Algorithm):
    """modelfree/td3.py:104-188 over ddpg.py:213-264: attribute names as the reference stores them."""

    def __init__(self, *, policy, critic, critic2, lr=1e-3, critic_lr=None, tau=0.005, gamma=0.99, policy_noise=0.2,
                 update_actor_freq=2, noise_clip=0.5, n_step_return_horizon=1):
        import copy

        super().__init__(policy)
        self.policy_optim = self._create_optimizer(policy, lr)
        self.critic, self.critic2 = critic, critic2
        self.critic_old = EvalModeModuleWrapper(copy.deepcopy(critic))
        self.critic2_old = EvalModeModuleWrapper(copy.deepcopy(critic2))
        self.actor_old = EvalModeModuleWrapper(copy.deepcopy(policy.actor))
        self.critic_optim = self._create_optimizer(critic, critic_lr or lr)
        self.critic2_optim = self._create_optimizer(critic2, critic_lr or lr)
        self.tau, self.gamma, self.n_step_return_horizon = tau, gamma, n_step_return_horizon
        self.policy_noise, self.update_actor_freq, self.noise_clip = policy_noise, update_actor_freq, noise_clip
        self._cnt, self._last = 0, 0

    def update(self, buffer, sample_size):
        return self._update(sample_size, buffer, lambda batch: self._update_with_batch(batch))


class DDPG(Algorithm):
    """modelfree/ddpg.py:343-395 over :213-264: attribute names as the reference stores them."""

    def __init__(self, *, policy, critic, lr=1e-3, critic_lr=None, tau=0.005, gamma=0.99, n_step_return_horizon=1):
        import copy

        super().__init__(policy)
        self.policy_optim = self._create_optimizer(policy, lr)
        self.critic = critic
        self.critic_old = EvalModeModuleWrapper(copy.deepcopy(critic))
        self.critic_optim = self._create_optimizer(critic, critic_lr or lr)
        self.tau, self.gamma, self.n_step_return_horizon = tau, gamma, n_step_return_horizon
        self.actor_old = EvalModeModuleWrapper(copy.deepcopy(policy.actor))

    def update(self, buffer, sample_size):
        return self._update(sample_size, buffer, lambda batch: self._update_with_batch(batch))


class DQNet(nn.Module):
    """env/atari/atari_network.py:60-122 without extra layers: `net` = Sequential(Sequential(conv, ReLU, conv, ReLU, conv,
    ReLU, Flatten), Linear, ReLU, Linear) -- the nesting that yields the reference's state_dict keys."""

    def __init__(self, c, h, w, n_act):
        super().__init__()
        cnn = nn.Sequential(nn.Conv2d(c, 32, 8, 4), nn.ReLU(), nn.Conv2d(32, 64, 4, 2), nn.ReLU(), nn.Conv2d(64, 64, 3, 1),
                            nn.ReLU(), nn.Flatten())
        with torch.no_grad():
            feat = int(cnn(torch.zeros(1, c, h, w)).shape[1])
        self.net = nn.Sequential(cnn, nn.Linear(feat, 512), nn.ReLU(), nn.Linear(512, n_act))


class DiscreteQLearningPolicy(nn.Module):
    """modelfree/dqn.py:39-143: `model`, `is_within_training_step`."""

    def __init__(self, model):
        super().__init__()
        self.model = model
        self.is_within_training_step = False

    def compute_q_value(self, logits, mask):
        """dqn.py:145-151."""
        if mask is not None:
            min_value = logits.min() - logits.max() - 1.0
            logits = logits + torch.as_tensor(1 - np.asarray(mask), dtype=logits.dtype, device=logits.device) * min_value
        return logits


class DQN(Algorithm):
    """modelfree/dqn.py:190-285, 330-363: attribute names as the reference stores them."""

    def __init__(self, *, policy, lr=1e-4, gamma=0.99, n_step_return_horizon=1, target_update_freq=0, is_double=True,
                 huber_loss_delta=None, max_grad_norm=None):
        import copy

        super().__init__(policy)
        self.optim = self._create_optimizer(policy, lr, max_grad_norm)
        self.gamma, self.n_step, self.target_update_freq = gamma, n_step_return_horizon, target_update_freq
        self.is_double, self.huber_loss_delta = is_double, huber_loss_delta
        self._iter = 0
        self.model_old = EvalModeModuleWrapper(copy.deepcopy(policy.model)) if target_update_freq > 0 else None

    def update(self, buffer, sample_size):
        return self._update(sample_size, buffer, lambda batch: self._update_with_batch(batch))


class DQNetFeatures(nn.Module):
    """env/atari/atari_network.py:60-122 with features_only=True, output_dim_added_layer=512 (atari_ppo.py:106-114):
    `net` = Sequential(Sequential(conv, ReLU, conv, ReLU, conv, ReLU, Flatten), Linear, ReLU); `output_dim`."""

    def __init__(self, c, h, w, output_dim=512):
        super().__init__()
        cnn = nn.Sequential(nn.Conv2d(c, 32, 8, 4), nn.ReLU(), nn.Conv2d(32, 64, 4, 2), nn.ReLU(), nn.Conv2d(64, 64, 3, 1),
                            nn.ReLU(), nn.Flatten())
        with torch.no_grad():
            feat = int(cnn(torch.zeros(1, c, h, w)).shape[1])
        self.net = nn.Sequential(cnn, nn.Linear(feat, output_dim), nn.ReLU())
        self.output_dim = output_dim


class DiscreteActor(nn.Module):
    """utils/net/discrete.py:22-101: `preprocess`, `last` = MLP without hidden layers, `softmax_output`."""

    def __init__(self, preprocess_net, n_act, softmax_output=True):
        super().__init__()
        self.preprocess = preprocess_net
        self.last = _MLP([preprocess_net.output_dim, n_act], None)
        self.softmax_output = softmax_output


class DiscreteCritic(nn.Module):
    """utils/net/discrete.py:104-163: `preprocess`, `last` (`last_size` outputs: 1 for V(s), n_act for DiscreteSAC's Q(s, .))."""

    def __init__(self, preprocess_net, last_size=1):
        super().__init__()
        self.preprocess = preprocess_net
        self.last = _MLP([preprocess_net.output_dim, last_size], None)


class Recurrent(nn.Module):
    """utils/net/common.py:372-452: `nn` = LSTM(hidden, hidden, layer_num, batch_first), `fc1`, `fc2` in that order."""

    def __init__(self, layer_num, obs_dim, n_act, hidden):
        super().__init__()
        self.nn = nn.LSTM(input_size=hidden, hidden_size=hidden, num_layers=layer_num, batch_first=True)
        self.fc1 = nn.Linear(obs_dim, hidden)
        self.fc2 = nn.Linear(hidden, n_act)


class QRDQNet(DQNet):
    """env/atari/atari_network.py:211-235: DQNet with n_act * num_quantiles outputs (same state_dict keys)."""

    def __init__(self, c, h, w, n_act, num_quantiles):
        super().__init__(c, h, w, n_act * num_quantiles)
        self.action_num, self.num_quantiles = n_act, num_quantiles


class QRDQN(DQN):
    """modelfree/qrdqn.py:40-91: DQN's attributes plus `num_quantiles`."""

    def __init__(self, *, policy, lr=1e-4, gamma=0.99, num_quantiles=200, n_step_return_horizon=1, target_update_freq=0,
                 max_grad_norm=None):
        super().__init__(policy=policy, lr=lr, gamma=gamma, n_step_return_horizon=n_step_return_horizon,
                         target_update_freq=target_update_freq, max_grad_norm=max_grad_norm)
        self.num_quantiles = num_quantiles


class C51Net(DQNet):
    """env/atari/atari_network.py:125-151: DQNet with n_act * num_atoms outputs (same state_dict keys; the softmax of its
    forward is the engine's business)."""

    def __init__(self, c, h, w, n_act, num_atoms):
        super().__init__(c, h, w, n_act * num_atoms)
        self.action_num, self.num_atoms = n_act, num_atoms


class C51Policy(DiscreteQLearningPolicy):
    """modelfree/c51.py:16-67: `num_atoms`, `v_min`, `v_max` and the `support` parameter (requires_grad=False)."""

    def __init__(self, model, num_atoms=51, v_min=-10.0, v_max=10.0):
        super().__init__(model)
        self.num_atoms, self.v_min, self.v_max = num_atoms, v_min, v_max
        self.support = nn.Parameter(torch.linspace(v_min, v_max, num_atoms), requires_grad=False)


class C51(DQN):
    """modelfree/c51.py:70-118: the attributes of QLearningOffPolicyAlgorithm plus `delta_z`; the optimizer is built over
    the policy, whose only trainable parameters are the model's (the support does not require grad)."""

    def __init__(self, *, policy, lr=1e-4, gamma=0.99, n_step_return_horizon=1, target_update_freq=0, max_grad_norm=None):
        super().__init__(policy=policy, lr=lr, gamma=gamma, n_step_return_horizon=n_step_return_horizon,
                         target_update_freq=target_update_freq, max_grad_norm=max_grad_norm)
        self.delta_z = (policy.v_max - policy.v_min) / (policy.num_atoms - 1)


class NoisyLinear(nn.Module):
    """utils/net/discrete.py:317-374: mu / sigma parameters, the factorised noise vectors `eps_p`, `eps_q` (parameters that
    do not require grad, so they are part of state_dict()), `sample()` redraws them from torch's default generator."""

    def __init__(self, in_features, out_features, noisy_std=0.5):
        super().__init__()
        self.mu_W = nn.Parameter(torch.empty(out_features, in_features))
        self.sigma_W = nn.Parameter(torch.empty(out_features, in_features))
        self.mu_bias = nn.Parameter(torch.empty(out_features))
        self.sigma_bias = nn.Parameter(torch.empty(out_features))
        self.eps_p = nn.Parameter(torch.empty(in_features), requires_grad=False)
        self.eps_q = nn.Parameter(torch.empty(out_features), requires_grad=False)
        self.in_features, self.out_features, self.sigma = in_features, out_features, noisy_std
        bound = 1 / np.sqrt(in_features)
        self.mu_W.data.uniform_(-bound, bound)
        self.mu_bias.data.uniform_(-bound, bound)
        self.sigma_W.data.fill_(noisy_std / np.sqrt(in_features))
        self.sigma_bias.data.fill_(noisy_std / np.sqrt(in_features))
        self.sample()

    @staticmethod
    def f(x):
        x = torch.randn(x.size(0), device=x.device)
        return x.sign().mul_(x.abs().sqrt_())

    def sample(self):
        self.eps_p.copy_(self.f(self.eps_p))
        self.eps_q.copy_(self.f(self.eps_q))


class RainbowNet(nn.Module):
    """env/atari/atari_network.py:154-208 with is_dueling = is_noisy = True: `net` = the conv stack itself (features_only
    without an added layer), `Q` and `V` = Sequential(NoisyLinear, ReLU, NoisyLinear)."""

    def __init__(self, c, h, w, n_act, num_atoms, noisy_std=0.5):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(c, 32, 8, 4), nn.ReLU(), nn.Conv2d(32, 64, 4, 2), nn.ReLU(), nn.Conv2d(64, 64, 3, 1),
                                 nn.ReLU(), nn.Flatten())
        with torch.no_grad():
            feat = int(self.net(torch.zeros(1, c, h, w)).shape[1])
        self.action_num, self.num_atoms = n_act, num_atoms
        self.Q = nn.Sequential(NoisyLinear(feat, 512, noisy_std), nn.ReLU(), NoisyLinear(512, n_act * num_atoms, noisy_std))
        self.V = nn.Sequential(NoisyLinear(feat, 512, noisy_std), nn.ReLU(), NoisyLinear(512, num_atoms, noisy_std))
        self.output_dim = n_act * num_atoms


class RainbowDQN(C51):
    """modelfree/rainbow.py:18-76: C51 whose `model_old` is the bare module (not the eval-mode wrapper), plus `_sample_noise`."""

    def __init__(self, **kw):
        super().__init__(**kw)
        if self.use_target_network:
            self.model_old = self.model_old.module

    @property
    def use_target_network(self):
        return self.target_update_freq > 0

    @staticmethod
    def _sample_noise(model):
        sampled = False
        for m in model.modules():
            if isinstance(m, NoisyLinear):
                m.sample()
                sampled = True
        return sampled


# ------------------------------------------------------------------------------------------------ replay buffer
class _SubBuffer:
    """The per-environment ReplayBuffer inside a manager: `_insertion_idx`, `__len__`, `maxsize`."""

    def __init__(self, size):
        self.maxsize, self._insertion_idx, self._size = size, 0, 0

    def __len__(self):
        return self._size


class _Meta:
    def __init__(self, keys):
        self._keys = tuple(keys)

    def get_keys(self):
        return self._keys


class VectorReplayBuffer:
    """data/buffer/vecbuf.py + manager.py:23-234: `buffer_num` equal sub-buffers of ceil(total / n) slots; the
    bookkeeping (`_extend_offset`, `_lengths`, `last_index`, sub-buffer `_insertion_idx`), `add`, `reset`,
    `unfinished_index`, `sample_indices` and `sample` the wrappers rely on."""

    def __init__(self, total_size, buffer_num, *, obs_shape, act_shape, obs_dtype=np.float32, act_dtype=np.float32,
                 seed=0, stack_num=1):
        self.stack_num = stack_num
        size = int(np.ceil(total_size / buffer_num))
        self.buffer_num, self.maxsize = buffer_num, size * buffer_num
        self.buffers = [_SubBuffer(size) for _ in range(buffer_num)]
        self._offset = np.arange(buffer_num, dtype=np.int64) * size
        self._extend_offset = np.arange(buffer_num + 1, dtype=np.int64) * size
        self._lengths = np.zeros(buffer_num, dtype=np.int64)
        self.last_index = self._offset.copy()
        B = self.maxsize
        self.obs = np.zeros((B, *obs_shape), obs_dtype)
        self.obs_next = np.zeros((B, *obs_shape), obs_dtype)
        self.act = np.zeros((B, *act_shape), act_dtype)
        self.rew = np.zeros(B, np.float64)                                  # buffer_base.py:492
        self.terminated = np.zeros(B, bool)
        self.truncated = np.zeros(B, bool)
        self.done = np.zeros(B, bool)
        self._meta = _Meta(("obs", "act", "rew", "terminated", "truncated", "done", "obs_next"))
        self._rng = np.random.RandomState(seed)

    def __len__(self):
        return int(self._lengths.sum())

    def reset(self, keep_statistics=False):
        for b in self.buffers:
            b._insertion_idx = b._size = 0
        self._lengths[:] = 0
        self.last_index = self._offset.copy()

    def add(self, batch, buffer_ids=None):
        ids = np.arange(self.buffer_num) if buffer_ids is None else np.asarray(buffer_ids)
        ptrs = []
        for row, e in enumerate(ids):
            sb = self.buffers[e]
            ptr = int(self._offset[e]) + sb._insertion_idx
            sb._insertion_idx = (sb._insertion_idx + 1) % sb.maxsize
            sb._size = min(sb._size + 1, sb.maxsize)
            self._lengths[e] = sb._size
            self.last_index[e] = ptr
            for key in ("obs", "act", "rew", "terminated", "truncated", "obs_next"):
                getattr(self, key)[ptr] = getattr(batch, key)[row]
            self.done[ptr] = bool(batch.terminated[row]) or bool(batch.truncated[row])
            ptrs.append(ptr)
        return np.asarray(ptrs)

    def unfinished_index(self):
        return np.asarray([int(self.last_index[e]) for e in range(self.buffer_num)
                           if self._lengths[e] > 0 and not self.done[self.last_index[e]]], dtype=np.int64)

    def sample_indices(self, batch_size):
        if batch_size == 0:
            out = []
            for e, sb in enumerate(self.buffers):
                n, off = len(sb), int(self._offset[e])
                if n < sb.maxsize:
                    out.append(off + np.arange(n))
                else:
                    out.append(off + (sb._insertion_idx + np.arange(n)) % sb.maxsize)
            return np.concatenate(out).astype(np.int64)
        lens = self._lengths.astype(np.float64)
        sub = self._rng.choice(self.buffer_num, batch_size, p=lens / lens.sum())
        cnt = np.bincount(sub, minlength=self.buffer_num)
        return np.concatenate([int(self._offset[e]) + self._rng.choice(int(self._lengths[e]), c)
                               for e, c in enumerate(cnt) if c]).astype(np.int64)

    def sample(self, batch_size):
        idx = self.sample_indices(batch_size)
        return Batch(obs=self.obs[idx], act=self.act[idx], rew=self.rew[idx], terminated=self.terminated[idx],
                     truncated=self.truncated[idx], done=self.done[idx], obs_next=self.obs_next[idx]), idx


class PrioritizedVectorReplayBuffer(VectorReplayBuffer):
    """data/buffer/prio.py:17-107 as the hooks see it: `sample` attaches the importance weights
    (weight / min_prio) ** -beta / max (prio.py:69-79, 104-106), `update_weight(index, new_weight)` stores
    (|new_weight| + eps) ** alpha and tracks max / min of the un-exponentiated values (prio.py:81-94); new slots get
    max_prio ** alpha (prio.py:44-47).  Sampling itself is the uniform stand-in of the base class: the sum-tree descent is
    pinned elsewhere (tests/golden/segtree_per.npz)."""

    def __init__(self, *a, alpha=0.6, beta=0.4, **kw):
        super().__init__(*a, **kw)
        self._alpha, self._beta = alpha, beta
        self._max_prio = self._min_prio = 1.0
        self.prio = np.zeros(self.maxsize)
        self.__eps = np.finfo(np.float32).eps.item()
        self.weight_updates = []

    def add(self, batch, buffer_ids=None):
        ptrs = super().add(batch, buffer_ids)
        self.prio[ptrs] = self._max_prio ** self._alpha
        return ptrs

    _weight_norm = True

    def get_weight(self, index):
        """prio.py:69-79."""
        return (self.prio[index] / self._min_prio) ** (-self._beta)

    def sample(self, batch_size):
        batch, idx = super().sample(batch_size)
        w = self.get_weight(idx)
        batch.weight = w / np.max(w)
        return batch, idx

    def update_weight(self, index, new_weight):
        w = np.abs(new_weight.detach().cpu().numpy() if isinstance(new_weight, torch.Tensor) else np.asarray(new_weight)) + self.__eps
        self.prio[np.asarray(index)] = w ** self._alpha
        self._max_prio, self._min_prio = max(self._max_prio, float(w.max())), min(self._min_prio, float(w.min()))
        self.weight_updates.append((np.asarray(index).copy(), w.copy()))
