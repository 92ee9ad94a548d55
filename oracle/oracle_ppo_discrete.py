"""TEST INFRASTRUCTURE ONLY - CPU (torch fp32 autograd) restatement of the reference's PPO learn() path for the
CartPole-shape configuration (BASELINE.json configs[0]): a ReLU MLP trunk shared by a discrete actor and a critic.

Never imported by the product (`tianshou_amd/`).  Pinned against the UNMODIFIED reference through
tests/golden/ppo_discrete_*.npz (oracle/gen_golden.py::gen_ppo_discrete).

Follows test/discrete/test_ppo_discrete.py:88-127:
  nets      Net(obs, [h, h]) (utils/net/common.py:343-369) shared by DiscreteActor(action_shape) -- default
            softmax_output=True, so dist_fn = torch.distributions.Categorical receives PROBABILITIES -- and
            DiscreteCritic (utils/net/discrete.py:27-123); orthogonal weights, zero biases (:99-102)
  learn     the PPO restatement of oracle_ppo_cnn (a2c.py:115-153, ppo.py:146-224, algorithm_base.py:484-500),
            with this module's `MlpNet` as the network
`softmax_output=False` gives the logits variant (DiscreteActor(softmax_output=False) + dist_fn_categorical_from_logits).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch.distributions import Categorical

PARAM_ORDER = ["l1.w", "l1.b", "l2.w", "l2.b", "actor.w", "actor.b", "critic.w", "critic.b"]
TRUNK_KEYS = ["preprocess.model.model.0.weight", "preprocess.model.model.0.bias",
              "preprocess.model.model.2.weight", "preprocess.model.model.2.bias"]
HEAD_KEYS = ["last.model.0.weight", "last.model.0.bias"]


def init_params(obs_dim: int, hidden: int, n_act: int, seed: int, orthogonal: bool = True):
    """Same RNG consumption as torch.manual_seed(seed); Net; DiscreteActor; DiscreteCritic; then the orthogonal
    initialisation loop over ActorCritic(actor, critic).modules() (shared modules visited once)."""
    torch.manual_seed(seed)
    lins = [torch.nn.Linear(obs_dim, hidden), torch.nn.Linear(hidden, hidden), torch.nn.Linear(hidden, n_act),
            torch.nn.Linear(hidden, 1)]
    if orthogonal:
        for m in lins:
            torch.nn.init.orthogonal_(m.weight)
            torch.nn.init.zeros_(m.bias)
    vals = [x for lin in lins for x in (lin.weight, lin.bias)]
    return {k: v.detach().clone() for k, v in zip(PARAM_ORDER, vals)}


def flatten_params(p) -> torch.Tensor:
    return torch.cat([p[k].reshape(-1) for k in PARAM_ORDER])


def features(p, obs) -> torch.Tensor:
    x = torch.as_tensor(obs, dtype=torch.float32).flatten(1)
    x = F.relu(F.linear(x, p["l1.w"], p["l1.b"]))
    return F.relu(F.linear(x, p["l2.w"], p["l2.b"]))


class MlpNet:
    def __init__(self, softmax_output: bool = True):
        self.softmax_output = softmax_output

    def logits(self, p, obs) -> torch.Tensor:
        return F.linear(features(p, obs), p["actor.w"], p["actor.b"])

    def dist(self, p, obs) -> Categorical:
        lg = self.logits(p, obs)
        if self.softmax_output:                                # discrete.py:87-88, then Categorical(probs)
            return Categorical(F.softmax(lg, dim=-1))
        return Categorical(logits=lg)

    @staticmethod
    def critic_forward(p, obs) -> torch.Tensor:
        return F.linear(features(p, obs), p["critic.w"], p["critic.b"])
