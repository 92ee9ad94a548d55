"""TEST INFRASTRUCTURE ONLY - CPU (torch fp32 autograd) restatement of the reference's Rainbow learn() path.

Never imported by the product (`tianshou_amd/`).  Pinned against the UNMODIFIED reference through
tests/golden/rainbow_*.npz (oracle/gen_golden.py::gen_rainbow).

Follows:
  net       RainbowNet env/atari/atari_network.py:154-208: DQNet(features_only=True) trunk; Q = NoisyLinear(F, 512) - ReLU -
            NoisyLinear(512, n_act * n_atoms); dueling V = NoisyLinear(F, 512) - ReLU - NoisyLinear(512, n_atoms);
            logits = q - q.mean(1) + v; softmax over the atoms
  NoisyLinear  utils/net/discrete.py:317-374: weight = mu_W + sigma_W * (eps_q x eps_p), bias = mu_bias + sigma_bias * eps_q
            in training mode, the mu parts alone otherwise; sample() redraws eps = sign(x) sqrt|x|, x ~ N(0, 1)
  RainbowDQN   modelfree/rainbow.py:77-101: fresh noise for the online and the lagged network, then C51's update
            (c51.py:120-160, restated in oracle_distq); the periodic hard sync (dqn.py:277-285) copies the WHOLE
            state_dict, noise included
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O
from . import oracle_distq as OQ
from . import oracle_dqn as OD

NOISY = ["Q0", "Q2", "V0", "V2"]
CONV_ORDER = ["conv1.w", "conv1.b", "conv2.w", "conv2.b", "conv3.w", "conv3.b"]
PARAM_ORDER = CONV_ORDER + [f"{L}.{t}" for L in NOISY for t in ("mu_W", "sigma_W", "mu_b", "sigma_b")]
# RainbowNet.state_dict() order of the trainable tensors (eps_p / eps_q sit behind each layer's four)
TIANSHOU_KEYS = ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias"] + \
    [f"{m}.{i}.{t}" for m, i in (("Q", 0), ("Q", 2), ("V", 0), ("V", 2)) for t in ("mu_W", "sigma_W", "mu_bias", "sigma_bias")]
NOISE_KEYS = [f"{m}.{i}.{t}" for m, i in (("Q", 0), ("Q", 2), ("V", 0), ("V", 2)) for t in ("eps_p", "eps_q")]


def feature_dim(h: int, w: int) -> int:
    oh, ow = OD.conv_out_hw(h, w)[-1]
    return 64 * oh * ow


def layer_dims(h: int, w: int, n_act: int, n_atoms: int) -> dict[str, tuple[int, int]]:
    f = feature_dim(h, w)
    return {"Q0": (f, 512), "Q2": (512, n_act * n_atoms), "V0": (f, 512), "V2": (512, n_atoms)}


def _f(n: int) -> torch.Tensor:
    x = torch.randn(n)                                          # discrete.py:357-359
    return x.sign().mul_(x.abs().sqrt_())


def init_params(c: int, h: int, w: int, n_act: int, n_atoms: int, seed: int, noisy_std: float = 0.5):
    """Same RNG consumption as torch.manual_seed(seed); RainbowNet(c, h, w, [n_act], n_atoms) -> (params, noise)."""
    torch.manual_seed(seed)
    convs = [torch.nn.Conv2d(c, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1)]
    p = {}
    for name, m in zip(["conv1", "conv2", "conv3"], convs):
        p[name + ".w"], p[name + ".b"] = m.weight.detach().clone(), m.bias.detach().clone()
    noise = {}
    for L, (fin, fout) in layer_dims(h, w, n_act, n_atoms).items():
        bound = 1 / np.sqrt(fin)
        p[L + ".mu_W"] = torch.empty(fout, fin).uniform_(-bound, bound)
        p[L + ".mu_b"] = torch.empty(fout).uniform_(-bound, bound)
        p[L + ".sigma_W"] = torch.full((fout, fin), noisy_std / np.sqrt(fin))
        p[L + ".sigma_b"] = torch.full((fout,), noisy_std / np.sqrt(fin))
        noise[L + ".eps_p"], noise[L + ".eps_q"] = _f(fin), _f(fout)
    return p, noise


def sample_noise(h: int, w: int, n_act: int, n_atoms: int) -> dict:
    """RainbowDQN._sample_noise (rainbow.py:77-91): module order Q.0, Q.2, V.0, V.2; eps_p then eps_q."""
    noise = {}
    for L, (fin, fout) in layer_dims(h, w, n_act, n_atoms).items():
        noise[L + ".eps_p"], noise[L + ".eps_q"] = _f(fin), _f(fout)
    return noise


def flatten_params(p) -> torch.Tensor:
    return torch.cat([p[k].reshape(-1) for k in PARAM_ORDER])


def features(p, obs) -> torch.Tensor:
    x = torch.as_tensor(np.asarray(obs) if not isinstance(obs, torch.Tensor) else obs, dtype=torch.float32)
    x = F.relu(F.conv2d(x, p["conv1.w"], p["conv1.b"], stride=4))
    x = F.relu(F.conv2d(x, p["conv2.w"], p["conv2.b"], stride=2))
    x = F.relu(F.conv2d(x, p["conv3.w"], p["conv3.b"], stride=1))
    return x.flatten(1)


def noisy_linear(p, noise, L: str, x: torch.Tensor) -> torch.Tensor:
    if noise is None:                                           # eval mode
        return F.linear(x, p[L + ".mu_W"], p[L + ".mu_b"])
    weight = p[L + ".mu_W"] + p[L + ".sigma_W"] * noise[L + ".eps_q"].ger(noise[L + ".eps_p"])
    bias = p[L + ".mu_b"] + p[L + ".sigma_b"] * noise[L + ".eps_q"].clone()
    return F.linear(x, weight, bias)


def dist(p, noise, obs, n_act: int, n_atoms: int) -> torch.Tensor:
    """RainbowNet.forward -> probabilities [B, n_act, n_atoms]."""
    f = features(p, obs)
    q = noisy_linear(p, noise, "Q2", F.relu(noisy_linear(p, noise, "Q0", f))).view(-1, n_act, n_atoms)
    v = noisy_linear(p, noise, "V2", F.relu(noisy_linear(p, noise, "V0", f))).view(-1, 1, n_atoms)
    logits = q - q.mean(dim=1, keepdim=True) + v
    return logits.softmax(dim=2)


class RainbowState:
    def __init__(self, params, noise, cfg: OQ.DistQConfig):
        self.dqn = OD.DQNState.create(params, cfg.dqn())
        self.noise = {k: v.clone() for k, v in noise.items()}
        self.noise_old = {k: v.clone() for k, v in noise.items()} if cfg.target_update_freq > 0 else None   # deepcopy


def next_dist(st: RainbowState, cfg: OQ.DistQConfig, obs_next, n_act: int, old_training: bool) -> torch.Tensor:
    """c51.py:124-132 with the noisy nets: greedy action of the (training-mode) online net, its distribution under the
    lagged net."""
    with torch.no_grad():
        d_online = dist(st.dqn.params, st.noise, obs_next, n_act, cfg.n_atoms)
        act = (d_online * OQ.support(cfg)).sum(2).argmax(dim=1)
        if st.dqn.params_old is not None:
            d = dist(st.dqn.params_old, st.noise_old if old_training else None, obs_next, n_act, cfg.n_atoms)
        else:
            d = d_online
        return d[torch.arange(len(act)), act, :]


def preprocess(cfg: OQ.DistQConfig, bstate: O.BufferState, indices) -> np.ndarray:
    """C51._target_q (c51.py:120-121): the support itself goes through the n-step return."""
    ret, _ = O.compute_nstep_return(bstate, indices, lambda after: OQ.support(cfg).repeat(len(after), 1).numpy(), cfg.gamma,
                                    cfg.n_step)
    return ret.astype(np.float32)


def update_with_batch(st: RainbowState, cfg: OQ.DistQConfig, obs, act, returns, obs_next, n_act: int, noise, noise_old,
                      weight=None, old_training: bool = True, collect: dict | None = None):
    """rainbow.py:93-101 + c51.py:143-160 -> (loss float, new batch.weight float32[B]).  `noise` / `noise_old`: the draws
    of this update's _sample_noise calls."""
    st.noise = {k: torch.as_tensor(v).clone() for k, v in noise.items()}
    if st.dqn.params_old is not None and noise_old is not None:
        st.noise_old = {k: torch.as_tensor(v).clone() for k, v in noise_old.items()}
    d = st.dqn
    if d.params_old is not None and d.iter % cfg.target_update_freq == 0:         # load_state_dict: noise included
        d.params_old = {k: v.clone() for k, v in d.params.items()}
        st.noise_old = {k: v.clone() for k, v in st.noise.items()}
    d.iter += 1
    ret = torch.as_tensor(np.asarray(returns), dtype=torch.float32)
    act_t = torch.as_tensor(np.asarray(act), dtype=torch.int64)
    w = 1.0 if weight is None else torch.as_tensor(np.asarray(weight), dtype=torch.float32)
    with torch.no_grad():                                                         # C51._target_dist
        nd = next_dist(st, cfg, obs_next, n_act, old_training)
        delta_z = (cfg.v_max - cfg.v_min) / (cfg.n_atoms - 1)
        ts = ret.clamp(cfg.v_min, cfg.v_max)
        tgt = ((1 - (ts.unsqueeze(1) - OQ.support(cfg).view(1, -1, 1)).abs() / delta_z).clamp(0, 1) * nd.unsqueeze(1)).sum(-1)
    p = {k: v.clone().requires_grad_(True) for k, v in d.params.items()}
    d_all = dist(p, st.noise, obs, n_act, cfg.n_atoms)
    curr = d_all[torch.arange(len(act_t)), act_t, :]
    ce = -(tgt * torch.log(curr + 1e-8)).sum(1)
    loss = (ce * w).mean()
    loss.backward()
    grads = {k: v.grad for k, v in p.items()}
    if collect is not None:
        collect.update(dist=d_all.detach().clone(), target_dist=tgt.clone(), grads={k: g.clone() for k, g in grads.items()})
    OD._adam(d, cfg.dqn(), grads)
    return float(loss.item()), ce.detach().clone()
