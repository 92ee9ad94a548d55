"""Random initial parameters for the bench scripts, in the reference's (torch) tensor layouts.

The measured legs of bench_*.py build their engines from these tensors through the product's own converters
(`*_flat_from_torch`); nothing under oracle/ is involved.  Every builder returns a dict whose insertion order is the order the
converters take, so `list(d.values())` is the converter input; the keys equal the oracle's parameter names so that the
cpu_baseline legs can start the oracle from the same weights (tests/test_bench_init.py checks keys, order and shapes against
the oracle's own initialisers).

Initialisation: weights uniform(+-1 / sqrt(fan_in)) and biases uniform(+-1 / sqrt(fan_in)) like torch.nn.Linear / Conv2d;
throughput does not depend on the values.
"""
from __future__ import annotations

import numpy as np
import torch


def _gen(seed: int) -> torch.Generator:
    return torch.Generator().manual_seed(seed)


def _uniform(shape, bound: float, g: torch.Generator) -> torch.Tensor:
    return ((torch.rand(shape, generator=g) * 2 - 1) * bound).contiguous()


def _linear(d: dict, wk: str, bk: str, n_out: int, n_in: int, g: torch.Generator) -> None:
    bound = 1.0 / np.sqrt(n_in)
    d[wk], d[bk] = _uniform((n_out, n_in), bound, g), _uniform((n_out,), bound, g)


def _mlp3(keys, dims, g) -> dict:
    """keys = (w1, b1, w2, b2, w3, b3), dims = (in, h1, h2, out)."""
    d: dict = {}
    for i in range(3):
        _linear(d, keys[2 * i], keys[2 * i + 1], dims[i + 1], dims[i], g)
    return d


# ---- SAC / TD3 / DDPG / REDQ (examples/mujoco nets: Net[256, 256] + heads) ----------------------------------------------
def det_actor(obs: int, act: int, seed: int = 0, hidden: int = 256) -> dict:
    return _mlp3(("w1", "b1", "w2", "b2", "wa", "ba"), (obs, hidden, hidden, act), _gen(seed))


def q_critic(obs: int, act: int, seed: int = 0, hidden: int = 256) -> dict:
    return _mlp3(("w1", "b1", "w2", "b2", "wq", "bq"), (obs + act, hidden, hidden, 1), _gen(seed))


def sac_actor(obs: int, act: int, seed: int = 0, hidden: int = 256) -> dict:
    g = _gen(seed)
    d = _mlp3(("w1", "b1", "w2", "b2", "wmu", "bmu"), (obs, hidden, hidden, act), g)
    _linear(d, "wsig", "bsig", act, hidden, g)
    return d


def td3_nets(obs: int, act: int, seed: int = 0, twin: bool = True):
    return det_actor(obs, act, seed), q_critic(obs, act, seed + 1), q_critic(obs, act, seed + 2) if twin else None


def sac_nets(obs: int, act: int, seed: int = 0):
    return sac_actor(obs, act, seed), q_critic(obs, act, seed + 1), q_critic(obs, act, seed + 2)


def redq_ensemble(obs: int, act: int, n: int, seed: int = 0, hidden: int = 256) -> dict:
    """EnsembleLinear layout (utils/net/common.py:518): weight [E, in, out], bias [E, 1, out]."""
    g = _gen(seed)
    d: dict = {}
    for wk, bk, n_in, n_out in (("w1", "b1", obs + act, hidden), ("w2", "b2", hidden, hidden), ("wq", "bq", hidden, 1)):
        bound = 1.0 / np.sqrt(n_in)
        d[wk], d[bk] = _uniform((n, n_in, n_out), bound, g), _uniform((n, 1, n_out), bound, g)
    return d


def dsac_nets(obs: int, n_act: int, hidden: int, seed: int = 0) -> list[dict]:
    keys = ("l1.w", "l1.b", "l2.w", "l2.b", "head.w", "head.b")
    return [_mlp3(keys, (obs, hidden, hidden, n_act), _gen(seed + i)) for i in range(3)]


# ---- MuJoCo / CartPole on-policy nets --------------------------------------------------------------------------------------
def ppo_nets(obs: int, act: int, seed: int = 0, hidden: int = 64, sigma0: float = -0.5) -> dict:
    g = _gen(seed)
    d = _mlp3(("a_w1", "a_b1", "a_w2", "a_b2", "a_wmu", "a_bmu"), (obs, hidden, hidden, act), g)
    d["a_sigma"] = torch.full((act,), sigma0)
    d.update(_mlp3(("c_w1", "c_b1", "c_w2", "c_b2", "c_wv", "c_bv"), (obs, hidden, hidden, 1), g))
    return d


def ppo_discrete_net(obs: int, hidden: int, n_act: int, seed: int = 0) -> dict:
    g = _gen(seed)
    d: dict = {}
    _linear(d, "l1.w", "l1.b", hidden, obs, g)
    _linear(d, "l2.w", "l2.b", hidden, hidden, g)
    _linear(d, "actor.w", "actor.b", n_act, hidden, g)
    _linear(d, "critic.w", "critic.b", 1, hidden, g)
    return d


# ---- Atari nets (env/atari/atari_network.py) ----------------------------------------------------------------------------------
def _conv_out(c: int, h: int, w: int):
    oh, ow = (h - 8) // 4 + 1, (w - 8) // 4 + 1
    oh, ow = (oh - 4) // 2 + 1, (ow - 4) // 2 + 1
    return 64 * (oh - 2) * (ow - 2)


def _nature_trunk(c: int, g: torch.Generator) -> dict:
    d: dict = {}
    for name, oc, ic, k in (("conv1", 32, c, 8), ("conv2", 64, 32, 4), ("conv3", 64, 64, 3)):
        bound = 1.0 / np.sqrt(ic * k * k)
        d[f"{name}.w"], d[f"{name}.b"] = _uniform((oc, ic, k, k), bound, g), _uniform((oc,), bound, g)
    return d


def dqnet(c: int, h: int, w: int, n_out: int, seed: int = 0) -> dict:
    g = _gen(seed)
    d = _nature_trunk(c, g)
    _linear(d, "fc1.w", "fc1.b", 512, _conv_out(c, h, w), g)
    _linear(d, "fc2.w", "fc2.b", n_out, 512, g)
    return d


def cnn_actor_critic(c: int, h: int, w: int, n_act: int, seed: int = 0) -> dict:
    g = _gen(seed)
    d = _nature_trunk(c, g)
    _linear(d, "fc.w", "fc.b", 512, _conv_out(c, h, w), g)
    _linear(d, "actor.w", "actor.b", n_act, 512, g)
    _linear(d, "critic.w", "critic.b", 1, 512, g)
    return d


def rainbow_net(c: int, h: int, w: int, n_act: int, n_atoms: int, seed: int = 0, sigma0: float = 0.5):
    """-> (parameters, noise): NoisyLinear mu / sigma of the Q and V branches (utils/net/discrete.py NoisyLinear) and one draw
    of the factorised noise vectors."""
    g = _gen(seed)
    d = _nature_trunk(c, g)
    feat = _conv_out(c, h, w)
    noise: dict = {}
    for name, n_out, n_in in (("Q0", 512, feat), ("Q2", n_act * n_atoms, 512), ("V0", 512, feat), ("V2", n_atoms, 512)):
        bound = 1.0 / np.sqrt(n_in)
        d[f"{name}.mu_W"], d[f"{name}.sigma_W"] = _uniform((n_out, n_in), bound, g), torch.full((n_out, n_in), sigma0 * bound)
        d[f"{name}.mu_b"], d[f"{name}.sigma_b"] = _uniform((n_out,), bound, g), torch.full((n_out,), sigma0 * bound)
        for key, n in (("eps_p", n_in), ("eps_q", n_out)):
            x = torch.randn(n, generator=g)
            noise[f"{name}.{key}"] = x.sign() * x.abs().sqrt()
    return d, noise


# ---- Recurrent (utils/net/common.py:372) ------------------------------------------------------------------------------------
def recurrent_net(obs: int, hidden: int, layers: int, n_act: int, seed: int = 0) -> dict:
    g = _gen(seed)
    d: dict = {}
    bound = 1.0 / np.sqrt(hidden)
    for k in range(layers):
        d[f"nn.weight_ih_l{k}"], d[f"nn.weight_hh_l{k}"] = _uniform((4 * hidden, hidden), bound, g), _uniform((4 * hidden, hidden), bound, g)
        d[f"nn.bias_ih_l{k}"], d[f"nn.bias_hh_l{k}"] = _uniform((4 * hidden,), bound, g), _uniform((4 * hidden,), bound, g)
    _linear(d, "fc1.weight", "fc1.bias", hidden, obs, g)
    _linear(d, "fc2.weight", "fc2.bias", n_act, hidden, g)
    return d


def warm_clocks(device=None, seconds: float = 0.3) -> None:
    """Keeps the GPU busy for `seconds` before a bench's warm-up steps: a fresh process starts at idle clocks, and a
    timed region of a few tens of milliseconds that begins inside the ramp measures the ramp (one `--workload sac` run
    in five came out 3x slow before this).  Dense fp32 work only -- no state of the benchmarked engine is touched."""
    import time

    dev = torch.device("cuda") if device is None else device
    a = torch.randn(2048, 2048, device=dev)
    b = torch.randn(2048, 2048, device=dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            a = torch.mm(a, b).clamp_(-1.0, 1.0)
        torch.cuda.synchronize(dev)


def pmc_traffic(json_path: str, match: tuple, grid_threads=None):
    """`roofline.traffic` of the bench lines whose figure covers several kernels (C3: the conv kernels of a kind, C5: every
    linear-layer GEMM of an update): launch-weighted mean of (2 * FETCH_SIZE + WRITE_SIZE) KiB per launch over the kernels of
    a scripts/rocprof_pmc.py --json file whose name contains one of `match` -- per launch, like `achieved`.  FETCH_SIZE is
    doubled as MI355X_MICROARCH.md "HBM" prescribes for gfx950.  -> (bytes per launch, {kernel: [launches, bytes]}) or
    (None, {}) when the profile is absent."""
    import json

    try:
        with open(json_path) as f:
            prof = json.load(f)
    except (OSError, ValueError):
        return None, {}
    tot_b = tot_n = 0.0
    parts = {}
    for name, e in prof.items():
        if not any(m in name for m in match):
            continue
        for gk, g in (e.get("by_grid") or {"": e}).items():
            if "FETCH_SIZE" not in g or "WRITE_SIZE" not in g:
                continue
            if grid_threads is not None and gk and int(np.prod([int(x) for x in gk.split(",")])) != int(grid_threads):
                continue
            b = (2.0 * g["FETCH_SIZE"] + g["WRITE_SIZE"]) * 1024.0
            tot_b += b * g["launches"]
            tot_n += g["launches"]
            k = parts.setdefault(name, [0, 0.0])
            k[0] += int(g["launches"])
            k[1] += b * g["launches"]
    if not tot_n:
        return None, {}
    return int(tot_b / tot_n), {k: [v[0], int(v[1] / v[0])] for k, v in parts.items()}
