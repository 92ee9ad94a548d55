"""Shared replay of tests/golden/dqn_*.npz (used by the oracle pin test and the GPU parity test)."""
from __future__ import annotations

import os

import numpy as np

from oracle import oracle as O
from oracle import oracle_dqn as OD

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(tag: str):
    g = np.load(os.path.join(GOLDEN, f"dqn_{tag}.npz"))
    E, slots, steps, c, h, w, n_act, batch, n_updates, seed, per, stack = (int(x) for x in g["dims"])
    cfgd = dict(zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist()))
    cfg = OD.DQNConfig(gamma=cfgd["gamma"], n_step=int(cfgd["n_step"]),
                       target_update_freq=int(cfgd["target_update_freq"]), is_double=bool(cfgd["is_double"]),
                       huber_delta=None if cfgd["huber_delta"] < 0 else cfgd["huber_delta"], lr=cfgd["lr"])
    dims = dict(E=E, slots=slots, steps=steps, c=c, h=h, w=w, n_act=n_act, batch=batch, n_updates=n_updates,
                seed=seed, per=bool(per), stack=bool(stack))
    bstate = O.BufferState(g["buf_offset"], g["buf_last_index"], g["buf_lengths"], g["buf_insertion"],
                           g["rew"], g["terminated"], g["truncated"])
    return g, dims, cfg, bstate


def torch_order_flat(p) -> np.ndarray:
    return OD.flatten_params(p).numpy()


def load_distq(kind: str):
    """tests/golden/qrdqn.npz | c51.npz (oracle/gen_golden.py::gen_distq)."""
    from oracle import oracle_distq as OQ

    g = np.load(os.path.join(GOLDEN, "qrdqn.npz" if kind == "qr" else "c51.npz"))
    E, slots, steps, c, h, w, n_act, n_atoms, batch, n_updates, seed = (int(x) for x in g["dims"])
    cd = dict(zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist()))
    cfg = OQ.DistQConfig(kind=kind, n_atoms=n_atoms, v_min=cd["v_min"], v_max=cd["v_max"], gamma=cd["gamma"],
                         n_step=int(cd["n_step"]), target_update_freq=int(cd["target_update_freq"]), lr=cd["lr"])
    dims = dict(E=E, slots=slots, steps=steps, c=c, h=h, w=w, n_act=n_act, n_atoms=n_atoms, batch=batch,
                n_updates=n_updates, seed=seed)
    bstate = O.BufferState(g["buf_offset"], g["buf_last_index"], g["buf_lengths"], g["buf_insertion"],
                           g["rew"], g["terminated"], g["truncated"])
    return g, dims, cfg, bstate


def load_rainbow(tag: str):
    """tests/golden/rainbow_*.npz (oracle/gen_golden.py::gen_rainbow)."""
    from oracle import oracle_distq as OQ

    g = np.load(os.path.join(GOLDEN, f"rainbow_{tag}.npz"))
    E, slots, steps, c, h, w, n_act, n_atoms, batch, n_updates, seed = (int(x) for x in g["dims"])
    cd = dict(zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist()))
    cfg = OQ.DistQConfig(kind="c51", n_atoms=n_atoms, v_min=cd["v_min"], v_max=cd["v_max"], gamma=cd["gamma"],
                         n_step=int(cd["n_step"]), target_update_freq=int(cd["target_update_freq"]), lr=cd["lr"])
    dims = dict(E=E, slots=slots, steps=steps, c=c, h=h, w=w, n_act=n_act, n_atoms=n_atoms, batch=batch,
                n_updates=n_updates, seed=seed)
    bstate = O.BufferState(g["buf_offset"], g["buf_last_index"], g["buf_lengths"], g["buf_insertion"],
                           g["rew"], g["terminated"], g["truncated"])
    return g, dims, cfg, bstate


def rainbow_noise(g, u: int, old: bool = False):
    pre = f"u{u}_noise_old_" if old else f"u{u}_noise_"
    d = {k[len(pre):]: g[k] for k in g.files if k.startswith(pre) and (old or not k.startswith(f"u{u}_noise_old_"))}
    return d or None


def load_drqn(tag: str):
    """tests/golden/drqn_*.npz (oracle/gen_golden.py::gen_drqn)."""
    g = np.load(os.path.join(GOLDEN, f"drqn_{tag}.npz"))
    E, slots, steps, obs_dim, hidden, layers, n_act, stack_num, batch, n_updates, seed, per = (int(x) for x in g["dims"])
    cd = dict(zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist()))
    cfg = OD.DQNConfig(gamma=cd["gamma"], n_step=int(cd["n_step"]), target_update_freq=int(cd["target_update_freq"]),
                       is_double=bool(cd["is_double"]), huber_delta=None if cd["huber_delta"] < 0 else cd["huber_delta"],
                       lr=cd["lr"])
    dims = dict(E=E, slots=slots, steps=steps, obs_dim=obs_dim, hidden=hidden, layers=layers, n_act=n_act, stack_num=stack_num,
                batch=batch, n_updates=n_updates, seed=seed, per=bool(per))
    bstate = O.BufferState(g["buf_offset"], g["buf_last_index"], g["buf_lengths"], g["buf_insertion"],
                           g["rew"], g["terminated"], g["truncated"])
    return g, dims, cfg, bstate


def drqn_small(p: dict, layers: int) -> np.ndarray:
    """The tensors the fixtures keep in full: everything but the LSTM weight matrices (state_dict order)."""
    import torch

    from oracle import oracle_drqn as ORQ

    return torch.cat([p[k].detach().reshape(-1) for k in ORQ.param_keys(layers) if "weight_" not in k]).numpy()
