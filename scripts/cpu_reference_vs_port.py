"""CPU calibration of bench.py's `cpu_baseline` (kind "port") against the reference itself.

Runs in the AUTHORING container only (it imports the unmodified reference from /root/reference through
oracle/ref_shim.py; the GPU box has no /root/reference):

    python scripts/cpu_reference_vs_port.py [--envs 512 --steps 2048 --repeat 10 --batch 65536 --threads 32]
        -> profiles/r05_cpu_reference_vs_port.json

Times, on the same synthetic VectorReplayBuffer contents, same initial weights, same thread count, in one process:
  * reference:  tianshou PPO.update(buffer, batch_size, repeat)  (algorithm_base.py:586-631 -> a2c.py:115-153,
                ppo.py:146-224), twice: as the shim imports it (`numba.njit` = identity, the eight njit bodies
                interpreted) and with `_gae` bound to the oracle's -O3 C restatement (what a numba-compiled body costs);
  * port:       oracle_ppo.preprocess + oracle_ppo.update, the code bench.py's cpu_baseline leg times.
Both run the same 160 (default) minibatch steps; the final parameters are compared as a sanity check that both did the
same work.  TEST / MEASUREMENT INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=512)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--repeat", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--threads", type=int, default=max(1, min(32, (os.cpu_count() or 2) // 2)))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_cpu_reference_vs_port.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)

    from oracle import ref_shim

    assert ref_shim.reference_available(), "needs /root/reference (authoring container)"
    ref_shim.install()
    import gymnasium as gym
    from torch import nn
    from torch.distributions import Independent, Normal

    import tianshou.algorithm.algorithm_base as AB
    from tianshou.algorithm.modelfree.ppo import PPO
    from tianshou.algorithm.modelfree.reinforce import ProbabilisticActorPolicy
    from tianshou.algorithm.optim import AdamOptimizerFactory
    from tianshou.data import Batch, VectorReplayBuffer
    from tianshou.utils.net.common import ActorCritic, Net
    from tianshou.utils.net.continuous import ContinuousActorProbabilistic, ContinuousCritic
    from tianshou.utils.torch_utils import policy_within_training_step

    from oracle import oracle as O
    from oracle import oracle_ppo as OP

    E, T, OBS, ACT = a.envs, a.steps, 17, 6
    N = E * T
    kw = dict(eps_clip=0.2, value_clip=True, advantage_normalization=False, vf_coef=0.25, ent_coef=0.0,
              max_grad_norm=0.5, gae_lambda=0.95, gamma=0.99, return_scaling=True, max_batchsize=65536)   # mujoco_ppo.py defaults

    def build():
        torch.manual_seed(0)
        actor = ContinuousActorProbabilistic(preprocess_net=Net(state_shape=(OBS,), hidden_sizes=[64, 64], activation=nn.Tanh),
                                             action_shape=(ACT,), unbounded=True)
        critic = ContinuousCritic(preprocess_net=Net(state_shape=(OBS,), hidden_sizes=[64, 64], activation=nn.Tanh))
        torch.nn.init.constant_(actor.sigma_param, -0.5)
        for m in ActorCritic(actor, critic).modules():
            if isinstance(m, nn.Linear):
                nn.init.orthogonal_(m.weight, gain=np.sqrt(2))
                nn.init.zeros_(m.bias)
        for m in actor.mu.modules():
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.bias)
                m.weight.data.copy_(0.01 * m.weight.data)
        policy = ProbabilisticActorPolicy(actor=actor, dist_fn=lambda ls: Independent(Normal(*ls), 1), action_scaling=True,
                                          action_bound_method="clip", action_space=gym.spaces.Box(low=-1.0, high=1.0, shape=(ACT,)))
        return PPO(policy=policy, critic=critic, optim=AdamOptimizerFactory(lr=3e-4), **kw), actor, critic

    rng = np.random.default_rng(0)
    buf = VectorReplayBuffer(N, E)
    obs = rng.normal(size=(T + 1, E, OBS)).astype(np.float32)
    act = rng.normal(size=(T, E, ACT)).astype(np.float32)
    rew = rng.normal(size=(T, E)).astype(np.float32)
    term = rng.random((T, E)) < 0.005
    trunc = np.zeros((T, E), bool)
    t0 = time.perf_counter()
    for t in range(T):
        buf.add(Batch(obs=obs[t], act=act[t], rew=rew[t], terminated=term[t], truncated=trunc[t], obs_next=obs[t + 1]))
    t_fill = time.perf_counter() - t0

    def flat(actor, critic):                 # the engine's / the port's flat order (oracle/gen_golden.py _flat_from_modules)
        a, c = actor.state_dict(), critic.state_dict()
        parts = [a["preprocess.model.model.0.weight"], a["preprocess.model.model.0.bias"],
                 a["preprocess.model.model.2.weight"], a["preprocess.model.model.2.bias"],
                 a["mu.model.0.weight"], a["mu.model.0.bias"], a["sigma_param"],
                 c["preprocess.model.model.0.weight"], c["preprocess.model.model.0.bias"],
                 c["preprocess.model.model.2.weight"], c["preprocess.model.model.2.bias"],
                 c["last.model.0.weight"], c["last.model.0.bias"]]
        return torch.cat([q.detach().reshape(-1) for q in parts]).numpy().astype(np.float32)

    def run_reference(compiled_gae: bool):
        algo, actor, critic = build()
        orig = AB._gae
        if compiled_gae:
            AB._gae = lambda v_s, v_s_, rew, end_flag, gamma, lam: O._gae(v_s, v_s_, rew, end_flag, gamma, lam)
        marks = {}
        orig_pre = PPO._preprocess_batch

        def timed_pre(self, batch, buffer, indices):
            t = time.perf_counter()
            out = orig_pre(self, batch, buffer, indices)
            marks["preprocess_s"] = time.perf_counter() - t
            return out

        PPO._preprocess_batch = timed_pre
        perms.clear()
        orig_perm = np.random.permutation

        def rec_perm(n):                      # Batch.split's draws (batch.py:1209), replayed by the port below
            q = orig_perm(n)
            perms.append(np.asarray(q, np.int64))
            return q

        np.random.permutation = rec_perm
        try:
            np.random.seed(1)
            t = time.perf_counter()
            with policy_within_training_step(algo.policy):
                stats = algo.update(buffer=buf, batch_size=a.batch, repeat=a.repeat)
            total = time.perf_counter() - t
        finally:
            AB._gae = orig
            PPO._preprocess_batch = orig_pre
            np.random.permutation = orig_perm
        n_steps = int(stats.gradient_steps)
        return dict(update_s=total, gradient_steps=n_steps, **marks,
                    steps_per_s=n_steps / total), flat(actor, critic)

    perms: list[np.ndarray] = []
    O.build()
    O._gae(np.zeros(8, np.float32), np.zeros(8, np.float32), np.zeros(8), np.zeros(8, bool), 0.99, 0.95)
    ref_interp, p_ref = run_reference(False)
    ref_comp, p_ref2 = run_reference(True)

    # the port, on the same buffer contents / weights / permutation seed
    _, actor, critic = build()
    params = {k: v.clone() for k, v in OP.unflatten_params(torch.from_numpy(flat(actor, critic)), OBS, ACT).items()}
    st = OP.PPOState(params=params)
    ocfg = OP.PPOConfig(gamma=0.99, gae_lambda=0.95, eps_clip=0.2, value_clip=True, advantage_normalization=False,
                        vf_coef=0.25, ent_coef=0.0, max_grad_norm=0.5, return_scaling=True, lr=3e-4, max_batchsize=65536)
    b_obs, b_next, b_act = (torch.from_numpy(np.asarray(x, np.float32)) for x in (buf.obs, buf.obs_next, buf.act))
    b_rew = np.asarray(buf.rew, np.float64)
    b_term, b_trunc = np.asarray(buf.terminated, bool), np.asarray(buf.truncated, bool)
    idx = np.asarray(buf.sample_indices(0), np.int64)
    unf = np.asarray(buf.unfinished_index(), np.int64)
    perms = [q.copy() for q in perms]          # the reference's own minibatch orders (second run)
    assert len(perms) == a.repeat
    t = time.perf_counter()
    pre = OP.preprocess(st, ocfg, b_obs, b_next, b_act, b_rew, b_term, b_trunc, idx, unf)
    t_pre = time.perf_counter() - t
    t = time.perf_counter()
    OP.update(st, ocfg, {"obs": b_obs[idx], "act": b_act[idx]}, pre, a.batch, a.repeat, perms)
    t_upd = time.perf_counter() - t
    n_steps = a.repeat * (-(-len(idx) // a.batch))
    port = dict(update_s=t_pre + t_upd, preprocess_s=t_pre, gradient_steps=n_steps, steps_per_s=n_steps / (t_pre + t_upd))
    p_port = OP.flatten_params(st.params).numpy()

    out = {
        "what": "tianshou PPO.update() (the reference itself, imported through oracle/ref_shim.py) vs bench.py's cpu_baseline port, "
                "same buffer contents, weights, thread count, one process",
        "shape": {"envs": E, "steps": T, "transitions": N, "obs": OBS, "act": ACT, "batch": a.batch, "repeat": a.repeat},
        "threads": a.threads, "host_cpus": os.cpu_count(), "torch": torch.__version__,
        "buffer_fill_s": t_fill,
        "reference_njit_interpreted": ref_interp,
        "reference_gae_compiled": ref_comp,
        "port": port,
        "ratio_port_over_reference_compiled": port["steps_per_s"] / ref_comp["steps_per_s"],
        "ratio_port_over_reference_interpreted": port["steps_per_s"] / ref_interp["steps_per_s"],
        "max_abs_param_diff_reference_vs_port": float(np.max(np.abs(p_ref - p_port))),
        "max_abs_param_diff_reference_two_runs": float(np.max(np.abs(p_ref - p_ref2))),
        "note": "the shim replaces numba.njit by the identity (numba is not installed): `reference_njit_interpreted` runs the GAE "
                "recurrence as Python bytecode, `reference_gae_compiled` binds algorithm_base._gae to the -O3 C restatement "
                "(oracle/ts_oracle.c) and is the figure to compare the port with",
    }
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
