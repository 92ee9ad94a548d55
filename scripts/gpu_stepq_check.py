"""GPU check of the feature-split step kernel (ts_ppo_q.h, TS_PPO_STEPQ=1) against the 128-sample kernel (TS_PPO_STEPQ=0)
and the CPU oracle: unclipped gradient of one minibatch step per parameter block, the four loss figures, and the
parameters after a few Adam steps -- over shapes / hyper-parameter branches -- then kernel timings (HIP events through
ts_profile) of step / reduce / Adam at 65,536 / 32,768 / 16,384 / 8,192 rows.

    python scripts/gpu_stepq_check.py [check] [time]
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_ppo as OP          # noqa: E402  (checker)
from tianshou_amd import _lib                # noqa: E402
from tianshou_amd import ppo as P            # noqa: E402

DEV = "cuda:0"


def blocks(obs_dim, act_dim):
    off = 0
    for k, shp in P.param_shapes(obs_dim, act_dim).items():
        n = int(np.prod(shp))
        yield k, off, off + n
        off += n


def make(n, obs_dim, act_dim, seed):
    rng = np.random.default_rng(seed)
    params = OP.init_params(obs_dim, act_dim, seed=seed)
    # wider heads than the 0.01-scaled init so that every branch of the loss is exercised
    params["a_wmu"] = params["a_wmu"] * 30.0
    params["a_bmu"] = torch.from_numpy(rng.normal(size=act_dim).astype(np.float32) * 0.1)
    params["c_bv"] = torch.from_numpy(rng.normal(size=1).astype(np.float32) * 0.1)
    params["a_b2"] = torch.from_numpy(rng.normal(size=64).astype(np.float32) * 0.1)
    params["c_b1"] = torch.from_numpy(rng.normal(size=64).astype(np.float32) * 0.1)
    b = dict(obs=rng.normal(size=(n, obs_dim)).astype(np.float32), act=rng.normal(size=(n, act_dim)).astype(np.float32),
             adv=rng.normal(size=n).astype(np.float32), returns=rng.normal(size=n).astype(np.float32),
             logp_old=(rng.normal(size=n) * 0.3 - 1.2 * act_dim).astype(np.float32), v_s=rng.normal(size=n).astype(np.float32))
    return params, b


def run_engine(variant, params, b, obs_dim, act_dim, kw, batch, repeat, perms):
    os.environ["TS_PPO_STEPQ"] = str(variant)
    eng = P.PPOEngine(obs_dim, act_dim, OP.flatten_params(params).to(DEV), P.PPOConfig(**kw))
    db = {k: torch.as_tensor(v, device=DEV) for k, v in b.items()}
    losses, steps, grads = eng.update(db, batch, repeat, perms, want_grad=True)
    torch.cuda.synchronize()
    return losses.cpu().numpy().astype(np.float64), grads.cpu().numpy(), eng.params.cpu().numpy()


def check():
    cases = [
        ("c2-like", 4096, 17, 6, 1024, 2, dict(eps_clip=0.2, vf_coef=0.25, ent_coef=0.0, max_grad_norm=0.5, value_clip=True,
                                                advantage_normalization=False, lr=3e-4)),
        ("ragged n, adv_norm, dual clip, entropy", 3001, 17, 6, 1000, 1, dict(eps_clip=0.2, dual_clip=3.0, vf_coef=0.5,
                                                ent_coef=0.01, max_grad_norm=None, value_clip=False,
                                                advantage_normalization=True, lr=1e-3)),
        ("a2c", 2048, 11, 3, 512, 1, dict(algo="a2c", vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5, lr=7e-4,
                                          advantage_normalization=False)),
        ("obs 3 act 1", 777, 3, 1, 256, 1, dict(eps_clip=0.2, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5, value_clip=True,
                                               advantage_normalization=True, lr=3e-4)),
        ("obs 8 act 2", 1500, 8, 2, 500, 1, dict(eps_clip=0.1, vf_coef=0.5, ent_coef=0.02, max_grad_norm=1.0, value_clip=True,
                                                advantage_normalization=False, lr=3e-4)),
        ("obs 27 act 8", 1024, 27, 8, 1024, 1, dict(eps_clip=0.2, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5, value_clip=True,
                                                   advantage_normalization=False, lr=3e-4)),
        ("obs 21 act 1 (last k-step leaves the record)", 1000, 21, 1, 500, 1, dict(eps_clip=0.2, vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5,
                                                                                  value_clip=True, advantage_normalization=False, lr=3e-4)),
        ("obs 13 act 4", 900, 13, 4, 300, 1, dict(eps_clip=0.2, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5, value_clip=False,
                                                 advantage_normalization=True, lr=3e-4)),
        ("one tile", 20, 17, 6, 20, 1, dict(eps_clip=0.2, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5, value_clip=True,
                                           advantage_normalization=False, lr=3e-4)),
        ("many tiles per workgroup", 65536 + 40, 17, 6, 65536 + 40, 1, dict(eps_clip=0.2, vf_coef=0.25, ent_coef=0.0, max_grad_norm=0.5,
                                                                           value_clip=True, advantage_normalization=False, lr=3e-4)),
        ("actor only (nets=1)", 2048, 17, 6, 512, 1, dict(algo="a2c", nets=1, vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5, lr=7e-4,
                                                          advantage_normalization=False)),
        ("critic only (nets=2)", 2048, 17, 6, 512, 1, dict(algo="a2c", nets=2, vf_coef=0.5, ent_coef=0.01, max_grad_norm=0.5, lr=7e-4,
                                                           advantage_normalization=False)),
    ]
    bad = 0
    for name, n, obs_dim, act_dim, batch, repeat, kw in cases:
        params, b = make(n, obs_dim, act_dim, seed=n)
        rng = np.random.default_rng(1)
        perms = [rng.permutation(n) for _ in range(repeat)]
        okw = {k: v for k, v in kw.items() if k != "nets"}
        st = OP.PPOState(params={k: v.clone() for k, v in params.items()})
        tb = {k: torch.from_numpy(v) for k, v in b.items()}
        lo, go = OP.update(st, OP.PPOConfig(**okw), {"obs": tb["obs"], "act": tb["act"]},
                           {k: tb[k] for k in ("adv", "returns", "logp_old", "v_s")}, batch, repeat, perms, collect_grads=True)
        go = go.numpy()
        res = {v: run_engine(v, params, b, obs_dim, act_dim, kw, batch, repeat, perms) for v in (0, 1, 2)}
        print(f"== {name}: n={n} obs={obs_dim} act={act_dim} batch={batch} steps={len(lo)}")
        nets = kw.get("nets", 0)
        for v in (0, 1, 2):
            l, g, p = res[v]
            line = []
            worst = 0.0
            for k, a, e in blocks(obs_dim, act_dim):
                if nets == 1 and k.startswith("c_") or nets == 2 and k.startswith("a_"):
                    continue
                ref = go[a:e]
                scale = max(np.abs(ref).max(), 1e-12)
                err = np.abs(g[a:e] - ref).max() / scale
                worst = max(worst, err)
                line.append(f"{k}:{err:.1e}")
            if nets == 0:
                lerr = np.abs(l - lo).max() / max(np.abs(lo).max(), 1e-12)
            else:       # the oracle has both networks; compare the live network's loss column only
                col = 1 if nets == 1 else 2
                lerr = np.abs(l[:, col] - lo[:, col]).max() / max(np.abs(lo[:, col]).max(), 1e-12)
            flag = "" if worst < 2e-4 and lerr < 2e-5 else "   <<<<<< MISMATCH"
            bad += bool(flag)
            print(f"  variant {v}: loss rel err {lerr:.1e}; grad rel err per block  " + " ".join(line) + flag)
        d01, d02 = np.abs(res[0][2] - res[1][2]).max(), np.abs(res[0][2] - res[2][2]).max()
        print(f"  params after {len(lo)} steps: max |variant0 - variant1| = {d01:.2e}, |variant0 - variant2| = {d02:.2e}")
    print("CHECK", "FAILED" if bad else "ok", f"({bad} mismatching lines)")
    return bad


def timing():
    lib = _lib.load()
    obs_dim, act_dim = 17, 6
    n_total = 1 << 20
    params, b = make(n_total, obs_dim, act_dim, seed=3)
    db = {k: torch.as_tensor(v, device=DEV) for k, v in b.items()}
    kw = dict(eps_clip=0.2, vf_coef=0.25, ent_coef=0.0, max_grad_norm=0.5, value_clip=True, advantage_normalization=False, lr=3e-4)
    g = torch.Generator(device=DEV).manual_seed(0)
    perm = torch.randperm(n_total, device=DEV, generator=g)
    print("rows      variant pairs  step_us  reduce_us  adam_us   wall_us_per_step (16 steps back to back)")
    for rows in (65536, 32768, 16384, 8192, 4096):
        for variant, pairs in ((0, 0), (1, 0), (1, 256), (2, 0), (2, 384), (2, 128), (2, 64)):
            if pairs * 32 > rows:
                continue
            os.environ["TS_PPO_STEPQ"] = str(variant)
            if pairs:
                os.environ["TS_PPO_STEPQ_PAIRS"] = str(pairs)
            else:
                os.environ.pop("TS_PPO_STEPQ_PAIRS", None)
            eng = P.PPOEngine(obs_dim, act_dim, OP.flatten_params(params).to(DEV), P.PPOConfig(**kw))
            steps = 16
            sub = {k: v[: rows * steps] for k, v in db.items()}
            pm = perm[perm < rows * steps][: rows * steps].contiguous()
            offs = [k * rows for k in range(steps + 1)]
            eng._run_steps(sub, pm, offs)            # warm-up
            torch.cuda.synchronize()
            eng._ws.profile_begin()
            eng._run_steps(sub, pm, offs)
            prof = eng._ws.profile_end()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(4):
                eng._run_steps(sub, pm, offs)
            t1.record(); torch.cuda.synchronize()
            wall = t0.elapsed_time(t1) * 1e3 / (4 * steps)
            us = lambda k: prof[k][0] / max(prof[k][1], 1) * 1e3     # noqa: E731
            print(f"{rows:8d}  {variant}       {pairs:4d}  {us('ppo_step'):7.2f}  {us('ppo_reduce'):8.2f}  {us('ppo_adam'):7.2f}   {wall:7.2f}")
    os.environ.pop("TS_PPO_STEPQ_PAIRS", None)
    os.environ.pop("TS_PPO_STEPQ", None)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    rc = 0
    if "check" in what:
        rc = check()
    if "time" in what:
        timing()
    sys.exit(1 if rc else 0)
