"""C5 workload: SAC updates/s on a synthetic Humanoid-shape replay (2^21 slots, obs f32[376],
act f32[17], twin critics + lagged copies, hidden [256, 256], auto alpha, B=4096).

    python bench.py --workload sac [--steps K] [--warmup W]        (or: python bench_sac.py)

One "step" = one SAC.update(): uniform sample -> gather (obs, act, obs_next) -> a' ~ pi(s'), min Q_old - alpha logp
-> 1-step return -> critic1, critic2, actor, alpha steps -> Polyak.  Everything device-resident; rsample() noise
from the device RNG.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

OBS, ACT, BATCH = 376, 17, 4096
PEAK_F32_MFMA_TFLOPS = 157.3
FLOP_PER_SAMPLE = 4.76e6          # algorithmic minimum of one update


def cpu_baseline(updates: int = 3, budget_s: float = 10.0):
    from oracle import oracle_sac as OS

    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    cfg = OS.SACConfig(auto_alpha=True, target_entropy=-float(ACT))
    st = OS.SACState.create(*OS.init_sac_params(OBS, ACT, 0), cfg)
    g = torch.Generator().manual_seed(0)
    obs, obs_next = torch.randn(BATCH, OBS, generator=g), torch.randn(BATCH, OBS, generator=g)
    act = torch.rand(BATCH, ACT, generator=g) * 2 - 1
    rew = torch.randn(BATCH, generator=g)
    noise = torch.randn(BATCH, ACT, generator=g)

    def one():
        ret = rew + cfg.gamma * OS.target_q(st, cfg, obs_next, noise).flatten()
        OS.update_with_batch(st, cfg, obs, act, ret, noise)

    one()
    t0, done = time.perf_counter(), 0
    while done < updates or time.perf_counter() - t0 < budget_s:         # at least `updates`, then up to ~budget_s of CPU work
        one()
        done += 1
    dt, updates = time.perf_counter() - t0, done
    return {"value": updates / dt, "unit": "updates/s", "cores": threads, "kind": "port",
            "sample": f"{updates} updates of B={BATCH} (target + 3 optimizer steps + Polyak), torch fp32 CPU oracle"}


def hook_level(updates: int = 200, slots: int = 1 << 14) -> dict:
    """The drop-in as Tianshou calls it: `HipSAC.update(buffer, 4096)` (tianshou_amd/integration.py) on a HOST replay buffer of
    the C5 shape -- `sample_indices`, `_preprocess_batch`, `_update_with_batch`, `_postprocess_batch`, statistics as Python
    floats (one device synchronisation per update, as the reference's `.item()` calls have).  The reference package is not on the
    GPU box, so the subclass is built over the stand-ins of tests/standin.py (same attribute surface,
    tests/test_standin_surface.py); hook bodies, device mirror, engine and write-back are the production code.  Two modes: the
    defaults (index-only sampling, engine noise, write-back when the torch state is read) and the reference-exact mode
    (`host_batch=True, update_noise="torch", write_back="eager"`: the reference's own `Algorithm._update` with its host copy of
    the batch, torch's host generator, five networks + four optimizers written back after every update)."""
    from torch import nn

    from tests import standin as SI
    from tianshou_amd.integration import make_hip_sac

    E = 16
    n = E * slots
    rng = np.random.default_rng(0)
    out = {}
    for mode, kw in (("default", {}), ("reference_exact", dict(host_batch=True, update_noise="torch", write_back="eager"))):
        torch.manual_seed(0)
        actor = SI.ContinuousActorProbabilistic(SI.Net(OBS, [256, 256], nn.ReLU), ACT, unbounded=True, conditioned_sigma=True)
        c1 = SI.ContinuousCritic(SI.Net(OBS + ACT, [256, 256], nn.ReLU))
        c2 = SI.ContinuousCritic(SI.Net(OBS + ACT, [256, 256], nn.ReLU))
        algo = make_hip_sac(ref=SI)(policy=SI.Policy(actor), critic=c1, critic2=c2, lr=1e-3, critic_lr=1e-3, tau=0.005, gamma=0.99,
                                    alpha=SI.AutoAlpha(-float(ACT), 0.0, 3e-4), n_step_return_horizon=1, device="cuda", **kw).to("cuda")
        buf = SI.VectorReplayBuffer(n, E, obs_shape=(OBS,), act_shape=(ACT,))
        buf.obs[:] = rng.standard_normal((n, OBS), dtype=np.float32)
        buf.obs_next[:] = rng.standard_normal((n, OBS), dtype=np.float32)
        buf.act[:] = rng.uniform(-1, 1, (n, ACT)).astype(np.float32)
        buf.rew[:] = rng.standard_normal(n, dtype=np.float32)
        buf.terminated[:] = rng.random(n) < 0.001
        buf.done[:] = buf.terminated
        for e, sb in enumerate(buf.buffers):
            sb._size, sb._insertion_idx = slots, 0
            buf._lengths[e] = slots
            buf.last_index[e] = (e + 1) * slots - 1
        algo.policy.is_within_training_step = True
        for _ in range(20):
            algo.update(buf, BATCH)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(updates):
            stats = algo.update(buf, BATCH)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        algo.hip_sync()
        torch.cuda.synchronize()
        out[mode] = {"updates_per_s": updates / dt, "ms_per_update": dt / updates * 1e3, "sync_ms_after": (time.perf_counter() - t1) * 1e3,
                     "critic1_loss": float(stats.critic1_loss)}
    out["note"] = ("HipSAC.update() over a host-filled VectorReplayBuffer stand-in (production hook code); `sync_ms_after` = one "
                   "hip_sync() (the deferred write-back of five networks and four optimizers) after the timed loop")
    return out


def run(steps: int, warmup: int, slots: int = 1 << 21, with_cpu: bool = True) -> dict:
    import bench_init as BI

    from tianshou_amd import _lib
    from tianshou_amd import sac as S
    from tianshou_amd.buffer import DeviceReplayBuffer, gather_rows

    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    obs = torch.randn(slots, OBS, generator=g, device=dev)
    obs_next = torch.randn(slots, OBS, generator=g, device=dev)
    act = torch.rand(slots, ACT, generator=g, device=dev) * 2 - 1
    rew = torch.randn(slots, generator=g, device=dev).double()
    term = torch.rand(slots, generator=g, device=dev) < 0.001
    E = 16
    T = slots // E
    offset = np.arange(E + 1, dtype=np.int64) * T
    buf = DeviceReplayBuffer(offset=offset, last_index=offset[:-1] + T - 1, lengths=np.full(E, T, np.int64),
                             insertion=np.zeros(E, np.int64), rew=rew, terminated=term,
                             truncated=torch.zeros(slots, dtype=torch.bool, device=dev), obs=obs, act=act,
                             obs_next=obs_next)
    actor, c1, c2 = BI.sac_nets(OBS, ACT, 0)
    cfg = S.SACConfig(gamma=0.99, tau=0.005, n_step=1, auto_alpha=True, target_entropy=-float(ACT), log_alpha0=0.0,
                      actor_lr=1e-3, critic_lr=1e-3, alpha_lr=3e-4)
    eng = S.SACEngine(OBS, ACT, S.actor_flat_from_torch(list(actor.values()), OBS, ACT),
                      S.critic_flat_from_torch(list(c1.values()), OBS, ACT),
                      S.critic_flat_from_torch(list(c2.values()), OBS, ACT), cfg)

    from tianshou_amd.buffer import normal_noise

    n_upd = [0]

    two_calls = bool(os.environ.get("TS_SAC_TWO_CALLS"))         # A/B: the separate entry points of rounds 2-5

    def update():
        n_upd[0] += 1
        idx = buf.sample_indices(BATCH, seed=(0x5A7, n_upd[0]))   # manager.py:216-234: sub-buffer by length, uniform inside
        if not two_calls:     # one library call (ts_sac_learn_rows): rsample() eps of a' ~ pi(s') and a ~ pi(s) drawn inside the packing
            return eng.learn_rows(buf, idx, noise_key=(0x5AC, n_upd[0]))[0]      # launch, target pass, 1-step return, update
        noise = normal_noise((2, BATCH, ACT), 0x5AC, n_upd[0], dev)
        ret = eng.preprocess(buf, idx, noise[0])                  # n_step 1: gather + _target_q + 1-step return, one sequence
        stats, _ = eng.update_with_rows(buf, idx, ret, noise[1])  # the input packing reads the rows (TS_SAC_NO_ROWS=1: gathers)
        return stats

    BI.warm_clocks(dev)
    for _ in range(warmup):
        update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        stats = update()
    t_host = time.perf_counter() - t0            # the host's share: enqueueing `steps` updates (no synchronisation inside)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    ws = _lib.default_workspace(0)
    n_prof = 10
    ws.profile_begin()
    for _ in range(n_prof):
        update()
    torch.cuda.synchronize()
    prof = ws.profile_end()
    gemm_ms = sum(prof[k][0] for k in ("conv_fwd", "conv_wgrad", "conv_dgrad")) / n_prof
    launches = sum(prof[k][1] for k in ("conv_fwd", "conv_wgrad", "conv_dgrad")) // n_prof
    tf = FLOP_PER_SAMPLE * BATCH / (gemm_ms * 1e-3) / 1e12
    # HBM bytes per GEMM launch from the TCC counters of this very command (profiles/r06_pmc_sac.json, written by
    # scripts/gpu_r6_pmc.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes), launch-weighted over the same kernels
    traffic, traffic_parts = BI.pmc_traffic(os.path.join(ROOT, "profiles", "r06_pmc_sac.json"), ("mlp3_fwd", "mlp3_bwd", "conv_wgrad_group"))
    roof = {"bound": "mfma", "kernel": "mlp3_fwd_kernel / mlp3_bwd_kernel / conv_wgrad_group_kernel (all linear-layer GEMMs of one update)",
            "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F32_MFMA_TFLOPS,
            "traffic": traffic, "traffic_by_kernel": traffic_parts,
            "traffic_source": "profiles/r06_pmc_sac.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `bench.py --workload sac`)",
            "avg_launch_us": gemm_ms * 1e3 / launches, "launches_per_update": launches,
            "gemm_us_per_update": gemm_ms * 1e3,
            "kernel_us_per_update": {k: prof[k][0] * 1e3 / n_prof for k in ("conv_fwd", "conv_wgrad", "conv_dgrad")}}
    return {
        "metric": "SAC learn() updates/sec (B=4096, obs 376, act 17, hidden 256x256, auto alpha)",
        "value": steps / dt, "unit": "updates/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C5 SAC Humanoid-shape replay: {slots} slots, obs f32[376], act f32[17], "
                               "actor 171,042 + 2 x 166,913 critic parameters, B=4096", "parallelism": "dp1"},
        "roofline": roof,
        "whole_update_mfma_frac": FLOP_PER_SAMPLE * BATCH * steps / dt / 1e12 / PEAK_F32_MFMA_TFLOPS,
        "host_enqueue_ms_per_step": t_host / steps * 1e3,
        "cpu_baseline": cpu_baseline() if with_cpu else None,
        "hook_level": hook_level() if with_cpu else None,
        "final_stats": [float(x) for x in stats.tolist()],
    }


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--slots", type=int, default=1 << 21)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a.steps, a.warmup, a.slots, not a.no_cpu_baseline)), flush=True)
