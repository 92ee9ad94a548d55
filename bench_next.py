"""Measurement of the rows SURVEY 8f adds to the hot path (N3) and of BASELINE.json configs[0], on the same contract
as bench.py: TD3 / DDPG / DiscreteSAC (MLP learners), QRDQN / C51 (NatureCNN learners) and the CartPole-shape PPO.

    python bench.py --workload td3|ddpg|redq|dsac|qrdqn|c51|rainbow|npg|trpo|ppo_discrete [--steps K] [--warmup W]   (or: python bench_next.py W)

One "step" = one reference update(): sample -> gather -> target / n-step return -> optimizer steps (-> Polyak), everything
device-resident; for ppo_discrete one update() = preprocessing + repeat x ceil(N / 64) minibatch steps (value = minibatch
gradient steps/s).  roofline: the linear / conv GEMM kernels (fp32 MFMA), achieved = algorithmic flop of one update (the
formulas below) / their measured time (HIP events on the launch stream); cpu_baseline: the oracle's restatement of the
same update on the host cores.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

import bench_init as BI

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = 157.3                    # fp32 MFMA TFLOP/s (MI355X_MICROARCH.md)
GEMM_KINDS = ("conv_fwd", "conv_wgrad", "conv_dgrad")


# ---- algorithmic flop ------------------------------------------------------------------------------------------------
def mlp_flop(dims, wgrad=False, dgrad_layers=0, first_dx_cols=0):
    """2 * MACs per sample of a dense net with layer widths `dims`: forward, + weight gradients (same again),
    + input gradients of the last `dgrad_layers` layers, + the first layer's input gradient for `first_dx_cols` columns."""
    macs = [a * b for a, b in zip(dims[:-1], dims[1:])]
    f = sum(macs)
    if wgrad:
        f += sum(macs)
    f += sum(macs[len(macs) - dgrad_layers:]) if dgrad_layers else 0
    f += first_dx_cols * dims[1]
    return 2 * f


def cnn_flop(n_out):
    """NatureCNN trunk + fc1 + a head of n_out outputs: (forward, weight-gradient, input-gradient) flop."""
    conv = [3_276_800, 2_654_208, 1_806_336]
    fc, head = 3136 * 512, 512 * n_out
    fwd = 2 * (sum(conv) + fc + head)
    return fwd, fwd, fwd - 2 * conv[0]


# ---- shared pieces ---------------------------------------------------------------------------------------------------
def _threads():
    t = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(t)
    return t


def _flat_buffer(slots, E, dev, g, **cols):
    from tianshou_amd.buffer import DeviceReplayBuffer

    T = slots // E
    offset = np.arange(E + 1, dtype=np.int64) * T
    rew = torch.randn(slots, generator=g, device=dev).double()
    term = torch.rand(slots, generator=g, device=dev) < 0.002
    return DeviceReplayBuffer(offset=offset, last_index=offset[:-1] + T - 1, lengths=np.full(E, T, np.int64),
                              insertion=np.zeros(E, np.int64), rew=rew, terminated=term,
                              truncated=torch.zeros(slots, dtype=torch.bool, device=dev), **cols)


_TICK = [0]


def _tick() -> int:
    _TICK[0] += 1
    return _TICK[0]


def _perm(n: int, dev) -> torch.Tensor:
    """Minibatch order of one repeat: the engine's keyed device permutation (ts_random_permutation, as bench.py's next_perm) --
    the device stand-in for Batch.split's np.random.permutation (batch.py:1209); a sort-based torch.randperm costs ~250 us at
    2^20 entries."""
    from tianshou_amd.buffer import random_permutation

    return random_permutation(n, 0x51ED + 0x9E3779B97F4A7C15 * _tick(), dev)


_HOST_MS = [None]
DISTQ_REPLAY_STREAM = not os.environ.get("TS_DISTQ_NO_REPLAY_STREAM")   # A/B switch (QRDQN / C51 / Rainbow), as bench_dqn's
_LAST_PROF: dict = {}


def _time(update, steps, warmup):
    import bench_init as BI
    from tianshou_amd import _lib

    BI.warm_clocks()
    for _ in range(warmup):
        update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = update()
    _HOST_MS[0] = (time.perf_counter() - t0) / steps * 1e3          # the host's share: enqueueing (no synchronisation inside)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ws = _lib.default_workspace(0)
    n_prof = 10
    ws.profile_begin()
    for _ in range(n_prof):
        update()
    torch.cuda.synchronize()
    prof = ws.profile_end()
    _LAST_PROF.clear()
    _LAST_PROF.update({k: (v[0] / n_prof, v[1] // n_prof) for k, v in prof.items()})       # every kernel kind, per update
    return dt, last, {k: (prof[k][0] / n_prof, prof[k][1] // n_prof) for k in GEMM_KINDS}


def _roofline(prof, flop_per_update, what, kinds=GEMM_KINDS, kernel=None):
    ms = sum(prof[k][0] for k in kinds)
    n = sum(prof[k][1] for k in kinds)
    tf = flop_per_update / (ms * 1e-3) / 1e12
    kernel = kernel or "linear / conv layer GEMM kernels: conv_rows / conv_wgrad(_group), fused mlp3_fwd / mlp3_bwd where the shape allows"
    return {"bound": "mfma", "kernel": f"{kernel} ({what})", "achieved": tf, "peak": PEAK,
            "unit": "TFLOP/s", "frac": tf / PEAK, "traffic": None, "avg_launch_us": ms * 1e3 / max(n, 1),
            "launches_per_update": n, "gemm_us_per_update": ms * 1e3, "algorithmic_flop_per_update": flop_per_update,
            "kernel_us_per_update": {k: prof[k][0] * 1e3 for k in kinds}}


def _line(metric, value, unit, steps, warmup, dt, workload, roof, cpu, extra=None):
    out = {"metric": metric, "value": value, "unit": unit, "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "config": {"workload": workload, "parallelism": "dp1"},
           "roofline": roof, "cpu_baseline": cpu, "host_enqueue_ms_per_step": _HOST_MS[0]}
    out.update(extra or {})
    return out


# ---- TD3 / DDPG (Humanoid shape, as C5) ------------------------------------------------------------------------------
def run_td3(steps, warmup, with_cpu, twin=True, slots=1 << 21):
    from tianshou_amd import td3 as T
    from tianshou_amd.buffer import gather_rows_multi

    OBS, ACT, B, dev = 376, 17, 4096, torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    buf = _flat_buffer(slots, 16, dev, g, obs=torch.randn(slots, OBS, generator=g, device=dev),
                       act=torch.rand(slots, ACT, generator=g, device=dev) * 2 - 1,
                       obs_next=torch.randn(slots, OBS, generator=g, device=dev))
    actor, c1, c2 = BI.td3_nets(OBS, ACT, 0, twin=twin)
    cf = lambda c: None if c is None else T.critic_flat_from_torch(list(c.values()), OBS, ACT)  # noqa: E731
    eng = T.TD3Engine(OBS, ACT, T.actor_flat_from_torch(list(actor.values()), OBS, ACT), cf(c1), cf(c2),
                      T.TD3Config(twin=twin))

    from tianshou_amd.buffer import normal_noise

    n_upd = [0]

    def update():
        idx = buf.sample_indices(B, seed=(0x5A7, _tick()))  # manager.py:216-234; draws inside the sampling kernel
        n_upd[0] += 1
        noise = normal_noise((B, ACT), 0x7D3, n_upd[0], dev) if twin else None
        ret = eng.preprocess(buf, idx, noise)
        return eng.update_with_batch(*gather_rows_multi([buf.obs, buf.act], idx), ret)[0]    # Batch.__getitem__: both keys, one launch

    dt, stats, prof = _time(update, steps, warmup)
    nc = 2 if twin else 1
    a_dims, c_dims = [OBS, 256, 256, ACT], [OBS + ACT, 256, 256, 1]
    every = 0.5 if twin else 1.0                                   # delayed actor updates (td3.py:215)
    flop = B * (mlp_flop(a_dims) + nc * mlp_flop(c_dims)                                   # target
                + nc * mlp_flop(c_dims, wgrad=True, dgrad_layers=2)                          # critic steps
                + every * (mlp_flop(a_dims, wgrad=True, dgrad_layers=2)                      # actor step ...
                           + mlp_flop(c_dims, dgrad_layers=2, first_dx_cols=ACT)))           # ... through critic 1
    cpu = None
    if with_cpu:
        from oracle import oracle_sac as OS

        cfg = OS.TD3Config(twin=twin)
        st = OS.TD3State.create(actor, c1, c2, cfg)
        gc = torch.Generator().manual_seed(0)
        obs, obs_next = torch.randn(B, OBS, generator=gc), torch.randn(B, OBS, generator=gc)
        act, rew, noise = torch.rand(B, ACT, generator=gc) * 2 - 1, torch.randn(B, generator=gc), torch.randn(B, ACT, generator=gc)
        th = _threads()

        def one():
            ret = rew + cfg.gamma * OS.td3_target_q(st, cfg, obs_next, noise).flatten()
            OS.td3_update_with_batch(st, cfg, obs, act, ret)

        one()
        t0 = time.perf_counter()
        for _ in range(40):
            one()
        cpu = {"value": 40 / (time.perf_counter() - t0), "unit": "updates/s", "cores": th, "kind": "port",
               "sample": f"40 updates of B={B} (target + critic steps + delayed actor step + Polyak), torch fp32 CPU oracle"}
    name = "TD3" if twin else "DDPG"
    return _line(f"{name} learn() updates/sec (B=4096, obs 376, act 17, hidden 256x256)", steps / dt, "updates/s", steps,
                 warmup, dt, f"{name} on the C5 Humanoid-shape replay: {slots} slots, obs f32[376], act f32[17], B=4096",
                 _roofline(prof, flop, "all linear-layer GEMMs of one update"), cpu,
                 {"final_stats": [float(x) for x in stats.tolist()]})


# ---- REDQ (Humanoid shape, as C5) ------------------------------------------------------------------------------------------
def run_redq(steps, warmup, with_cpu, slots=1 << 21):
    from tianshou_amd import redq as RQ
    from tianshou_amd import sac as S
    from tianshou_amd.buffer import gather_rows_multi

    OBS, ACT, B, E, SUB, DELAY, dev = 376, 17, 4096, 10, 2, 20, torch.device("cuda")      # the REDQ paper's ensemble / delay
    g = torch.Generator(device=dev).manual_seed(0)
    buf = _flat_buffer(slots, 16, dev, g, obs=torch.randn(slots, OBS, generator=g, device=dev),
                       act=torch.rand(slots, ACT, generator=g, device=dev) * 2 - 1,
                       obs_next=torch.randn(slots, OBS, generator=g, device=dev))
    actor, critic = BI.sac_actor(OBS, ACT, 0), BI.redq_ensemble(OBS, ACT, E, 1)
    cfg = RQ.REDQConfig(auto_alpha=True, target_entropy=-float(ACT), ensemble_size=E, subset_size=SUB, actor_delay=DELAY)
    eng = RQ.REDQEngine(OBS, ACT, S.actor_flat_from_torch(list(actor.values()), OBS, ACT),
                        RQ.ensemble_flat_from_torch(list(critic.values()), OBS, ACT), cfg)
    rng = np.random.default_rng(0)

    from tianshou_amd.buffer import normal_noise

    n_upd = [0]

    def update():
        idx = buf.sample_indices(B, seed=(0x5A7, _tick()))  # manager.py:216-234; draws inside the sampling kernel
        n_upd[0] += 1
        noise = normal_noise((2, B, ACT), 0x2ED0, n_upd[0], dev)
        ret = eng.preprocess(buf, idx, noise[0], rng.choice(E, SUB, replace=False))
        return eng.update_with_batch(*gather_rows_multi([buf.obs, buf.act], idx), ret,
                                     noise[1] if eng.will_update_actor() else None)[0]

    dt, stats, prof = _time(update, steps, warmup)
    a_dims, c_dims = [OBS, 256, 256, 2 * ACT], [OBS + ACT, 256, 256, 1]
    flop = B * (mlp_flop(a_dims) + SUB * mlp_flop(c_dims) + E * mlp_flop(c_dims, wgrad=True, dgrad_layers=2)
                + (mlp_flop(a_dims, wgrad=True, dgrad_layers=2) + E * mlp_flop(c_dims, dgrad_layers=2, first_dx_cols=ACT)) / DELAY)
    cpu = None
    if with_cpu:
        from oracle import oracle_redq as OR

        ocfg = OR.REDQConfig(auto_alpha=True, target_entropy=-float(ACT), ensemble_size=E, subset_size=SUB, actor_delay=DELAY)
        st = OR.REDQState.create(actor, critic, ocfg)
        gc = torch.Generator().manual_seed(0)
        obs, obs_next = torch.randn(B, OBS, generator=gc), torch.randn(B, OBS, generator=gc)
        act, rew, noise = torch.rand(B, ACT, generator=gc) * 2 - 1, torch.randn(B, generator=gc), torch.randn(B, ACT, generator=gc)
        th = _threads()

        def one():
            ret = rew + ocfg.gamma * OR.target_q(st, ocfg, obs_next, noise, rng.choice(E, SUB, replace=False)).flatten()
            OR.update_with_batch(st, ocfg, obs, act, ret, noise)

        one()
        t0 = time.perf_counter()
        for _ in range(10):
            one()
        cpu = {"value": 10 / (time.perf_counter() - t0), "unit": "updates/s", "cores": th, "kind": "port",
               "sample": f"10 updates of B={B} (subset target, 10-member ensemble step, Polyak), torch fp32 CPU oracle"}
    return _line("REDQ learn() updates/sec (B=4096, obs 376, act 17, 10 critics, subset 2, actor delay 20)", steps / dt,
                 "updates/s", steps, warmup, dt, f"REDQ on the C5 Humanoid-shape replay: {slots} slots, B=4096, ensemble 10",
                 _roofline(prof, flop, "all linear-layer GEMMs of one update"), cpu,
                 {"final_stats": [float(x) for x in stats.tolist()]})


# ---- DiscreteSAC ---------------------------------------------------------------------------------------------------------
def run_dsac(steps, warmup, with_cpu, slots=1 << 21):
    from tianshou_amd import dsac as DS
    from tianshou_amd.buffer import gather_rows
    from tianshou_amd.sac import SACConfig

    OBS, A, HID, B, dev = 128, 18, 256, 4096, torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    buf = _flat_buffer(slots, 16, dev, g, obs=torch.randn(slots, OBS, generator=g, device=dev),
                       act=torch.randint(0, A, (slots,), generator=g, device=dev),
                       obs_next=torch.randn(slots, OBS, generator=g, device=dev))
    nets = BI.dsac_nets(OBS, A, HID, 0)
    flats = [DS.net_flat_from_torch(list(p.values()), OBS, A, HID) for p in nets]
    te = 0.98 * float(np.log(A))
    eng = DS.DiscreteSACEngine(OBS, A, HID, *flats, SACConfig(auto_alpha=True, target_entropy=te, actor_lr=1e-4, critic_lr=1e-3))

    def update():
        idx = buf.sample_indices(B, seed=(0x5A7, _tick()))  # manager.py:216-234; draws inside the sampling kernel
        ret = eng.preprocess(buf, idx)
        return eng.update_with_batch(gather_rows(buf.obs, idx), buf.act[idx], ret)[0]

    dt, stats, prof = _time(update, steps, warmup)
    dims = [OBS, HID, HID, A]
    flop = B * (3 * mlp_flop(dims)                                       # target: actor + 2 lagged critics
                + 2 * mlp_flop(dims, wgrad=True, dgrad_layers=2)         # critic steps
                + 2 * mlp_flop(dims) + mlp_flop(dims, wgrad=True, dgrad_layers=2))   # actor step with both critics' Q
    cpu = None
    if with_cpu:
        from oracle import oracle_dsac as ODS
        from oracle import oracle_sac as OS

        cfg = OS.SACConfig(auto_alpha=True, target_entropy=te, actor_lr=1e-4, critic_lr=1e-3)
        st = OS.SACState.create(*nets, cfg)
        gc = torch.Generator().manual_seed(0)
        obs, obs_next = torch.randn(B, OBS, generator=gc), torch.randn(B, OBS, generator=gc)
        act, rew = torch.randint(0, A, (B,), generator=gc), torch.randn(B, generator=gc)
        th = _threads()

        def one():
            ODS.update_with_batch(st, cfg, obs, act, rew + cfg.gamma * ODS.target_q(st, cfg, obs_next))

        one()
        t0 = time.perf_counter()
        for _ in range(40):
            one()
        cpu = {"value": 40 / (time.perf_counter() - t0), "unit": "updates/s", "cores": th, "kind": "port",
               "sample": f"40 updates of B={B} (target + 3 optimizer steps + alpha + Polyak), torch fp32 CPU oracle"}
    return _line("DiscreteSAC learn() updates/sec (B=4096, obs 128, 18 actions, hidden 256x256, auto alpha)", steps / dt,
                 "updates/s", steps, warmup, dt, f"DiscreteSAC, {slots}-slot replay, obs f32[128], 18 actions, B=4096",
                 _roofline(prof, flop, "all linear-layer GEMMs of one update"), cpu,
                 {"final_stats": [float(x) for x in stats.tolist()]})


# ---- QRDQN / C51 (Atari shape, as C3) ----------------------------------------------------------------------------------
def run_distq(steps, warmup, with_cpu, kind, slots=1 << 20):
    import bench_dqn as BD
    from tianshou_amd import distq as Q
    from tianshou_amd import dqn as D

    C, H, W, A, B = BD.C, BD.H, BD.W, BD.N_ACT, BD.BATCH
    N = 200 if kind == "qr" else 51
    frames, act, buf, per = BD.build(slots, 16)
    p = BI.dqnet(C, H, W, A * N, 0)
    cfg = Q.DistQConfig(kind=kind, n_atoms=N, gamma=0.99, n_step=3, target_update_freq=500, lr=5e-5, v_min=-10.0, v_max=10.0)
    eng = Q.DistQEngine(C, H, W, A, Q.flat_from_torch(list(p.values()), C, H, W, A, N), cfg)
    gen = torch.Generator(device="cuda").manual_seed(1)

    draw = lambda: torch.rand(B, generator=gen, device="cuda", dtype=torch.float64)   # prio.py:65 draws  # noqa: E731
    # priority update + the next batch (draws, sum-tree descent, both gathers, C51's support returns) beside the backward pass
    replay = (D.ReplayStream(eng, buf, frames, per, C, draw, lambda i: act[i], prepare=Q.replay_prepare(eng, buf, frames, C))
              if DISTQ_REPLAY_STREAM else None)

    def update():
        if replay is None:
            idx, wt = per.sample(draw())
            ret = eng.preprocess(buf, frames, idx, C)
            obs = D.gather_obs_nhwc(frames, buf, idx, C, as_u8=True)
            obs_next = D.gather_obs_nhwc(frames, buf, buf.next(idx), C, as_u8=True) if kind == "c51" else None
            a = act[idx]
        else:
            idx, wt, a, obs, obs_next, ret = replay.take()
            if kind == "qr":
                ret, obs_next = eng.returns_from_obs_next(buf, idx, obs_next), None
        loss, prio = eng.update_with_batch(obs, a, ret, wt, obs_next_nhwc=obs_next)
        if replay is None:
            per.update_weight(idx, prio)
        else:
            replay.give(idx, prio)
        return loss

    dt, loss, prof = _time(update, steps, warmup)
    fwd, wg, dg = cnn_flop(A * N)
    flop = B * (3 * fwd + wg + dg)             # online + lagged pass on s', forward / backward on s
    cpu = None
    if with_cpu:
        from oracle import oracle_distq as OQ
        from oracle import oracle_dqn as OD

        ocfg = OQ.DistQConfig(kind=kind, n_atoms=N, n_step=3, target_update_freq=500, lr=5e-5)
        st = OD.DQNState.create(p, ocfg.dqn())
        rng = np.random.default_rng(0)
        obs = rng.integers(0, 256, size=(B, C, H, W), dtype=np.uint8)
        obs_next = rng.integers(0, 256, size=(B, C, H, W), dtype=np.uint8)
        a = rng.integers(0, A, size=B)
        ret = rng.normal(size=(B, N)).astype(np.float32)
        th = _threads()

        def one():
            if kind == "qr":
                OQ.next_dist(st, ocfg, obs_next, A)
            OQ.update_with_batch(st, ocfg, obs, a, ret, A, obs_next=obs_next)

        one()
        t0 = time.perf_counter()
        for _ in range(12):
            one()
        cpu = {"value": 12 / (time.perf_counter() - t0), "unit": "updates/s", "cores": th, "kind": "port",
               "sample": f"12 updates of B={B} (2 target forwards + fwd/bwd/Adam), torch fp32 CPU oracle"}
    name = "QRDQN" if kind == "qr" else "C51"
    return _line(f"{name} learn() updates/sec (B=512, NatureCNN, {N} {'quantiles' if kind == 'qr' else 'atoms'}, n-step 3, PER)",
                 steps / dt, "updates/s", steps, warmup, dt,
                 f"{name} on the C3 Atari-shape replay: {slots} slots of u8[84,84] frames, stack 4, 6 actions, B=512",
                 _roofline(prof, flop, "conv / linear GEMMs of one update"), cpu, {"final_loss": float(loss)})


# ---- Rainbow (Atari shape, as C3) -----------------------------------------------------------------------------------------
def run_rainbow(steps, warmup, with_cpu, slots=1 << 20):
    import bench_dqn as BD
    from tianshou_amd import distq as Q
    from tianshou_amd import dqn as D
    from tianshou_amd import rainbow as RB

    C, H, W, A, B, N = BD.C, BD.H, BD.W, BD.N_ACT, BD.BATCH, 51
    dims = (C, H, W, A, N)
    frames, act, buf, per = BD.build(slots, 16)
    p, n0 = BI.rainbow_net(*dims, 0)
    cfg = Q.DistQConfig(kind="c51", n_atoms=N, gamma=0.99, n_step=3, target_update_freq=500, lr=6.25e-5, v_min=-10.0, v_max=10.0)
    eng = RB.RainbowEngine(C, H, W, A, RB.flat_from_torch(list(p.values()), *dims),
                           RB.noise_from_torch(list(n0.values()), *dims), cfg)
    gen = torch.Generator(device="cuda").manual_seed(1)
    nn = eng.lay["noise_count"]

    def draw():                                              # f(x) = sign(x) sqrt|x| (discrete.py:357-359), on the device
        x = torch.randn(nn, generator=gen, device="cuda")
        return x.sign() * x.abs().sqrt()

    base = Q.replay_prepare(eng, buf, frames, C)
    # the two noise draws of an update (rainbow.py:97-100) need no network either: they join the batch on the replay stream
    replay = (D.ReplayStream(eng, buf, frames, per, C, lambda: torch.rand(B, generator=gen, device="cuda", dtype=torch.float64),
                             lambda i: act[i], prepare=lambda i: base(i) + (draw(), draw()))
              if DISTQ_REPLAY_STREAM else None)

    def update():
        if replay is None:
            u = torch.rand(B, generator=gen, device="cuda", dtype=torch.float64)
            idx, wt = per.sample(u)
            ret = eng.preprocess(buf, idx)
            eng.set_noise(draw(), draw())
            obs = D.gather_obs_nhwc(frames, buf, idx, C, as_u8=True)
            obs_next = D.gather_obs_nhwc(frames, buf, buf.next(idx), C, as_u8=True)
            a = act[idx]
        else:
            idx, wt, a, obs, obs_next, ret, n1, n2 = replay.take()
            eng.set_noise(n1, n2)
        loss, prio = eng.update_with_batch(obs, a, ret, obs_next, wt)
        if replay is None:
            per.update_weight(idx, prio)
        else:
            replay.give(idx, prio)
        return loss

    dt, loss, prof = _time(update, steps, warmup)
    conv = [3_276_800, 2_654_208, 1_806_336]
    heads = 2 * 3136 * 512 + 512 * A * N + 512 * N
    fwd = 2 * (sum(conv) + heads)
    flop = B * (3 * fwd + fwd + (fwd - 2 * conv[0]))
    cpu = None
    if with_cpu:
        from oracle import oracle_distq as OQ
        from oracle import oracle_rainbow as ORB

        ocfg = OQ.DistQConfig(kind="c51", n_atoms=N, n_step=3, target_update_freq=500, lr=6.25e-5)
        st = ORB.RainbowState(p, n0, ocfg)
        rng = np.random.default_rng(0)
        obs = rng.integers(0, 256, size=(B, C, H, W), dtype=np.uint8)
        obs_next = rng.integers(0, 256, size=(B, C, H, W), dtype=np.uint8)
        a = rng.integers(0, A, size=B)
        ret = rng.normal(size=(B, N)).astype(np.float32)
        th = _threads()

        def one():
            ORB.update_with_batch(st, ocfg, obs, a, ret, obs_next, A, ORB.sample_noise(H, W, A, N), ORB.sample_noise(H, W, A, N))

        one()
        t0 = time.perf_counter()
        for _ in range(8):
            one()
        cpu = {"value": 8 / (time.perf_counter() - t0), "unit": "updates/s", "cores": th, "kind": "port",
               "sample": f"8 updates of B={B} (noise draws, 2 target forwards, fwd/bwd/Adam), torch fp32 CPU oracle"}
    return _line("Rainbow learn() updates/sec (B=512, RainbowNet: noisy + dueling, 51 atoms, n-step 3, PER)", steps / dt,
                 "updates/s", steps, warmup, dt,
                 f"Rainbow on the C3 Atari-shape replay: {slots} slots of u8[84,84] frames, stack 4, 6 actions, B=512",
                 _roofline(prof, flop, "conv / linear GEMMs of one update"), cpu, {"final_loss": float(loss)})


# ---- NPG / TRPO (MuJoCo shape, as C2) ---------------------------------------------------------------------------------------
def run_natural(steps, warmup, with_cpu, algo="npg"):
    from tianshou_amd import npg as NG

    OBS, ACT, HID, E, T, MB, dev = 17, 6, 64, 512, 512, 65536, torch.device("cuda")         # 2^18 transitions, 4 minibatches
    n = E * T
    g = torch.Generator(device=dev).manual_seed(0)
    obs, obs_next = torch.randn(n, OBS, generator=g, device=dev), torch.randn(n, OBS, generator=g, device=dev)
    act = torch.randn(n, ACT, generator=g, device=dev) * 0.6
    rew = torch.randn(n, generator=g, device=dev).double()
    term = torch.rand(n, generator=g, device=dev) < 0.002
    trunc = torch.zeros(n, dtype=torch.bool, device=dev)
    cut = (torch.arange(E, device=dev) + 1) * T - 1
    p = BI.ppo_nets(OBS, ACT, 0)
    a_keys = ("a_w1", "a_b1", "a_w2", "a_b2", "a_wmu", "a_bmu", "a_sigma")
    c_keys = ("c_w1", "c_b1", "c_w2", "c_b2", "c_wv", "c_bv")
    cfg = NG.NPGConfig(algo=algo, trust_region_size=0.1, optim_critic_iters=5, lr=1e-3)
    eng = NG.NPGEngine(OBS, ACT, HID, NG.actor_flat_from_torch([p[k] for k in a_keys], OBS, HID, ACT),
                       NG.critic_flat_from_torch([p[k] for k in c_keys], OBS, HID), cfg)
    count = [0]

    def update():
        pre = eng.preprocess(obs, obs_next, act, rew, term, trunc, cut)
        stats, k = eng.update(pre, MB, 1, [_perm(n, dev)])
        count[0] = k
        return stats

    dt, stats, prof = _time(update, steps, warmup)
    k = count[0]
    a_dims, c_dims = [OBS, HID, HID, ACT], [OBS, HID, HID, 1]
    per_pass = mlp_flop(a_dims)
    # per minibatch sample: gradient (fwd + bwd), 10 (+1 TRPO) Fisher-vector products (forward-mode pass ~ 2 fwd, reverse pass),
    # candidate evaluations, 5 critic iterations (fwd + bwd)
    fvps = 11 if algo == "trpo" else 10
    evals = 10 if algo == "trpo" else 1
    flop_mb = MB * (mlp_flop(a_dims, wgrad=True, dgrad_layers=2) + fvps * (2 * per_pass + mlp_flop(a_dims, wgrad=True, dgrad_layers=2)
                                                                           - per_pass) + evals * per_pass
                    + 5 * mlp_flop(c_dims, wgrad=True, dgrad_layers=2))
    flop = k * flop_mb + n * (mlp_flop(a_dims) + 2 * mlp_flop(c_dims))
    cpu = None
    if with_cpu:
        from oracle import oracle_npg as ON
        from oracle import oracle_ppo as OP

        ocfg = ON.NPGConfig(algo=algo, trust_region_size=0.1, optim_critic_iters=5, lr=1e-3)
        st = OP.PPOState(params={kk: v.clone() for kk, v in p.items()})
        gc = torch.Generator().manual_seed(0)
        o, a = torch.randn(MB, OBS, generator=gc), torch.randn(MB, ACT, generator=gc) * 0.6
        adv, ret = torch.randn(MB, generator=gc), torch.randn(MB, generator=gc)
        with torch.no_grad():
            lp = OP.dist_of(*OP.actor_forward(p, o)).log_prob(a)
        th = _threads()
        ON.minibatch_step(st, ocfg, o[:4096], a[:4096], adv[:4096], ret[:4096], lp[:4096])
        t0 = time.perf_counter()
        ON.minibatch_step(st, ocfg, o, a, adv, ret, lp)
        cpu = {"value": 1 / (time.perf_counter() - t0), "unit": "update-steps/s", "cores": th, "kind": "port",
               "sample": f"one minibatch step of {MB} samples (gradient, 10 CG iterations by double backward, critic iterations), "
                         "torch fp32 CPU oracle"}
    name = algo.upper()
    return _line(f"{name} learn() update-steps/sec (minibatch 65536, obs 17, act 6, MLP[64,64], preprocessing incl.)",
                 steps * k / dt, "update-steps/s", steps, warmup, dt,
                 f"{name} on a C2-shape rollout: {E} envs x {T} steps = {n} transitions, minibatch {MB}, 5 critic iterations",
                 _roofline({kk: _LAST_PROF[kk] for kk in GEMM_KINDS + ("ppo_step",)}, flop,
                           "gradient, Fisher-vector products, candidate evaluations, critic steps; preprocessing passes",
                           kinds=GEMM_KINDS + ("ppo_step",),
                           kernel="npg_fvp_kernel / npg_grad_kernel / npg_eval_kernel (ts_npg_q.h: one launch per pass of the actor), "
                                  "ppo_step1_kernel (critic iterations), conv_rows GEMMs (preprocessing)"), cpu,
                 {"gradient_steps_per_update": k, "final_stats": [float(x) for x in stats[-1].tolist()]})


# ---- PPO, CartPole shape (BASELINE.json configs[0]) ------------------------------------------------------------------------
def run_ppo_discrete(steps, warmup, with_cpu):
    from tianshou_amd import ppo_discrete as PD
    from tianshou_amd.ppo import PPOConfig

    OBS, HID, A, E, T, BS, REPEAT, dev = 4, 64, 2, 20, 100, 64, 10, torch.device("cuda")   # test_ppo_discrete.py defaults
    n = E * T
    g = torch.Generator(device=dev).manual_seed(0)
    buf = _flat_buffer(n, E, dev, g, obs=torch.randn(n, OBS, generator=g, device=dev),
                       act=torch.randint(0, A, (n,), generator=g, device=dev),
                       obs_next=torch.randn(n, OBS, generator=g, device=dev))
    p = BI.ppo_discrete_net(OBS, HID, A, 1626)
    kw = dict(gamma=0.99, gae_lambda=0.95, eps_clip=0.2, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5, value_clip=False,
              advantage_normalization=False, return_scaling=False, lr=3e-4)
    eng = PD.DiscretePPOEngine(OBS, HID, A, PD.flat_from_torch(list(p.values()), OBS, HID, A), PPOConfig(**kw))
    count = [0]

    def update():
        pre = eng.preprocess(buf)
        perms = [_perm(n, dev) for _ in range(REPEAT)]
        losses, k = eng.update(buf, pre, BS, REPEAT, perms)
        count[0] = k
        return losses

    dt, losses, prof = _time(update, steps, warmup)
    k = count[0]
    dims = [OBS, HID, HID, A + 1]
    flop = 2 * n * mlp_flop(dims) + k * BS * mlp_flop(dims, wgrad=True, dgrad_layers=2)
    # the one-launch update kernel alone (ts_mlp_ppo_update: every minibatch step of the update in one persistent workgroup)
    pre = eng.preprocess(buf)
    perms = [_perm(n, dev) for _ in range(REPEAT)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    eng.update(buf, pre, BS, REPEAT, perms)
    e1.record()
    torch.cuda.synchronize()
    upd_us = e0.elapsed_time(e1) * 1e3
    cpu = None
    if with_cpu:
        from oracle import oracle_ppo as OP
        from oracle import oracle_ppo_cnn as OC
        from oracle import oracle_ppo_discrete as OD

        cfg = OP.PPOConfig(**kw)
        st = OP.PPOState(params={kk: v.clone() for kk, v in p.items()})
        net = OD.MlpNet(True)
        cb = {kk: (v.cpu().numpy() if torch.is_tensor(v) else v) for kk, v in
              dict(obs=buf.obs, obs_next=buf.obs_next, act=buf.act, rew=buf.rew, term=buf.terminated).items()}
        th = _threads()
        t0 = time.perf_counter()
        idx = np.arange(n)
        unf = np.arange(E) * T + T - 1
        pre = OC.preprocess(st, cfg, cb["obs"], cb["obs_next"], cb["act"], cb["rew"], cb["term"], np.zeros(n, bool), idx, unf,
                            net=net)
        OC.update(st, cfg, cb["obs"], cb["act"], pre, BS, REPEAT, [np.random.permutation(n) for _ in range(REPEAT)], net=net)
        cpu = {"value": k / (time.perf_counter() - t0), "unit": "update-steps/s", "cores": th, "kind": "port",
               "sample": f"one update(): preprocessing of {n} transitions + {k} minibatch steps of 64, torch fp32 CPU oracle"}
    return _line("PPO learn() update-steps/sec, CartPole shape (obs 4, MLP[64,64], minibatch 64, preprocessing incl.)",
                 steps * k / dt, "update-steps/s", steps, warmup, dt,
                 f"BASELINE configs[0] shape: {E} envs x {T} steps = {n} transitions, obs f32[4], 2 actions, repeat {REPEAT}",
                 _roofline(prof, flop, "linear-layer GEMMs of the preprocessing passes; the gradient steps are the one-launch kernel, see `update_launch`"), cpu,
                 {"gradient_steps_per_update": k, "final_loss": float(losses[-1, 0]),
                  "update_launch": {"kernel": "mlp_ppo_update_small_kernel (one workgroup, all steps)", "us": upd_us,
                                    "us_per_gradient_step": upd_us / max(k, 1),
                                    "algorithmic_flop": k * BS * mlp_flop(dims, wgrad=True, dgrad_layers=2),
                                    "note": "events around DiscretePPOEngine.update(): row gather + ts_mlp_ppo_update; one CU of 256 -- "
                                            "a latency figure, not a roofline fraction of the chip"}})


def run_reinforce(steps, warmup, with_cpu):
    from tianshou_amd import npg as NG
    from tianshou_amd import reinforce as RF

    OBS, ACT, HID, E, T, MB, dev = 17, 6, 64, 512, 2048, 65536, torch.device("cuda")        # the C2 rollout: 2^20 transitions
    n = E * T
    g = torch.Generator(device=dev).manual_seed(0)
    obs = torch.randn(n, OBS, generator=g, device=dev)
    act = torch.randn(n, ACT, generator=g, device=dev) * 0.6
    rew = torch.randn(n, generator=g, device=dev).double()
    term = torch.rand(n, generator=g, device=dev) < 0.002
    trunc = torch.zeros(n, dtype=torch.bool, device=dev)
    cut = (torch.arange(E, device=dev) + 1) * T - 1
    p = BI.ppo_nets(OBS, ACT, 0)
    a_keys = ("a_w1", "a_b1", "a_w2", "a_b2", "a_wmu", "a_bmu", "a_sigma")
    eng = RF.ReinforceEngine(OBS, ACT, HID, NG.actor_flat_from_torch([p[k] for k in a_keys], OBS, HID, ACT),
                             RF.ReinforceConfig(return_standardization=True, lr=1e-3))
    count = [0]

    def update():
        ret = eng.preprocess(rew, term, trunc, cut)
        losses, k = eng.update(obs, act, ret, MB, 1, [_perm(n, dev)])
        count[0] = k
        return losses

    dt, losses, prof = _time(update, steps, warmup)
    k = count[0]
    flop = k * MB * mlp_flop([OBS, HID, HID, ACT], wgrad=True, dgrad_layers=2)
    cpu = None
    if with_cpu:
        from oracle import oracle_ppo as OP
        from oracle import oracle_reinforce as OR

        st = OP.PPOState(params={kk: p[kk].clone() for kk in OR.ACTOR_KEYS})
        gc = torch.Generator().manual_seed(0)
        o, a, r = torch.randn(4 * MB, OBS, generator=gc), torch.randn(4 * MB, ACT, generator=gc) * 0.6, torch.randn(4 * MB, generator=gc)
        ocfg = OR.ReinforceConfig(lr=1e-3)
        OR.update(st, ocfg, o[:4096], a[:4096], r[:4096], None, 1, [np.arange(4096)])
        t0 = time.perf_counter()
        OR.update(st, ocfg, o, a, r, MB, 1, [np.arange(4 * MB)])
        cpu = {"value": 4 / (time.perf_counter() - t0), "unit": "update-steps/s", "cores": _threads(), "kind": "port",
               "sample": f"4 minibatch steps of {MB} samples, torch fp32 CPU oracle (returns precomputed)"}
    if eng.fused_supported():       # the minibatch loop runs on ts_ppo.hip's fused step kernel (A2C actor loss, zero critic beside it)
        ms, launches = _LAST_PROF["ppo_step"]
        tf = flop / (ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "ppo_step2_kernel (algo a2c, adv := returns, vf_coef 0; the launch also evaluates a zero "
                                           "critic, which the algorithmic flops do not count)",
                "achieved": tf, "peak": PEAK, "unit": "TFLOP/s", "frac": tf / PEAK, "traffic": None,
                "avg_launch_us": ms * 1e3 / max(launches, 1), "launches_per_update": launches, "algorithmic_flop_per_update": flop}
    else:
        roof = _roofline(prof, flop, "linear-layer GEMMs of the policy-gradient steps")
    return _line("Reinforce learn() update-steps/sec (minibatch 65536, obs 17, act 6, MLP[64,64], preprocessing incl.)",
                 steps * k / dt, "update-steps/s", steps, warmup, dt,
                 f"Reinforce on the C2 rollout: {E} envs x {T} steps = {n} transitions, minibatch {MB}, return standardisation",
                 roof, cpu, {"gradient_steps_per_update": k, "final_loss": float(losses[-1]),
                             "path": "fused step kernel" if eng.fused_supported() else "per-layer GEMMs"})


def run_drqn(steps, warmup, with_cpu, slots=20000):
    """test/discrete/test_drqn.py's learner: Recurrent(2 LSTM layers of 128) on CartPole observations, stack_num 4, batch 128,
    n-step 3, double-Q with a lagged network, 20000-slot buffer of 16 envs without obs_next."""
    from tianshou_amd import dqn as D
    from tianshou_amd import drqn as R

    OBS, H, L, A, T, B, dev = 4, 128, 2, 2, 4, 128, torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    buf = _flat_buffer(slots, 16, dev, g, obs=torch.randn(slots, OBS, generator=g, device=dev),
                       act=torch.randint(0, A, (slots,), generator=g, device=dev))
    p = BI.recurrent_net(OBS, H, L, A, 0)
    kw = dict(gamma=0.95, n_step=3, target_update_freq=320, is_double=True, lr=1e-3)
    eng = R.RecurrentDQNEngine(OBS, H, L, A, R.flat_from_torch(list(p.values()), OBS, H, L, A), D.DQNConfig(**kw))

    draw = lambda: buf.sample_indices(B, seed=(0x5A7, _tick()))  # manager.py:216-234; draws inside the sampling kernel  # noqa: E731
    prefetch = not os.environ.get("TS_DRQN_NO_PREFETCH")
    # uniform buffer: the next batch (indices, both stacked gathers, actions, n-step coefficients) depends on nothing of the
    # update and is prepared on a second stream beside it
    one_call = not os.environ.get("TS_DRQN_NO_LEARN_STEP")      # sample + preprocess + update as ONE library call (default)
    replay = (D.ReplayStream(eng, buf, buf.obs, None, T, draw, None, prepare=R.replay_prepare(eng, buf, buf.obs, T, buf.act))
              if not one_call and not os.environ.get("TS_DRQN_NO_REPLAY_STREAM") else None)

    def update():
        if one_call:
            return eng.learn_step(buf, buf.obs, buf.act, B, T, (0x5A7, _tick()))[0]
        if replay is None:
            idx = draw()
            # batch.obs first: its forward pass runs on a side stream beside the two s_{t+n} passes of _target_q
            obs, ret = eng.preprocess_with_obs(buf, buf.obs, idx, T, prefetch=prefetch)
            return eng.update_with_batch(obs, buf.act[idx], ret)[0]
        idx, _, _, pair, coef = replay.take()
        obs, ret = eng.preprocess_with_obs(buf, buf.obs, idx, T, prefetch=prefetch, pair=pair, coef=coef)
        loss = eng.update_with_batch(obs, pair[2], ret)[0]
        replay.give(idx, None)
        return loss

    dt, loss, prof = _time(update, steps, warmup)
    fwd = 2 * (T * OBS * H + L * T * 2 * H * 4 * H + H * A)          # per sample: fc1 per step, W_ih + W_hh per layer and step, fc2
    flop = B * (2 * fwd + 3 * fwd)                                    # target: online + lagged forward; update: forward + backward
    cpu = None
    if with_cpu:
        from oracle import oracle_dqn as OD
        from oracle import oracle_drqn as ORQ

        ocfg = OD.DQNConfig(**kw)
        st = OD.DQNState.create(p, ocfg)
        gc = torch.Generator().manual_seed(0)
        obs, obs_next = torch.randn(B, T, OBS, generator=gc), torch.randn(B, T, OBS, generator=gc)
        act, rew = torch.randint(0, A, (B,), generator=gc), torch.randn(B, generator=gc)
        th = _threads()

        def one():
            ORQ.update_with_batch(st, ocfg, obs, act, rew + ocfg.gamma * ORQ.target_q(st, ocfg, obs_next))

        one()
        t0 = time.perf_counter()
        for _ in range(100):
            one()
        cpu = {"value": 100 / (time.perf_counter() - t0), "unit": "updates/s", "cores": th, "kind": "port",
               "sample": f"100 updates of B={B} (double-Q target + one optimizer step), torch fp32 CPU oracle"}
    return _line("DRQN learn() updates/sec (B=128, stack 4, obs 4, 2 actions, 2 LSTM layers of 128, n-step 3)", steps / dt,
                 "updates/s", steps, warmup, dt,
                 f"DQN on Recurrent (test_drqn.py), {slots}-slot replay of 16 envs, stack_num 4, B=128: launch-latency-bound",
                 _roofline(prof, flop, "all GEMMs of one update: input / recurrent projections, their gradients, fc1, fc2"), cpu,
                 {"final_loss": float(loss)})


RUNNERS = {
    "drqn": run_drqn,
    "reinforce": run_reinforce,
    "td3": lambda s, w, c: run_td3(s, w, c, twin=True), "ddpg": lambda s, w, c: run_td3(s, w, c, twin=False),
    "dsac": run_dsac, "qrdqn": lambda s, w, c: run_distq(s, w, c, "qr"), "c51": lambda s, w, c: run_distq(s, w, c, "c51"),
    "ppo_discrete": run_ppo_discrete, "rainbow": run_rainbow, "redq": run_redq,
    "npg": lambda s, w, c: run_natural(s, w, c, "npg"), "trpo": lambda s, w, c: run_natural(s, w, c, "trpo"),
}


def run(workload: str, steps: int, warmup: int, with_cpu: bool = True) -> dict:
    return RUNNERS[workload](steps, warmup, with_cpu)


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=sorted(RUNNERS))
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a.workload, a.steps, a.warmup, not a.no_cpu_baseline)), flush=True)
