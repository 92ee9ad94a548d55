"""C3 workload: DQN updates/s on a synthetic Atari-layout replay (2^20 slots, u8 84x84 frames,
frame stack 4 through prev(), PER alpha 0.6 / beta 0.4, n-step 3, double-Q with a lagged net, Huber, B=512).

    python bench.py --workload dqn [--steps K] [--warmup W]        (or: python bench_dqn.py)

One "step" = one DQN.update(): PER sample -> frame-stack gather of s and s_{t+n} -> Q_online(s'), Q_target(s')
-> n-step return -> Q(s), TD loss, backward, Adam -> PER priority update.  Everything device-resident.
Prints one JSON line with the same keys as bench.py plus per-kernel-kind roofline figures.
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

C, H, W, N_ACT, BATCH = 4, 84, 84, 6, 512
PEAK_F32_MFMA_TFLOPS = 157.3
# algorithmic flop per sample: one forward 18.69 MFLOP; weight gradients the same again;
# input gradients for every layer but conv1
FWD_FLOP = 2 * 9_346_048
CONV1_FLOP = 2 * 3_276_800
FLOP_BY_KIND = {"conv_fwd": 3 * FWD_FLOP, "conv_wgrad": FWD_FLOP, "conv_dgrad": FWD_FLOP - CONV1_FLOP}


def build(slots: int, E: int, seed: int = 0):
    from tianshou_amd.buffer import DeviceReplayBuffer
    from tianshou_amd.segtree import PrioritizedWeights

    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(seed)
    frames = torch.empty((slots, H, W), dtype=torch.uint8, device=dev)
    step = 1 << 16
    for lo in range(0, slots, step):
        frames[lo:lo + step] = torch.randint(0, 256, (min(step, slots - lo), H, W), generator=g, device=dev,
                                             dtype=torch.uint8)
    rew = torch.randn(slots, generator=g, device=dev).double()
    term = torch.rand(slots, generator=g, device=dev) < 0.005
    trunc = torch.zeros(slots, dtype=torch.bool, device=dev)
    act = torch.randint(0, N_ACT, (slots,), generator=g, device=dev)
    T = slots // E
    offset = np.arange(E + 1, dtype=np.int64) * T
    buf = DeviceReplayBuffer(offset=offset, last_index=offset[:-1] + T - 1, lengths=np.full(E, T, np.int64),
                             insertion=np.zeros(E, np.int64), rew=rew, terminated=term, truncated=trunc)
    per = PrioritizedWeights(slots, 0.6, 0.4)
    per.init_weight(torch.arange(slots, device=dev))
    return frames, act, buf, per


def torch_layers(seed: int = 0):
    torch.manual_seed(seed)
    return [torch.nn.Conv2d(C, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1),
            torch.nn.Linear(3136, 512), torch.nn.Linear(512, N_ACT)]


def cpu_baseline(updates: int = 3, budget_s: float = 10.0):
    """The oracle's restatement of the same update (torch fp32 on the host cores): two no-grad forwards on
    s_{t+n}, forward + backward + Adam on s; sampling / gather / n-step excluded (they favour the CPU)."""
    from oracle import oracle_dqn as OD

    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    p = OD.init_params(C, H, W, N_ACT, 0)
    cfg = OD.DQNConfig(gamma=0.99, n_step=3, target_update_freq=500, is_double=True, huber_delta=1.0, lr=1e-4)
    st = OD.DQNState.create(p, cfg)
    rng = np.random.default_rng(0)
    obs = rng.integers(0, 256, size=(BATCH, C, H, W), dtype=np.uint8)
    obs_next = rng.integers(0, 256, size=(BATCH, C, H, W), dtype=np.uint8)
    act = rng.integers(0, N_ACT, size=BATCH)
    ret = rng.normal(size=BATCH).astype(np.float32)
    OD.target_q(st, cfg, obs_next)
    OD.update_with_batch(st, cfg, obs, act, ret)
    t0, done = time.perf_counter(), 0
    while done < updates or time.perf_counter() - t0 < budget_s:         # at least `updates`, then up to ~budget_s of CPU work
        OD.target_q(st, cfg, obs_next)
        OD.update_with_batch(st, cfg, obs, act, ret)
        done += 1
    dt, updates = time.perf_counter() - t0, done
    return {"value": updates / dt, "unit": "updates/s", "cores": threads, "kind": "port",
            "sample": f"{updates} updates of B={BATCH} (2 target forwards + fwd/bwd/Adam), torch fp32 CPU oracle"}


PREFETCH = not os.environ.get("TS_DQN_NO_PREFETCH")      # A/B switch of the side-stream forward pass
REPLAY_STREAM = not os.environ.get("TS_DQN_NO_REPLAY_STREAM")      # A/B switch: priority update + next batch beside the backward pass
LEARN_STEP = not os.environ.get("TS_DQN_NO_LEARN_STEP")      # A/B switch: the whole update as one library call (ts_dqn_learn_step)


def hook_level(updates: int = 200, slots: int = 1 << 12) -> dict:
    """The drop-in as Tianshou calls it: `HipDQN.update(buffer, 512)` (tianshou_amd/integration.py) on a HOST prioritized frame
    buffer of the C3 layout (single uint8 frames, stack_num 4, 16 sub-buffers) -- `sample_indices`, importance weights,
    `_preprocess_batch`, `_update_with_batch`, `_postprocess_batch` (priorities back into the host buffer), the loss as a Python
    float (one device synchronisation per update, as the reference's `.item()` has).  The reference package is not on the GPU
    box: the subclass is built over tests/standin.py (same attribute surface, tests/test_standin_surface.py); hook bodies, device
    mirror, engine and write-back are the production code.  Two modes: the defaults (index-only sampling, write-back when the
    torch state is read) and the reference-exact mode (`host_batch=True, write_back="eager"`: the reference's own
    `Algorithm._update` with its host copy of the batch -- two stacked observations per transition -- and both networks + the
    optimizer state written back after every update)."""
    from tests import standin as SI
    from tianshou_amd.integration import make_hip_dqn

    E = 16
    n = E * slots
    rng = np.random.default_rng(0)
    out = {}
    for mode, kw, n_upd in (("default", {}, updates), ("reference_exact", dict(host_batch=True, write_back="eager"), max(updates // 8, 10))):
        torch.manual_seed(0)
        algo = make_hip_dqn(ref=SI)(policy=SI.DiscreteQLearningPolicy(SI.DQNet(C, H, W, N_ACT)), lr=1e-4, gamma=0.99,
                                    n_step_return_horizon=3, target_update_freq=500, is_double=True, huber_loss_delta=1.0,
                                    device="cuda", **kw).to("cuda")
        buf = SI.PrioritizedVectorReplayBuffer(n, E, obs_shape=(H, W), act_shape=(), obs_dtype=np.uint8, act_dtype=np.int64,
                                               stack_num=C, alpha=0.6, beta=0.4)
        buf.obs[:] = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
        buf.obs_next[:] = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
        buf.act[:] = rng.integers(0, N_ACT, n)
        buf.rew[:] = rng.standard_normal(n)
        buf.terminated[:] = rng.random(n) < 0.002
        buf.done[:] = buf.terminated
        buf.prio[:] = 1.0
        buf._meta = SI._Meta(("obs", "act", "rew", "terminated", "truncated", "done"))   # ignore_obs_next=True (atari_dqn.py)
        for e, sb in enumerate(buf.buffers):
            sb._size, sb._insertion_idx = slots, 0
            buf._lengths[e] = slots
            buf.last_index[e] = (e + 1) * slots - 1
        algo.policy.is_within_training_step = True
        for _ in range(5):
            algo.update(buf, BATCH)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_upd):
            stats = algo.update(buf, BATCH)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        algo.hip_sync()
        torch.cuda.synchronize()
        out[mode] = {"updates_per_s": n_upd / dt, "ms_per_update": dt / n_upd * 1e3, "updates": n_upd,
                     "sync_ms_after": (time.perf_counter() - t1) * 1e3, "loss": float(stats.loss)}
    out["note"] = ("HipDQN.update() over a host-filled PrioritizedVectorReplayBuffer stand-in (production hook code); `sync_ms_after` = "
                   "one hip_sync() (the deferred write-back of two networks and the optimizer state) after the timed loop")
    return out


def run(steps: int, warmup: int, slots: int = 1 << 20, with_cpu: bool = True) -> dict:
    import bench_init as BI
    from tianshou_amd import _lib
    from tianshou_amd import dqn as D

    frames, act, buf, per = build(slots, 16)
    tensors = [t for m in torch_layers() for t in (m.weight, m.bias)]
    cfg = D.DQNConfig(gamma=0.99, n_step=3, target_update_freq=500, is_double=True, huber_delta=1.0, lr=1e-4)
    eng = D.DQNEngine(C, H, W, N_ACT, D.flat_from_torch(tensors, C, H, W, N_ACT), cfg)
    gen = torch.Generator(device="cuda").manual_seed(1)

    draw = lambda: torch.rand(BATCH, generator=gen, device="cuda", dtype=torch.float64)   # prio.py:65 draws  # noqa: E731
    replay = D.ReplayStream(eng, buf, frames, per, C, draw, lambda i: act[i]) if REPLAY_STREAM and not LEARN_STEP else None
    count = [0]

    def update():
        if LEARN_STEP:      # the same cycle as below in one call: draws from the engine's Philox stream (key 1, update number)
            count[0] += 1
            return eng.learn_step(buf, frames, act, per, BATCH, (1, count[0]))[0]
        if replay is None:
            idx, wt = per.sample(draw())
            a, pair, coef = act[idx], None, None
        else:               # sampled and gathered on the replay stream while the previous update ran its backward pass
            idx, wt, a, pair, coef = replay.take()
        # both stacked gathers in one launch, Q_online(s) on a side stream beside the two s_{t+n} passes of _target_q
        obs, ret = eng.preprocess_with_obs(buf, frames, idx, C, prefetch=PREFETCH, pair=pair, coef=coef)
        loss, td = eng.update_with_batch(obs, a, ret, wt)
        if replay is None:
            per.update_weight(idx, td)
        else:
            replay.give(idx, td)
        return loss

    BI.warm_clocks()
    for _ in range(warmup):
        update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = update()
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
    kinds = {}
    for k, flop in FLOP_BY_KIND.items():
        ms, n = prof[k]
        tf = flop * BATCH * n_prof / (ms * 1e-3) / 1e12
        kinds[k] = {"achieved": tf, "frac": tf / PEAK_F32_MFMA_TFLOPS, "launches_per_update": n // n_prof,
                    "us_per_update": ms * 1e3 / n_prof, "algorithmic_flop_per_update": flop * BATCH}
    dom = max(kinds, key=lambda k: kinds[k]["us_per_update"])
    # HBM bytes per launch of the dominant kind's kernels from the TCC counters of this very command (profiles/r06_pmc_dqn.json,
    # scripts/gpu_r6_pmc.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes), launch-weighted
    names = {"conv_fwd": ("conv_rows_kernel<false", "conv_rows_kernel<0"), "conv_wgrad": ("conv_wgrad",),
             "conv_dgrad": ("conv_rows_kernel<true", "conv_rows_kernel<1")}[dom]
    traffic, traffic_parts = BI.pmc_traffic(os.path.join(ROOT, "profiles", "r06_pmc_dqn.json"), names)
    roof = {"bound": "mfma", "kernel": {"conv_fwd": "conv_rows_kernel<false,...> (3 forwards x 4 layers)",
                                        "conv_wgrad": "conv_wgrad_kernel", "conv_dgrad": "conv_rows_kernel<true,...>"}[dom],
            "achieved": kinds[dom]["achieved"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": kinds[dom]["frac"], "traffic": traffic, "traffic_by_kernel": traffic_parts,
            "traffic_source": "profiles/r06_pmc_dqn.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `bench.py --workload dqn`)",
            "avg_launch_us": kinds[dom]["us_per_update"] / kinds[dom]["launches_per_update"]}
    total_flop = sum(FLOP_BY_KIND.values()) * BATCH
    out = {
        "metric": "DQN learn() updates/sec (B=512, NatureCNN, n-step 3, PER, double-Q, Huber)",
        "value": steps / dt, "unit": "updates/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C3 DQN Atari-shape replay: {slots} slots of u8[84,84] frames, stack 4, 6 actions, "
                               "NatureCNN (1,687,206 params), B=512, n-step 3, PER, target sync every 500",
                   "parallelism": "dp1"},
        "roofline": roof, "roofline_by_kind": kinds,
        "whole_update_mfma_frac": total_flop * steps / dt / 1e12 / PEAK_F32_MFMA_TFLOPS,
        "host_enqueue_ms_per_step": t_host / steps * 1e3,
        "update_path": "ts_dqn_learn_step (one library call per update)" if LEARN_STEP else
                       "separate calls" + (" + replay stream" if REPLAY_STREAM else ""),
        "cpu_baseline": cpu_baseline() if with_cpu else None, "final_loss": float(loss),
        "hook_level": hook_level() if with_cpu else None,
    }
    return out


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--slots", type=int, default=1 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a.steps, a.warmup, a.slots, not a.no_cpu_baseline)), flush=True)
