#!/usr/bin/env python
"""bench.py -- PPO learn() hot path on MI355X (BASELINE.json metric:
"learn() update-steps/sec + GAE transitions/sec, PPO batch=65536, 1/2/4/8 GPU").

Workload (BASELINE.json configs[1], SURVEY 8d "C2"): synthetic MuJoCo-shape rollout, 512 envs x
2048 steps = 2^20 transitions per GPU, obs f32[17], act f32[6], MLP[64,64] tanh actor-critic,
PPO hyper-parameters of examples/mujoco/mujoco_ppo.py (gamma .99, lambda .95, eps .2, vf .25,
ent 0, max_grad_norm .5, value_clip, return_scaling, lr 3e-4), minibatch 65536, repeat 10.

One bench "step" = one whole update() of the reference (Algorithm._update, algorithm_base.py:586-631)
on data already resident in HBM: V(s), V(s'), GAE over the 2^20 transitions, logp_old, then
repeat x 16 = 160 minibatch gradient steps (forward, loss, backward, grad-clip, Adam).
value = minibatch gradient steps per second over the whole job (all ranks), preprocessing included.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak|both]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N   (N > 1, one rank per GPU)
    (`python bench.py --gpus N` without WORLD_SIZE in the environment starts that launcher itself)

N > 1, strong scaling (BASELINE.json configs[3], the line's `value`): ONE 2^20-transition rollout whose 512
sub-buffers are sharded over the ranks by env id (`shard_envs`), global minibatch 65,536 =
65,536 / N rows per rank per gradient step, 160 gradient steps per update() whatever N; per step the flat
fp32 gradient + loss parts (11,089 floats) are all-reduced.  The weak-scaling figure (every rank its own
2^20-transition rollout and a 65,536-row local minibatch: the per-GPU work of N = 1) is measured in the same
run and printed beside it (`weak_scaling`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ENV, T_STEPS, OBS, ACT = 512, 2048, 17, 6
N_TRANS = N_ENV * T_STEPS
MINIBATCH, REPEAT = 65536, 10
FLOP_PER_SAMPLE_STEP = 60544          # fwd 21,632 + bwd dW 21,632 + dX 17,280
GAE_BYTES_PER_TRANSITION = 26         # 22 + 4: rew kept float64 as the reference stores it
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBPS = 8000.0
# HBM bytes per launch from the TCC counters: profiles/r05_pmc_hbm_traffic.json, written by scripts/gpu_pmc_traffic.sh (two
# separate rocprofv3 --pmc passes over this very command, mean per launch per kernel; scripts/rocprof_pmc.py --json).
# FETCH_SIZE / WRITE_SIZE are in KiB and, on gfx950, FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane loads at
# 64 bytes -> doubled (MI355X_MICROARCH.md "HBM").  Counters cannot be read live from inside bench.py, so the line carries
# the profiled value of the same workload, looked up by the name of the kernel that ran; no entry -> null.
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r06_pmc_hbm_traffic.json")
TRAFFIC_SOURCE = "profiles/r06_pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, C2 workload, scripts/gpu_pmc_traffic.sh)"


def hbm_traffic_bytes(kernel, grid_threads=None, path=None):
    """(2 * FETCH_SIZE + WRITE_SIZE) KiB of the profiled `kernel`, or None when the profile has no such kernel.
    `grid_threads`: total work-items of the launch that is being quoted -- the profile keeps every launch geometry of a kernel
    name apart (`by_grid`), and a line about the 2^20-transition scan must not quote the 2^24 one's counters (round 5's file
    did: VERDICT r5).  With a grid given, ONLY that geometry is accepted (None rather than another size's bytes)."""
    try:
        with open(path or TRAFFIC_JSON) as f:
            e = json.load(f).get(kernel)
    except (OSError, ValueError):
        return None
    if e and grid_threads is not None:
        e = next((v for k, v in (e.get("by_grid") or {}).items() if int(np.prod([int(x) for x in k.split(",")])) == int(grid_threads)),
                 None)
    if not e or "FETCH_SIZE" not in e or "WRITE_SIZE" not in e:
        return None
    return int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)


H2D_BYTES_PER_UPDATE = N_TRANS * (2 * OBS * 4 + ACT * 4 + 8 + 2)   # obs, obs_next, act f32; rew f64; two flag bytes


def make_rollout(device, seed, env_range=None):
    """The synthetic rollout of `seed` (512 sub-buffers x 2048 slots, env-major); env_range = (lo, hi): only those
    sub-buffers of the SAME rollout (every rank draws the whole thing and keeps its shard)."""
    full = _make_rollout(device, seed)
    if env_range is None or env_range == (0, N_ENV):
        return full
    lo, hi = env_range
    return tuple(t[lo * T_STEPS: hi * T_STEPS].clone() for t in full)


def _make_rollout(device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    obs = torch.randn(N_TRANS, OBS, device=device, generator=g)
    obs_next = torch.randn(N_TRANS, OBS, device=device, generator=g)
    act = torch.randn(N_TRANS, ACT, device=device, generator=g)
    rew = torch.randn(N_TRANS, device=device, generator=g).double()   # generated f32, stored f64
    term = (torch.rand(N_TRANS, device=device, generator=g) < 0.005).to(torch.uint8)
    trunc = torch.zeros(N_TRANS, dtype=torch.uint8, device=device)
    trunc.view(N_ENV, T_STEPS)[:, 999::1000] = 1                      # MuJoCo 1000-step time limit
    trunc &= (1 - term)
    return obs, obs_next, act, rew, term, trunc


def mujoco_cfg():
    from tianshou_amd.ppo import PPOConfig

    return PPOConfig(gamma=0.99, gae_lambda=0.95, eps_clip=0.2, dual_clip=None, value_clip=True,
                     advantage_normalization=False, recompute_advantage=False, vf_coef=0.25,
                     ent_coef=0.0, max_grad_norm=0.5, return_scaling=True, lr=3e-4)


def init_flat_params(seed=0):
    """orthogonal(sqrt 2) / zero bias / mu head x0.01 / sigma -0.5 (mujoco_ppo.py:108-120)."""
    g = torch.Generator().manual_seed(seed)

    def ortho(rows, cols, gain):
        a = torch.randn(rows, cols, generator=g)
        flat = a if rows >= cols else a.t()
        q, r = torch.linalg.qr(flat)
        q = q * torch.sign(torch.diagonal(r)).unsqueeze(0)
        return (q if rows >= cols else q.t()) * gain

    s2 = 2.0 ** 0.5
    parts = [ortho(64, OBS, s2), torch.zeros(64), ortho(64, 64, s2), torch.zeros(64),
             ortho(ACT, 64, s2) * 0.01, torch.zeros(ACT), torch.full((ACT,), -0.5),
             ortho(64, OBS, s2), torch.zeros(64), ortho(64, 64, s2), torch.zeros(64),
             ortho(1, 64, s2), torch.zeros(1)]
    return torch.cat([p.reshape(-1) for p in parts]).float()


class Learner:
    """One rank: device-resident rollout shard + PPO engine.  world_size > 1 splits every gradient step around an
    all-reduce of the flat gradient.  scaling = "strong": this rank's sub-buffers of the one global rollout and its
    65,536 / world rows of every global minibatch; "weak": a whole rollout of its own and 65,536-row local minibatches."""

    def __init__(self, device, rank, world, scaling="strong", allreduce=None, exchange="none"):
        from tianshou_amd import _lib
        from tianshou_amd.distributed import shard_envs
        from tianshou_amd.ppo import PPOEngine

        self.device, self.rank, self.world, self.scaling = device, rank, world, scaling
        if scaling == "strong" and world > 1:
            lo, hi = shard_envs(N_ENV, rank, world)
            self.data = make_rollout(device, seed=1000, env_range=(lo, hi))
            self.n_env, self.minibatch = hi - lo, MINIBATCH // world
        else:
            self.data = make_rollout(device, seed=1000 + rank)
            self.n_env, self.minibatch = N_ENV, MINIBATCH
        self.n_trans = self.n_env * T_STEPS
        self.cfg = mujoco_cfg()
        self.eng = PPOEngine(OBS, ACT, init_flat_params(0).to(device), self.cfg)
        self.cut = (torch.arange(self.n_env, device=device) + 1) * T_STEPS - 1   # last slot of every env
        self.perm_seed = 1234 + 7919 * rank
        self._lib = _lib
        self.ws = _lib.default_workspace(device.index)
        self.dp = None
        self._ar = allreduce
        self.exchange = exchange

    def next_perm(self):
        from tianshou_amd.buffer import random_permutation

        self.perm_seed += 0x9E3779B97F4A7C15
        return random_permutation(self.n_trans, self.perm_seed, self.device)

    def _dp(self):
        """world > 1: every minibatch is ONE C call (ts_ppo_dp_step: gradient -> exchange -> clip + Adam on the stream); the
        exchange is the C-ABI all-reduce (`make_exchange`), torch.distributed if that could not be set up."""
        if self.dp is None:
            from tianshou_amd.distributed import DataParallelPPO

            self.dp = DataParallelPPO(self.eng, allreduce=self._ar)
        return self.dp

    def preprocess(self, local_only=False):
        """local_only: no collective inside (the rank-0-only measurements after the timed region)."""
        obs, obs_next, act, rew, term, trunc = self.data
        if self.world > 1 and not local_only:   # shard-local values / GAE / logp_old; ret_rms from the global return statistics
            return self._dp().preprocess(obs, obs_next, act, rew, term, trunc, self.cut)
        return self.eng.preprocess(obs, obs_next, act, rew, term, trunc, self.cut)

    def update_once(self):
        """one reference update(): preprocess + REPEAT x (transitions / minibatch) gradient steps."""
        b = self.preprocess()
        # minibatch order: keyed device-side permutations (ts_random_permutation); the reference draws
        # np.random.permutation on the host (~10 ms per 2^20 entries), a sort-based torch.randperm
        # costs ~0.25 ms - either would be a visible part of the 13 ms update
        perms = [self.next_perm() for _ in range(REPEAT)]
        if self.world == 1 and not os.environ.get("TS_BENCH_FORCE_DP"):    # (env: time the DP host path at N = 1)
            losses, steps = self.eng.update(b, self.minibatch, REPEAT, perms)
            return losses, steps
        return self._dp().update(b, self.minibatch, REPEAT, perms)


def make_exchange(device, rank, world):
    """The gradient exchange of the N > 1 legs: the C-ABI all-reduce, whose one-shot path (peer buffers mapped through HIP
    IPC, one single-workgroup kernel) carries the 44 KB payload; torch.distributed (three Python calls per step) only if
    that cannot be set up on EVERY rank.  -> (allreduce or None, description)"""
    if world == 1:
        return None, "none"
    import torch.distributed as dist

    ar, name = None, "torch.distributed all_reduce (RCCL)"
    if not os.environ.get("TS_BENCH_TORCH_ALLREDUCE"):
        try:
            from tianshou_amd.collective import NativeAllReduce

            ar = NativeAllReduce(device, rccl=not os.environ.get("TS_BENCH_ONE_GPU"))
        except Exception as e:      # noqa: BLE001 - any failure of the native set-up: the proven path
            print(f"[bench] rank {rank}: native all-reduce unavailable ({e}); using torch.distributed", file=sys.stderr)
            ar = None
        ok = torch.tensor([0 if ar is None else 1], device=device)       # all ranks or none
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and ar is not None:
            ar.close()
            ar = None
    if ar is not None:
        name = "ts_allreduce one-shot (HIP IPC)" if ar.small_capacity >= 11085 + 4 else "ts_allreduce (RCCL)"
    return ar, name


def exchange_selftest(device, rank, world, ar, floats, rounds):
    """Before anything is timed at N > 1: `rounds` all-reduces of per-rank random payloads through the exchange the legs
    will use, each checked bit for bit against torch.distributed's all_reduce of the same payload (integer-valued floats:
    every summation order gives the same bits).  A mismatch or an error on ANY rank makes EVERY rank drop the native
    exchange (the caller falls back to torch.distributed / RCCL).  Also collected per rank for the JSON line: the device,
    hipDeviceCanAccessPeer to every other visible device, whether the one-shot IPC path came up.
    -> (ok on every rank, report dict on rank 0 / None elsewhere)."""
    import torch.distributed as dist

    on_gpu = dist.get_backend() == "nccl"
    bad, err = 0, None
    if ar is not None:
        g = torch.Generator(device=device).manual_seed(977 * rank + 13)
        try:
            for r in range(rounds):
                x = torch.randint(-1024, 1025, (floats,), generator=g, device=device).float()
                want = x.clone() if on_gpu else x.cpu()
                dist.all_reduce(want)
                ar(x)
                if not torch.equal(x.cpu(), want.cpu()):
                    bad += 1
                    if bad >= 3:
                        break
        except Exception as e:      # noqa: BLE001 - a failed hand-off of the one-shot path raises: same consequence
            err = repr(e)
            bad += 1
    flag = torch.tensor([1 if bad == 0 else 0], dtype=torch.int32, device=device if on_gpu else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    n_dev = torch.cuda.device_count()
    me = {"rank": rank, "device": str(device), "visible_devices": n_dev,
          "can_access_peer": [bool(torch.cuda.can_device_access_peer(device.index, j)) if j != device.index else None
                              for j in range(n_dev)],
          "one_shot_ipc": bool(ar is not None and ar.small_capacity > 0), "native": ar is not None,
          "mismatching_rounds": bad, "error": err}
    gathered = [None] * world
    dist.all_gather_object(gathered, me)
    ok = int(flag.item()) == 1
    report = {"rounds": rounds if ar is not None else 0, "floats": floats, "passed_on_every_rank": ok, "ranks": gathered} if rank == 0 else None
    return ok, report


def time_exchange(device, world, ar, floats, iters=200):
    """Mean time of one all-reduce of the per-step payload ([gradient | loss parts]), HIP events on the launch stream around
    `iters` back-to-back calls on every rank (collective: all ranks call it)."""
    if world == 1:
        return None
    import torch.distributed as dist

    buf = torch.zeros(floats, dtype=torch.float32, device=device)
    call = ar if ar is not None else (lambda t: dist.all_reduce(t))
    for _ in range(20):
        call(buf)
    torch.cuda.synchronize()
    dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call(buf)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def time_gae(learner, iters=50):
    """GAE scan alone (inputs resident in HBM -> adv, returns), HIP events on the launch stream."""
    from tianshou_amd.returns import gae_scan

    obs, obs_next, act, rew, term, trunc = learner.data
    v = torch.randn(learner.n_trans, device=learner.device)
    vn = torch.randn(learner.n_trans, device=learner.device)
    for _ in range(5):
        gae_scan(v, vn, rew, term, trunc, learner.cut)
    torch.cuda.synchronize()
    learner.ws.profile_begin()
    for _ in range(iters):
        gae_scan(v, vn, rew, term, trunc, learner.cut)
    prof = learner.ws.profile_end()
    ms = (prof["gae_maps"][0] + prof["gae_apply"][0]) / iters
    return ms * 1e-3


def time_gae_large(learner, log2n=24, iters=10, n_env=8192):
    """The same scan at 2^24 transitions (8192 envs x 2048 steps): the bandwidth the kernel reaches once the launch
    and hand-off latencies are amortised ('also report N = 2^24 ... for asymptotic GB/s')."""
    from tianshou_amd.returns import gae_scan

    n, dev = 1 << log2n, learner.device
    g = torch.Generator(device=dev).manual_seed(1)
    v, vn = torch.randn(n, generator=g, device=dev), torch.randn(n, generator=g, device=dev)
    rew = torch.randn(n, generator=g, device=dev).double()
    term = torch.rand(n, generator=g, device=dev) < 0.002
    trunc = torch.zeros(n, dtype=torch.bool, device=dev)
    per = n // n_env
    cut = torch.arange(per - 1, n, per, device=dev, dtype=torch.int64)              # last slot of every sub-buffer
    for _ in range(2):
        gae_scan(v, vn, rew, term, trunc, cut)
    torch.cuda.synchronize()
    learner.ws.profile_begin()
    for _ in range(iters):
        gae_scan(v, vn, rew, term, trunc, cut)
    prof = learner.ws.profile_end()
    return n, (prof["gae_maps"][0] + prof["gae_apply"][0]) / iters * 1e-3


def step_plan(learner, rows):
    """(kernel name, workgroups, gradient slabs) of the fused step for a minibatch of `rows` rows (ts_ppo_step_plan)."""
    import ctypes as C

    v, g, sl = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    learner._lib.check(learner._lib.load().ts_ppo_step_plan(learner._lib.i64(OBS), learner._lib.i64(ACT), learner._lib.i64(rows),
                                                          C.c_int32(0), C.byref(v), C.byref(g), C.byref(sl)))
    return ("ppo_step2_kernel", "ppo_stepq_kernel", "ppo_stepq2_kernel")[v.value], g.value, sl.value


def strong_scaling_projection(learner, b, steps=16):
    """BASELINE configs[3] on the one GPU that is here: a rank of an N-GPU strong-scaling run takes 65,536 / N rows of
    every global minibatch, so its per-step device time is the step kernel + slab reduction + Adam at that row count (plus
    the gradient exchange, which needs the node).  Measured per row count: the three kernels by HIP events (ts_profile)
    and the wall time per step of `steps` back-to-back steps on the stream, with the slab count the reduction reads."""
    from tianshou_amd.ppo import split_offsets  # noqa: F401  (same module as the engine; keeps the import local)

    out = []
    eng = learner.eng
    for world in (1, 2, 4, 8):
        rows = MINIBATCH // world
        n = rows * steps
        sub = {k: b[k][:n] for k in ("obs", "act", "adv", "returns", "logp_old", "v_s")}
        full = learner.next_perm()
        perm = full[full < n].contiguous()                  # a permutation of range(n)
        offs = [k * rows for k in range(steps + 1)]
        saved = (eng.params.clone(), eng.adam_m.clone(), eng.adam_v.clone(), eng.adam_step)
        eng._run_steps(sub, perm, offs)                     # warm-up
        torch.cuda.synchronize()
        learner.ws.profile_begin()
        eng._run_steps(sub, perm, offs)
        prof = learner.ws.profile_end()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(4):
            eng._run_steps(sub, perm, offs)
        t1.record()
        torch.cuda.synchronize()
        eng.params.copy_(saved[0]); eng.adam_m.copy_(saved[1]); eng.adam_v.copy_(saved[2]); eng.adam_step = saved[3]
        us = lambda k: prof[k][0] / max(prof[k][1], 1) * 1e3     # noqa: E731
        wall = t0.elapsed_time(t1) * 1e3 / (4 * steps)
        kernel, grid, slabs = step_plan(learner, rows)
        out.append({"n_gpus": world, "rows_per_rank": rows, "step_kernel": kernel, "workgroups": grid, "slabs": slabs,
                    "step_kernel_us": us("ppo_step"), "reduce_us": us("ppo_reduce"), "adam_us": us("ppo_adam"),
                    "wall_us_per_step": wall, "speedup_over_n1_without_exchange": None})
    for e in out:
        e["speedup_over_n1_without_exchange"] = out[0]["wall_us_per_step"] / e["wall_us_per_step"]
    return {"what": "per-rank device time of one gradient step at 65,536 / N rows, measured on ONE GPU (no gradient exchange: "
                    "add `exchange_us` of a --gpus N run)", "steps_timed": steps, "by_world_size": out}


def cpu_baseline(sample_steps=None):
    """The oracle (CPU port of the reference path: torch-fp32 ops + C restatement of the numba
    kernels) on the host cores.  Default sample: ONE whole update() = full preprocess of the 2^20
    rollout in max_batchsize=65536 chunks + all 160 minibatch gradient steps of 65536 (10 repeats
    over fresh permutations), i.e. one step of this bench (~11 s on 32 threads); `sample_steps` < 160
    times that many gradient steps and extrapolates 160 / (t_pre + 160 * t_step)."""
    from oracle import oracle as O
    from oracle import oracle_ppo as OP

    # torch's intra-op pool on every hardware thread is far slower than a moderate pool for these
    # small GEMMs (oversubscription); use at most 32 threads and report the count actually used.
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 2) // 2)))
    rng = np.random.default_rng(0)
    obs = torch.from_numpy(rng.normal(size=(N_TRANS, OBS)).astype(np.float32))
    obs_next = torch.from_numpy(rng.normal(size=(N_TRANS, OBS)).astype(np.float32))
    act = torch.from_numpy(rng.normal(size=(N_TRANS, ACT)).astype(np.float32))
    rew = rng.normal(size=N_TRANS).astype(np.float32).astype(np.float64)
    term = rng.random(N_TRANS) < 0.005
    trunc = np.zeros(N_TRANS, bool)
    c = mujoco_cfg()
    ocfg = OP.PPOConfig(gamma=c.gamma, gae_lambda=c.gae_lambda, eps_clip=c.eps_clip, value_clip=True,
                        advantage_normalization=False, vf_coef=c.vf_coef, ent_coef=c.ent_coef,
                        max_grad_norm=c.max_grad_norm, return_scaling=True, lr=c.lr, max_batchsize=65536)
    flat = init_flat_params(0)
    st = OP.PPOState(params=OP.unflatten_params(flat, OBS, ACT))
    idx = np.arange(N_TRANS)
    unf = (np.arange(N_ENV) + 1) * T_STEPS - 1
    O.compute_episodic_return(rew[:4096], term[:4096], trunc[:4096], idx[:4096], unf[:1], np.zeros(4096, np.float32),
                              np.zeros(4096, np.float32), 0.99, 0.95)       # library load / first-touch outside the clock
    OP.preprocess(OP.PPOState(params=OP.unflatten_params(flat, OBS, ACT)), ocfg, obs[:65536], obs_next[:65536], act[:65536],
                  rew[:65536], term[:65536], trunc[:65536], idx[:65536], unf[:32])
    t0 = time.perf_counter()
    pre = OP.preprocess(st, ocfg, obs, obs_next, act, rew, term, trunc, idx, unf)
    t_pre = time.perf_counter() - t0
    t_gae = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        O.compute_episodic_return(rew, term, trunc, idx, unf, pre["v_s"].numpy(), pre["v_s"].numpy(), 0.99, 0.95)
        t_gae = min(t_gae, time.perf_counter() - t0)
    steps_per_update = REPEAT * (N_TRANS // MINIBATCH)
    if sample_steps is None or sample_steps >= steps_per_update:       # the whole update(), as the reference runs it
        sample_steps = steps_per_update
        perms = [rng.permutation(N_TRANS) for _ in range(REPEAT)]
        t0 = time.perf_counter()
        OP.update(st, ocfg, {"obs": obs, "act": act}, pre, MINIBATCH, REPEAT, perms)
    else:
        sub = rng.permutation(N_TRANS)[: MINIBATCH * sample_steps]
        data = {"obs": obs[sub], "act": act[sub]}
        pre_s = {k: pre[k][sub] for k in ("v_s", "returns", "adv", "logp_old")}
        t0 = time.perf_counter()
        OP.update(st, ocfg, data, pre_s, MINIBATCH, 1, [np.arange(MINIBATCH * sample_steps)])
    t_step = (time.perf_counter() - t0) / sample_steps
    value = steps_per_update / (t_pre + steps_per_update * t_step)
    # the port calibrated against the reference itself where the reference can run (the authoring container: 8 CPUs, so 4
    # threads in round 5 and all 8 in round 6): profiles/r0{5,6}_cpu_reference_vs_port.json, scripts/cpu_reference_vs_port.py
    calib = ""
    try:
        cal_name = next(n for n in ("r06_cpu_reference_vs_port.json", "r05_cpu_reference_vs_port.json")
                        if os.path.exists(os.path.join(ROOT, "profiles", n)))
        with open(os.path.join(ROOT, "profiles", cal_name)) as f:
            c = json.load(f)
        calib = (f"; calibration against tianshou's own PPO.update() on the same buffer / weights / seed ({c['threads']} threads of a "
                 f"{c['host_cpus']}-CPU container, {c['shape']['transitions']} transitions, {c['port']['gradient_steps']} steps): reference "
                 f"{c['reference_gae_compiled']['steps_per_s']:.1f} steps/s (njit bodies compiled), port {c['port']['steps_per_s']:.1f} = "
                 f"{c['ratio_port_over_reference_compiled']:.2f} x the reference, final parameters differ by "
                 f"{c['max_abs_param_diff_reference_vs_port']:.1e} (profiles/{cal_name}); CAVEAT: the calibration ran at "
                 f"{c['threads']} threads, this line at {torch.get_num_threads()} -- the reference's share of Python / "
                 "single-thread NumPy work grows with the thread count, so the port's advantage at this line's thread count is "
                 "at least the calibrated ratio, not equal to it")
    except (OSError, ValueError, KeyError):
        pass
    return {
        "value": value, "unit": "update-steps/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": (f"PORT of the reference path, not the reference itself (oracle/: torch-fp32 CPU ops in the reference's "
                   f"order + a -O3 C restatement of its numba kernels): 1 full preprocess of 2^20 transitions "
                   f"({t_pre:.2f} s, of which the GAE scan {t_gae * 1e3:.1f} ms single-thread = "
                   f"{N_TRANS / t_gae / 1e6:.0f} M transitions/s) + {sample_steps} of the 160 gradient steps of 65536 "
                   f"({t_step * 1e3:.1f} ms each, {torch.get_num_threads()} threads); whole-update rate = 160 / (t_pre + 160 t_step)" + calib),
        "gae_transitions_per_s": N_TRANS / t_gae,
        "inner_update_steps_per_s": 1.0 / t_step,
    }


def rocm_eager_baseline(device, steps=16):
    """SURVEY 8d's second baseline: what a Tianshou user has on this GPU today - the reference's own sequence of torch
    operations (oracle/oracle_ppo.py, pinned to the reference by tests/golden) with every tensor on the MI355X, i.e.
    PyTorch-ROCm eager kernels, the GAE recurrence on the host like the reference's numba kernel.  Bounded sample:
    one preprocess of the 2^20 rollout (data resident) + `steps` minibatch steps of 65536."""
    from oracle import oracle_ppo as OP

    g = torch.Generator(device=device).manual_seed(5)
    obs = torch.randn(N_TRANS, OBS, device=device, generator=g)
    obs_next = torch.randn(N_TRANS, OBS, device=device, generator=g)
    act = torch.randn(N_TRANS, ACT, device=device, generator=g)
    rng = np.random.default_rng(0)
    rew = rng.normal(size=N_TRANS).astype(np.float32).astype(np.float64)
    term = rng.random(N_TRANS) < 0.005
    trunc = np.zeros(N_TRANS, bool)
    c = mujoco_cfg()
    ocfg = OP.PPOConfig(gamma=c.gamma, gae_lambda=c.gae_lambda, eps_clip=c.eps_clip, value_clip=True,
                        advantage_normalization=False, vf_coef=c.vf_coef, ent_coef=c.ent_coef,
                        max_grad_norm=c.max_grad_norm, return_scaling=True, lr=c.lr, max_batchsize=65536)
    st = OP.PPOState(params={k: v.to(device) for k, v in OP.unflatten_params(init_flat_params(0), OBS, ACT).items()})
    idx = np.arange(N_TRANS)
    unf = (np.arange(N_ENV) + 1) * T_STEPS - 1
    pre = OP.preprocess(st, ocfg, obs, obs_next, act, rew, term, trunc, idx, unf)          # warm-up (kernel loading)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pre = OP.preprocess(st, ocfg, obs, obs_next, act, rew, term, trunc, idx, unf)
    torch.cuda.synchronize()
    t_pre = time.perf_counter() - t0
    n_sub = MINIBATCH * 2
    sub = torch.randperm(N_TRANS, device=device, generator=g)[:n_sub]
    data = {"obs": obs[sub], "act": act[sub]}
    pre_s = {k: pre[k][sub] for k in ("v_s", "returns", "adv", "logp_old")}
    perms = [np.arange(n_sub)] * (steps // 2 + 1)
    OP.update(st, ocfg, data, pre_s, MINIBATCH, 1, perms[:1])                              # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    OP.update(st, ocfg, data, pre_s, MINIBATCH, steps // 2, perms[:steps // 2])
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / (2 * (steps // 2))
    n_steps = REPEAT * (N_TRANS // MINIBATCH)
    return {"value": n_steps / (t_pre + n_steps * t_step), "unit": "update-steps/s", "kind": "port",
            "device": "MI355X, PyTorch-ROCm eager (same torch build the engine links against)",
            "sample": (f"oracle_ppo with tensors on cuda: 1 preprocess of 2^20 transitions ({t_pre * 1e3:.0f} ms, GAE "
                       f"recurrence on the host as in the reference) + {2 * (steps // 2)} gradient steps of 65536 "
                       f"({t_step * 1e3:.2f} ms each); whole-update rate = 160 / (t_pre + 160 t_step)"),
            "inner_update_steps_per_s": 1.0 / t_step}


def h2d_seconds(device, reps=3):
    """Wall time of moving one update()'s 2^20-transition batch (obs, obs_next, act, rew, flags: 179 MB) from pinned
    host memory to HBM - what `value` would additionally pay if the boundary handed over host buffers."""
    host = [torch.empty((N_TRANS, OBS), dtype=torch.float32).pin_memory(), torch.empty((N_TRANS, OBS), dtype=torch.float32).pin_memory(),
            torch.empty((N_TRANS, ACT), dtype=torch.float32).pin_memory(), torch.empty(N_TRANS, dtype=torch.float64).pin_memory(),
            torch.empty(N_TRANS, dtype=torch.uint8).pin_memory(), torch.empty(N_TRANS, dtype=torch.uint8).pin_memory()]
    dst = [torch.empty_like(h, device=device) for h in host]
    best = 1e9
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for h, d in zip(host, dst):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def hook_level(device, updates=3, permutations="device"):
    """The drop-in as Tianshou calls it: `HipPPO.update(buffer, batch_size, repeat)` (tianshou_amd/integration.py) on a
    HOST replay buffer of the C2 shape.  The reference package is not on the GPU box, so the subclass is built over the
    stand-ins of tests/standin.py (same attribute surface, tests/test_standin_surface.py); the hook bodies, the device
    mirror, the engine and the write-back are the production code.  Reports the first update (whole buffer crosses
    PCIe: mirror snapshot) and the steady state (nothing new to copy)."""
    from torch import nn

    from tests import standin as SI
    from tianshou_amd.integration import make_hip_ppo

    HipPPO = make_hip_ppo("ppo", ref=SI)
    torch.manual_seed(0)
    actor = SI.ContinuousActorProbabilistic(SI.Net(OBS, [64, 64], nn.Tanh), ACT, unbounded=True)
    critic = SI.ContinuousCritic(SI.Net(OBS, [64, 64], nn.Tanh))
    with torch.no_grad():
        actor.sigma_param.fill_(-0.5)
    c = mujoco_cfg()
    policy = SI.Policy(actor, action_space=SI.Box(-1.0, 1.0, (ACT,)), action_scaling=True, action_bound_method="clip")
    algo = HipPPO(policy=policy, critic=critic, device=str(device), permutations=permutations, lr=c.lr, eps_clip=c.eps_clip, value_clip=True,
                  advantage_normalization=False, vf_coef=c.vf_coef, ent_coef=c.ent_coef, max_grad_norm=c.max_grad_norm,
                  return_scaling=True, gamma=c.gamma, gae_lambda=c.gae_lambda).to(device)
    buf = SI.VectorReplayBuffer(N_TRANS, N_ENV, obs_shape=(OBS,), act_shape=(ACT,))
    rng = np.random.default_rng(3)
    # filled like N_ENV x T_STEPS add() calls would (env-major storage, every sub-buffer full and unwrapped)
    buf.obs[:] = rng.standard_normal((N_TRANS, OBS), dtype=np.float32)
    buf.obs_next[:] = rng.standard_normal((N_TRANS, OBS), dtype=np.float32)
    buf.act[:] = rng.standard_normal((N_TRANS, ACT), dtype=np.float32)
    buf.rew[:] = rng.standard_normal(N_TRANS, dtype=np.float32)
    buf.terminated[:] = rng.random(N_TRANS) < 0.005
    buf.done[:] = buf.terminated
    for e, sb in enumerate(buf.buffers):
        sb._size, sb._insertion_idx = T_STEPS, 0
        buf._lengths[e] = T_STEPS
        buf.last_index[e] = (e + 1) * T_STEPS - 1
    algo.policy.is_within_training_step = True
    times = []
    for _ in range(updates + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stats = algo.update(buf, MINIBATCH, REPEAT)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    n_steps = stats.gradient_steps
    # the Collector's step on the other side of the buffer (data/collector.py:735-744) for the C2 vector of 512
    # environments -- policy(batch) (host observations in, sampled action out) + policy.map_action(act) -- with the forward on
    # the engine's inference kernels reading the engine's parameters (tianshou_amd/policy.py; the default), beside the same
    # sequence as the reference's torch modules run it on the GPU (ROCm eager: Net -> mu head -> Normal.sample -> NumPy clip /
    # scale on the host)
    obs_np = rng.standard_normal((N_ENV, OBS), dtype=np.float32)
    batch = SI.Batch(obs=obs_np, info={})

    def engine_step():
        res = algo.policy(batch, None)
        return algo.policy.map_action(res.act.detach().cpu().numpy())

    def eager_step():
        with torch.no_grad():
            h = actor.preprocess.model.model(torch.as_tensor(obs_np, device=device, dtype=torch.float32))
            mu = actor.mu.model(h)
            sigma = (actor.sigma_param.view(1, -1) + torch.zeros_like(mu)).exp()
            act = torch.distributions.Independent(torch.distributions.Normal(mu, sigma), 1).sample()
        return SI.Policy.map_action(algo.policy, act.detach().cpu().numpy())

    def per_step_us(fn, n=300):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    collector = {"envs": N_ENV, "engine_forward_us_per_step": per_step_us(engine_step), "torch_eager_us_per_step": per_step_us(eager_step),
                 "policy_class": type(algo.policy).__name__, "reads_engine_parameters": algo.policy._hip_engine() is algo._hip_engine,
                 "note": "policy(batch) + policy.map_action(act) for one vector step of 512 environments, host observations in, "
                         "host actions out; engine = ts_ppo_policy_forward_bounded (forward, sample, bound, scale: 2 launches, "
                         "1 H2D, 1 D2H)"}
    return {"first_update_steps_per_s": n_steps / times[0], "first_update_ms": times[0] * 1e3,
            "steady_update_steps_per_s": n_steps / min(times[1:]), "steady_update_ms": min(times[1:]) * 1e3,
            "permutations": permutations, "collector_step": collector,
            "note": "HipPPO.update() over a host-filled VectorReplayBuffer stand-in; first = full mirror upload + engine "
                    "creation, steady = no new slots to copy"}


def other_workloads(with_cpu: bool = True):
    """Short runs of the C3 / C5 / Atari-shape PPO / NPG / TRPO rows so that they are measured by the same driver command."""
    out = {}
    for name, mod, args in (("dqn", "bench_dqn", (150, 30)), ("sac", "bench_sac", (300, 50)), ("ppo_atari", "bench_ppo_cnn", (1, 1))):
        try:
            import importlib

            # C3 / C5 carry their own `cpu_baseline` (the torch-fp32 CPU port, a bounded sample of ~10 s each) and
            # `roofline.traffic` (TCC counters of their own commands, profiles/r06_pmc_{dqn,sac}.json) -- SURVEY 8d asks for
            # both per configuration; the Atari-shape PPO port needs ~50 s per gradient step and stays out of the default run
            r = importlib.import_module(mod).run(*args, with_cpu=with_cpu and name in ("dqn", "sac"))
            out[name] = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r.get("ms_per_step"),
                         "roofline_frac": (r.get("roofline") or {}).get("frac"), "roofline": r.get("roofline"),
                         "cpu_baseline": r.get("cpu_baseline"), "host_enqueue_ms_per_step": r.get("host_enqueue_ms_per_step"),
                         "config": r.get("config")}
        except Exception as e:                                   # a side leg must not take the headline line down
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    # NPG / TRPO on the C2 shape (bench_next.py): the rows whose network passes became one-launch kernels in round 5
    for name in ("npg", "trpo"):
        try:
            import bench_next

            r = bench_next.RUNNERS[name](5, 2, False)
            out[name] = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r.get("ms_per_step"),
                         "roofline_frac": (r.get("roofline") or {}).get("frac"), "config": r.get("config")}
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", default="both", choices=["strong", "weak", "both"],
                    help="N > 1: strong = one 2^20-transition rollout sharded over the ranks, global minibatch 65536 (BASELINE "
                         "configs[3]); weak = a 2^20 rollout and 65536 rows per rank; both (default) = strong is `value`, the weak "
                         "figure is measured in the same run and printed beside it.  N = 1: the two are the same workload.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--selftest-rounds", type=int, default=1000,
                    help="N > 1: all-reduces of random payloads checked against torch.distributed before the timed legs "
                         "(a mismatch on any rank drops the native exchange on every rank)")
    ap.add_argument("--no-extras", action="store_true", help="skip the ROCm-eager baseline, the hook-level leg, the H2D "
                    "measurement and the other workloads (profiling runs)")
    ap.add_argument("--workload", default="ppo",
                    choices=["ppo", "dqn", "sac", "ppo_atari", "td3", "ddpg", "dsac", "qrdqn", "c51", "rainbow", "npg", "trpo", "redq", "ppo_discrete", "reinforce", "drqn"],
                    help="ppo = BASELINE.json's metric on C2 (default); dqn / sac = the C3 / C5 rows (bench_dqn.py, "
                         "bench_sac.py); ppo_atari = the north star's Atari-shape PPO (bench_ppo_cnn.py); the rest = "
                         "the SURVEY 8f rows and BASELINE configs[0] (bench_next.py)")
    args = ap.parse_args()
    if args.workload != "ppo":
        import importlib

        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")
        if args.workload in ("td3", "ddpg", "dsac", "qrdqn", "c51", "rainbow", "npg", "trpo", "redq", "ppo_discrete", "reinforce", "drqn"):
            import bench_next

            k = 1 if args.workload in ("ppo_discrete", "npg", "trpo", "reinforce") else 10
            # sub-millisecond updates: 20 warm-up updates, so that a 20 ms timed region does not start inside the clock /
            # allocator / lazy-load transients of a fresh process (2 warm-up updates gave one outlier run in four)
            print(json.dumps(bench_next.run(args.workload, max(args.steps, 1) * k, max(args.warmup, 1) * (20 if k == 10 else 2),
                                            with_cpu=not args.no_cpu_baseline)), flush=True)
            return
        if args.workload == "ppo_atari":
            import bench_ppo_cnn

            print(json.dumps(bench_ppo_cnn.run(1, 1, with_cpu=not args.no_cpu_baseline)), flush=True)
            return
        mod = importlib.import_module("bench_" + args.workload)
        print(json.dumps(mod.run(max(args.steps, 1) * 10, max(args.warmup, 1) * 20,
                                 with_cpu=not args.no_cpu_baseline)), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started like the N = 1 bench (`python bench.py --gpus N ...`): become the launcher of the N ranks
        import socket
        import subprocess

        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")
    # TS_BENCH_ONE_GPU=1: a dry run of the N > 1 path on a single GPU (every rank on cuda:0, rendezvous over gloo, the
    # exchange on the one-shot IPC all-reduce -- ranks that share a GPU cannot form an RCCL communicator).  Not a
    # measurement: the ranks time-share the chip.
    one_gpu = bool(os.environ.get("TS_BENCH_ONE_GPU")) and world > 1
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    import bench_init

    ar, exchange = make_exchange(device, rank, world)
    xch = {"ar": ar, "name": exchange, "note": None}
    selftest = None
    if world > 1:
        ok, selftest = exchange_selftest(device, rank, world, ar, 11085 + 4, args.selftest_rounds)
        if not ok and ar is not None:
            try:
                ar.close()
            except Exception:       # noqa: BLE001
                pass
            xch["ar"], xch["name"] = None, "torch.distributed all_reduce (RCCL)"
            xch["note"] = "the native exchange failed the pre-flight self-test; every leg runs on torch.distributed"
    legs = ["strong", "weak"] if (args.scaling == "both" and world > 1) else [("strong" if args.scaling == "both" else args.scaling)]

    def timed_leg(scaling):
        """W warm-up + exactly K timed update()s, barrier + synchronize on both sides, MAX over the ranks.  If the native
        exchange reports a failed hand-off on ANY rank (bounded spin of the one-shot path: never seen on one device, unproven
        across xGMI), every rank drops it and the leg is measured again on torch.distributed -- a slower number instead of
        no number."""
        out = _timed_leg(scaling)
        if out is None:
            out = _timed_leg(scaling)
        return out

    def _timed_leg(scaling):
        ar, exchange = xch["ar"], xch["name"]
        learner = Learner(device, rank, world, scaling, allreduce=ar, exchange=exchange)
        bench_init.warm_clocks(device)        # idle clocks -> load clocks before the W warm-up steps (not an update step)
        for _ in range(args.warmup):
            learner.update_once()
        barrier()
        t0 = time.perf_counter()
        total = 0
        for _ in range(args.steps):
            losses, steps = learner.update_once()
            total += steps
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist

            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        if ar is not None:
            import torch.distributed as dist

            ok = 1
            try:
                ar.check()                    # a one-shot exchange whose peer never arrived would have left stale sums
            except Exception as e:            # noqa: BLE001
                ok = 0
                print(f"[bench] rank {rank}: native exchange failed its check ({e})", file=sys.stderr)
            t = torch.tensor([ok], device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 0:
                ar.close()
                xch["ar"], xch["name"] = None, "torch.distributed all_reduce (RCCL)"
                xch["note"] = "the native exchange failed its hand-off check; legs re-measured on torch.distributed"
                return None
        return learner, el, total, [float(x) for x in losses[-1].tolist()]

    learner, elapsed, total_steps, final_loss = timed_leg(legs[0])
    head = legs[0]
    # strong: `total_steps` gradient steps of the GLOBAL minibatch; weak: every rank ran that many on its own 65,536 rows
    value = total_steps / elapsed * (world if head == "weak" else 1)
    beside = None
    if len(legs) > 1:
        keep = learner
        l2, el2, st2, _ = timed_leg(legs[1])
        beside = {"scaling": legs[1], "value": world * st2 / el2, "unit": "update-steps/s", "ms_per_step": el2 / args.steps * 1e3,
                  "workload": f"every rank its own 2^20-transition rollout, local minibatch {MINIBATCH} (the per-GPU work of N = 1)"}
        del l2
        torch.cuda.empty_cache()
        learner = keep
    ar, exchange = xch["ar"], xch["name"]
    exchange_us = time_exchange(device, world, ar, learner.eng.P + 4)
    rccl_ranks = None
    if ar is not None:
        rccl_ranks = {"communicator": ar.ranks()[0], "rccl_reported": ar.ranks()[1]}
    elif world > 1:
        import torch.distributed as dist

        rccl_ranks = {"communicator": dist.get_world_size(), "rccl_reported": dist.get_world_size() if dist.get_backend() == "nccl" else 0}

    # per-kernel durations with HIP events on the launch stream, one more (rank-local) update()
    roof, extra = None, {}
    n_chunks = learner.n_trans // learner.minibatch
    if rank == 0:
        b = learner.preprocess(local_only=True)
        torch.cuda.synchronize()
        # every world size: rank 0 times the kernels of one local update() on its own shard (no collective inside)
        if learner.eng is not None:
            learner.ws.profile_begin()
            perms = [learner.next_perm() for _ in range(REPEAT)]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            learner.eng.update(b, learner.minibatch, REPEAT, perms)
            torch.cuda.synchronize()
            t_inner = time.perf_counter() - t1
            prof = learner.ws.profile_end()
            step_ms, step_n = prof["ppo_step"]
            avg_s = step_ms / max(step_n, 1) * 1e-3
            achieved = FLOP_PER_SAMPLE_STEP * learner.minibatch / avg_s / 1e12
            roof = {"bound": "mfma", "kernel": step_plan(learner, learner.minibatch)[0], "achieved": achieved,
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                    # (looked up by kernel name AND launch geometry: workgroups of the plan x 256 threads)
                    "traffic": hbm_traffic_bytes(step_plan(learner, learner.minibatch)[0], step_plan(learner, learner.minibatch)[1] * 256)
                    if learner.minibatch == MINIBATCH else None,
                    "traffic_source": TRAFFIC_SOURCE,
                    "avg_launch_us": avg_s * 1e6, "launches": step_n, "rows_per_launch": learner.minibatch,
                    "algorithmic_flop_per_launch": FLOP_PER_SAMPLE_STEP * learner.minibatch}
            extra["kernel_us"] = {k: (v[0] / v[1] * 1e3 if v[1] else None) for k, v in prof.items()}
            extra["inner_update_steps_per_s"] = REPEAT * n_chunks / t_inner
            if world == 1:
                extra["strong_scaling_projection"] = strong_scaling_projection(learner, b)
        t_gae = time_gae(learner)
        gbps = GAE_BYTES_PER_TRANSITION * learner.n_trans / t_gae / 1e9
        extra["gae_transitions_per_s"] = learner.n_trans / t_gae
        extra["roofline_gae"] = {"bound": "hbm", "kernel": "gae_single_pass", "achieved": gbps,
                                 "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                                 # (the 2^20 launch's own counters: tiles of 2,048 transitions x 256 threads)
                                 "traffic": hbm_traffic_bytes("gae_single_pass", -(-learner.n_trans // 2048) * 256),
                                 "traffic_source": TRAFFIC_SOURCE,
                                 "avg_launch_us": t_gae * 1e6, "transitions": learner.n_trans,
                                 "algorithmic_bytes_per_launch": GAE_BYTES_PER_TRANSITION * learner.n_trans}
        for key, envs in (("roofline_gae_2p24", 8192), ("roofline_gae_2p24_c2_layout", N_ENV)):
            # 2^24 transitions as 8192 sub-buffers x 2048 slots (C2's slot count) and as C2's own 512 sub-buffers x 32768
            n_l, t_l = time_gae_large(learner, n_env=envs)
            gbps_l = GAE_BYTES_PER_TRANSITION * n_l / t_l / 1e9
            extra[key] = {"bound": "hbm", "kernel": "gae_single_pass", "achieved": gbps_l,
                          "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps_l / PEAK_HBM_GBPS,
                          "traffic": hbm_traffic_bytes("gae_single_pass", -(-n_l // 2048) * 256),
                          "traffic_source": TRAFFIC_SOURCE, "avg_launch_us": t_l * 1e6, "transitions": n_l, "sub_buffers": envs,
                          "transitions_per_s": n_l / t_l,
                          "algorithmic_bytes_per_launch": GAE_BYTES_PER_TRANSITION * n_l}
        t1 = time.perf_counter()
        for _ in range(3):
            learner.preprocess(local_only=True)
        torch.cuda.synchronize()
        extra["preprocess_transitions_per_s"] = 3 * learner.n_trans / (time.perf_counter() - t1)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
    if rank == 0 and world == 1 and not args.no_extras and not args.no_cpu_baseline:
        t_h2d = h2d_seconds(device)
        t_update = elapsed / args.steps
        extra["value_incl_h2d"] = {"value": total_steps / args.steps / (t_update + t_h2d), "unit": "update-steps/s",
                                   "h2d_ms": t_h2d * 1e3, "bytes": H2D_BYTES_PER_UPDATE,
                                   "note": "`value` with one pinned-host -> HBM copy of the 2^20-transition batch added to every update()"}
        extra["rocm_eager_baseline"] = rocm_eager_baseline(device)
        extra["recompute_advantage"] = recompute_leg(learner, args.steps)
        del learner
        torch.cuda.empty_cache()
        extra["hook_level"] = hook_level(device)                                   # device-side minibatch permutations
        extra["hook_level_host_perms"] = hook_level(device, updates=1, permutations="host")   # the reference's np.random draws
        torch.cuda.empty_cache()
        extra["other_workloads"] = other_workloads()

    if rank == 0:
        if head == "strong":
            shard = (f"ONE rollout of 512 envs x 2048 steps = 2^20 transitions" +
                     (f" sharded over {world} ranks by env id ({learner_desc(world)}), global minibatch {MINIBATCH} = "
                      f"{MINIBATCH // world} rows per rank per step" if world > 1 else f", minibatch {MINIBATCH}"))
        else:
            shard = f"512 envs x 2048 steps = 2^20 transitions PER GPU, local minibatch {MINIBATCH}"
        out = {
            "metric": "PPO learn() update-steps/sec (minibatch 65536, preprocessing incl.) + GAE transitions/sec; "
                      "cpu_baseline / rocm_eager_baseline are PORTS of the reference path (oracle/), kind=port",
            "value": value,
            "unit": "update-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": head, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2 PPO MuJoCo-shape rollout ({'BASELINE configs[3]' if world > 1 and head == 'strong' else 'BASELINE configs[1]'}): "
                                   f"{shard}, obs 17, act 6, MLP[64,64] actor-critic, repeat 10, recompute_advantage off",
                       "gradient_steps_per_step": total_steps // max(args.steps, 1),
                       "transitions_per_step": N_TRANS * (world if head == "weak" else 1),
                       "parallelism": f"dp{world}", "exchange": exchange},
            "roofline": roof, "cpu_baseline": cpu, "final_losses": final_loss,
        }
        if world > 1:
            out["exchange_us"] = exchange_us
            out["exchange_ranks"] = rccl_ranks
            out["exchange_selftest"] = selftest
            if xch["note"]:
                out["exchange_note"] = xch["note"]
        if beside is not None:
            out["weak_scaling" if beside["scaling"] == "weak" else "strong_scaling"] = beside
        out.update(extra)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


def learner_desc(world):
    from tianshou_amd.distributed import shard_envs

    sizes = sorted({shard_envs(N_ENV, r, world)[1] - shard_envs(N_ENV, r, world)[0] for r in range(world)})
    return " / ".join(str(x) for x in sizes) + " sub-buffers per rank"


def recompute_leg(learner, steps):
    """The same update() with recompute_advantage=True (the default of examples/mujoco/mujoco_ppo.py:56; ppo.py:174-178:
    V(s), V(s'), GAE and the return normalisation are redone before every repeat after the first).  The headline runs
    with it off (SURVEY 8d's C2 definition); this is the second number."""
    import dataclasses

    from tianshou_amd.ppo import PPOEngine

    cfg = dataclasses.replace(learner.cfg, recompute_advantage=True)
    eng = PPOEngine(OBS, ACT, init_flat_params(0).to(learner.device), cfg)
    obs, obs_next, act, rew, term, trunc = learner.data

    def once():
        b = eng.preprocess(obs, obs_next, act, rew, term, trunc, learner.cut)
        perms = [learner.next_perm() for _ in range(REPEAT)]
        return eng.update(b, learner.minibatch, REPEAT, perms)

    once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = 0
    for _ in range(max(1, steps)):
        total += once()[1]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"value": total / el, "unit": "update-steps/s", "ms_per_step": el / max(1, steps) * 1e3,
            "note": "recompute_advantage=True: 9 more passes of V(s), V(s'), GAE, return normalisation per update()"}


if __name__ == "__main__":
    main()
