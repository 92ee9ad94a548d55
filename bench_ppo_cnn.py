"""North-star Atari shape: PPO update-steps/s on a synthetic 2^20-transition rollout of u8 84x84 frames
(512 envs x 2048 steps, frame stack 4 through prev()), shared NatureCNN actor-critic, minibatch 65536.

    python bench.py --workload ppo_atari [--steps K] [--warmup W]        (or: python bench_ppo_cnn.py)

One "step" = one PPO.update(): V(s), V(s'), log pi_old over the whole rollout (one trunk pass per observation),
GAE, then `repeat` x 16 minibatch gradient steps of 65536 (gather + stack frames, forward, Categorical PPO loss,
backward, clip + Adam).  value = gradient steps / s (preprocessing inside the timed region).
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

C, H, W, N_ACT = 4, 84, 84, 6
N_ENV, T = 512, 2048
MINIBATCH = 65536
PEAK_F32_MFMA_TFLOPS = 157.3
FWD_FLOP = 2 * (3_276_800 + 2_654_208 + 1_806_336 + 1_605_632 + 512 * (N_ACT + 1))   # per sample, one trunk + heads
CONV1_FLOP = 2 * 3_276_800
STEP_FLOP = 3 * FWD_FLOP - CONV1_FLOP        # forward + weight gradients + input gradients (not for conv1)


def cpu_baseline(batch: int = 256):
    from oracle import oracle_ppo as OP
    from oracle import oracle_ppo_cnn as OC

    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    rng = np.random.default_rng(0)
    p = OC.init_params(C, H, W, N_ACT, 0)
    cfg = OP.PPOConfig(eps_clip=0.1, value_clip=True, advantage_normalization=True, vf_coef=0.25, ent_coef=0.01,
                       max_grad_norm=0.5, lr=2.5e-4, adam_eps=1e-5)
    st = OP.PPOState(params=p)
    obs = rng.integers(0, 256, size=(batch, C, H, W), dtype=np.uint8)
    act = rng.integers(0, N_ACT, size=batch)
    pre = {"adv": torch.randn(batch), "returns": torch.randn(batch), "logp_old": torch.full((batch,), -1.8),
           "v_s": torch.zeros(batch)}
    OC.update(st, cfg, obs, act, pre, batch, 1, [np.arange(batch)])
    t0 = time.perf_counter()
    n = 3
    OC.update(st, cfg, obs, act, pre, batch, n, [np.arange(batch)] * n)
    dt = (time.perf_counter() - t0) / n
    return {"value": 1.0 / (dt * MINIBATCH / batch), "unit": "update-steps/s", "cores": threads, "kind": "port",
            "sample": f"{n} gradient steps on {batch} samples (torch fp32 CPU oracle, two trunk passes like the "
                      f"reference), scaled to the 65536-sample minibatch; preprocessing excluded"}


def run(steps: int, warmup: int, repeat: int = 2, with_cpu: bool = True) -> dict:
    from tianshou_amd import _lib
    from tianshou_amd import ppo_cnn as PC
    from tianshou_amd.buffer import DeviceReplayBuffer, random_permutation
    from tianshou_amd.ppo import PPOConfig

    dev = torch.device("cuda")
    n = N_ENV * T
    g = torch.Generator(device=dev).manual_seed(0)
    frames = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    for lo in range(0, n, 1 << 16):
        frames[lo:lo + (1 << 16)] = torch.randint(0, 256, (min(1 << 16, n - lo), H, W), generator=g, device=dev,
                                                  dtype=torch.uint8)
    act = torch.randint(0, N_ACT, (n,), generator=g, device=dev)
    rew = torch.randn(n, generator=g, device=dev).double()
    term = torch.rand(n, generator=g, device=dev) < 0.002
    buf = DeviceReplayBuffer.from_vector_fill(N_ENV, rew=rew, terminated=term,
                                              truncated=torch.zeros(n, dtype=torch.bool, device=dev))
    torch.manual_seed(0)
    mods = [torch.nn.Conv2d(C, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1),
            torch.nn.Linear(3136, 512), torch.nn.Linear(512, N_ACT), torch.nn.Linear(512, 1)]
    tensors = [t for m in mods for t in (m.weight, m.bias)]
    cfg = PPOConfig(gamma=0.99, gae_lambda=0.95, eps_clip=0.1, value_clip=True, advantage_normalization=True,
                    vf_coef=0.25, ent_coef=0.01, max_grad_norm=0.5, return_scaling=False, lr=2.5e-4, adam_eps=1e-5)
    eng = PC.CnnPPOEngine(C, H, W, N_ACT, PC.flat_from_torch(tensors, C, H, W, N_ACT), cfg)
    seed = [0]

    def update():
        pre = eng.preprocess(buf, frames, act, C)
        perms = []
        for _ in range(repeat):
            seed[0] += 1
            perms.append(random_permutation(n, seed[0], dev))
        return eng.update(buf, frames, pre, C, MINIBATCH, repeat, perms)

    import bench_init as _BI

    _BI.warm_clocks(dev)
    for _ in range(warmup):
        update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = 0
    for _ in range(steps):
        losses, k = update()
        total += k
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    # per-kernel-kind timing of ONE minibatch step
    pre = eng.preprocess(buf, frames, act, C)
    rows = random_permutation(n, 12345, dev)[:MINIBATCH]
    obs = PC.gather_obs_nhwc(frames, buf, pre["indices"][rows], C, as_u8=True)
    ws = _lib.default_workspace(0)
    torch.cuda.synchronize()
    ws.profile_begin()
    t1 = time.perf_counter()
    eng.step(obs, pre["act"][rows], pre["adv"][rows], pre["returns"][rows], pre["logp_old"][rows], pre["v_s"][rows])
    torch.cuda.synchronize()
    t_step = time.perf_counter() - t1
    prof = ws.profile_end()
    gemm_ms = sum(prof[k][0] for k in ("conv_fwd", "conv_wgrad", "conv_dgrad"))
    launches = sum(prof[k][1] for k in ("conv_fwd", "conv_wgrad", "conv_dgrad"))
    tf = STEP_FLOP * MINIBATCH / (gemm_ms * 1e-3) / 1e12
    roof = {"bound": "mfma", "kernel": "conv_rows2_kernel / conv_wgrad2_kernel (14 GEMM launches of one minibatch step; second generation)",
            "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F32_MFMA_TFLOPS,
            "traffic": None, "avg_launch_us": gemm_ms * 1e3 / max(launches, 1), "launches": launches,
            "kernel_ms_per_step": {k: prof[k][0] for k in ("conv_fwd", "conv_wgrad", "conv_dgrad")},
            "algorithmic_flop_per_step": STEP_FLOP * MINIBATCH, "one_step_wall_ms": t_step * 1e3}
    return {
        "metric": "PPO learn() update-steps/sec, Atari shape (minibatch 65536, preprocessing incl.)",
        "value": total / dt, "unit": "update-steps/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"PPO Atari-shape rollout: {N_ENV} envs x {T} steps = 2^20 transitions of u8[84,84] frames "
                               f"(stack 4), shared NatureCNN actor-critic (1,687,719 params), minibatch 65536, "
                               f"repeat {repeat}", "gradient_steps_per_step": repeat * (n // MINIBATCH),
                   "transitions_per_step": n, "parallelism": "dp1"},
        "roofline": roof, "cpu_baseline": cpu_baseline() if with_cpu else None,
        "final_losses": [float(x) for x in losses[-1].tolist()],
    }


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a.steps, a.warmup, a.repeat, not a.no_cpu_baseline)), flush=True)
