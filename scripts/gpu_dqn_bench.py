"""C3 workload: DQN updates/s at B=512 on a 2^20-slot Atari-layout PER buffer (synthetic).
    python scripts/gpu_dqn_bench.py [--slots 1048576] [--updates 50] [--batch 512]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tianshou_amd import dqn as D  # noqa: E402
from tianshou_amd.buffer import DeviceReplayBuffer  # noqa: E402
from tianshou_amd.segtree import PrioritizedWeights  # noqa: E402


def build(slots: int, E: int, n_act: int, seed: int = 0):
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(seed)
    frames = torch.empty((slots, 84, 84), dtype=torch.uint8, device=dev)
    step = 1 << 16
    for lo in range(0, slots, step):
        frames[lo:lo + step] = torch.randint(0, 256, (min(step, slots - lo), 84, 84), generator=g, device=dev,
                                             dtype=torch.uint8)
    rew = torch.randn(slots, generator=g, device=dev).double()
    term = torch.rand(slots, generator=g, device=dev) < 0.005
    trunc = torch.zeros(slots, dtype=torch.bool, device=dev)
    act = torch.randint(0, n_act, (slots,), generator=g, device=dev)
    T = slots // E
    offset = np.arange(E + 1, dtype=np.int64) * T
    buf = DeviceReplayBuffer(offset=offset, last_index=offset[:-1] + T - 1, lengths=np.full(E, T, np.int64),
                             insertion=np.zeros(E, np.int64), rew=rew, terminated=term, truncated=trunc)
    per = PrioritizedWeights(slots, 0.6, 0.4)
    per.init_weight(torch.arange(slots, device=dev))
    return frames, act, buf, per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=1 << 20)
    ap.add_argument("--updates", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--huber", type=float, default=1.0)
    a = ap.parse_args()
    c, h, w, A = 4, 84, 84, 6
    frames, act, buf, per = build(a.slots, 16, A)
    torch.manual_seed(0)
    net = [torch.nn.Conv2d(c, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1),
           torch.nn.Linear(3136, 512), torch.nn.Linear(512, A)]
    tensors = [t for m in net for t in (m.weight, m.bias)]
    cfg = D.DQNConfig(gamma=0.99, n_step=3, target_update_freq=500, is_double=True,
                      huber_delta=a.huber if a.huber > 0 else None, lr=1e-4)
    eng = D.DQNEngine(c, h, w, A, D.flat_from_torch(tensors, c, h, w, A), cfg)
    gen = torch.Generator(device="cuda").manual_seed(1)

    def update():
        u = torch.rand(a.batch, generator=gen, device="cuda", dtype=torch.float64)
        idx, wt = per.sample(u)
        ret = eng.preprocess(buf, frames, idx, c)
        obs = D.gather_obs_nhwc(frames, buf, idx, c)
        loss, td = eng.update_with_batch(obs, act[idx], ret, wt)
        per.update_weight(idx, td)
        return loss

    for _ in range(a.warmup):
        update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.updates):
        loss = update()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flop = 86.9e6 * a.batch
    print(json.dumps({"metric": "DQN updates/s (C3: NatureCNN, n-step 3, PER, double-Q, B=%d)" % a.batch,
                      "value": a.updates / dt, "ms_per_update": 1e3 * dt / a.updates,
                      "mfma_frac_whole_update": flop * a.updates / dt / 157.3e12, "loss": float(loss)}))


if __name__ == "__main__":
    main()
