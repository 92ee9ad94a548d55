"""conv2's input gradient at the Atari-shape minibatch: pixel-shuffle form (TS_DGRAD_PS=1) against one GEMM per parity, over the
rows2 variants (waves per workgroup, row tiles per wave).  Usage: python scripts/gpu_r6_dgrad_ps_sweep.py [batch]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from tianshou_amd import _lib
from tianshou_amd import dqn as D

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


lib.ts_conv_set_generation(1)
IH = IW = 20; IC = 32; K = 4; S = 2; OC = 64
x = torch.randn(B, IH, IW, IC, device="cuda").clamp_(min=0)
wb = torch.randn(K * K * IC + 1, OC, device="cuda") * 0.05
dy = torch.randn(B, 9, 9, OC, device="cuda")
gf = 2.0 * B * 81 * OC * K * K * IC / 1e9
t_w = timeit(lambda: D.conv_backward(x, wb, dy, K, K, S, need_dx=False), 5)
for ps in ("1", "0"):
    os.environ["TS_DGRAD_PS"] = ps
    for waves, tm in ((0, 0), (8, 2), (16, 1)) + (((16, 2),) if ps == "0" else ()):
        for k in ("TS_R2_WAVES", "TS_R2_TM"):
            os.environ.pop(k, None)
        if waves:
            os.environ["TS_R2_WAVES"], os.environ["TS_R2_TM"] = str(waves), str(tm)
        try:
            t = timeit(lambda: D.conv_backward(x, wb, dy, K, K, S, mask=x, need_dx=True), 5) - t_w
            print(f"B={B} conv2 dgrad ps={ps} waves {waves or 'default'} tm {tm or 'default'}: {t:8.1f} us  {gf / t * 1e3:6.1f} TF/s (algorithmic {gf:.0f} GF)", flush=True)
        except Exception as e:
            print(f"ps={ps} waves {waves} tm {tm}: {type(e).__name__}: {e}")
