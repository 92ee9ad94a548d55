"""A few launches of every second-generation conv kernel at the Atari-shape minibatch, for rocprofv3 (--stats / --pmc):
    python scripts/gpu_conv2_one.py [batch] [layers,comma,separated]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from tianshou_amd import _lib
from tianshou_amd import dqn as D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
want = sys.argv[2].split(",") if len(sys.argv) > 2 else None
_lib.load().ts_conv_set_generation(int(os.environ.get("GEN", "1")))
layers = [("conv1u8", 84, 84, 4, 8, 4, 32, True), ("conv2", 20, 20, 32, 4, 2, 64, False),
          ("conv3", 9, 9, 64, 3, 1, 64, False), ("fc1", 1, 1, 3136, 1, 1, 512, False)]
for name, IH, IW, IC, K, S, OC, u8 in layers:
    if want and name not in want:
        continue
    x = torch.randint(0, 256, (B, IH, IW, IC), device="cuda", dtype=torch.uint8) if u8 else \
        torch.randn(B, IH, IW, IC, device="cuda").clamp_(min=0)
    wb = torch.randn(K * K * IC + 1, OC, device="cuda") * 0.05
    oh, ow = (IH - K) // S + 1, (IW - K) // S + 1
    dy = torch.randn(B, oh, ow, OC, device="cuda")
    for _ in range(3):
        D.conv_forward(x, wb, K, K, S, True)
        D.conv_backward(x, wb, dy, K, K, S, mask=None if u8 else x, need_dx=not u8)
    torch.cuda.synchronize()
    del x, dy
    torch.cuda.empty_cache()
print("done")
